"""Single-kernel GPflow-style conditional -- counterpart of vfegpssm/conditionals.py (jitter 1e-7, :101).

R independent GPs (columns of f) share ONE kernel; arithmetic identical to conditionals_multi_output."""
from __future__ import annotations

from . import conditionals_multi_output as _cmo

JITTER = 1e-7


def conditional(Xnew, X, kern, f, *, full_cov=False, q_sqrt=None, white=False):
    import numpy as np
    R = np.asarray(f).shape[1]
    return _cmo.conditional(Xnew, X, [kern] * R, f, full_cov=full_cov, q_sqrt=q_sqrt, white=white, jitter=JITTER)
