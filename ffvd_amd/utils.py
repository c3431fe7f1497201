"""Counterpart of vfegpssm/utils.py: the reparameterised Monte-Carlo draw, with the N(0,1) sample injected
(TensorFlow's random stream is not reproducible outside TensorFlow, SURVEY 7 'RNG parity')."""
from __future__ import annotations

import numpy as np

from . import _lib


def get_rand(x, eps, full_cov=False):
    """mean + eps * sqrt(var) for x = (mean, var) (utils.py:11); eps has the shape of mean."""
    if full_cov:
        raise NotImplementedError("full_cov draws are not used by the GP-SSM path (FFVD_Main.py:266)")
    lib = _lib.load()
    mean = _lib.as_f64(x[0])
    var = _lib.as_f64(x[1], mean.shape, "var")
    eps = _lib.as_f64(eps, mean.shape, "eps")
    out = np.empty_like(mean)
    _lib.check(lib.ffvd_op_get_rand(_lib.dptr(mean), _lib.dptr(var), _lib.dptr(eps), mean.size, _lib.dptr(out)), None,
               "get_rand")
    return out
