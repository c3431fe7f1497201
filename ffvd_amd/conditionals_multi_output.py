"""GP conditional operators for D independent kernels -- counterpart of vfegpssm/conditionals_multi_output.py.

Same function names and argument order as the reference; NumPy in/out; all arithmetic in libffvd_hip.so.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .kernels import stack_hypers

JITTER = 1e-5      # conditionals_multi_output.py:108,159


def kernel_pre_cal(X, kern):
    """Per kernel d: L_d = chol(K_d(X) + 1e-5 I); returns the list of L_d^{-T} (upper triangular)
    (conditionals_multi_output.py:124-169)."""
    lib = _lib.load()
    kind, _, logvar, loglen = stack_hypers(kern)
    X = _lib.as_f64(X)
    M, P = X.shape
    D = len(kern)
    out = np.empty((D, M, M))
    rc = lib.ffvd_op_kernel_pre_cal(kind, _lib.dptr(X), M, P, D, _lib.dptr(logvar),
                                    None if loglen is None else _lib.dptr(loglen), JITTER, _lib.dptr(out))
    _lib.check(rc, None, "kernel_pre_cal")
    return [out[d] for d in range(D)]


def collapse_after_kernel_precalculation(Lm_inverse_seq, X_combine, X, Z, kern, Q, batch_size, Y_N):
    """Collapsed-U ELBO terms (-term1/Y_N, -term2/Y_N, -trace/Y_N) (conditionals_multi_output.py:230-257)."""
    lib = _lib.load()
    kind, _, logvar, loglen = stack_hypers(kern)
    D = len(kern)
    Z = _lib.as_f64(Z)
    M, P = Z.shape
    X = _lib.as_f64(X)
    T = X.shape[0] - 1
    Xc = _lib.as_f64(X_combine, (T, P), "X_combine")
    X = _lib.as_f64(X, (T + 1, D), "X")
    W = _lib.as_f64(np.stack([np.asarray(w) for w in Lm_inverse_seq]), (D, M, M), "Lm_inverse_seq")
    Q = _lib.as_f64(Q, (D,), "Q")
    out = np.zeros(3)
    rc = lib.ffvd_op_collapse(kind, _lib.dptr(W), _lib.dptr(Xc), _lib.dptr(X), _lib.dptr(Z), T, M, P, D,
                              _lib.dptr(logvar), None if loglen is None else _lib.dptr(loglen), _lib.dptr(Q),
                              float(batch_size), float(Y_N), _lib.dptr(out))
    _lib.check(rc, None, "collapse_after_kernel_precalculation")
    return float(out[0]), float(out[1]), float(out[2])


def conditional(Xnew, X, kern, f, *, full_cov=False, q_sqrt=None, white=False, return_Lm=False, jitter=JITTER):
    """Mean and variance (N x D each) of D independent GPs at Xnew given whitened values f at X
    (conditionals_multi_output.py:73-120 -> base_conditional :6-70).

    Only the configuration the GP-SSM path uses is implemented: white=True, full_cov=False, q_sqrt=None,
    return_Lm=False (return_Lm=True is broken in the reference)."""
    if full_cov or q_sqrt is not None or not white or return_Lm:
        raise NotImplementedError("conditional: only white=True, full_cov=False, q_sqrt=None, return_Lm=False")
    lib = _lib.load()
    kind, _, logvar, loglen = stack_hypers(kern)
    D = len(kern)
    X = _lib.as_f64(X)
    M, P = X.shape
    Xnew = _lib.as_f64(Xnew)
    if Xnew.ndim != 2 or Xnew.shape[1] != P:
        raise ValueError(f"Xnew: expected (N, {P}), got {Xnew.shape}")
    N = Xnew.shape[0]
    f = _lib.as_f64(f, (M, D), "f")
    mean, var = np.empty((N, D)), np.empty((N, D))
    rc = lib.ffvd_op_conditional(kind, _lib.dptr(Xnew), N, _lib.dptr(X), M, P, D, _lib.dptr(logvar),
                                 None if loglen is None else _lib.dptr(loglen), _lib.dptr(f), float(jitter),
                                 _lib.dptr(mean), _lib.dptr(var))
    _lib.check(rc, None, "conditional")
    return mean, var


def collapse_u_mean_after_kernel_precalculation(Lm_inverse_seq, X_combine, X, Z, kern, Q):
    """Posterior mean of the whitened inducing outputs (M x D) and the stack of L_H^{-T} (D x M x M)
    (conditionals_multi_output.py:206-227)."""
    lib = _lib.load()
    kind, _, logvar, loglen = stack_hypers(kern)
    D = len(kern)
    Z = _lib.as_f64(Z)
    M, P = Z.shape
    X = _lib.as_f64(X)
    T = X.shape[0] - 1
    Xc = _lib.as_f64(X_combine, (T, P), "X_combine")
    X = _lib.as_f64(X, (T + 1, D), "X")
    W = _lib.as_f64(np.stack([np.asarray(w) for w in Lm_inverse_seq]), (D, M, M), "Lm_inverse_seq")
    Q = _lib.as_f64(Q, (D,), "Q")
    U_mean, Hinv = np.empty((M, D)), np.empty((D, M, M))
    rc = lib.ffvd_op_collapse_u_mean(kind, _lib.dptr(W), _lib.dptr(Xc), _lib.dptr(X), _lib.dptr(Z), T, M, P, D,
                                     _lib.dptr(logvar), None if loglen is None else _lib.dptr(loglen), _lib.dptr(Q),
                                     _lib.dptr(U_mean), _lib.dptr(Hinv))
    _lib.check(rc, None, "collapse_u_mean_after_kernel_precalculation")
    return U_mean, Hinv


def conditional_after_kernel_precalculation(Lm_inverse_seq, Xnew, Z, kern, f, *, full_cov=False, q_sqrt=None,
                                            white=False, return_Lm=False):
    """conditional() with the pre-computed L^{-T} stack (conditionals_multi_output.py:306-387); mean, var N x D.

    q_sqrt may be a D x M x M stack: as in the reference, slice d = 0 inflates the variance of EVERY dim
    (the stack is handed to every dim at :317 and `[:, :, 0]` at :322 keeps slice 0)."""
    if full_cov or not white or return_Lm:
        raise NotImplementedError("conditional_after_kernel_precalculation: only white=True, full_cov=False")
    lib = _lib.load()
    kind, _, logvar, loglen = stack_hypers(kern)
    D = len(kern)
    Z = _lib.as_f64(Z)
    M, P = Z.shape
    Xnew = _lib.as_f64(Xnew)
    if Xnew.ndim != 2 or Xnew.shape[1] != P:
        raise ValueError(f"Xnew: expected (N, {P}), got {Xnew.shape}")
    N = Xnew.shape[0]
    f = _lib.as_f64(f, (M, D), "f")
    W = _lib.as_f64(np.stack([np.asarray(w) for w in Lm_inverse_seq]), (D, M, M), "Lm_inverse_seq")
    qs = None
    if q_sqrt is not None:
        q = np.asarray(q_sqrt, dtype=np.float64)
        if q.ndim != 3 or q.shape[1:] != (M, M):
            raise ValueError("Bad dimension for q_sqrt: expected (D, M, M)")
        qs = np.ascontiguousarray(q[0])
    mean, var = np.empty((N, D)), np.empty((N, D))
    rc = lib.ffvd_op_conditional_precalc(kind, _lib.dptr(W), _lib.dptr(Xnew), N, _lib.dptr(Z), M, P, D,
                                         _lib.dptr(logvar), None if loglen is None else _lib.dptr(loglen),
                                         _lib.dptr(f), None if qs is None else _lib.dptr(qs), _lib.dptr(mean),
                                         _lib.dptr(var))
    _lib.check(rc, None, "conditional_after_kernel_precalculation")
    return mean, var
