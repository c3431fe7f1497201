"""Posterior rollouts / prediction: the loop of BaseModel.collect_samples_formal
(vfegpssm/base_model.py:197-350) with the R posterior rollouts advanced side by side on the GPU.

The reference builds `num` rollouts one after another, each `test_len` sequential calls of
conditional_after_kernel_precalculation at ONE point; with the collapsed U (case 4) they differ only by their
noise draws, so here every step is one batched call at the R current states: `ffvd_op_rollout` enqueues
steps x (projection, q_sqrt inflation, conditional, update) on one stream without host round trips.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .kernels import stack_hypers


def rollout(Lm_inverse_seq, Z, kern, U_val, q_sqrt, x_last, control_inputs, ctrl_offset, steps, Q, eps):
    """predict_x, predict_x_var (R, steps, D) of base_model.py:288-314.

    q_sqrt: None or the (D, M, M) stack U_variance_cholesky -- slice d = 0 is used for every dim;
    x_last: (D,) = layers[-1].X[-1]; control_inputs: (n, C), row ctrl_offset + t feeds step t;
    eps: (steps, R, D) standard-normal draws (tf.random.normal at :306, injected)."""
    kind, _, logvar, loglen = stack_hypers(kern)
    D = len(kern)
    Z = _lib.as_f64(Z)
    M, P = Z.shape
    eps = _lib.as_f64(eps)
    if eps.ndim != 3 or eps.shape[2] != D or eps.shape[0] != steps:
        raise ValueError(f"eps: expected ({steps}, R, {D}), got {eps.shape}")
    R = eps.shape[1]
    C = P - D
    ctrl = None
    if C > 0:
        ci = _lib.as_f64(control_inputs)
        if ci.ndim != 2 or ci.shape[1] != C or ci.shape[0] < ctrl_offset + steps:
            raise ValueError(f"control_inputs: need at least {ctrl_offset + steps} rows of {C} columns")
        ctrl = np.ascontiguousarray(ci[ctrl_offset: ctrl_offset + steps])
    W = _lib.as_f64(np.stack([np.asarray(w) for w in Lm_inverse_seq]), (D, M, M), "Lm_inverse_seq")
    f = _lib.as_f64(U_val, (M, D), "U_val")
    qs = None
    if q_sqrt is not None:
        q = np.asarray(q_sqrt, dtype=np.float64)
        if q.ndim != 3 or q.shape[1:] != (M, M):
            raise ValueError("Bad dimension for q_sqrt: expected (D, M, M)")
        qs = np.ascontiguousarray(q[0])
    x_last = _lib.as_f64(x_last, (D,), "x_last")
    log_Q = np.log(_lib.as_f64(Q, (D,), "Q"))
    px, pv = np.empty((R, steps, D)), np.empty((R, steps, D))
    rc = _lib.load().ffvd_op_rollout(kind, _lib.dptr(W), _lib.dptr(Z), M, P, D, _lib.dptr(logvar),
                                     None if loglen is None else _lib.dptr(loglen), _lib.dptr(f),
                                     None if qs is None else _lib.dptr(qs), _lib.dptr(x_last), R,
                                     None if ctrl is None else _lib.dptr(ctrl), C, steps, _lib.dptr(log_Q), _lib.dptr(eps),
                                     _lib.dptr(px), _lib.dptr(pv))
    _lib.check(rc, None, "ffvd_op_rollout")
    return px, pv


def predict_y_summary(predict_x, predict_x_var, CC, DD, log_Rchols, Y_test=None, Y_train_std=1.0):
    """base_model.py:330-348 (host-side: a (num, test_len, D) x (D, Ydim) contraction and three means)."""
    CC, DD = np.asarray(CC, dtype=np.float64), np.asarray(DD, dtype=np.float64)
    predict_y = (np.mean(np.einsum("ijk,kl->ijl", predict_x, CC), axis=0) + DD[None, :]).reshape(-1)
    predict_y_var = (np.mean(np.einsum("ijk,kl->ijl", predict_x_var, CC ** 2), axis=0).reshape(-1)
                     + np.exp(2 * np.asarray(log_Rchols, dtype=np.float64))).reshape(-1)
    out = {"predict_y": predict_y, "predict_y_var": predict_y_var}
    if Y_test is not None:
        y30, p30 = np.asarray(Y_test, dtype=np.float64)[:30].reshape(-1), predict_y[:30]
        out["RMSE"] = float(np.sqrt(np.mean((y30 - p30) ** 2)) * Y_train_std)
    return out


def pg_sweep(Lm_inverse_seq, Z, kern, U_val, X_ref, Y, control_inputs, CC, DD, Rchols, Q, x0, eps, unif):
    """One particle-Gibbs sweep over the latent trajectory: the INTENT of BaseModel.PG_for_X_speedup
    (base_model.py:78-138; the op as written never updates X, see include/ffvd_abi.h `ffvd_op_pg_sweep`).

    X_ref (X_N, D): the current trajectory = the reference particle; Y (>= X_N-1, Ydim); control_inputs (>= X_N-1, C) or
    None; Rchols (Ydim, Ydim) = likelihood.Rchols = exp(log_Rchols), lower triangular; Q (D,);
    x0 (PG_particles-1, D): the N(0, I) start of :79; eps (X_N-1, PG_particles-1, D): the normal draws of :101;
    unif (X_N-1, PG_particles-1) in [0, 1): the categorical draws of :113 as inverse-CDF uniforms.
    Returns particles (X_N, PG_particles-1, D) (`resampled_X`, :133) and idx (X_N-1, PG_particles-1)."""
    kind, _, logvar, loglen = stack_hypers(kern)
    D = len(kern)
    Z = _lib.as_f64(Z)
    M, P = Z.shape
    C = P - D
    X_ref = _lib.as_f64(X_ref)
    if X_ref.ndim != 2 or X_ref.shape[1] != D:
        raise ValueError(f"X_ref: expected (X_N, {D}), got {X_ref.shape}")
    XN = X_ref.shape[0]
    steps = XN - 1
    x0 = _lib.as_f64(x0)
    if x0.ndim != 2 or x0.shape[1] != D or x0.shape[0] < 1:
        raise ValueError(f"x0: expected (PG_particles - 1, {D}), got {x0.shape}")
    R = x0.shape[0]
    eps = _lib.as_f64(eps, (steps, R, D), "eps")
    unif = _lib.as_f64(unif, (steps, R), "unif")
    if unif.size and (unif.min() < 0.0 or unif.max() >= 1.0):
        raise ValueError("unif: the categorical draws are inverse-CDF uniforms in [0, 1)")
    Y = _lib.as_f64(Y)
    if Y.ndim != 2 or Y.shape[0] < steps:
        raise ValueError(f"Y: need at least {steps} rows")
    Ydim = Y.shape[1]
    Yc = np.ascontiguousarray(Y[:steps])
    ctrl = None
    if C > 0:
        ci = _lib.as_f64(control_inputs)
        if ci.ndim != 2 or ci.shape[1] != C or ci.shape[0] < steps:
            raise ValueError(f"control_inputs: need at least {steps} rows of {C} columns")
        ctrl = np.ascontiguousarray(ci[:steps])
    W = _lib.as_f64(np.stack([np.asarray(w) for w in Lm_inverse_seq]), (D, M, M), "Lm_inverse_seq")
    f = _lib.as_f64(U_val, (M, D), "U_val")
    CC = _lib.as_f64(CC, (D, Ydim), "CC")
    DD = _lib.as_f64(np.asarray(DD).reshape(-1), (Ydim,), "DD")
    Rch = _lib.as_f64(Rchols, (Ydim, Ydim), "Rchols")
    log_Q = np.log(_lib.as_f64(Q, (D,), "Q"))
    parts = np.empty((XN, R, D))
    idx = np.zeros((max(steps, 0), R), dtype=np.int32)
    rc = _lib.load().ffvd_op_pg_sweep(kind, _lib.dptr(W), _lib.dptr(Z), M, P, D, _lib.dptr(logvar),
                                      None if loglen is None else _lib.dptr(loglen), _lib.dptr(f), _lib.dptr(X_ref), XN,
                                      _lib.dptr(Yc), Ydim, None if ctrl is None else _lib.dptr(ctrl), C, _lib.dptr(CC),
                                      _lib.dptr(DD), _lib.dptr(Rch), _lib.dptr(log_Q), R, _lib.dptr(x0), _lib.dptr(eps),
                                      _lib.dptr(unif), _lib.dptr(parts), idx.ctypes.data)
    _lib.check(rc, None, "ffvd_op_pg_sweep")
    return parts, idx
