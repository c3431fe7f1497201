"""Second, independently arranged CPU restatement (torch fp64) -- TEST INFRASTRUCTURE ONLY.

Purpose: cross-check `oracle/ffvd_oracle.py` (the line-by-line NumPy
restatement) with a differently organised computation of the same quantities
(batched over latent dims, triangular solves instead of explicit inverses,
Cholesky solves instead of LU) and provide reverse-mode gradients of `nll` for
later rounds (the reference obtains them from `tf.gradients`, base_model.py:148).
Parity status: unpinned (see the header of ffvd_oracle.py).

Same import rule as ffvd_oracle.py: tests / smoke / bench cpu_baseline only.
"""
from __future__ import annotations

import math

import torch


def _se_K(X, X2, logvar, loglen):
    """SE/ARD kernel batched over D: X (N,P), X2 (N2,P), logvar (D,), loglen (D,P) -> (D,N,N2).
    Expanded-form distance as kernels_multi_output.py:163-182."""
    ell = torch.exp(loglen)                       # D,P
    Xs = X[None] / ell[:, None, :]                # D,N,P
    X2s = X2[None] / ell[:, None, :]
    r2 = -2.0 * Xs @ X2s.transpose(1, 2) + (Xs * Xs).sum(-1, keepdim=True) \
        + (X2s * X2s).sum(-1, keepdim=True).transpose(1, 2)
    return torch.exp(logvar)[:, None, None] * torch.exp(-0.5 * r2)


def _lin_K(X, X2, logvar):
    """LinearK batched over D (kernels.py:270-276)."""
    return torch.exp(logvar)[:, None, None] * (X @ X2.T)[None]


def nll_terms(params, Y, control_inputs, *, U_collapse=True, kernel_type="SquaredExponential",
              prior_type="normal", jitter=1e-5):
    """Same contract as ffvd_oracle.nll_terms but with torch tensors (fp64); returns
    a dict of 0-d tensors so that autograd can differentiate `nll`."""
    X = params["X"]
    T = X.shape[0] - 1
    D = X.shape[1]
    Z = params["Z"]
    M = Z.shape[0]
    Q = torch.exp(params["log_Q"])
    Rrow = torch.exp(params["log_Rchols"])[0]
    logvar, loglen = params["logvariance"], params["loglengthscales"]
    se = kernel_type == "SquaredExponential"

    resid = (Y - (X[1:] @ params["CC"] + params["DD"])) / Rrow[None, :]
    out = {}
    out["nll_log_likelihood"] = -(-0.5 * (resid ** 2).sum() - T * torch.log(Rrow).sum()) / T
    if control_inputs is not None and control_inputs.shape[0] > 0:
        xc = torch.cat((X[:-1], control_inputs[:T]), dim=1)
    else:
        xc = X[:-1]
    Kuu = (_se_K(Z, Z, logvar, loglen) if se else _lin_K(Z, Z, logvar)) + jitter * torch.eye(M, dtype=X.dtype)
    Kuf = _se_K(Z, xc, logvar, loglen) if se else _lin_K(Z, xc, logvar)        # D,M,T
    L = torch.linalg.cholesky(Kuu)
    A = torch.linalg.solve_triangular(L, Kuf, upper=False)                      # D,M,T = F^T
    if se:
        kdiag = torch.exp(logvar)[:, None].expand(D, T)
    else:
        kdiag = torch.exp(logvar)[:, None] * (xc * xc).sum(-1)[None]
    delta = (X[1:] - X[:-1]).T                                                   # D,T

    hyp = -0.5 * ((params["log_Q"] ** 2).sum() + (params["CC"] ** 2).sum()
                  + (params["DD"] ** 2).sum() + (params["log_Rchols"] ** 2).sum())
    if se:
        # dgp_model.py:127: tf.cast(tf.math.log(0.05), tf.float64) -- a float32 logarithm, widened
        c_se = float(torch.log(torch.tensor(0.05, dtype=torch.float32)).double())
        p_hyper = -0.5 * (loglen ** 2).sum() - 0.5 * ((logvar - c_se) ** 2).sum()
    else:
        p_hyper = -0.5 * ((logvar - math.log(0.05)) ** 2).sum()
    p_Z = -0.5 * (Z ** 2).sum() if prior_type == "normal" else torch.zeros((), dtype=X.dtype)
    p_x0 = -0.5 * (X[0] ** 2).sum()

    if U_collapse:
        H = (A @ A.transpose(1, 2)) / Q[:, None, None] + torch.eye(M, dtype=X.dtype)
        b = (A @ delta[:, :, None])[:, :, 0] / Q[:, None]                        # D,M
        LH = torch.linalg.cholesky(H)
        logdet = 2.0 * torch.log(torch.diagonal(LH, dim1=1, dim2=2)).sum(-1)     # D
        y = torch.linalg.solve_triangular(LH, b[:, :, None], upper=False)[:, :, 0]
        quad = (y * y).sum(-1)
        out["later_term1"] = (0.5 * logdet).sum() / T
        out["later_term2"] = (-0.5 * quad).sum() / T
        out["nll_reg_trace_inverse_Q_B"] = (0.5 * ((kdiag - (A * A).sum(1)) / Q[:, None]).sum()) / T
        out["x_t_prior_Q"] = -((-0.5 * delta ** 2 / Q[:, None]).sum() - 0.5 * T * torch.log(Q).sum()) / T
        out["nll_part_prior"] = -(p_hyper + p_Z + p_x0 + hyp) / T
        out["nll"] = (out["nll_part_prior"] + out["nll_log_likelihood"] + out["x_t_prior_Q"]
                      + out["nll_reg_trace_inverse_Q_B"] + out["later_term1"] + out["later_term2"])
    else:
        U = params["U"]
        fvar = kdiag - (A * A).sum(1)                                            # D,T
        fmean = (A * U.T[:, :, None]).sum(1)                                     # D,T
        out["nll_reg_trace_inverse_Q_B"] = (0.5 * (fvar / Q[:, None]).sum()) / T
        r = (delta - fmean) / torch.sqrt(Q)[:, None]
        out["x_t_prior_Q"] = -((-0.5 * r ** 2).sum() - 0.5 * T * torch.log(Q).sum()) / T
        out["nll_part_prior"] = -(-0.5 * (U ** 2).sum() + p_hyper + p_Z + p_x0 + hyp) / T
        out["nll"] = (out["nll_part_prior"] + out["nll_log_likelihood"] + out["x_t_prior_Q"]
                      + out["nll_reg_trace_inverse_Q_B"])
    return out


def to_torch(params, Y, control_inputs, requires_grad=()):
    tp = {}
    for k, v in params.items():
        t = torch.as_tensor(v, dtype=torch.float64).clone()
        if k in requires_grad:
            t.requires_grad_(True)
        tp[k] = t
    tY = torch.as_tensor(Y, dtype=torch.float64)
    tc = None if control_inputs is None else torch.as_tensor(control_inputs, dtype=torch.float64)
    return tp, tY, tc


def nll_and_grad(params, Y, control_inputs, wrt, **kw):
    """nll plus d nll / d params[k] for k in wrt (single chain: params['X'] is (T+1, D))."""
    tp, tY, tc = to_torch(params, Y, control_inputs, requires_grad=wrt)
    out = nll_terms(tp, tY, tc, **kw)
    grads = torch.autograd.grad(out["nll"], [tp[k] for k in wrt])
    return {k: float(v) for k, v in out.items()}, {k: g.numpy() for k, g in zip(wrt, grads)}
