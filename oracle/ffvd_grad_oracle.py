"""Closed-form reverse-mode gradient of `nll` (collapsed-U branch, SE kernel) -- TEST INFRASTRUCTURE ONLY.

The reference obtains d nll / d variables from TensorFlow autodiff (`tf.gradients(nll, vars)`,
base_model.py:148; `AdamOptimizer.minimize(nll)`, dgp_model.py:303-305).  This module restates that gradient in
closed form (re-derived in the K_uu + K_uf K_fu / Q form) in NumPy, so that the HIP
backward kernels have an op-by-op CPU twin; `tests/test_oracle.py` checks it against torch autograd of the
independent restatement (`oracle/ffvd_oracle_torch.py`).  Parity status: unpinned (see ffvd_oracle.py).

Notation per latent dim d (index dropped):  alpha = 1/Q,  K = K_uu + jitter I,  Kf = K_fu (T x M),
G = Kf^T Kf,  g = Kf^T delta,  A = K + alpha G,  c = alpha g,  u = A^-1 c,
l = -1/2 (log|A| - log|K|) + 1/2 c^T u - 1/2 alpha (T sigma^2 - tr(K^-1 G)),   nll contribution = -l / T.

  dl/dG     = Gamma = 1/2 alpha (K^-1 - A^-1 - u u^T)        (evaluated as W N W^T in whitened variables, see nll_grad)
  dl/dg     = alpha u
  dl/dK     = Psi   = 1/2 (K^-1 - A^-1 - u u^T) - 1/2 alpha K^-1 G K^-1
  dl/dalpha = -1/2 tr(A^-1 G) + u^T g - 1/2 u^T G u - 1/2 (T sigma^2 - tr(K^-1 G))
  dl/dKf    = 2 Kf Gamma + delta (alpha u)^T ,   dl/ddelta = alpha Kf u
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import cho_factor, cho_solve, solve_triangular

from . import ffvd_oracle as orc


def _se_chain(E, X1, X2, ell, same):
    """Chain E = dL/dK o K through K(x, z) = s2 exp(-1/2 sum_p ((x_p - z_p)/ell_p)^2).
    Returns (dX1, dX2, dlogell, dlogs2).  If `same`, X1 is X2 (K(Z, Z)) and dX1 already holds both roles."""
    inv2 = 1.0 / ell ** 2
    r = E.sum(axis=1)                       # row sums
    cs = E.sum(axis=0)                      # column sums
    dX1 = -(X1 * r[:, None] - E @ X2) * inv2[None, :]
    dX2 = (E.T @ X1 - X2 * cs[:, None]) * inv2[None, :]
    dlogell = ((r[:, None] * X1 ** 2).sum(0) - 2.0 * np.einsum("tm,tp,mp->p", E, X1, X2) + (cs[:, None] * X2 ** 2).sum(0)) * inv2
    dlogs2 = E.sum()
    if same:
        return dX1 + dX2, None, dlogell, dlogs2
    return dX1, dX2, dlogell, dlogs2


def nll_grad(params, Y, control_inputs, jitter=orc.JITTER_MULTI_OUTPUT, prior_type="normal"):
    """Gradient of the single-chain nll (dgp_model.py:288, collapsed branch, SE kernel, full batch).

    Returns dict with the keys of `params` (X, Z, logvariance, loglengthscales, log_Q, CC, DD, log_Rchols)."""
    X, Z = params["X"], params["Z"]
    T, D = X.shape[0] - 1, X.shape[1]
    M, P = Z.shape
    c_in = control_inputs[:T] if control_inputs is not None and control_inputs.shape[0] > 0 else np.zeros((T, 0))
    xc = np.concatenate((X[:-1], c_in), axis=1)
    Q = np.exp(params["log_Q"])
    R = np.exp(params["log_Rchols"])[0]
    CC, DD = params["CC"], params["DD"]
    g = {k: np.zeros_like(np.asarray(v, dtype=np.float64)) for k, v in params.items() if k != "U"}

    # ---- likelihood (dgp_model.py:248-250,264) -------------------------------------------------
    r = (Y - (X[1:] @ CC + DD)) / R[None, :]                  # T x Ydim
    g["X"][1:] += -(r / R[None, :]) @ CC.T / T
    g["CC"] += -(X[1:].T @ (r / R[None, :])) / T
    g["DD"] += -(r / R[None, :]).sum(0) / T
    g["log_Rchols"][0] += -((r ** 2).sum(0) - T) / T
    # ---- transition prior with Q (dgp_model.py:283-284) ------------------------------------------
    delta = X[1:] - X[:-1]                                    # T x D
    g["X"][1:] += delta / Q[None, :] / T
    g["X"][:-1] -= delta / Q[None, :] / T
    g["log_Q"] += (0.5 * T - 0.5 * (delta ** 2).sum(0) / Q) / T
    # ---- priors (dgp_model.py:105-130,252,286,326-334) -------------------------------------------
    g["loglengthscales"] += params["loglengthscales"] / T
    g["logvariance"] += (params["logvariance"] - orc.LOG_PRIOR_VARIANCE_SE) / T
    if prior_type == "normal":
        g["Z"] += Z / T
    g["X"][0] += X[0] / T
    g["log_Q"] += params["log_Q"] / T
    g["CC"] += CC / T
    g["DD"] += DD / T
    g["log_Rchols"] += params["log_Rchols"] / T
    # ---- collapsed GP terms (conditionals_multi_output.py:230-257) -------------------------------
    for d in range(D):
        ell = np.exp(params["loglengthscales"][d])
        s2 = np.exp(params["logvariance"][d])
        kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
        alpha = 1.0 / Q[d]
        Kuu = kern.K(Z)
        K = Kuu + jitter * np.eye(M)
        Kf = kern.K(xc, Z)
        G = Kf.T @ Kf
        gv = Kf.T @ delta[:, d]
        A = K + alpha * G
        # The closed form above, evaluated in whitened variables: with K = L L^T, W = L^-T and H = W^T A W (= I + alpha
        # W^T G W, condition ~1e4 where A and K have ~1e7),  K^-1 - A^-1 = W (I - H^-1) W^T.  Forming that difference from
        # two explicit inverses loses eps * cond(K) * |K^-1| in the near-null directions of K_uu and costs dZ three
        # digits at M = 512 (central differences of the nll arbitrate: tools/grad_check_full.py); this order does not.
        L = np.linalg.cholesky(K)
        W = solve_triangular(L, np.eye(M), lower=True).T
        H = W.T @ A @ W
        H = 0.5 * (H + H.T)
        cH = cho_factor(H, lower=True)
        w = cho_solve(cH, alpha * (W.T @ gv))
        Hinv = cho_solve(cH, np.eye(M))
        u = W @ w                                               # = A^-1 c
        Nw = np.eye(M) - Hinv - np.outer(w, w)
        Gam = 0.5 * alpha * (W @ Nw @ W.T)                      # = 1/2 alpha (K^-1 - A^-1 - u u^T)
        Psi = 0.5 * (W @ (Nw - (H - np.eye(M))) @ W.T)          # alpha K^-1 G K^-1 = W (H - I) W^T
        trAinvG = (M - np.trace(Hinv)) / alpha                  # tr(A^-1 G),  G = (A - K) / alpha
        trKinvG = (np.trace(H) - M) / alpha                     # tr(K^-1 G) = tr(W^T G W)
        dalpha = (-0.5 * trAinvG + u @ gv - 0.5 * (w @ (H - np.eye(M)) @ w) / alpha - 0.5 * (T * s2 - trKinvG))
        dKf = 2.0 * Kf @ Gam + np.outer(delta[:, d], alpha * u)
        ddelta = alpha * (Kf @ u)
        # chain rule through the kernel matrices; everything below is d l, the nll gets -1/T of it
        dxc, dZ1, dll1, dls1 = _se_chain(dKf * Kf, xc, Z, ell, same=False)
        dZ2, _, dll2, dls2 = _se_chain(Psi * Kuu, Z, Z, ell, same=True)
        dls = dls1 + dls2 - 0.5 * alpha * T * s2              # Kdiag = sigma^2 enters the trace term directly
        g["X"][:-1, :] += -dxc[:, :D] / T
        g["X"][1:, d] += -ddelta / T
        g["X"][:-1, d] -= -ddelta / T
        g["Z"] += -(dZ1 + dZ2) / T
        g["loglengthscales"][d] += -(dll1 + dll2) / T
        g["logvariance"][d] += -dls / T
        g["log_Q"][d] += -(dalpha * (-alpha)) / T
    return g


def nll_grad_explicit_u(params, Y, control_inputs, jitter=orc.JITTER_MULTI_OUTPUT, prior_type="normal"):
    """Gradient of the single-chain nll of the EXPLICIT-U branch (dgp_model.py:289-297 with regularizer :337-359 and
    conditional / base_conditional, conditionals_multi_output.py:6-120), SE kernels, full batch -- what
    tf.gradients(nll, vars) returns in the reference's cases 1, 2, 3, 6.  Returns the keys of nll_grad plus 'U'.

    Per latent dim (index dropped):  alpha = 1/Q,  K = K_uu + jitter I = L L^T,  W = L^-T,  F = K_fu W,
    mean = F u,  var_t = sigma^2 - |F_t|^2,  r = delta - mean,
        l = sum_t [ -1/2 alpha r_t^2 - 1/2 alpha var_t ] + T/2 log alpha,      nll contribution = -l / T.
    With beta = W u,  g_r = K_uf r,  G = K_uf K_fu:
        dl/dK_fu = alpha (r beta^T + K_fu K^-1)            (same shape as the collapsed branch: Gamma -> alpha K^-1 / 2)
        dl/dW    = alpha (g_r u^T + G W),   dl/du = alpha W^T g_r,   dl/ddelta = -alpha r
        dl/dL    = -tril(W (dl/dW)^T W)                    (W = L^-T)
        dl/dK    = W Phi W^T,  Phi = sym(tril(L^T dl/dL) with its diagonal halved)     (Cholesky adjoint)
        dl/dalpha = -1/2 sum r^2 - 1/2 sum var + T / (2 alpha)."""
    X, Z, U = params["X"], params["Z"], params["U"]
    T, D = X.shape[0] - 1, X.shape[1]
    M, P = Z.shape
    c_in = control_inputs[:T] if control_inputs is not None and control_inputs.shape[0] > 0 else np.zeros((T, 0))
    xc = np.concatenate((X[:-1], c_in), axis=1)
    Q = np.exp(params["log_Q"])
    R = np.exp(params["log_Rchols"])[0]
    CC, DD = params["CC"], params["DD"]
    g = {k: np.zeros_like(np.asarray(v, dtype=np.float64)) for k, v in params.items()}
    # likelihood and priors: identical to the collapsed branch, except that the transition prior lives in l below
    rl = (Y - (X[1:] @ CC + DD)) / R[None, :]
    g["X"][1:] += -(rl / R[None, :]) @ CC.T / T
    g["CC"] += -(X[1:].T @ (rl / R[None, :])) / T
    g["DD"] += -(rl / R[None, :]).sum(0) / T
    g["log_Rchols"][0] += -((rl ** 2).sum(0) - T) / T
    g["loglengthscales"] += params["loglengthscales"] / T
    g["logvariance"] += (params["logvariance"] - orc.LOG_PRIOR_VARIANCE_SE) / T
    if prior_type == "normal":
        g["Z"] += Z / T
    g["X"][0] += X[0] / T
    g["log_Q"] += params["log_Q"] / T
    g["CC"] += CC / T
    g["DD"] += DD / T
    g["log_Rchols"] += params["log_Rchols"] / T
    g["U"] += U / T                                              # prior_U, choice 1 (dgp_model.py:134-135)
    delta = X[1:] - X[:-1]
    for d in range(D):
        ell = np.exp(params["loglengthscales"][d])
        s2 = np.exp(params["logvariance"][d])
        kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
        alpha = 1.0 / Q[d]
        Kuu = kern.K(Z)
        K = Kuu + jitter * np.eye(M)
        L = np.linalg.cholesky(K)
        W = np.linalg.inv(L).T
        Kf = kern.K(xc, Z)
        F = Kf @ W
        u = U[:, d]
        beta = W @ u
        mean = Kf @ beta
        r = delta[:, d] - mean
        var = s2 - (F ** 2).sum(1)
        gr = Kf.T @ r
        G = Kf.T @ Kf
        Kinv = W @ W.T
        dKf = alpha * (np.outer(r, beta) + Kf @ Kinv)
        dW = alpha * (np.outer(gr, u) + G @ W)
        dL = -np.tril(W @ dW.T @ W)
        S = L.T @ dL
        Phi = np.tril(S)
        Phi[np.diag_indices(M)] *= 0.5
        Phi = 0.5 * (Phi + Phi.T)
        dK = W @ Phi @ W.T
        du = alpha * (W.T @ gr)
        dalpha = -0.5 * np.sum(r ** 2) - 0.5 * np.sum(var) + T / (2.0 * alpha)
        dxc, dZ1, dll1, dls1 = _se_chain(dKf * Kf, xc, Z, ell, same=False)
        dZ2, _, dll2, dls2 = _se_chain(dK * Kuu, Z, Z, ell, same=True)
        dls = dls1 + dls2 - 0.5 * alpha * T * s2
        g["X"][:-1, :] += -dxc[:, :D] / T
        g["X"][1:, d] += alpha * r / T                       # d(-l/T)/d delta_t = alpha r_t / T, delta_t = x_{t+1} - x_t
        g["X"][:-1, d] -= alpha * r / T
        g["Z"] += -(dZ1 + dZ2) / T
        g["loglengthscales"][d] += -(dll1 + dll2) / T
        g["logvariance"][d] += -dls / T
        g["log_Q"][d] += -(dalpha * (-alpha)) / T
        g["U"][:, d] += -du / T
    return g
