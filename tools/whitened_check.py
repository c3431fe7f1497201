"""CPU experiment behind DESIGN.md section 7: dZ at T=4096, M=512 (two chains) from four ways of forming Gamma / Psi,
against central differences of the nll measured on the GPU (tools/grad_check_full.py printed them; hard-coded below).
explicit inverses lose 3 digits; every whitened variant (H from F, from G or from A; Gamma as W N W^T or as K^-1 - B^T B)
reaches the accuracy of the finite differences themselves."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.linalg import solve_triangular, cho_factor, cho_solve
from ffvd_amd import synthetic
from oracle import ffvd_oracle as orc
from oracle.ffvd_grad_oracle import _se_chain
params, Y, c, meta = synthetic.make_named("c2", S=2)
S=2
T, D, M = meta["T"], meta["D"], meta["M"]
Z = params["Z"]; Q = np.exp(params["log_Q"]); jitter = orc.JITTER_MULTI_OUTPUT
def dz_variants(X):
    xc = np.concatenate((X[:-1], c[:T]), axis=1)
    delta = X[1:] - X[:-1]
    out = {"explicit": np.zeros_like(Z), "whitened_gamma": np.zeros_like(Z), "whitened_ordered": np.zeros_like(Z)}
    for d in range(D):
        ell = np.exp(params["loglengthscales"][d])
        kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
        alpha = 1.0 / Q[d]
        Kuu = kern.K(Z); K = Kuu + jitter * np.eye(M); Kf = kern.K(xc, Z)
        G = Kf.T @ Kf; gv = Kf.T @ delta[:, d]
        # explicit-inverse closed form (what the GPU and the oracle do)
        A = K + alpha * G
        Kinv = np.linalg.inv(K); Ainv = np.linalg.inv(A)
        u = Ainv @ (alpha * gv)
        Gam = 0.5 * alpha * (Kinv - Ainv - np.outer(u, u))
        Psi = 0.5 * (Kinv - Ainv - np.outer(u, u)) - 0.5 * alpha * (Kinv @ G @ Kinv)
        dKf = 2.0 * Kf @ Gam + np.outer(delta[:, d], alpha * u)
        _, dZ1, _, _ = _se_chain(dKf * Kf, xc, Z, ell, same=False)
        dZ2, _, _, _ = _se_chain(Psi * Kuu, Z, Z, ell, same=True)
        out["explicit"] += dZ1 + dZ2
        # whitened: W = L^-T, F = Kf W (triangular solve), H = I + alpha F^T F (well conditioned)
        L = np.linalg.cholesky(K)
        F = solve_triangular(L, Kf.T, lower=True).T            # T x M
        H = np.eye(M) + alpha * (F.T @ F)
        b = alpha * (F.T @ delta[:, d])
        cH = cho_factor(H)
        w = cho_solve(cH, b)
        Hinv = cho_solve(cH, np.eye(M))
        N = np.eye(M) - Hinv - np.outer(w, w)
        W = solve_triangular(L, np.eye(M), lower=True).T       # L^-T
        # (a) Gamma / Psi formed explicitly from the whitened pieces, contractions as before
        Gam_w = 0.5 * alpha * (W @ N @ W.T)
        Psi_w = 0.5 * (W @ (2 * np.eye(M) - H - Hinv - np.outer(w, w)) @ W.T)
        uw = W @ w
        dKf_a = 2.0 * Kf @ Gam_w + np.outer(delta[:, d], alpha * uw)
        _, dZ1a, _, _ = _se_chain(dKf_a * Kf, xc, Z, ell, same=False)
        dZ2a, _, _, _ = _se_chain(Psi_w * Kuu, Z, Z, ell, same=True)
        out["whitened_gamma"] += dZ1a + dZ2a
        # (b) products ordered through F: (F N) W^T
        dKf_b = alpha * ((F @ N) @ W.T) + np.outer(delta[:, d], alpha * uw)
        _, dZ1b, _, _ = _se_chain(dKf_b * Kf, xc, Z, ell, same=False)
        out["whitened_ordered"] += dZ1b + dZ2a
    return {k: -v / T for k, v in out.items()}
acc = None
for s in range(S):
    r = dz_variants(params["X"][s])
    acc = r if acc is None else {k: acc[k] + r[k] for k in r}
dz = {k: v / S + Z / T for k, v in acc.items()}
fd = {(287,4): -2.279856e-04, (303,2): -3.154887e-04, (493,0): -1.903354e-04, (236,3): 1.061184e-04, (329,4): -6.528010e-05, (271,4): 4.992696e-05}
for k, v in dz.items():
    print(k, " ".join("%+.2e" % (v[i] - f) for i, f in fd.items()))

def dz_from_G(X, mode):
    xc = np.concatenate((X[:-1], c[:T]), axis=1)
    delta = X[1:] - X[:-1]
    out = np.zeros_like(Z)
    for d in range(D):
        ell = np.exp(params["loglengthscales"][d])
        kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
        alpha = 1.0 / Q[d]
        Kuu = kern.K(Z); K = Kuu + jitter * np.eye(M); Kf = kern.K(xc, Z)
        G = Kf.T @ Kf; gv = Kf.T @ delta[:, d]
        L = np.linalg.cholesky(K)
        W = solve_triangular(L, np.eye(M), lower=True).T
        if mode == "G":
            H = np.eye(M) + alpha * (W.T @ G @ W)
        else:
            A = K + alpha * G
            H = W.T @ A @ W
        H = 0.5 * (H + H.T)
        b = alpha * (W.T @ gv)
        cH = cho_factor(H)
        w = cho_solve(cH, b)
        Hinv = cho_solve(cH, np.eye(M))
        N = np.eye(M) - Hinv - np.outer(w, w)
        Gam_w = 0.5 * alpha * (W @ N @ W.T)
        Psi_w = 0.5 * (W @ (2 * np.eye(M) - H - Hinv - np.outer(w, w)) @ W.T)
        uw = W @ w
        dKf_a = 2.0 * Kf @ Gam_w + np.outer(delta[:, d], alpha * uw)
        _, dZ1a, _, _ = _se_chain(dKf_a * Kf, xc, Z, ell, same=False)
        dZ2a, _, _, _ = _se_chain(Psi_w * Kuu, Z, Z, ell, same=True)
        out += dZ1a + dZ2a
    return -out / T
for mode in ("G", "A"):
    v = sum(dz_from_G(params["X"][s], mode) for s in range(S)) / S + Z / T
    print("H from", mode, " ".join("%+.2e" % (v[i] - f) for i, f in fd.items()))

def dz_alt(X):
    xc = np.concatenate((X[:-1], c[:T]), axis=1)
    delta = X[1:] - X[:-1]
    out = np.zeros_like(Z)
    for d in range(D):
        ell = np.exp(params["loglengthscales"][d])
        kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
        alpha = 1.0 / Q[d]
        Kuu = kern.K(Z); K = Kuu + jitter * np.eye(M); Kf = kern.K(xc, Z)
        G = Kf.T @ Kf; gv = Kf.T @ delta[:, d]
        L = np.linalg.cholesky(K)
        Linv = solve_triangular(L, np.eye(M), lower=True)
        W = Linv.T
        A = K + alpha * G
        H = W.T @ A @ W; H = 0.5 * (H + H.T)
        LH = np.linalg.cholesky(H)
        LHinv = solve_triangular(LH, np.eye(M), lower=True)
        B = LHinv @ Linv                           # lower triangular, A^-1 = B^T B
        w = LHinv.T @ (LHinv @ (alpha * (W.T @ gv)))
        u = W @ w
        Kinv_w = W @ W.T
        Ainv_w = B.T @ B
        Gam = 0.5 * alpha * (Kinv_w - Ainv_w - np.outer(u, u))
        Psi = 0.5 * (Kinv_w - Ainv_w - np.outer(u, u)) - 0.5 * alpha * (W @ (W.T @ G @ W) @ W.T)
        dKf = 2.0 * Kf @ Gam + np.outer(delta[:, d], alpha * u)
        _, dZ1, _, _ = _se_chain(dKf * Kf, xc, Z, ell, same=False)
        dZ2, _, _, _ = _se_chain(Psi * Kuu, Z, Z, ell, same=True)
        out += dZ1 + dZ2
    return -out / T
v = sum(dz_alt(params["X"][s]) for s in range(S)) / S + Z / T
print("K^-1 - B^T B ", " ".join("%+.2e" % (v[i] - f) for i, f in fd.items()))

# Round 3: the same whitened quantities WITHOUT forming H.  L_H = L^-1 L_A (both lower triangular, so the product is H's Cholesky
# factor), hence L_H^-T = L^T L_A^-T: the factorisation of A with L^T in the extension rows (instead of I) leaves L_H^-T there, and
# y = L_A^-1 c equals L_H^-1 W^T c.  No W^T A W products.
def dz_lt(X):
    xc = np.concatenate((X[:-1], c[:T]), axis=1)
    delta = X[1:] - X[:-1]
    out = np.zeros_like(Z)
    for d in range(D):
        ell = np.exp(params["loglengthscales"][d])
        kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
        alpha = 1.0 / Q[d]
        Kuu = kern.K(Z); K = Kuu + jitter * np.eye(M); Kf = kern.K(xc, Z)
        G = Kf.T @ Kf; gv = Kf.T @ delta[:, d]
        L = np.linalg.cholesky(K)
        Linv = solve_triangular(L, np.eye(M), lower=True)
        W = Linv.T
        A = K + alpha * G
        LA = np.linalg.cholesky(A)
        LHinvT = solve_triangular(LA, L, lower=True).T      # rows: L^T L_A^-T (what the extension rows hold)
        y = solve_triangular(LA, alpha * gv, lower=True)
        B = LHinvT.T @ Linv
        w = LHinvT @ y
        u = W @ w
        Kinv_w = W @ W.T
        Ainv_w = B.T @ B
        Gam = 0.5 * alpha * (Kinv_w - Ainv_w - np.outer(u, u))
        Psi = 0.5 * (Kinv_w - Ainv_w - np.outer(u, u)) - 0.5 * alpha * (W @ (W.T @ G @ W) @ W.T)
        dKf = 2.0 * Kf @ Gam + np.outer(delta[:, d], alpha * u)
        _, dZ1, _, _ = _se_chain(dKf * Kf, xc, Z, ell, same=False)
        dZ2, _, _, _ = _se_chain(Psi * Kuu, Z, Z, ell, same=True)
        out += dZ1 + dZ2
    return -out / T
v = sum(dz_lt(params["X"][s]) for s in range(S)) / S + Z / T
print("L^T rows     ", " ".join("%+.2e" % (v[i] - f) for i, f in fd.items()))
