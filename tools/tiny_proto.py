"""NumPy prototype of the one-launch small-problem iteration (ffvd_amd/csrc/tiny.hip): the same decomposition into a head role
per (chain, latent dim) unit and 64-row strip roles, forward and backward, in the reference's op order (F = K_fu L^-T,
H = I + F^T F / Q; conditionals_multi_output.py:230-257) -- checked against the oracle.  CPU-only design aid: tools/ only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.linalg import solve_triangular
from ffvd_amd import synthetic
from oracle import ffvd_oracle as orc, ffvd_grad_oracle as gorc

STRIP = 64


def tiny_iteration(params, Y, c, jitter=1e-5, S_total=None):
    X = params["X"]
    S, T1, D = X.shape
    T = T1 - 1
    Z = params["Z"]; M, P = Z.shape
    S_total = S_total or S
    Q = np.exp(params["log_Q"]); R = np.exp(params["log_Rchols"])[0]
    CC, DD = params["CC"], params["DD"]
    nstrips = (T + STRIP - 1) // STRIP
    terms = np.zeros(7)
    g = {k: np.zeros_like(np.asarray(v, dtype=np.float64)) for k, v in params.items() if k != "U"}
    for s in range(S):
        xs = X[s]
        xc = np.concatenate((xs[:-1], c[:T]), axis=1)
        delta = xs[1:] - xs[:-1]
        lik = xq = tr = 0.0
        t1 = t2 = 0.0
        gX = np.zeros_like(xs)
        for d in range(D):
            ell = np.exp(params["loglengthscales"][d]); s2 = np.exp(params["logvariance"][d]); alpha = 1.0 / Q[d]
            inv2 = 1.0 / ell ** 2
            kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
            # ---- head, phase 0: K_uu, L, W = L^-T ------------------------------------------------
            Kuu = kern.K(Z)
            L = np.linalg.cholesky(Kuu + jitter * np.eye(M))
            W = solve_triangular(L, np.eye(M), lower=True).T
            # ---- strips, phase 1 -------------------------------------------------------------------
            Psum = np.zeros((M, M)); bsum = np.zeros(M); fsq = 0.0
            strips = []
            for i in range(nstrips):
                rows = slice(i * STRIP, min(T, (i + 1) * STRIP))
                Kf = kern.K(xc[rows], Z)
                F = Kf @ W
                Psum += F.T @ F
                bsum += F.T @ delta[rows, d]
                rs = (F ** 2).sum(1)
                tr += np.sum(-0.5 * ((s2 - rs) / Q[d]))
                xq += np.sum(-0.5 * (delta[rows, d] / np.sqrt(Q[d])) ** 2)
                if d == 0:
                    r = (Y[rows] - (xs[1:][rows] @ CC + DD)) / R[None, :]
                    lik += np.sum(-0.5 * r ** 2)
                strips.append((rows, Kf, F))
            # ---- head, phase 1: H, Cholesky, solve --------------------------------------------------
            H = np.eye(M) + alpha * Psum
            b = alpha * bsum
            LH = np.linalg.cholesky(H)
            y = solve_triangular(LH, b, lower=True)
            t1 += -0.5 * 2.0 * np.sum(np.log(np.diag(LH)))
            t2 += 0.5 * (y @ y)
            # backward: H^-1, w, the two M x M matrices the strips need
            LHinv = solve_triangular(LH, np.eye(M), lower=True)
            Hinv = LHinv.T @ LHinv
            w = LHinv.T @ y
            Nw = np.eye(M) - Hinv - np.outer(w, w)
            Nm2 = Nw - (H - np.eye(M))
            # dl/dalpha (whitened): G-terms through H - I = alpha W^T G W
            trAinvG = (M - np.trace(Hinv)) / alpha
            trKinvG = (np.trace(H) - M) / alpha
            gv_w = b / alpha                        # W^T g
            dalpha = -0.5 * trAinvG + w @ gv_w - 0.5 * (w @ (H - np.eye(M)) @ w) / alpha - 0.5 * (T * s2 - trKinvG)
            # ---- strips, phase 2 --------------------------------------------------------------------
            cs = np.zeros(M); etx = np.zeros((M, P)); rx2 = np.zeros(P); esum = 0.0
            for rows, Kf, F in strips:
                Rm = F @ Nw + np.outer(delta[rows, d], w)
                dKf = alpha * (Rm @ W.T)
                E = dKf * Kf
                r = E.sum(1)
                EZ = E @ Z
                cs += E.sum(0)
                etx += E.T @ xc[rows]
                rx2 += (r[:, None] * xc[rows] ** 2).sum(0)
                esum += r.sum()
                dxc = -(xc[rows] * r[:, None] - EZ) * inv2[None, :]
                ddelta = alpha * (F @ w)
                gX[:-1][rows] += -dxc[:, :D] / T
                gX[1:][rows, d] += -ddelta / T
                gX[:-1][rows, d] -= -ddelta / T
            dZ1 = (etx - Z * cs[:, None]) * inv2[None, :]
            dll1 = (rx2 - 2.0 * np.einsum("mp,mp->p", etx, Z) + (cs[:, None] * Z ** 2).sum(0)) * inv2
            # K_uu side, by 16-row blocks of Psi (dealt to the strips' workgroups): E_u = Psi o Kuu, symmetric
            Psi = 0.5 * (W @ Nm2 @ W.T)
            Eu = Psi * Kuu
            ru = Eu.sum(1)
            EuZ = Eu @ Z
            dZ2 = -2.0 * (Z * ru[:, None] - EuZ) * inv2[None, :]
            dll2 = 2.0 * ((ru[:, None] * Z ** 2).sum(0) - np.einsum("mp,mp->p", EuZ, Z)) * inv2
            dls = esum + ru.sum() - 0.5 * alpha * T * s2
            g["Z"] += -(dZ1 + dZ2) / T / S_total
            g["loglengthscales"][d] += -(dll1 + dll2) / T / S_total
            g["logvariance"][d] += -dls / T / S_total
            g["log_Q"][d] += -(dalpha * (-alpha)) / T / S_total
        # ---- per-chain closing terms (likelihood, transition prior, priors) ---------------------------
        r = (Y - (xs[1:] @ CC + DD)) / R[None, :]
        gX[1:] += -(r / R[None, :]) @ CC.T / T
        g["CC"] += -(xs[1:].T @ (r / R[None, :])) / T / S_total
        g["DD"] += -(r / R[None, :]).sum(0) / T / S_total
        g["log_Rchols"][0] += -((r ** 2).sum(0) - T) / T / S_total
        gX[1:] += delta / Q[None, :] / T
        gX[:-1] -= delta / Q[None, :] / T
        g["log_Q"] += (0.5 * T - 0.5 * (delta ** 2).sum(0) / Q) / T / S_total
        gX[0] += xs[0] / T
        g["X"][s] = gX / S_total
        # priors of the shared parameters: once per chain (every chain's nll carries them)
        g["loglengthscales"] += params["loglengthscales"] / T / S_total
        g["logvariance"] += (params["logvariance"] - orc.LOG_PRIOR_VARIANCE_SE) / T / S_total
        g["Z"] += Z / T / S_total
        g["log_Q"] += params["log_Q"] / T / S_total
        g["CC"] += CC / T / S_total
        g["DD"] += DD / T / S_total
        g["log_Rchols"] += params["log_Rchols"] / T / S_total
        # nll terms of the chain
        logR = np.sum(np.log(R)); logsqQ = np.sum(np.log(np.sqrt(Q)))
        prior = (-0.5 * np.sum(params["loglengthscales"] ** 2) - 0.5 * np.sum((params["logvariance"] - orc.LOG_PRIOR_VARIANCE_SE) ** 2)
                 - 0.5 * np.sum(Z ** 2) - 0.5 * np.sum(xs[0] ** 2)
                 - 0.5 * (np.sum(params["log_Q"] ** 2) + np.sum(CC ** 2) + np.sum(DD ** 2) + np.sum(params["log_Rchols"] ** 2)))
        tt = np.array([-prior / T, -(lik + T * (-logR)) / T, -(xq + T * (-logsqQ)) / T, -tr / T, -t1 / T, -t2 / T, 0.0])
        tt[6] = tt[:6].sum()
        terms += tt
    return terms / S, g


if __name__ == "__main__":
    for name in ("tiny", "small", "ragged"):
        params, Y, c, meta = synthetic.make_named(name)
        terms, g = tiny_iteration(params, Y, c)
        ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
        print(name, "nll", terms[6], ref["nll"], abs(terms[6] - ref["nll"]))
        S = meta["S"]
        want = None
        for s in range(S):
            p = dict(params); p["X"] = params["X"][s]
            gs = gorc.nll_grad(p, Y, c)
            if want is None:
                want = {k: np.zeros_like(v) for k, v in gs.items() if k != "X"}
                want["X"] = np.zeros_like(params["X"])
            for k in gs:
                if k == "X":
                    want["X"][s] = gs["X"] / S
                else:
                    want[k] += gs[k] / S
        for k in want:
            err = np.max(np.abs(g[k] - want[k])) / max(1e-300, np.max(np.abs(want[k])))
            print("   d%-16s rel.err %.2e" % (k, err))
