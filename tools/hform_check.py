"""CPU experiment behind the one-launch iteration's forward: the three collapsed terms of every latent dim from
H = I + F^T F / Q with F = K_fu W (the reference's op order, conditionals_multi_output.py:243-254) against H = I + W^T (K_uf K_fu) W / Q
(the raw Gram matrix first, whitened afterwards -- what lets the strips work while the head still factorises K_uu), both in fp64,
against the reference order evaluated in longdouble."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ffvd_oracle as orc
from ffvd_amd import synthetic


def chol(A):
    A = A.copy(); n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        v = A[j, j] - L[j, :j] @ L[j, :j]
        L[j, j] = np.sqrt(v)
        L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def lsolve(L, B):
    X = np.zeros_like(B)
    for i in range(L.shape[0]):
        X[i] = (B[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X


def terms(Kuu_j, Kf, delta, Q, dt, form):
    Kuu_j, Kf, delta = Kuu_j.astype(dt), Kf.astype(dt), delta.astype(dt)
    M = Kuu_j.shape[0]
    L = chol(Kuu_j)
    W = lsolve(L, np.eye(M, dtype=dt)).T                 # L^-T
    if form == "F":
        F = Kf @ W
        H = np.eye(M, dtype=dt) + (F.T @ F) / dt(Q)
        b = (F.T @ delta) / dt(Q)
    else:
        G = Kf.T @ Kf
        H = np.eye(M, dtype=dt) + (W.T @ (G @ W)) / dt(Q)
        b = (W.T @ (Kf.T @ delta)) / dt(Q)
    LH = chol(H)
    y = lsolve(LH, b[:, None])[:, 0]
    return np.array([2 * np.log(np.diag(LH)).sum(), y @ y, np.trace(H) - M], dtype=dt)


def run(name, params, Y, c):
    X = params["X"] if params["X"].ndim == 2 else params["X"][0]
    T = X.shape[0] - 1
    xc = np.concatenate((X[:-1], c[:T]), axis=1)
    worst = {"F": 0.0, "G": 0.0}
    for d in range(X.shape[1]):
        kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
        Z = params["Z"]
        Kj = kern.K(Z) + orc.JITTER_MULTI_OUTPUT * np.eye(Z.shape[0])
        Kf = kern.K(xc, Z)
        delta = X[1:, d] - X[:-1, d]
        Q = np.exp(params["log_Q"][d])
        ref = terms(Kj, Kf, delta, Q, np.longdouble, "F")
        for form in ("F", "G"):
            got = terms(Kj, Kf, delta, Q, np.float64, form)
            rel = np.abs((got - ref) / ref).astype(float)
            worst[form] = max(worst[form], rel.max())
            print(f"{name} dim {d} H from {form}: rel. error of logdet / quad / trace = " + " ".join("%.2e" % r for r in rel), flush=True)
    print(f"{name}: worst H-from-F {worst['F']:.2e}, worst H-from-G {worst['G']:.2e}")


if __name__ == "__main__":
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "actuator_slim.npz"))
    params = {k: z[k] for k in ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")}
    run("actuator", params, z["Y"], z["control_inputs"])
    for nm in ("small", "ragged"):
        p, Y, c, meta = synthetic.make_named(nm)
        run(nm, p, Y, c)
