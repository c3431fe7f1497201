"""CPU experiment behind the fp32-contraction backward pass: how the fp32 product in dl/dK_fu = 2 K_fu Gamma +
delta (alpha u)^T can be arranged, and what each arrangement costs in accuracy (NumPy float32 matmul as the stand-in for the fp32
matrix cores; errors of the K_fu-side gradients relative to the largest entry).  v0 is what ships (ONE fp32 product K_fu Gamma);
v1 splits Gamma into hi + lo fp32 parts (two products); v2 / v4 / v5 reuse the forward pass's fp32 F = K_fu L^-T, whose own error
(|L^-T| up to 300) then dominates; v3 shows that an F accurate to fp32 rounding would give 1e-5 -- it needs the fp64 product."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.linalg import cho_factor, cho_solve, solve_triangular
from ffvd_amd import synthetic
from oracle import ffvd_oracle as orc
from oracle.ffvd_grad_oracle import _se_chain
for ov in (dict(), dict(T=1000, M=600, D=2, C=1, S=2)):
    params, Y, c, meta = synthetic.make_named("small", **ov)
    X = params["X"][0]; Z = params["Z"]; T, D = X.shape[0]-1, X.shape[1]; M = Z.shape[0]
    xc = np.concatenate((X[:-1], c[:T]), axis=1); delta = X[1:] - X[:-1]; Q = np.exp(params["log_Q"])
    d = 0
    ell = np.exp(params["loglengthscales"][d]); kern = orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d])
    alpha = 1/Q[d]; Kuu = kern.K(Z); K = Kuu + 1e-5*np.eye(M); Kf = kern.K(xc, Z)
    L = np.linalg.cholesky(K); W = solve_triangular(L, np.eye(M), lower=True).T
    F = Kf @ W; H = np.eye(M) + alpha * F.T @ F; cH = cho_factor(H, lower=True)
    w = cho_solve(cH, alpha * (F.T @ delta[:, d])); Hinv = cho_solve(cH, np.eye(M)); u = W @ w
    N = np.eye(M) - Hinv - np.outer(w, w); Gam = 0.5*alpha*(W @ N @ W.T)
    def grads(twoR, Kfq):
        E = (twoR + np.outer(delta[:, d], alpha*u)) * Kfq
        dxc, dZ1, dll, dls = _se_chain(E, xc, Z, ell, same=False)
        return dxc, dZ1, dll, dls
    ref = grads(2*Kf@Gam, Kf)
    f32 = np.float32
    Kf32 = Kf.astype(f32); Gam32 = Gam.astype(f32)
    v0 = grads(2*(Kf32 @ Gam32).astype(np.float64), Kf32.astype(np.float64))
    Glo = (Gam - Gam32.astype(np.float64)).astype(f32)
    v1 = grads(2*((Kf32 @ Gam32).astype(np.float64) + (Kf32 @ Glo).astype(np.float64)), Kf32.astype(np.float64))
    F32 = (Kf32 @ W.astype(f32))
    B2 = (alpha * N @ W.T)
    v2 = grads((F32 @ B2.astype(f32)).astype(np.float64), Kf32.astype(np.float64))
    Fx = F.astype(f32)
    v3 = grads((Fx @ B2.astype(f32)).astype(np.float64), Kf32.astype(np.float64))
    B2lo = (B2 - B2.astype(f32).astype(np.float64)).astype(f32)
    v4 = grads((F32 @ B2.astype(f32)).astype(np.float64) + (F32 @ B2lo).astype(np.float64), Kf32.astype(np.float64))
    # v5: two-stage in fp32: P = F32 @ (alpha N) (well-conditioned), then R = P @ W^T
    P32 = (F32 @ (alpha*N).astype(f32))
    v5 = grads((P32 @ W.T.astype(f32)).astype(np.float64), Kf32.astype(np.float64))
    def err(v):
        return ["%.1e" % (np.max(np.abs(a-b))/np.max(np.abs(b))) for a, b in zip(v, ref)]
    print(ov, "|Gam|max %.1e |W|max %.1e |2KfGam|max %.1e" % (np.abs(Gam).max(), np.abs(W).max(), np.abs(2*Kf@Gam).max()))
    for name, v in (("v0 Kf32@Gam32", v0), ("v1 Gam hi+lo", v1), ("v2 F32@B2", v2), ("v3 exactF@B2", v3), ("v4 F32@B2 hi+lo", v4), ("v5 two-stage", v5)):
        print("  ", name, "dxc dZ dlogell dlogs2:", err(v))
