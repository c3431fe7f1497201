"""Randomised shapes through the dataflow Cholesky (operator call with >= 32 matrices) and the ELBO of both branches / routes
against the oracle; FFVD_CHOL=flow forces the one-launch variant everywhere (tools helper, run on the GPU box)."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from ffvd_amd import _lib, synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_oracle as orc
lib = _lib.load()
rng = np.random.default_rng(123)
# 1. Cholesky op, dataflow variant (batch >= 32), random sizes
worst = 0.0
for it in range(12):
    n = int(rng.integers(1, 700)); batch = int(rng.integers(32, 90))
    B = rng.standard_normal((batch, n, n + 3))
    A = B @ np.swapaxes(B, 1, 2) + 0.5 * np.eye(n)
    L = np.empty_like(A); info = np.zeros(batch, dtype=np.int32)
    rc = lib.ffvd_op_cholesky(_lib.dptr(A), n, batch, _lib.dptr(L), info.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int32)))
    assert rc == 0 and not info.any(), (n, batch, rc)
    ref = np.linalg.cholesky(A)
    err = np.max(np.abs(L - ref)) / np.max(np.abs(ref))
    worst = max(worst, err)
print("cholesky worst rel err", worst)
assert worst < 1e-9
# 2. ELBO on random shapes, both branches/routes, against the oracle (FFVD_CHOL env decides the variant)
w2 = 0.0
for it in range(10):
    T_ = int(rng.integers(40, 400))
    ov = dict(T=T_, M=int(rng.integers(10, min(330, T_))), D=int(rng.integers(1, 5)), C=int(rng.integers(0, 3)), S=int(rng.integers(1, 5)))
    params, Y, c, meta = synthetic.make_named("tiny", **ov)
    for collapse, route in ((True, "gram"), (True, "reference"), (False, "reference")):
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], U_collapse=collapse, route=route) as e:
            e.set_data(Y, c)
            got = e.nll_terms(params)
        ref = orc.nll_terms_chains(params, Y, c, U_collapse=collapse)
        err = abs(got["nll"] - ref["nll"]) / max(1.0, abs(ref["nll"]))
        w2 = max(w2, err)
        assert err < 2e-7, (ov, collapse, route, got["nll"], ref["nll"])
print("elbo worst rel err", w2)
# 3. (round 3) gradient of the collapsed branch on random shapes, Gram route (L^T rows, training forward) and reference route,
#    against the closed-form oracle; LinearK explicit-U forward through its rank against the oracle
from oracle import ffvd_grad_oracle as gorc
KEYS = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")
w3 = 0.0
for it in range(6):
    T_ = int(rng.integers(60, 300))
    ov = dict(T=T_, M=int(rng.integers(10, min(200, T_))), D=int(rng.integers(1, 5)), C=int(rng.integers(0, 3)), S=int(rng.integers(1, 4)))
    params, Y, c, meta = synthetic.make_named("tiny", **ov)
    S = meta["S"]
    ref = {k: np.zeros_like(np.asarray(params[k], dtype=np.float64)) for k in KEYS}
    for s in range(S):
        p = dict(params); p["X"] = params["X"][s]
        ga = gorc.nll_grad(p, Y, c)
        ref["X"][s] = ga["X"] / S
        for k in KEYS[1:]: ref[k] += ga[k] / S
    for route in ("gram", "reference"):
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route=route, grad=True) as e:
            e.set_data(Y, c)
            _, g = e.nll_and_grad(params)
        for k in KEYS:
            err = np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)
            w3 = max(w3, err)
            assert err < 2e-6, (ov, route, k, err)
print("gradient worst rel err", w3)
w4 = 0.0
for it in range(6):
    T_ = int(rng.integers(60, 300))
    ov = dict(T=T_, M=int(rng.integers(10, min(200, T_))), D=int(rng.integers(1, 7)), C=int(rng.integers(0, 3)), S=int(rng.integers(1, 4)))
    params, Y, c, meta = synthetic.make_named("small_lin", **ov)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], U_collapse=False, kernel_type="LinearK") as e:
        e.set_data(Y, c)
        got = e.nll_terms(params)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=False, kernel_type="LinearK")
    err = abs(got["nll"] - ref["nll"]) / max(1.0, abs(ref["nll"]))
    w4 = max(w4, err)
    assert err < 1e-7, (ov, got["nll"], ref["nll"])
print("LinearK through its rank, worst rel err", w4)
