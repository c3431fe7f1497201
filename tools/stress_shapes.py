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
