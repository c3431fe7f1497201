"""Soak of the rollout loop with resident operands: the same call many times, every result bit-identical to the first (its hand-offs have
no fences: a stale read would show up as a different trajectory).  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FFVD_STEP_LOOP"] = "2"
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd import conditionals_multi_output as cmo
from ffvd_amd.kernels_multi_output import SquaredExponential
from ffvd_amd.prediction import rollout
params, Y, c, meta = synthetic.make_named("c2", S=1)
D, M, C, T = meta["D"], meta["M"], meta["C"], meta["T"]
kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
X = params["X"][0]
L = cmo.kernel_pre_cal(params["Z"], kern)
U, H = cmo.collapse_u_mean_after_kernel_precalculation(L, np.concatenate((X[:-1], c), axis=1), X, params["Z"], kern, np.exp(params["log_Q"]))
rng = np.random.default_rng(0)
steps, n = 300, int(os.environ.get("SOAK_N", "150"))
for R, q in ((32, True), (16, False), (64, True)):
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
    eps = rng.standard_normal((steps, R, D))
    first = rollout(L, params["Z"], kern, U, H if q else None, X[-1], ctrl, T, steps, np.exp(params["log_Q"]), eps)
    os.environ["FFVD_STEP_LOOP"] = "0"
    ref = rollout(L, params["Z"], kern, U, H if q else None, X[-1], ctrl, T, steps, np.exp(params["log_Q"]), eps)
    os.environ["FFVD_STEP_LOOP"] = "2"
    t0 = time.perf_counter()
    for i in range(n):
        got = rollout(L, params["Z"], kern, U, H if q else None, X[-1], ctrl, T, steps, np.exp(params["log_Q"]), eps)
        assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1]), (R, q, i)
    print("R=%d q_sqrt=%d: %d calls of %d steps identical; against the launches max |dx| %.2e; %.2f ms per call" %
          (R, q, n, steps, np.abs(first[0] - ref[0]).max(), (time.perf_counter() - t0) / n * 1e3), flush=True)
