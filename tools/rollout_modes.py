"""Rollout loop at config 2's shapes (M = 512, D = 4, q_sqrt included): microseconds per step for the per-step launches (FFVD_STEP_LOOP=0),
the role pipelines (1) and the loop with resident operands (2, the default where it applies), and the largest difference of the results
against the launches.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
steps = int(os.environ.get("STEPS", "400"))
for R in tuple(int(x) for x in os.environ.get("RS", "16,32,64,100,128").split(",")):
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
    eps = rng.standard_normal((steps, R, D))
    ref = None
    for q in (True, False):
        for mode in ("0", "1", "2"):
            os.environ["FFVD_STEP_LOOP"] = mode
            rollout(L, params["Z"], kern, U, H if q else None, X[-1], ctrl, T, 2, np.exp(params["log_Q"]), eps[:2])
            t0 = time.perf_counter()
            rollout(L, params["Z"], kern, U, H if q else None, X[-1], ctrl, T, steps // 4, np.exp(params["log_Q"]), eps[:steps // 4])
            dq = time.perf_counter() - t0
            t0 = time.perf_counter()
            px, pv = rollout(L, params["Z"], kern, U, H if q else None, X[-1], ctrl, T, steps, np.exp(params["log_Q"]), eps)
            dt = time.perf_counter() - t0
            if mode == "0": ref = (px, pv)
            # (a call also uploads L^-T and forms W q_sqrt: the step cost is the slope between two step counts)
            print("R=%d q_sqrt=%d mode=%s: %.1f us per step (slope between %d and %d steps; whole call %.1f ms)  max |dx| vs launches %.2e, |dvar| %.2e" %
                  (R, q, mode, (dt - dq) / (steps - steps // 4) * 1e6, steps // 4, steps, dt * 1e3, np.abs(px - ref[0]).max(), np.abs(pv - ref[1]).max()), flush=True)
