"""Phase stamps of one step of the rollout loop with resident operands (FFVD_RR_STAMPS=1 makes the library print them).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FFVD_RR_STAMPS"] = "1"; os.environ["FFVD_STEP_LOOP"] = "2"
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
steps = 40
for R in (16, 32, 64):
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
    eps = rng.standard_normal((steps, R, D))
    for q in (True, False):
        rollout(L, params["Z"], kern, U, H if q else None, X[-1], ctrl, T, steps, np.exp(params["log_Q"]), eps)
