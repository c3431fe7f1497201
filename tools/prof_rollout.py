"""Rollout loop alone (for rocprofv3): 32 rollouts x 200 steps at the config-2 shapes, q_sqrt included (tools helper)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd import conditionals_multi_output as cmo
from ffvd_amd.kernels_multi_output import SquaredExponential
from ffvd_amd.prediction import rollout
params, Y, c, meta = synthetic.make_named("c2", S=1)
D, M, C, T = meta["D"], meta["M"], meta["C"], meta["T"]
kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d]))
        for d in range(D)]
X = params["X"][0]
L = cmo.kernel_pre_cal(params["Z"], kern)
U, H = cmo.collapse_u_mean_after_kernel_precalculation(L, np.concatenate((X[:-1], c), axis=1), X, params["Z"], kern,
                                                       np.exp(params["log_Q"]))
rng = np.random.default_rng(0)
R, steps = 32, 200
ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
eps = rng.standard_normal((steps, R, D))
for _ in range(2):
    px, pv = rollout(L, params["Z"], kern, U, H, X[-1], ctrl, T, steps, np.exp(params["log_Q"]), eps)
print("ok", px.shape)
