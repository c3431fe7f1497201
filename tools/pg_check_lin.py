import numpy as np, sys
sys.path.insert(0,'.')
from oracle import ffvd_oracle as orc, ffvd_pg_oracle as pgo
from ffvd_amd import synthetic, prediction
from ffvd_amd.kernels import LinearK
params, Y, c, meta = synthetic.make_named("small_lin")
D, C, T = meta["D"], meta["C"], meta["T"]
okern = orc.make_kernels(params, kernel_type=meta["kernel_type"])
Lm = orc.kernel_pre_cal(params["Z"], okern)
kern = [LinearK(D + C, variance=np.exp(params["logvariance"][d])) for d in range(D)]
X = params["X"][0]
rng = np.random.default_rng(3)
N = 7
x0 = rng.standard_normal((N-1, D)); eps = rng.standard_normal((T, N-1, D)); u = rng.random((T, N-1))
R = np.exp(params["log_Rchols"]); Q = np.exp(params["log_Q"])
pr, ir = pgo.pg_sweep(Lm, params["Z"], okern, params["U"], X, Y, c, params["CC"], params["DD"], R, Q, x0, eps, u)
pg, ig = prediction.pg_sweep(Lm, params["Z"], kern, params["U"], X, Y, c, params["CC"], params["DD"], R, Q, x0, eps, u)
print("LinearK idx equal", np.array_equal(ir, ig), "max |dparts|", np.max(np.abs(pr - pg)), "finite", np.isfinite(pg).all())
