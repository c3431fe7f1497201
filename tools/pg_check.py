import numpy as np, sys, time
sys.path.insert(0,'.')
from oracle import ffvd_oracle as orc, ffvd_pg_oracle as pgo
from ffvd_amd import synthetic, prediction
from ffvd_amd.kernels_multi_output import SquaredExponential
for name, ov in (("tiny", {}), ("small", {}), ("tiny", dict(C=0))):
    params, Y, c, meta = synthetic.make_named(name, **ov)
    D, C = meta["D"], meta["C"]
    okern = orc.make_kernels(params)
    Lm = orc.kernel_pre_cal(params["Z"], okern)
    kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
    X = params["X"][0]; T = meta["T"]
    rng = np.random.default_rng(3)
    N = 12
    x0 = rng.standard_normal((N-1, D)); eps = rng.standard_normal((T, N-1, D)); u = rng.random((T, N-1))
    R = np.exp(params["log_Rchols"]); Q = np.exp(params["log_Q"])
    pr, ir = pgo.pg_sweep(Lm, params["Z"], okern, params["U"], X, Y, c, params["CC"], params["DD"], R, Q, x0, eps, u)
    pg, ig = prediction.pg_sweep(Lm, params["Z"], kern, params["U"], X, Y, c, params["CC"], params["DD"], R, Q, x0, eps, u)
    print(name, ov, "idx equal", np.array_equal(ir, ig), "max |dparts|", np.max(np.abs(pr - pg)), "ref picked", int((ig == N-1).sum()))
