"""Wall time of one particle-Gibbs sweep at config 2's shape (T = 4096 steps, 100 particles, M = 512, D = 4), twice per setting of
FFVD_PG_FUSED (an experiment switch of round 5 -- the fused two-launch step, measured slower and removed again; the switch is ignored by
the shipped library: both settings time the four-launch step).  Results must be bit-identical.  GPU box helper; tools/pg_trace.sh
runs it under rocprofv3 for the per-kernel table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic, conditionals_multi_output as cmo
from ffvd_amd.kernels import SquaredExponential
from ffvd_amd.prediction import pg_sweep
params, Y, c, meta = synthetic.make_named("c2", S=1)
D, C, T = meta["D"], meta["C"], meta["T"]
X = params["X"][0]
kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
L = cmo.kernel_pre_cal(params["Z"], kern)
rng = np.random.default_rng(3)
N = 100
x0, eps, un = rng.standard_normal((N - 1, D)), rng.standard_normal((T, N - 1, D)), rng.random((T, N - 1))
args = (L, params["Z"], kern, params["U"], X, Y, c, params["CC"], params["DD"], np.exp(params["log_Rchols"]), np.exp(params["log_Q"]), x0, eps, un)
res = {}
for mode in ("1", "0", "1", "0"):
    os.environ["FFVD_PG_FUSED"] = mode
    pg_sweep(*[a[:33] if i in (4,) else a for i, a in enumerate(args[:11])], x0, eps[:32], un[:32])       # warm-up on 32 steps
    t0 = time.perf_counter()
    parts, idx = pg_sweep(*args)
    dt = time.perf_counter() - t0
    res.setdefault(mode, []).append((dt, parts, idx))
    print("PG sweep T=%d N=%d FFVD_PG_FUSED=%s: %.1f ms, %.2f us per step" % (T, N, mode, dt * 1e3, dt / T * 1e6), flush=True)
print("bit-identical:", np.array_equal(res["1"][0][1], res["0"][0][1]) and np.array_equal(res["1"][0][2], res["0"][0][2]))
