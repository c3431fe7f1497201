"""Soak of the one-launch iteration (tiny.hip): thousands of forward calls and Adam-free training evaluations at one and ten chains of the
actuator shape; every result bit-identical to the first, no bounded wait may fire.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd.engine import ElboEngine
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(ROOT, "tests", "golden", "actuator_slim.npz"), allow_pickle=False)
base = {k: z[k] for k in ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")}
Y, c = z["Y"], z["control_inputs"]
T, D = base["X"].shape[0] - 1, base["X"].shape[1]
M, C = base["Z"].shape[0], c.shape[1]
branches = [b == "B" for b in os.environ.get("SOAK_BRANCHES", "B,A").split(",")]          # B: collapsed U, A: explicit U (round 5)
for collapse in branches:
  for S, n in ((1, 6000), (10, 6000)):
    params = dict(base, X=np.repeat(base["X"][None], S, axis=0) + 1e-3 * np.random.default_rng(0).standard_normal((S,) + base["X"].shape))
    for grad in (False, True):
        e = ElboEngine(T, D, C, M, S, grad=grad, U_collapse=collapse)
        assert int(e.lib.ffvd_single_launch(e._h)) in (4, 8)
        e.set_data(Y, c); e.set_params(params)
        f = (lambda: e.nll_and_grad()) if grad else (lambda: (e.nll_terms(), None))
        first, g0 = f()
        t0 = time.perf_counter()
        for i in range(n if not grad else n // 4):
            got, g = f()
            assert got["nll"] == first["nll"], (S, grad, i, got["nll"], first["nll"])
            if grad: assert all(np.array_equal(g[k], g0[k]) for k in g0), (S, i)
        assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
        print("branch %s S=%d %s: identical (nll %.15g), %.3f ms each" % ("B" if collapse else "A", S, "nll + gradient" if grad else "forward", first["nll"], (time.perf_counter() - t0) / (n if not grad else n // 4) * 1e3), flush=True)
        e.close()
