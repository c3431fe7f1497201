"""Per-call latency of the forward ELBO and of the device-resident training step at the reference's own experiment size
(actuator: T=512, M=100, D=4, one chain) -- tools helper, run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ffvd_amd.engine import ElboEngine
z = np.load(os.path.join(ROOT, "tests", "golden", "actuator_slim.npz"), allow_pickle=False)
params = {k: z[k] for k in ("X","Z","U","logvariance","loglengthscales","log_Q","CC","DD","log_Rchols")}
Y, c = z["Y"], z["control_inputs"]
T, D = params["X"].shape[0]-1, params["X"].shape[1]
M, C = params["Z"].shape[0], c.shape[1]
S = int(os.environ.get("ACT_S", "1"))
params["X"] = np.repeat(params["X"][None], S, axis=0) + 1e-3 * np.random.default_rng(0).standard_normal((S,) + params["X"].shape)
modes = {'forward': (False,), 'train': (True,)}.get(sys.argv[1] if len(sys.argv) > 1 else '', (False, True))
collapse = os.environ.get("ACT_BRANCH", "B").upper() != "A"          # ACT_BRANCH=A: the explicit-U branch (FFVD_Main.py cases 1 / 2 / 3 / 6)
for grad in modes:
    e = ElboEngine(T, D, C, M, S, grad=grad, **(dict(route="gram") if collapse else dict(U_collapse=False)))
    e.set_data(Y, c); e.set_params(params)
    f = (lambda: e.adam_step(1e-9)) if grad else (lambda: e.nll_terms())
    for _ in range(10): f()
    n = 300; t0 = time.perf_counter()
    for _ in range(n): f()
    print("actuator T=%d M=%d D=%d S=%d branch %s (%s) %s: %.3f ms per call" % (T, M, D, S, "B" if collapse else "A", "one launch" if int(e.lib.ffvd_single_launch(e._h)) else "multi-kernel",
                                                                          "adam_step" if grad else "forward", (time.perf_counter()-t0)/n*1e3))
