import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_grad_oracle as gorc, ffvd_oracle_torch as orct, ffvd_oracle as orc
params, Y, c, meta = synthetic.make_named("tiny", C=0)
S = meta["S"]
with ElboEngine(meta["T"], meta["D"], 0, meta["M"], S, route="gram", grad=True) as e:
    e.set_data(Y, c); _, g = e.nll_and_grad(params)
keys = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")
cf = {k: 0 for k in keys}; ag = {k: 0 for k in keys}
cfX, agX = [], []
for s in range(S):
    p = dict(params); p["X"] = params["X"][s]
    a = gorc.nll_grad(p, Y, c); _, b = orct.nll_and_grad(p, Y, c, wrt=keys, U_collapse=True)
    cfX.append(a["X"] / S); agX.append(b["X"] / S)
    for k in keys[1:]:
        cf[k] = cf[k] + a[k] / S; ag[k] = ag[k] + b[k] / S
cf["X"], ag["X"] = np.stack(cfX), np.stack(agX)
kern = orc.make_kernels(params)
print("cond K_uu per dim", [float(np.linalg.cond(k.K(params["Z"]) + 1e-5 * np.eye(meta["M"]))) for k in kern])
for k in keys:
    sc = np.max(np.abs(ag[k])) + 1e-300
    print(k, "gpu-vs-autograd %.2e  gpu-vs-closed %.2e  closed-vs-autograd %.2e" % (np.max(np.abs(g[k] - ag[k])) / sc, np.max(np.abs(g[k] - cf[k])) / sc, np.max(np.abs(cf[k] - ag[k])) / sc))
