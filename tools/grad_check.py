import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_grad_oracle as go
KEYS = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")
def check(tag, **kw):
    params, Y, c, meta = synthetic.make_workload(**kw)
    S = params["X"].shape[0]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram", grad=True) as e:
        e.set_data(Y, c)
        terms, g = e.nll_and_grad(params)
    ref = {k: np.zeros_like(g[k]) for k in KEYS}
    for s in range(S):
        p = dict(params); p["X"] = params["X"][s]
        ga = go.nll_grad(p, Y, c)
        ref["X"][s] = ga["X"] / S
        for k in KEYS[1:]: ref[k] += ga[k] / S
    out = []
    for k in KEYS:
        err = np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)
        out.append("%s=%.1e" % (k, err))
    print(tag, " ".join(out), flush=True)
    return g, ref
check("base T=96 D=2 C=1 M=24 S=3", T=96, D=2, C=1, M=24, S=3)
check("C=2", T=96, D=2, C=2, M=24, S=3)
check("D=3", T=96, D=3, C=1, M=24, S=3)
check("M=77", T=96, D=2, C=1, M=77, S=3)
check("T=301", T=301, D=2, C=1, M=24, S=3)
check("S=2", T=96, D=2, C=1, M=24, S=2)
g, ref = check("ragged", T=301, D=3, C=2, M=77, S=2)
d = np.abs(g["X"] - ref["X"])
idx = np.unravel_index(np.argmax(d), d.shape)
print("worst X idx", idx, g["X"][idx], ref["X"][idx])
print("X err by dim", d.max(axis=(0, 1)), "by chain", d.max(axis=(1, 2)))
print("rows with err>1e-6:", np.where(d.max(axis=(0, 2)) > 1e-6)[0][:20])
