"""Gradient at the full T / M of config 2 (two chains) against the closed-form CPU oracle (tools helper; ~1 min of CPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_grad_oracle as go
KEYS = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")
params, Y, c, meta = synthetic.make_named("c2", S=2)
S = 2
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram", grad=True) as e:
    e.set_data(Y, c)
    terms, g = e.nll_and_grad(params)
ref = {k: np.zeros_like(g[k]) for k in KEYS}
t0 = time.perf_counter()
for s in range(S):
    p = dict(params); p["X"] = params["X"][s]
    ga = go.nll_grad(p, Y, c)
    ref["X"][s] = ga["X"] / S
    for k in KEYS[1:]: ref[k] += ga[k] / S
print("oracle %.1f s" % (time.perf_counter() - t0))
print(" ".join("%s=%.1e" % (k, np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)) for k in KEYS))
# which of the two is closer to the truth for dZ?  central differences of the reference-route nll (accurate to 1e-13)
dz_gpu, dz_ref = g["Z"], ref["Z"]
diff = np.abs(dz_gpu - dz_ref)
idx = np.dstack(np.unravel_index(np.argsort(diff.ravel())[::-1][:6], diff.shape))[0]
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="reference") as e:
    e.set_data(Y, c)
    for (m, p_) in idx:
        fd = []
        for h in (1e-4, 1e-5):
            zp = params["Z"].copy(); zp[m, p_] += h
            zm = params["Z"].copy(); zm[m, p_] -= h
            fd.append((e.nll_terms(dict(params, Z=zp))["nll"] - e.nll_terms(dict(params, Z=zm))["nll"]) / (2 * h))
        print("Z[%d,%d] gpu %.6e oracle %.6e fd(1e-4) %.6e fd(1e-5) %.6e" % (m, p_, dz_gpu[m, p_], dz_ref[m, p_], fd[0], fd[1]))
print("max|dZ| %.3e" % np.max(np.abs(dz_ref)))
