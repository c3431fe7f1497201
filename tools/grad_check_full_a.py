"""Explicit-U branch: dZ / dU at the full T / M of config 2 (two chains) against central differences of the GPU nll."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2", S=2)
S = 2
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, U_collapse=False, grad=True) as e:
    e.set_data(Y, c)
    terms, g = e.nll_and_grad(params)
rng = np.random.default_rng(0)
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, U_collapse=False) as e:
    e.set_data(Y, c)
    for key in ("Z", "U"):
        big = np.argsort(np.abs(g[key]).ravel())[::-1][:3]
        for flat in list(big) + list(rng.integers(g[key].size, size=3)):
            idx = np.unravel_index(flat, g[key].shape)
            fd = []
            for h in (1e-4, 1e-5):
                vp = params[key].copy(); vp[idx] += h
                vm = params[key].copy(); vm[idx] -= h
                fd.append((e.nll_terms(dict(params, **{key: vp}))["nll"] - e.nll_terms(dict(params, **{key: vm}))["nll"]) / (2 * h))
            print("%s%s gpu %.6e fd(1e-4) %.6e fd(1e-5) %.6e  rel.diff %.1e" % (key, idx, g[key][idx], fd[0], fd[1], abs(g[key][idx] - fd[0]) / np.max(np.abs(g[key]))))
