"""Gradient vs the closed-form oracle on shapes that leave partial 128-tiles (Mp = 192, Tp = 192 / 320)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_grad_oracle as gorc
for ov in (dict(T=170, M=150, S=2, D=3, C=1), dict(T=300, M=150, S=1, D=2, C=2), dict(T=130, M=40, S=2, D=5, C=2)):
    params, Y, c, meta = synthetic.make_named("tiny", **ov)
    S = meta["S"]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram", grad=True) as e:
        e.set_data(Y, c); _, g = e.nll_and_grad(params)
    ref = None
    for s in range(S):
        p = dict(params); p["X"] = params["X"][s]
        a = gorc.nll_grad(p, Y, c)
        if ref is None: ref = {k: (np.zeros((S,) + v.shape) if k == "X" else np.zeros_like(v)) for k, v in a.items()}
        ref["X"][s] = a["X"] / S
        for k in a:
            if k != "X": ref[k] += a[k] / S
    print(ov, " ".join("%s=%.1e" % (k, np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)) for k in ref if k in g))
