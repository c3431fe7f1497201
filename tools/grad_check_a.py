"""Explicit-U branch: HIP backward pass against the closed-form oracle (nll_grad_explicit_u)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_grad_oracle as gorc, ffvd_oracle as orc
for name, ov in (("tiny", {}), ("ragged", {}), ("small", {}), ("tiny", dict(C=0)), ("tiny", dict(T=170, M=150, S=2, D=3, C=1))):
    params, Y, c, meta = synthetic.make_named(name, **ov)
    S = meta["S"]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, U_collapse=False, grad=True) as e:
        e.set_data(Y, c)
        t, g = e.nll_and_grad(params)
    ref = None
    for s in range(S):
        p = dict(params); p["X"] = params["X"][s]
        a = gorc.nll_grad_explicit_u(p, Y, c)
        if ref is None: ref = {k: (np.zeros((S,) + v.shape) if k == "X" else np.zeros_like(v)) for k, v in a.items()}
        ref["X"][s] = a["X"] / S
        for k in a:
            if k != "X": ref[k] += a[k] / S
    nll_ref = orc.nll_terms_chains(params, Y, c, U_collapse=False)["nll"]
    print(name, ov, "nll %.1e" % (abs(t["nll"] - nll_ref) / abs(nll_ref)),
          " ".join("%s=%.1e" % (k, np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)) for k in g))
