"""Consistency probe of the one-launch path (GPU box): forward and backward launches in turn for several chain counts; every value must
repeat and the forward nll must equal the nll the backward launch reports."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
for S in (6, 10, 1, 3):
    params, Y, c, meta = synthetic.make_workload(T=512, D=4, C=1, M=100, S=S)
    with ElboEngine(512, 4, 1, 100, S, grad=True) as e:
        e.set_data(Y, c); e.set_params(params)
        f = e.nll_terms()["nll"]
        vals = []
        for _ in range(3):
            t, g = e.nll_and_grad(params)
            vals.append((t["nll"], float(np.abs(g["Z"]).sum()), float(np.abs(g["X"]).sum())))
            assert e.nll_terms()["nll"] == f
        print("S", S, "plan", e.lib.ffvd_single_launch(e._h), "fwd", f, "grad", vals[0], "repeatable", len(set(vals)) == 1 and vals[0][0] == f, flush=True)
