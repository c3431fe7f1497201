"""Accuracy of both routes against the CPU oracle on the ill-conditioned M=600 case and the small goldens."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_oracle as orc
for name, ov in (("small", {}), ("c2", dict(T=700, M=600, S=2)), ("c2", dict(T=1024, M=512, S=2))):
    params, Y, c, meta = synthetic.make_named(name, **ov)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True, kernel_type=meta["kernel_type"])
    for route in ("reference", "gram"):
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route=route) as e:
            e.set_data(Y, c)
            t = e.nll_terms(params)
        print(name, ov, route, "nll rel err %.2e" % (abs(t["nll"] - ref["nll"]) / abs(ref["nll"])),
              "lt1 %.2e" % (abs(t["later_term1"] - ref["later_term1"]) / abs(ref["later_term1"])),
              "trace abs %.2e" % abs(t["nll_reg_trace_inverse_Q_B"] - ref["nll_reg_trace_inverse_Q_B"]))
