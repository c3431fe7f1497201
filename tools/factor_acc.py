"""Gram-route nll of the Cholesky-heavy test shapes against the oracle and the reference op order, for the library FFVD_LIB names
(tools helper: A/B of the dataflow Cholesky's diagonal factor).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_oracle as orc
os.environ["FFVD_NO_TINY"] = "1"
for ov in (dict(M=768, T=832, S=1, D=2), dict(M=512, T=576, S=2, D=4), dict(M=1024, T=1088, S=1, D=2)):
    params, Y, c, meta = synthetic.make_named("c2", **ov)
    S = meta["S"]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram") as e:
        e.set_data(Y, c); tg = e.nll_terms(params)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S) as e:
        e.set_data(Y, c); tr = e.nll_terms(params)
    ref = orc.nll_terms(dict(params, X=params["X"][S - 1]), Y, c, U_collapse=True)
    g, r, o = tg["nll_per_chain"][S - 1], tr["nll_per_chain"][S - 1], ref["nll"]
    print(ov, "gram %.15f  reference-route %.15f  oracle %.15f | gram-oracle %.2e  ref-oracle %.2e  gram-ref %.2e" % (g, r, o, g - o, r - o, g - r))
    print("    oracle terms:", {k: float("%.6g" % v) for k, v in ref.items() if np.isscalar(v)})
