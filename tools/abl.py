import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2")
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"]); e.set_data(Y, c); e.set_params(params)
try:
    e.elbo_sums()
except Exception as ex: pass
st=[e.profile_stages() for _ in range(3)][-1]
print("RES", sys.argv[1:], "project_ms=%.3f gram_ms=%.3f" % (st["project_F"], st["gram_H"]))
