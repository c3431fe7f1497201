import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2")
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], U_collapse=False, grad=True)
e.set_data(Y, c); e.set_params(params)
for _ in range(3): e.adam_step(1e-9)
