"""One-off timing of the other BASELINE configs (4: T=16384 M=2048 D=8 S=64, in fp64; 5: LinearK D=16, explicit U)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
name = sys.argv[1]
route = sys.argv[2] if len(sys.argv) > 2 else "reference"
t0 = time.perf_counter()
params, Y, c, meta = synthetic.make_named(name)
t_gen = time.perf_counter() - t0
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], kernel_type=meta["kernel_type"],
               U_collapse=meta["U_collapse"], route=route)
e.set_data(Y, c); e.set_params(params)
t = e.nll_terms()
ms = e.time_elbo(3) / 3
print(json.dumps({"config": name, "route": route, "nll": t["nll"], "ms_per_iter": ms, "workspace_GiB": e.workspace_bytes / 2**30,
                  "W_alg": synthetic.algorithmic_flops(**meta), "gen_s": t_gen}))
