"""Timing of the explicit-U branch at the headline shape: forward, forward+backward, device-resident Adam step."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2")
out = {"workload": "synthetic T=4096 D=4 C=1 M=512 S=32 fp64, explicit-U branch"}
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], U_collapse=False) as e:
    e.set_data(Y, c); e.set_params(params); e.nll_terms()
    out["forward_ms"] = e.time_elbo(10) / 10
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], U_collapse=False, grad=True) as e:
    e.set_data(Y, c); e.set_params(params); e.nll_and_grad()
    t0 = time.perf_counter()
    for _ in range(5): e.nll_and_grad()
    out["fwd_bwd_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    first = e.adam_step(0.003)["nll"]
    t0 = time.perf_counter()
    for _ in range(10): last = e.adam_step(0.003)["nll"]
    out["adam_step_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    out["nll_first"], out["nll_after_11_steps"] = first, last
print(json.dumps(out))
