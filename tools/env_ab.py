"""Same-box, same-process A/B of environment switches that are read when a handle is created (FFVD_* in abi.hip): one engine per
variant, alternating timing rounds, every variant's terms and per-chain nll compared bit for bit with the first one's.
Usage: python tools/env_ab.py "NAME=1 OTHER=2" "NAME=0" ... [--workload c2] [--S 32] [--n 40] [--rounds 3]   ("-" = empty environment)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--workload", default="c2"); ap.add_argument("--S", type=int, default=0)
ap.add_argument("--n", type=int, default=40); ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
params, Y, c, meta = synthetic.make_named(a.workload, **({"S": a.S} if a.S else {}))
engines, base = [], None
for v in a.variants:
    env = dict(kv.split("=", 1) for kv in v.split()) if v != "-" else {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram")
    for k, o in old.items():
        if o is None: del os.environ[k]
        else: os.environ[k] = o
    e.set_data(Y, c); e.set_params(params)
    t = e.nll_terms()
    key = (tuple(sorted((k, float(x)) for k, x in t.items() if k != "nll_per_chain")), e.chain_nll().tobytes())
    if base is None: base = key
    engines.append((v, e, key == base, e.lib.ffvd_schedule_name(e._h).decode()[:70]))
res = {v: [] for v, *_ in engines}
for r in range(a.rounds):
    for v, e, same, name in engines:
        for _ in range(5): e.nll_terms()
        ts = []
        for _ in range(a.n):
            t0 = time.perf_counter(); e.nll_terms(); ts.append(time.perf_counter() - t0)
        res[v].append(float(np.median(ts)) * 1e3)
for v, e, same, name in engines:
    print("AB %-40s median ms/iter per round: %s  bit-identical: %s  [%s]" % (v, " ".join("%.3f" % x for x in res[v]), same, name), flush=True)
