"""A/B of the pass-pipelined forward iteration (enqueue_elbo_pipe, FFVD_PIPE / FFVD_PIPE_MODE) against the full-batch schedule at
BASELINE configs[1] (c2), on ONE box in ONE process: the switches are read when a handle is created.  Every variant must return the
baseline's terms and per-chain nll bit for bit.  Usage: python tools/pipe_ab.py [variants=0:0,2:0,4:0,...] [n=40] [rounds=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
variants = [tuple(int(x) for x in v.split(":")) for v in kw.get("variants", "0:0,2:0,4:0,8:0,4:1,4:2,2:2").split(",")]
n, rounds = int(kw.get("n", 40)), int(kw.get("rounds", 3))
params, Y, c, meta = synthetic.make_named("c2")
engines, base = [], None
for passes, mode in variants:
    os.environ["FFVD_PIPE"], os.environ["FFVD_PIPE_MODE"] = str(passes), str(mode)
    e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram")
    e.set_data(Y, c); e.set_params(params)
    t = e.nll_terms()
    key = (tuple(sorted((k, float(v)) for k, v in t.items() if k != "nll_per_chain")), e.chain_nll().tobytes())
    if base is None: base = key
    name = e.lib.ffvd_schedule_name(e._h).decode()[:60]
    engines.append((passes, mode, e, key == base, name))
best = {}
for r in range(rounds):                       # alternating rounds: box drift hits every variant alike
    for passes, mode, e, same, name in engines:
        for _ in range(5): e.nll_terms()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); e.nll_terms(); ts.append(time.perf_counter() - t0)
        ms = float(np.median(ts)) * 1e3
        best.setdefault((passes, mode), []).append(ms)
for passes, mode, e, same, name in engines:
    v = best[(passes, mode)]
    print("PIPE passes=%d mode=%d  median ms/iter per round: %s  bit-identical to baseline: %s  [%s]" %
          (passes, mode, " ".join("%.3f" % x for x in v), same, name), flush=True)
