"""Soak: thousands of forward iterations at two shard sizes; every result must be bit-identical to the first and no call may fail
(the dataflow Cholesky's bounded waits must never fire in normal operation).  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
for S, n in ((32, 2500), (4, 6000), (1, 4000)):
    params, Y, c, meta = synthetic.make_named("c2", S=S)
    e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram")
    e.set_data(Y, c); e.set_params(params)
    first = e.nll_terms()
    t0 = time.perf_counter()
    for i in range(n):
        got = e.nll_terms()
        assert got["nll"] == first["nll"], (S, i, got["nll"], first["nll"])
    print("S=%d: %d iterations identical (nll %.15g), %.3f ms each" % (S, n, first["nll"], (time.perf_counter() - t0) / n * 1e3), flush=True)
