"""Per-call time of one rank's share of BASELINE configs[4] (LinearK, explicit U, T=4096, M=512, x_dim=16) when the 16 latent dims are
sharded over 1, 2, 4, 8 ranks (d_count = 16, 8, 4, 2 dims on one GPU, result read back every call) -- what an 8-GPU dim-sharded job would
see per rank (tools helper, GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c5")
for dc in (16, 8, 4, 2):
    e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], kernel_type="LinearK", U_collapse=False, d_begin=0, d_count=dc,
                   shared_terms=True)
    e.set_data(Y, c); e.set_params(params)
    for _ in range(5): e.nll_terms()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n): e.nll_terms()
    print("C5 dims_per_rank=%d ms/call=%.3f" % (dc, (time.perf_counter() - t0) / n * 1e3))
    e.close()
