"""Per-call time of the forward ELBO when the caller reads the result back every iteration (tools helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
for S in [int(x) for x in kw.get("S", "4,8,32").split(",")]:
    params, Y, c, meta = synthetic.make_named("c2", S=S)
    e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route=kw.get("route", "gram"))
    e.set_data(Y, c); e.set_params(params)
    for _ in range(5): e.nll_terms()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): e.nll_terms()
    print("SYNC S=%d ms/call=%.3f" % (S, (time.perf_counter() - t0) / n * 1e3), os.environ.get("FFVD_NO_MAIN_FIRST", ""))
    if "enq" in kw:      # host time of the enqueue alone (all launches of the iteration submitted, nothing waited for)
        import ctypes
        from ffvd_amd import _lib
        f = ctypes.CDLL(_lib.LIB_PATH).ffvd_debug_enqueue_us
        f.restype = ctypes.c_double; f.argtypes = [ctypes.c_void_p]
        print("   host enqueue %.1f us per iteration" % f(e._h))
