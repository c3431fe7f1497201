"""Soak of the training forward + backward (L^T rows read in place, identity-structured rows behind all main rows, reference route
with its chain on the side stream, LinearK through its rank): every result bit-identical to the first, no stall recovery.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
KEYS = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")
for label, kw, ekw, n in (("gram route, 32 chains", {}, dict(route="gram", grad=True), 400),
                          ("gram route, 31 chains (ragged groups)", dict(S=31), dict(route="gram", grad=True), 200),
                          ("reference route, 32 chains", {}, dict(route="reference", grad=True), 150)):
    params, Y, c, meta = synthetic.make_named("c2", **kw)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], **ekw) as e:
        e.set_data(Y, c)
        t0, g0 = e.nll_and_grad(params)
        tt = time.perf_counter()
        for i in range(n):
            t, g = e.nll_and_grad(params)
            assert t["nll"] == t0["nll"], (label, i)
            for k in KEYS:
                assert np.array_equal(g[k], g0[k]), (label, i, k)
        rec = int(e.lib.ffvd_stall_recoveries(e._h))
        assert rec == 0, rec
    print("%s: %d forward+backward passes identical (nll %.15g), %.2f ms each, 0 stall recoveries" % (label, n, t0["nll"], (time.perf_counter() - tt) / n * 1e3), flush=True)
for label, name, ekw, n in (("forward, reference route", "c2", dict(route="reference"), 1500), ("forward, config 5 (LinearK through its rank)", "c5", dict(kernel_type="LinearK", U_collapse=False), 3000)):
    params, Y, c, meta = synthetic.make_named(name)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], **ekw) as e:
        e.set_data(Y, c); e.set_params(params)
        first = e.nll_terms()
        tt = time.perf_counter()
        for i in range(n):
            assert e.nll_terms()["nll"] == first["nll"], (label, i)
        assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
    print("%s: %d iterations identical (nll %.15g), %.3f ms each" % (label, n, first["nll"], (time.perf_counter() - tt) / n * 1e3), flush=True)
