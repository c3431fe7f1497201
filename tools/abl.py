import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
route = kw.get("route", "reference")
ov = {"S": int(kw["S"])} if "S" in kw else {}
params, Y, c, meta = synthetic.make_named(kw.get("workload", "c2"), **ov)
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route=route,
               chains_per_pass=int(kw.get("cpp", 0)))
e.set_data(Y, c); e.set_params(params)
nll = e.nll_terms()["nll"]
for _ in range(3): st = e.profile_stages()
ms = e.time_elbo(20) / 20
print("RES", sys.argv[1:], "nll=%.15g" % nll, "ms/iter=%.3f" % ms, {k: round(v, 3) for k, v in st.items()})
if kw.get("grad"):
    import time
    e2 = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True)
    e2.set_data(Y, c); e2.set_params(params)
    t, g = e2.nll_and_grad()
    t0 = time.perf_counter()
    for _ in range(5): e2.nll_and_grad()
    print("GRAD nll=%.15g fwd+bwd ms/iter=%.3f |dX|max=%.3e |dZ|max=%.3e" % (t["nll"], (time.perf_counter() - t0) / 5 * 1e3, abs(g["X"]).max(), abs(g["Z"]).max()))
