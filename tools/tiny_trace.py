"""Wall-clock stamps of every workgroup of the one-launch iteration (tiny.hip, debug build `python -m ffvd_amd.build --tinytrace`):
head / strip phases in microseconds since the earliest stamp of the launch.  GPU box.  usage: tiny_trace.py [forward|train] [S]"""
import ctypes as ct, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffvd_amd import build as fb
os.environ["FFVD_LIB"] = fb.build_variant(os.environ.get("TINY_TRACE_VARIANT", "tinytrace"))
import numpy as np
from ffvd_amd.engine import ElboEngine
z = np.load(os.path.join(ROOT, "tests", "golden", "actuator_slim.npz"), allow_pickle=False)
params = {k: z[k] for k in ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")}
Y, c = z["Y"], z["control_inputs"]
T, D = params["X"].shape[0] - 1, params["X"].shape[1]
M, C = params["Z"].shape[0], c.shape[1]
grad = len(sys.argv) > 1 and sys.argv[1] == "train"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
params["X"] = np.repeat(params["X"][None], S, axis=0) + 1e-3 * np.random.default_rng(0).standard_normal((S,) + params["X"].shape)
e = ElboEngine(T, D, C, M, S, grad=grad)
e.set_data(Y, c); e.set_params(params)
f = (lambda: e.adam_step(1e-9)) if grad else (lambda: e.nll_terms())
for _ in range(5): f()
buf = (ct.c_longlong * (1024 * 32))()
e.lib.ffvd_debug_tiny_trace.argtypes = [ct.c_void_p]
assert e.lib.ffvd_debug_tiny_trace(buf) == 0
a = np.array(buf[:]).reshape(1024, 32)
nunits = S * D
nst = (T + 63) // 64 if int(e.lib.ffvd_single_launch(e._h)) == 4 else (T + 127) // 128
NT = (M + 15) // 16
nwg = nunits * (1 + nst + (NT if grad else 0))
if not os.environ.get("FFVD_TINY_NO_XCD") and 8 * (1 + nst + (NT if grad else 0)) * ((nunits + 7) // 8) <= 256:        # undo the kernel's xcd_map: physical workgroup id -> role-major id
    wpu = 1 + nst + (NT if grad else 0)
    phys = a.copy(); a = np.zeros_like(phys)
    for b in range(8 * wpu * ((nunits + 7) // 8)):
        xcd, slot = b & 7, b >> 3
        uu, role = (slot // wpu) * 8 + xcd, slot % wpu
        if uu >= nunits or b >= 1024: continue
        vb = uu if role == 0 else (nunits + uu * nst + role - 1 if role <= nst else nunits * (1 + nst) + uu * NT + role - 1 - nst)
        a[vb] = phys[b]
a = a[:nwg].astype(np.float64)
t0 = a[a > 0].min()
us = np.where(a > 0, (a - t0) / 100.0, np.nan)
names_h = ["start", "K built", "chol(K) done", "W published", "strips in", "H summed", "chol(H) done", "terms", "done"]
names_s = ["start", "K_fu built", "W seen", "F in LDS", "partials out", "counted", "N seen", "E ready", "reduced", "K_uu rows", "last strip", "closed"]
print("mode", "train" if grad else "forward", "S", S, "workgroups", nwg, "wavefronts/wg", int(e.lib.ffvd_single_launch(e._h)))
print("heads (us):")
for u in range(min(nunits, 8)):
    print("  u%-3d" % u, " ".join("%s=%.1f" % (n, us[u, i]) for i, n in enumerate(names_h) if not np.isnan(us[u, i])))
print("head 0 detail (us): prologue %s | chol(H) columns %s | W stored %.1f" % (" ".join("%.1f" % us[0, i] for i in (16, 17)), " ".join("%.1f" % us[0, i] for i in range(18, 26) if not np.isnan(us[0, i])), us[0, 26]))
print("head 0, last column step of chol(H) (us): newest terms subtracted %.2f, 16-pivot chain done %.2f" % (us[0, 27], us[0, 29]))
cl = [int(np.nanargmax(us[:, 11]))] if np.any(~np.isnan(us[:, 11])) else []
for w in cl[:1]:
    print("closing workgroup %d (us): unit totals done %.1f, chain arrival %.1f, chain sums done %.1f, dX done %.1f, launch arrival %.1f, end %.1f" % (w, us[w, 12], us[w, 13], us[w, 30], us[w, 14], us[w, 15], np.nanmax(us[w, :12])))
print("strip 0 detail (us): F^T F tiles out %.1f, F^T delta out %.1f" % (us[nunits, 16], us[nunits, 17]))
print("strips of unit 0 (us):")
for i in range(nst):
    r = us[nunits + i]
    print("  s%-3d" % i, " ".join("%s=%.1f" % (n, r[k]) for k, n in enumerate(names_s) if not np.isnan(r[k])))
if grad and nwg > nunits * (1 + nst):
    print("side workgroups of unit 0 (us):")
    for rb in range(NT):
        r = us[nunits * (1 + nst) + rb]
        print("  k%-3d" % rb, " ".join("%s=%.1f" % (names_s[k], r[k]) for k in (0, 6, 9, 10, 11) if not np.isnan(r[k])))
print("span of the launch: %.1f us" % np.nanmax(us))
