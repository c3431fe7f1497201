"""Where the skinny product of a particle-Gibbs / rollout step spends its time (debug build `steptrace`): wall-clock stamps of every
workgroup of the LAST skinny launch of a short sweep.  Run on the GPU box:
    python -m ffvd_amd.build --steptrace && FFVD_LIB=ffvd_amd/libffvd_hip_steptrace.so python tools/step_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic, _lib, conditionals_multi_output as cmo
from ffvd_amd.kernels import SquaredExponential
from ffvd_amd.prediction import pg_sweep
params, Y, c, meta = synthetic.make_named("c2", S=1)
D, C, T = meta["D"], meta["C"], 64
X = params["X"][0][: T + 1]
kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
L = cmo.kernel_pre_cal(params["Z"], kern)
rng = np.random.default_rng(3)
N = int(os.environ.get("PG_N", "100"))
x0, eps, un = rng.standard_normal((N - 1, D)), rng.standard_normal((T, N - 1, D)), rng.random((T, N - 1))
pg_sweep(L, params["Z"], kern, params["U"], X, Y[:T], c[:T], params["CC"], params["DD"], np.exp(params["log_Rchols"]), np.exp(params["log_Q"]), x0, eps, un)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(4096 * 8, dtype=np.int64)
assert lib.ffvd_debug_step_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
raw = buf.reshape(4096, 8)
pg = raw[4095].astype(np.float64) / 100.0
raw = raw[:4095]
live = raw[:, 0] > 0
st = raw[live, :4].astype(np.float64) / 100.0      # us (100 MHz wall clock)
slab, iters = raw[live, 4], raw[live, 5]
t0 = st[:, 0].min()
print("workgroups stamped: %d; span first start -> last end: %.2f us" % (len(st), st[:, 3].max() - t0))
print("starts spread over %.2f us (median %.2f); per-workgroup duration: median %.2f, max %.2f us" %
      (st[:, 0].max() - t0, np.median(st[:, 0] - t0), np.median(st[:, 3] - st[:, 0]), (st[:, 3] - st[:, 0]).max()))
print("phases (median / max us): k loop %.2f / %.2f, partial sums through LDS %.2f / %.2f, epilogue %.2f / %.2f" %
      (np.median(st[:, 1] - st[:, 0]), (st[:, 1] - st[:, 0]).max(), np.median(st[:, 2] - st[:, 1]), (st[:, 2] - st[:, 1]).max(),
       np.median(st[:, 3] - st[:, 2]), (st[:, 3] - st[:, 2]).max()))
order = np.argsort(st[:, 3])
print("last five workgroups to end (start, k loop done, end; us after the first start):")
for i in order[-5:]:
    print("   %.2f  %.2f  %.2f" % (st[i, 0] - t0, st[i, 1] - t0, st[i, 3] - t0))
print("k loop by iterations per wavefront (workgroups, median us, us per iteration):")
for it in sorted(set(iters.tolist())):
    sel = iters == it
    kl = np.median(st[sel, 1] - st[sel, 0])
    print("   %d iterations: %4d workgroups, %.2f us, %.2f us / iteration" % (it, sel.sum(), kl, kl / max(it, 1)))
hw, xcc = raw[live, 6], raw[live, 7] & 0xf
cu = (xcc << 16) | (hw & 0xff00)                  # XCC, then SE / SH / CU bits of HW_ID (bits 8..15)
per_cu = {}
for c, it in zip(cu.tolist(), iters.tolist()): per_cu.setdefault(c, []).append(it)
loads = np.array([sum(v) for v in per_cu.values()])
print("compute units used: %d (XCCs %d); workgroups per CU: min %d max %d; k iterations (of a wavefront) per CU: min %d median %d max %d" %
      (len(per_cu), len(set(xcc.tolist())), min(len(v) for v in per_cu.values()), max(len(v) for v in per_cu.values()), loads.min(), np.median(loads), loads.max()))
print("first CUs' workgroups (iterations each):", [sorted(v) for v in list(per_cu.values())[:12]])
print("HW_ID samples:", [hex(int(x)) for x in hw[:8]], "XCC_ID:", [int(x) for x in xcc[:16]])
if pg[0] > 0:
    print("pg_step: constants + first loads %.2f, candidates %.2f, weights %.2f us" % (pg[6] - pg[0], pg[7] - pg[6], pg[1] - pg[7]))
    print("pg_step (one workgroup), us: candidates + weights %.2f, max %.2f, exp %.2f, cdf %.2f, resample + gather %.2f; total %.2f" %
          (pg[1] - pg[0], pg[2] - pg[1], pg[3] - pg[2], pg[4] - pg[3], pg[5] - pg[4], pg[5] - pg[0]))
sp = np.zeros(64 * 4 * 2, dtype=np.uint64)
if hasattr(lib, "ffvd_debug_step_spans") and lib.ffvd_debug_step_spans(sp.ctypes.data_as(ctypes.c_void_p)) == 0:
    sp = sp.reshape(64, 4, 2).astype(np.float64)
    start, end = sp[:, :, 0] / 100.0, sp[:, :, 1] / 100.0
    ok = np.arange(8, 63)                          # steady steps
    names = ("K build", "product", "epilogue", "step")
    print("timeline of a step, untraced (median over steps 8..62, us): " + ", ".join(
        "%s %.2f" % (names[k], np.median(end[ok, k] - start[ok, k])) for k in range(4)))
    print("   gaps: " + ", ".join("%s -> %s %.2f" % (names[k], names[(k + 1) % 4], np.median((start[ok, k + 1] if k < 3 else start[ok + 1, 0]) - end[ok, k])) for k in range(4)))
    print("   step period %.2f us" % np.median(start[ok + 1, 0] - start[ok, 0]))
