"""Where the fused backward product (bwd_fused_kernel, 58 % of a config-2 training step) spends its time: wall-clock stamps of every
workgroup of its last launch (debug build `bwdtrace`).  Run on the GPU box:
    python -m ffvd_amd.build --bwdtrace && FFVD_LIB=ffvd_amd/libffvd_hip_bwdtrace.so python tools/bwd_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic, _lib
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2")
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True)
e.set_data(Y, c); e.set_params(params)
for _ in range(3): e.adam_step(1e-9)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(16384 * 12, dtype=np.int64)
assert lib.ffvd_debug_bwd_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
raw = buf.reshape(16384, 12)
live = raw[:, 2] > 0
st = raw[live, :3].astype(np.float64) / 100.0          # us
where = raw[live, 3]
t0 = st[:, 0].min()
span = st[:, 2].max() - t0
main, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1]
print("workgroups %d; span %.1f us; per workgroup: main loop median %.2f (p10 %.2f p90 %.2f), epilogue median %.2f (p10 %.2f p90 %.2f) us" %
      (live.sum(), span, np.median(main), np.percentile(main, 10), np.percentile(main, 90), np.median(epi), np.percentile(epi, 10), np.percentile(epi, 90)))
print("sum of workgroup time %.1f ms = %.2f workgroups resident on average (512 = two per CU); epilogue share of workgroup time %.1f %%" %
      ((main + epi).sum() / 1e3, (main + epi).sum() / span, 100 * epi.sum() / (main + epi).sum()))
cu = ((where >> 32) << 16) | (where & 0xff00)
ncu = len(set(cu.tolist()))
per = np.bincount(np.unique(cu, return_inverse=True)[1])
print("CUs used %d; workgroups per CU min %d median %d max %d" % (ncu, per.min(), np.median(per), per.max()))
# concurrency over time
ev = np.concatenate([np.stack([st[:, 0], np.ones(len(st))], 1), np.stack([st[:, 2], -np.ones(len(st))], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
for frac in (0.05, 0.25, 0.5, 0.75, 0.95, 0.99):
    i = np.searchsorted(ev[:, 0], t0 + frac * span)
    print("   at %2.0f %% of the span: %d workgroups resident" % (100 * frac, conc[min(i, len(conc) - 1)]))
last = np.sort(st[:, 2] - t0)
print("last workgroup ends at %.1f us; 99 %% have ended by %.1f us, 90 %% by %.1f us" % (last[-1], last[int(0.99 * len(last))], last[int(0.9 * len(last))]))
# ideal: MFMA time of a workgroup's main loop: 8 wavefronts x 1024 MFMAs x 64 cycles / 4 SIMDs at 2.4 GHz
print("MFMA-bound main loop of one workgroup alone on a CU: %.2f us; with two resident: %.2f us each" % (8 * 1024 * 64 / 4 / 2400.0, 2 * 8 * 1024 * 64 / 4 / 2400.0))
ep = raw[live][:, [1, 4, 5, 6, 2]].astype(np.float64) / 100.0
d = np.diff(ep, axis=1)
print("epilogue phases of wavefront 0 (median us): staging + first K_fu loads %.2f, four strips %.2f, wait for the other wavefronts %.2f, combine + store %.2f" %
      tuple(np.median(d, axis=0)))
s0 = raw[live][:, [4, 7, 8, 9]].astype(np.float64) / 100.0
d0 = np.diff(s0, axis=1)
print("first strip of wavefront 0 (median us): e in place + kfu partial %.2f, column products (8 MFMAs) %.2f, patches + row products (8 MFMAs) + partial stores %.2f" % tuple(np.median(d0, axis=0)))
