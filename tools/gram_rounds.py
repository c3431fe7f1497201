"""Timeline of EVERY workgroup of the headline Gram launch (128 units x 10 tiles) from the debug build's stamps:
    python -m ffvd_amd.build --dftrace;  FFVD_LIB=ffvd_amd/libffvd_hip_dftrace.so python tools/gram_rounds.py
Prints how many tiles run at each moment, the duration of tiles by the round they started in, and how many workgroups each CU ran."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic, _lib
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2")
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram")
e.set_data(Y, c); e.set_params(params)
for _ in range(3): e.nll_terms()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(128 * 10 * 4, dtype=np.int64)
assert lib.ffvd_debug_gram_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(128, 10, 4)
start = t[:, :, 0].astype(np.float64) / 100.0          # us (100 MHz wall clock)
end = t[:, :, 1].astype(np.float64) / 100.0
hw = t[:, :, 2]
t0 = start.min()
start -= t0; end -= t0
dur = end - start
print("launch span %.0f us; first tile starts 0, last tile starts %.0f, last tile ends %.0f" % (end.max(), start.max(), end.max()))
diag = np.zeros(10, bool); diag[[0, 2, 5, 9]] = True
# running workgroups over time
grid = np.arange(0, end.max(), 50.0)
running = [(int(((start <= x) & (end > x)).sum())) for x in grid]
print("tiles in flight every 50 us:")
print(" ".join("%d" % r for r in running))
# rounds by start time
order = np.sort(start.ravel())
print("start-time quantiles (us): 512th %.0f  513th %.0f  1024th %.0f  1025th %.0f  last %.0f" % (order[511], order[512], order[1023], order[1024], order[-1]))
for name, lo, hi in (("round 1 (starts < 100 us)", -1, 100), ("round 2", 100, order[1023] + 1), ("round 3", order[1023] + 1, 1e9)):
    m = (start > lo) & (start <= hi) if lo >= 0 else (start <= hi)
    if m.sum():
        print("%s: %d tiles, duration mean %.0f us (diag %.0f, off %.0f), min %.0f max %.0f, ends %.0f .. %.0f" % (
            name, m.sum(), dur[m].mean(), dur[m & diag[None, :]].mean() if (m & diag[None, :]).any() else 0,
            dur[m & ~diag[None, :]].mean() if (m & ~diag[None, :]).any() else 0, dur[m].min(), dur[m].max(), end[m].min(), end[m].max()))
# HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ... (gfx9 layout); XCC in XCC_ID register (not read)
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
ids, counts = np.unique(cu, return_counts=True)
print("distinct (se, sh, cu) ids seen: %d; tiles per id: min %d max %d" % (len(ids), counts.min(), counts.max()))
# per XCD (blockIdx % 8) and per position in the CU
blk = t[:, :, 3]
xcd = blk & 7
r1 = start < 100
print("round-1 tile durations by XCD (blockIdx % 8): " + "  ".join("%d: %.0f (%.0f-%.0f)" % (x, dur[r1 & (xcd == x)].mean(), dur[r1 & (xcd == x)].min(), dur[r1 & (xcd == x)].max()) for x in range(8)))
print("all tile durations by XCD: " + "  ".join("%d: %.0f" % (x, dur[xcd == x].mean()) for x in range(8)))
print("tiles per XCD ending time: " + "  ".join("%d: %.0f" % (x, end[xcd == x].max()) for x in range(8)))
u = np.arange(128)[:, None] + 0 * blk
print("round-1 durations by position of the unit in its XCD's list (unit // 8): " + "  ".join("%d: %.0f" % (k, dur[r1 & (u // 8 == k)].mean()) for k in range(8) if (r1 & (u // 8 == k)).any()))
print("round-1 durations by tile index within the unit: " + "  ".join("%d: %.0f" % (k, dur[r1[:, k], k].mean()) for k in range(10) if r1[:, k].any()))
if len(sys.argv) > 1:
    np.save(sys.argv[1], t)
busy = dur.sum()
print("sum of tile durations %.0f us = %.1f slot-ms; over %d slots x span %.0f us: occupancy %.2f" % (busy, busy / 1e3, 512, end.max(), busy / (512 * end.max())))
