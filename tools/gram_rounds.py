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
used = t[:, :, 1] > 0                                   # 10 slots per unit; the round-3 schedule fills 9 (6 tiles + 3 combos)
start = np.where(used, t[:, :, 0].astype(np.float64) / 100.0, np.nan)          # us (100 MHz wall clock)
end = np.where(used, t[:, :, 1].astype(np.float64) / 100.0, np.nan)
t0 = np.nanmin(start)
start -= t0; end -= t0
dur = end - start
print("launch span %.0f us; first workgroup starts 0, last starts %.0f, last ends %.0f" % (np.nanmax(end), np.nanmax(start), np.nanmax(end)))
diagm = t[:, :, 2] == 1                                 # diagonal tile or combo workgroup
nwg = int(used.sum())
print("workgroups recorded: %d (%d per unit)" % (nwg, nwg // 128))
# running workgroups over time
grid = np.arange(0, np.nanmax(end), 50.0)
running = [(int(((start <= x) & (end > x)).sum())) for x in grid]
span_end = np.nanmax(end)
print("tiles in flight every 50 us:")
print(" ".join("%d" % r for r in running))
# rounds by start time
order = np.sort(start[used].ravel())
print("start-time quantiles (us): 512th %.0f  513th %.0f  1024th %.0f  1025th %.0f  last %.0f" % (order[511], order[512], order[1023], order[min(1024, len(order) - 1)], order[-1]))
for name, lo, hi in (("round 1 (starts < 100 us)", -1, 100), ("round 2", 100, order[1023] + 1), ("round 3", order[1023] + 1, 1e9)):
    m = used & ((start > lo) & (start <= hi) if lo >= 0 else (start <= hi))
    if m.sum():
        print("%s: %d workgroups, duration mean %.0f us (diagonal / combo %.0f, off-diagonal %.0f), min %.0f max %.0f, ends %.0f .. %.0f" % (
            name, m.sum(), dur[m].mean(), dur[m & diagm].mean() if (m & diagm).any() else 0,
            dur[m & ~diagm].mean() if (m & ~diagm).any() else 0, dur[m].min(), dur[m].max(), end[m].min(), end[m].max()))
blk = t[:, :, 3]
xcd = blk & 7
print("workgroup durations by XCD (blockIdx % 8): " + "  ".join("%d: %.0f" % (x, np.nanmean(dur[used & (xcd == x)])) for x in range(8)))
if len(sys.argv) > 1:
    np.save(sys.argv[1], t)
busy = np.nansum(dur)
print("sum of workgroup durations %.0f us = %.1f slot-ms; over %d slots x span %.0f us: occupancy %.2f" % (busy, busy / 1e3, 512, span_end, busy / (512 * span_end)))
