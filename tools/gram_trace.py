"""Start and duration of the Gram kernel's workgroups (tiles of the first 16 units) from the debug build's stamps
(python -m ffvd_amd.build --dftrace; FFVD_LIB=ffvd_amd/libffvd_hip_dftrace.so python tools/gram_trace.py)."""
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
buf = np.zeros(64 * 64, dtype=np.int64)
assert lib.ffvd_debug_df_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf[2048:2048 + 320].reshape(16, 10, 2).astype(np.float64) / 100.0
t0 = t[:, :, 0].min()
diag = [0, 2, 5, 9]
off = [1, 3, 4, 6, 7, 8]
dur = t[:, :, 1] - t[:, :, 0]
print("unit: start offsets of its 10 tiles (us) | durations (us)")
for u in range(16):
    print("%2d  " % u + " ".join("%7.0f" % (x - t0) for x in t[u, :, 0]) + "  |  " + " ".join("%6.0f" % x for x in dur[u]))
print("mean duration  diagonal tiles %.1f us   off-diagonal tiles %.1f us" % (dur[:, diag].mean(), dur[:, off].mean()))
