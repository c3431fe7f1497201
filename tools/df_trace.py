"""Where the time of one dataflow Cholesky goes: per block row of matrix 0, the wall-clock stamps the debug build
(tools/df_trace.sh, FFVD_LIB=ffvd_amd/libffvd_hip_dftrace.so) leaves behind.  Run on the GPU box:
    FFVD_LIB=$PWD/ffvd_amd/libffvd_hip_dftrace.so python tools/df_trace.py [S=32]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic, _lib
from ffvd_amd.engine import ElboEngine
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
S = int(kw.get("S", "32"))
params, Y, c, meta = synthetic.make_named("c2", S=S)
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram")
e.set_data(Y, c); e.set_params(params)
for _ in range(3): e.nll_terms()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(64 * 64, dtype=np.int64)
rc = lib.ffvd_debug_df_trace(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = buf.reshape(64, 64).astype(np.float64) / 100.0        # us (100 MHz wall clock)
nb = meta["M"] // 64
t0 = t[0, 51]
print("row  start |  per column j: wait-panels  gather  wait-diag  solve (us, end of phase since row 0 start) |  S_rr ready  factor done  published")
for r in range(nb + 1):
    cols = min(r, nb)
    line = "%3d %6.1f |" % (r, t[r, 51] - t0)
    for j in range(cols):
        w1 = t[r, 5 * j + 1] - t0 if j > 0 else float("nan")
        line += "  [%d] %6.1f %6.1f %6.1f (lds %6.1f) %6.1f" % (j, w1, t[r, 5 * j + 2] - t0, t[r, 5 * j + 3] - t0, t[r, 52 + j] - t0, t[r, 5 * j + 4] - t0)
    if r < nb:
        line += " | %6.1f %6.1f %6.1f" % (t[r, 48] - t0, t[r, 49] - t0, t[r, 50] - t0)
    print(line)

f = t[63, :17]
if f[16] > 0:
    print("blocked 64x64 factor of row 0 (us since its start): per 16-column step  [sums done, chain start, chain done, tiles start]")
    for s4 in range(4):
        print("   s=%d  " % s4 + "  ".join("%6.2f" % (f[4 * s4 + i] - f[0]) for i in range(4)))
    print("   end   %6.2f" % (f[16] - f[0]))
