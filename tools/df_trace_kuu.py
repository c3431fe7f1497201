"""The K_uu chain's dataflow launch (8 main block rows + 8 identity-structured rows that become L^-T) through kernel_pre_cal: wall-clock
stamps of matrix 0's rows from the debug build.  FFVD_LIB=$PWD/ffvd_amd/libffvd_hip_dftrace.so python tools/df_trace_kuu.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffvd_amd import synthetic, _lib
from ffvd_amd import conditionals_multi_output as cmo
from ffvd_amd.kernels_multi_output import SquaredExponential
params, Y, c, meta = synthetic.make_named("c2", S=1)
D, C = meta["D"], meta["C"]
kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
for _ in range(3): L = cmo.kernel_pre_cal(params["Z"], kern)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(64 * 64, dtype=np.int64)
assert lib.ffvd_debug_df_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(64, 64).astype(np.float64) / 100.0
nb = meta["M"] // 64
t0 = t[0, 51]
print("main rows: factor done / published (us since row 0 start)")
print("  " + "  ".join("%d: %.1f/%.1f" % (r, t[r, 49] - t0, t[r, 50] - t0) for r in range(nb)))
print("identity rows e: per column j >= e: gather done, diagonal block seen, solved")
for e in range(nb):
    r = nb + e
    line = "  e=%d start %.1f |" % (e, t[r, 51] - t0)
    for j in range(e, nb):
        line += "  [%d] %.1f %.1f %.1f" % (j, t[r, 5 * (j & 7) + 2] - t0, t[r, 5 * (j & 7) + 3] - t0, t[r, 5 * (j & 7) + 4] - t0)
    print(line)
