"""Diff of the one-launch path's scratch block between two builds (ffvd_debug_tiny_scratch dumps written by a run under FFVD_LIB=...),
region by region as tiny_carve lays them out -- the tool that located round 4's argument-block corruption.
usage: dbg_cmp.py bad.npy good.npy   (shape constants below: actuator size, S = 10)"""
import numpy as np, sys
bad, good = np.load(sys.argv[1]), np.load(sys.argv[2])
S, Dl, D, Mp, NT, nst, SR, nw, P, M, J = 10, 4, 4, 112, 7, 4, 128, 8, 5, 100, 1
nu, msq = S * Dl, Mp * Mp
al = lambda n: (n + 31) // 32 * 32
ntl = NT * (NT + 1) // 2
pstride, qstride = ntl * 256 + Mp + 8, 16 * Mp + 16
regs, o = [], 0
for name, n in (("Wg", nu * msq), ("Wt", nu * msq), ("Pp", nu * nst * pstride), ("hterms", nu * 2), ("cterms", S * 8), ("psums", 32), ("Hs", nu * msq),
                ("Nw", nu * msq), ("Nm2", nu * msq), ("wv", nu * Mp), ("uterms", nu * 8), ("Qp", nu * nst * qstride), ("dxc", nu * nst * SR * (P + 1)),
                ("kst", nu * nst * nw * 64 * 32), ("dz2", nu * Mp * 8), ("kuu", nu * NT * 9), ("uout", nu * (M * P + P + 2)), ("cpart", S * (D * J + 2 * J + Dl))):
    sz = 32 if name == "psums" else al(n)
    regs.append((name, o, n)); o += sz
print("total", o, len(bad))
for name, off, n in regs:
    a, b = bad[off:off + n], good[off:off + n]
    d = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0]
    print("%-7s n=%8d differing=%8d" % (name, n, len(d)), ("first at %d: %r vs %r" % (d[0], a[d[0]], b[d[0]])) if len(d) else "")
    if name == "Pp" and len(d):
        rel = d % pstride
        print("   Pp differing positions within a strip block: min %d max %d; scalars start at %d; units touched %s" % (rel.min(), rel.max(), ntl * 256 + Mp, sorted(set((d // pstride // nst).tolist()))[:12]))
    if name in ("cterms",) and len(d):
        print("   ", a[:16], b[:16])
