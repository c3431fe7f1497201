// ONE launch for the whole ELBO iteration -- and, with grad, its backward pass -- at the reference's own experiment size
// (FFVD_Main.py:356-369: T <= 512, M = 100, D = 4 latent dims, 1-10 chains; models.py:142-182 runs 4000 outer iterations,
// each 1 or 22 evaluations of nll and its gradient).  At that size the multi-kernel schedule of abi.hip is a chain of 17 (53
// with the backward pass) dependent launches of 4-20 us each on two streams; nothing in it is bound by the hardware.
//
// Arithmetic: the reference's own op order for the collapsed branch (conditionals_multi_output.py:124-169, :230-257):
//   L = chol(K(Z,Z) + jitter I),  W = L^-T,  F = K(x_comb, Z) W,  H = I + F^T F / Q,  b = delta^T F / Q,
//   -1/2 log|H|,  1/2 b H^-1 b^T,  -1/2 sum_t (sigma^2 - |F_t|^2) / Q        (H has condition ~1e4; nothing is formed in the
//   K_uu + K_uf K_fu / Q variables of the big-batch Gram route), likelihood / transition / prior terms as in dgp_model.py:248-288.
// Backward pass: the closed form of oracle/ffvd_grad_oracle.py (whitened variables), arranged as in tools/tiny_proto.py.
//
// Roles (one workgroup each, ALL resident at once -- tiny_plan checks that they fit; hand-offs through agent-scope flags exactly as
// in the dataflow Cholesky of kernels.hip: plain stores -> every wavefront's s_waitcnt vmcnt(0) -> barrier -> one lane's release +
// flag store; one lane polls -> acquire -> barrier -> plain loads; every wait is bounded by the wall clock):
//   head(u), one per (chain, latent dim) unit, blockIdx < nunits (dispatched first; a head never waits before it has published W):
//     K_uu in LDS -> blocked Cholesky (16 x 16 tiles: the 16-pivot chains on ONE wavefront, everything else on the matrix cores
//     of the others, L^-T riding along as identity-structured extension tiles) -> W, W^T to L2 -> flag W
//     ... waits for its strips ... H = I + alpha sum_strips F^T F -> the same factorisation -> log|H|, |L_H^-1 b|^2
//     (grad: H^-1, w = H^-1 b, N = I - H^-1 - w w^T, N - (H - I) -> L2 -> flag N).
//   strip(u, i), 16 rows per wavefront: K_fu rows generated into registers (MFMA accumulator layout) and LDS while the head
//     factorises -> wait W -> F = K W on the matrix cores -> row sums of F^2, F^T F tiles, F^T delta, chain-term partials -> count
//     (grad: wait N -> R = F N + delta w^T -> dl/dK_fu = alpha R W^T -> E = dl/dK_fu o K_fu in registers -> row sums, E Z, column
//     sums, E^T x as MFMA products against [1 | Z] and [1 | x] -> rows of dl/dX, partials of dl/dZ, dl/dloglen, dl/dlogvar; and one
//     16-row block of the K_uu side, Psi = 1/2 W (N - (H - I)) W^T o K_uu, per strip -> count).
//   closers: the workgroup that completes a unit adds its strips' partials; the one that completes a chain forms the chain's
//     terms (and dl/dX, the likelihood gradients); the one that completes the launch runs the nll assembly (finalize_body, the
//     code of finalize_kernel), the shared-parameter gradients, and re-arms the flags.  Fixed summation orders: results do not
//     depend on which workgroup closes.
#include "tiny.h"
#include "dev_common.h"
#include <cstring>

namespace ffvd {

constexpr int TNT = TINY_MPMAX / 16;            // tile columns at most
constexpr int TPP = TINY_PMAX;                  // GP input dimension at most
constexpr long long TINY_SPIN_TICKS = 100000000LL;   // 1 s of the 100 MHz wall clock

// Debug build only (-DFFVD_TINY_TRACE, tools/tiny_trace.py): wall-clock stamps of every workgroup's phases.
#ifdef FFVD_TINY_TRACE
__device__ long long tiny_trace_buf[1024 * 32];
#define TSTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 1024) tiny_trace_buf[blockIdx.x * 32 + (slot)] = wall_clock64(); } while (0)   // (by PHYSICAL id: tools/tiny_trace.py undoes xcd_map)
extern "C" int ffvd_debug_tiny_trace(long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tiny_trace_buf), sizeof(long long) * 1024 * 32);
}
#else
#define TSTAMP(slot) do { } while (0)
#endif

// ---- LDS layout (doubles), shared by host and device ----------------------------------------------------------------------
struct TinyLds {
    int ctl, tab, red, vec, mat, dinv, sc, xo, zo, misc, total;
};
// ctl / red / vec / mat are common to both roles; behind them the head keeps (dinv, sc) and a strip (xo, zo, misc) in the SAME
// space: a workgroup is one or the other (and a closer uses the common part only).
__host__ __device__ inline TinyLds tiny_lds(int Mp, int SR) {
    TinyLds l;
    const int LD = Mp + 1, NT = Mp / 16;
    int o = 0;
    l.ctl = o;  o += 8;                         // ints: wait slot, arrive slot
    l.tab = o;  o += 32;                        // ints: (i, j) of the lower-triangular tiles (<= 36 ints = 18 doubles), the lengthscales [8] behind them
    l.red = o;  o += 8 * 24;                    // reduction scratch [<= 8 wavefronts][<= 24 values]
    l.vec = o;  o += 4 * Mp;                    // head: b, y, w;  strip: w;  closer: column sums
    l.mat = o;                                  // head: the matrix being factorised [Mp][LD];  strip: K_fu / F / R / E rows [SR][LD]
    int oh = o + Mp * LD;
    l.dinv = oh; oh += (NT * 16 * 17 > Mp * 9) ? NT * 16 * 17 : Mp * 9;     // head: inverted diagonal tiles W(s,s); (K_uu build: Z / l and |.|^2)
    l.sc = oh;   oh += 2 * 16 * 17;             // head: the diagonal tile handed to the pivot chain, and behind it an identity tile (its lanes 16-31)
    int os = o + SR * LD;
    l.xo = os;   os += SR * 16;                 // strip: x / l rows + |.|^2 (K build), later [1 | x_comb] rows
    l.zo = os;   os += Mp * 16;                 // strip: Z / l rows + |.|^2 (K build), later [1 | Z] rows
    l.misc = os; os += 4 * SR + 64;             // strip: delta, per-row values
    l.total = oh > os ? oh : os;
    return l;
}

__host__ __device__ inline int tiny_ntl(int NT) { return NT * (NT + 1) / 2; }
__host__ __device__ inline int tiny_pstride(int Mp) { return tiny_ntl(Mp / 16) * 256 + Mp + 8; }
__host__ __device__ inline int tiny_qstride(int Mp) { return 16 * Mp + 16; }

// ---- hand-offs --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tiny_wait(int *flag, int need, int *abort_w, int *slot) {
    if (threadIdx.x == 0) {
        int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < need) {
            const long long t0 = wall_clock64();
            for (;;) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= need) break;
                if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { v = -1; break; }
                if (wall_clock64() - t0 > TINY_SPIN_TICKS) {
                    __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = -1;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *slot = v;
    }
    __syncthreads();
    const int v = *slot;
    __syncthreads();
    return v;
}
// every thread's stores are in memory; then ONE lane releases and sets the flag
__device__ __forceinline__ void tiny_publish(int *flag, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// ... or counts in; returns how many had arrived before (the same value in every thread).  The arrival that completes a count
// has acquired everything the others released.
__device__ __forceinline__ int tiny_arrive(int *cnt, int *slot) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int seen = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *slot = seen;
    }
    __syncthreads();
    const int v = *slot;
    __syncthreads();
    return v;
}

// fixed-order sums of N values over the workgroup (NW wavefronts)
template <int N, int NW>
__device__ __forceinline__ void tiny_sum(double (&v)[N], double *red) {
    block_sum_multi<N, NW>(v, reinterpret_cast<double(*)[N]>(red));
    __syncthreads();
}

// ---- blocked Cholesky of an n x n matrix in LDS (n = 16 NT), with L^-T -----------------------------------------------------
// 16-pivot chain on the tile in Sc (lower triangle valid): lanes 0-15 carry its rows, lanes 16-31 the rows of the identity through
// the same column operations -- they come out as L_ss^-T (kernels.hip, chol64_mfma_1w).  Writes L_ss (zeros above the diagonal)
// into the diagonal tile of Am and L_ss^-T into Dv.  Returns 0 or 1 + the first non-positive pivot of the tile.
// (tools/probes/lat_probe.hip: the bare right-looking chain is 1.06 us; as first written -- the identity rows chosen by a select
//  per element, i.e. an exec-masked load each; the first bad pivot tracked by a scalar compare per pivot; two exec-masked store
//  streams -- 1.92 us.  This form: 1.54 us.  Lanes 16-31 read an identity tile that sits in LDS behind Sc; a non-positive or NaN
//  pivot leaves NaN on the diagonal of L from there on, which is looked for once, behind the chain; one store stream.)
__device__ __forceinline__ int tiny_chain16(const double (*Sc)[17], double *Am, const int LD, const int s0, double (*Dv)[17], const int lane) {
    const int lr = lane & 15;
    double a[16];
    const double *src = &Sc[(lane < 16) ? lr : 16 + lr][0];             // rows 16-31 of Sc's block: the identity tile
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = src[c];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double ajj = readlane_f64(a[j], j);
        double piv, y;
        pivot_sqrt(ajj, piv, y);
        a[j] *= y;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = fma(-a[j], readlane_f64(a[j], c), a[c]);
    }
    double diag = a[0];
#pragma unroll
    for (int c = 1; c < 16; ++c) diag = (lr == c) ? a[c] : diag;
    if (lane < 32) {
        double *base = (lane < 16) ? Am + (size_t)(s0 + lr) * LD + s0 : &Dv[lr][0];
#pragma unroll
        for (int c = 0; c < 16; ++c) base[c] = (lane >= 16 || c <= lr) ? a[c] : 0.0;
    }
    const unsigned long long m = __ballot((lane < 16) & !(diag > 0.0));
    return m ? (int)__builtin_ctzll(m) + 1 : 0;
}

// S = A(s,s) - sum_{kbeg<=k<s} L(s,k) L(s,k)^T into Sc (one wavefront)
__device__ __forceinline__ void tiny_diag_gather(const double *Am, const int LD, const int s, const int kbeg, double (*Sc)[17], const int lane) {
    const int lr = lane & 15, lk = lane >> 4, s0 = 16 * s;
    d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
    for (int k = kbeg; k < s; ++k) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double v = Am[(size_t)(s0 + lr) * LD + 16 * k + 4 * t + lk];
            if (t & 1) a1 = mfma_f64(v, v, a1);
            else a0 = mfma_f64(v, v, a0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Sc[lk + 4 * r][lr] = Am[(size_t)(s0 + lk + 4 * r) * LD + s0 + lr] - (a0[r] + a1[r]);
}

// One 16 x 16 tile of block column s (one wavefront): X = (T - sum_{k0<=k<s} Xrow(k) L(s,k)^T) L_ss^-T, in place.
//   main row block rb > s:  T = A(rb,s), k from 0;      extension row block e = rb < s (identity-structured: it becomes
//   W(e,s) = (L^-T)(e,s)):  T = 0, k from e, and the k = e term reads W(e,e) = Dinv[e].
// The products run transposed in the accumulator layout, which is the B-operand layout of the next product (potrf_panel_kernel);
// one residual refinement step against L_ss restores substitution accuracy.
__device__ __forceinline__ void tiny_tile_solve(double *Am, const int LD, double (*Dinv)[16][17], const int s, const int rb, const bool ext,
                                                const int lane, const int kmain = 0 /* main rows: first term still to subtract */) {
    const int lr = lane & 15, lk = lane >> 4, s0 = 16 * s, r0 = 16 * rb;
    d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
    for (int k = ext ? rb : kmain; k < s; ++k) {
        const bool dk = ext && k == rb;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double av = Am[(size_t)(s0 + lr) * LD + 16 * k + 4 * t + lk];
            const double bv = dk ? Dinv[rb][lr][4 * t + lk] : Am[(size_t)(r0 + lr) * LD + 16 * k + 4 * t + lk];
            if (t & 1) a1 = mfma_f64(av, bv, a1);
            else a0 = mfma_f64(av, bv, a0);
        }
    }
    d4 Rt;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rt[r] = (ext ? 0.0 : Am[(size_t)(r0 + lr) * LD + s0 + lk + 4 * r]) - (a0[r] + a1[r]);
    d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) x = mfma_f64(Dinv[s][lk + 4 * t][lr], Rt[t], x);
    d4 res = Rt;
#pragma unroll
    for (int t = 0; t < 4; ++t) res = mfma_f64(-Am[(size_t)(s0 + lr) * LD + s0 + 4 * t + lk], x[t], res);
#pragma unroll
    for (int t = 0; t < 4; ++t) x = mfma_f64(Dinv[s][lk + 4 * t][lr], res[t], x);
#pragma unroll
    for (int r = 0; r < 4; ++r) Am[(size_t)(r0 + lr) * LD + s0 + lk + 4 * r] = x[r];
}

// Row block r is factorised by wavefront 0 in column step r - 1.  One step earlier (column r - 2) every term k <= r - 3 of its two
// tiles on the critical path is final: a helper subtracts them in place then,
//   A(r, r-1) -= sum_{k<kend} L(r,k) L(r-1,k)^T,   A(r, r) -= sum_{k<kend} L(r,k) L(r,k)^T,   kend = r - 2,
// and wavefront 0 is left with one term for the tile and two for the diagonal block whatever the column (the gathers were 40 % of
// its step at column 5: tools/tiny_trace.py).
__device__ __forceinline__ void tiny_pregather(double *Am, const int LD, const int r, const int kend, const int lane) {
    const int lr = lane & 15, lk = lane >> 4, r0 = 16 * r, q0 = 16 * (r - 1);
    d4 g0 = (d4){0.0, 0.0, 0.0, 0.0}, g1 = g0, h0 = g0, h1 = g0;
    for (int k = 0; k < kend; ++k) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double lq = Am[(size_t)(q0 + lr) * LD + 16 * k + 4 * t + lk], lrw = Am[(size_t)(r0 + lr) * LD + 16 * k + 4 * t + lk];
            if (t & 1) { g1 = mfma_f64(lq, lrw, g1); h1 = mfma_f64(lrw, lrw, h1); }
            else { g0 = mfma_f64(lq, lrw, g0); h0 = mfma_f64(lrw, lrw, h0); }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        Am[(size_t)(r0 + lr) * LD + q0 + lk + 4 * q] -= g0[q] + g1[q];        // (transposed in the accumulator layout, as in tiny_tile_solve)
        Am[(size_t)(r0 + lk + 4 * q) * LD + r0 + lr] -= h0[q] + h1[q];
    }
}

#ifndef TINY_CHOL_ROWS
#define TINY_CHOL_ROWS 1      // 0: the round-4 first form below (tile solves on the matrix cores behind every chain; A/B build `tinytiles`)
#endif
#if TINY_CHOL_ROWS
// One older term range of a tile of a LATER column, in place (one wavefront):  A(i, c) -= sum_{k<kend} L(i,k) L(c,k)^T
__device__ __forceinline__ void tiny_tile_sub(double *Am, const int LD, const int i, const int c, const int kbeg, const int kend, const int lane) {
    const int lr = lane & 15, lk = lane >> 4, i0 = 16 * i, c0 = 16 * c;
    d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
    for (int k = kbeg; k < kend; ++k) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double av = Am[(size_t)(i0 + lr) * LD + 16 * k + 4 * t + lk], bv = Am[(size_t)(c0 + lr) * LD + 16 * k + 4 * t + lk];
            if (t & 1) a1 = mfma_f64(av, bv, a1);
            else a0 = mfma_f64(av, bv, a0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Am[(size_t)(i0 + lk + 4 * r) * LD + c0 + lr] -= a0[r] + a1[r];
}
// In: lower triangle of Am.  Out: lower triangle = L (diagonal tiles with zeros above the diagonal), tiles above the diagonal =
// W = L^-T, Dinv[s] = W(s,s).
// Every tile BELOW diagonal tile s rides through that tile's 16-pivot chain (kernels.hip, chol64_mfma_4w has the argument: the chain is
// the unblocked right-looking elimination, a lane that starts from a row of T'(i,s) = A(i,s) - sum_{k<s} L(i,k) L(s,k)^T leaves it as
// that row of L(i,s)).  A chain wavefront carries the diagonal tile's rows in lanes 0-15 and three riders in lanes 16-63: wavefront
// 0 the identity (-> L_ss^-T = W(s,s)) and tiles (s+1,s), (s+2,s), wavefronts 1, 2 the tiles further down -- they repeat the pivots for
// themselves.  Column step s:
//   (a) every wavefront: the newest term, A(i,s) -= L(i,s-1) L(s,s-1)^T, for the tiles i >= s it is dealt (4 MFMAs each); barrier
//   (b) chain wavefronts: the chain;   the others, beside it: the older terms (k < s) of column s+1's tiles, in place, and the tiles
//       W(e, s-1), e < s-1, of the inverse (they need W(s-1,s-1), which chain s-1 left) as before on the matrix cores; barrier
// so that the latency chain of a column is one 4-MFMA product, two barriers and the pivot chain -- the first form had a tile solve
// (three dependent groups of four MFMAs), the sums of the next diagonal tile and two LDS round trips there: 3.15 -> 2.2 us per column.
template <int NW>
__device__ __forceinline__ void tiny_chol_inv(double *Am, const int LD, const int NT, double (*Dinv)[16][17], double (*Sc)[17],
                                              int32_t *info_word) {
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double (*Id)[17] = Sc + 16;                                        // the identity tile behind Sc
    // cnt[s]: chain wavefronts other than 0 that have read their rows in step s -- they read the diagonal tile that wavefront 0
    // overwrites with L_ss behind its chain (1.5 us later: the wait never spins, but the order is not left to timing)
    int *cnt = reinterpret_cast<int *>(&Sc[0][0]);
    if (tid < 16) cnt[tid] = 0;
    __syncthreads();
    int bad = 0;
    for (int s = 0; s < NT; ++s) {
        const int s0 = 16 * s;
        if (s > 0) {
            for (int i = s + wave; i < NT; i += NW) tiny_tile_sub(Am, LD, i, s, s - 1, s, lane);
            __syncthreads();
        }
        TSTAMP(27);
        const int nr = NT - s;                                               // riders: the identity, then tiles s+1 .. NT-1
        const int ncw = (nr + 2) / 3;                                        // chain wavefronts (<= 3: NT <= 8)
        if (wave < ncw) {
            __builtin_amdgcn_s_setprio(3);
            const int q = 3 * wave + lk - 1;                                 // this lane group's rider (lk = 0: the diagonal tile)
            const bool live = lk > 0 && q < nr;
            const double *src = (lk > 0 && q == 0) ? &Id[lr][0] : Am + (size_t)(s0 + (live ? 16 * q : 0) + lr) * LD + s0;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = src[c];
            if (wave > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(cnt + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double ajj = readlane_f64(a[j], j);
                double piv, y;
                pivot_sqrt(ajj, piv, y);
                a[j] *= y;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) a[c] = fma(-a[j], readlane_f64(a[j], c), a[c]);
            }
            double diag = a[0];
#pragma unroll
            for (int c = 1; c < 16; ++c) diag = (lr == c) ? a[c] : diag;
            if (wave == 0 && ncw > 1)
                while (__hip_atomic_load(cnt + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < ncw - 1) { }
            if (lk == 0 ? wave == 0 : live) {
                double *base = (lk > 0 && q == 0) ? &Dinv[s][lr][0] : Am + (size_t)(s0 + (lk > 0 ? 16 * q : 0) + lr) * LD + s0;
#pragma unroll
                for (int c = 0; c < 16; ++c) base[c] = (lk > 0 || c <= lr) ? a[c] : 0.0;
            }
            if (wave == 0) {
                const unsigned long long m = __ballot((lane < 16) & !(diag > 0.0));
                if (m && !bad) bad = s0 + (int)__builtin_ctzll(m) + 1;
            }
            __builtin_amdgcn_s_setprio(0);
            TSTAMP(29);
        } else {
            const int nid = NW - ncw, me = wave - ncw;
            int idx = 0;
            if (s >= 1)
                for (int i = s + 1; i < NT; ++i, ++idx)
                    if (idx % nid == me) tiny_tile_sub(Am, LD, i, s + 1, 0, s, lane);
            for (int e = 0; e + 1 < s; ++e, ++idx)
                if (idx % nid == me) tiny_tile_solve(Am, LD, Dinv, s - 1, e, true, lane);
        }
        __syncthreads();
        TSTAMP(18 + (s < 8 ? s : 7));
    }
    {   // the inverse's tiles of the last column
        for (int e = wave; e + 1 < NT; e += NW) tiny_tile_solve(Am, LD, Dinv, NT - 1, e, true, lane);
        __syncthreads();
    }
    if (wave == 0 && lane == 0 && bad && *info_word == 0) *info_word = bad;
}
#else
// In: lower triangle of Am.  Out: lower triangle = L (diagonal tiles with zeros above the diagonal), tiles above the diagonal =
// W = L^-T, Dinv[s] = W(s,s).  Left-looking by tile column; wavefront 0 owns the critical path (tile (s+1,s), then the gather and
// the 16-pivot chain of diagonal tile s+1), the others solve the remaining tiles of column s beside it; one barrier per column.
template <int NW>
__device__ __forceinline__ void tiny_chol_inv(double *Am, const int LD, const int NT, double (*Dinv)[16][17], double (*Sc)[17],
                                              int32_t *info_word) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bad = 0;
    if (wave == 0) {
        tiny_diag_gather(Am, LD, 0, 0, Sc, lane);
        wave_lds_order();
        bad = tiny_chain16(Sc, Am, LD, 0, Dinv[0], lane);
    }
    __syncthreads();
    for (int s = 0; s < NT; ++s) {
        if (wave == 0) {
            if (s + 1 < NT) {
                const int kb = s >= 1 ? s - 1 : 0;                            // terms k < s - 1 of row s + 1 were subtracted in step s - 1
                __builtin_amdgcn_s_setprio(3);
                tiny_tile_solve(Am, LD, Dinv, s, s + 1, false, lane, kb);
                wave_lds_order();
                TSTAMP(27);
                tiny_diag_gather(Am, LD, s + 1, kb, Sc, lane);
                wave_lds_order();
                TSTAMP(28);
                const int b2 = tiny_chain16(Sc, Am, LD, 16 * (s + 1), Dinv[s + 1], lane);
                TSTAMP(29);
                if (b2 && !bad) bad = 16 * (s + 1) + b2;
                __builtin_amdgcn_s_setprio(0);
            }
        } else {
            int idx = 0;
            if (s >= 1 && s + 2 < NT) {                                       // row s + 2: its terms k <= s - 1, one step ahead of wavefront 0
                if (idx % (NW - 1) == wave - 1) tiny_pregather(Am, LD, s + 2, s, lane);
                ++idx;
            }
            // (the helper that pre-gathered row s + 2 must not ALSO solve tile (s + 2, s) before it: that tile's gather reads what the
            //  pre-gather leaves alone -- columns < s of row s + 2 -- and writes column s; the pre-gather writes columns s + 1, s + 2)
            for (int i = s + 2; i < NT; ++i, ++idx)
                if (idx % (NW - 1) == wave - 1) tiny_tile_solve(Am, LD, Dinv, s, i, false, lane);
            for (int e = 0; e < s; ++e, ++idx)
                if (idx % (NW - 1) == wave - 1) tiny_tile_solve(Am, LD, Dinv, s, e, true, lane);
        }
        __syncthreads();
        TSTAMP(18 + (s < 8 ? s : 7));
    }
    if (wave == 0 && lane == 0 && bad && *info_word == 0) *info_word = bad;
}
#endif

// element (i, j) of W = L^-T as tiny_chol_inv leaves it (i <= j by tiles; zero below the block diagonal)
__device__ __forceinline__ double tiny_w_elem(const double *Am, const int LD, double (*Dinv)[16][17], const int i, const int j) {
    const int ti = i >> 4, tj = j >> 4;
    if (ti < tj) return Am[(size_t)i * LD + j];
    if (ti == tj) return Dinv[ti][i & 15][j & 15];
    return 0.0;
}

__device__ __forceinline__ void tiny_tile_ij(int tau, int &i, int &j) {      // lower-triangular tile index -> (i, j), j <= i
    int ii = 0;
    while ((ii + 1) * (ii + 2) / 2 <= tau) ++ii;
    i = ii;
    j = tau - ii * (ii + 1) / 2;
}

// ---- closers ------------------------------------------------------------------------------------------------------------------
struct TinyCtx {
    int *fW, *cP, *fN, *c2, *cchain, *call, *abort_w;
};
__device__ __forceinline__ TinyCtx tiny_ctx(const TinyArgs &a, int u, int s) {
    TinyCtx c;
    c.fW = a.flags + 4 * u; c.cP = c.fW + 1; c.fN = c.fW + 2; c.c2 = c.fW + 3;
    c.cchain = a.flags + 4 * a.nunits + s;
    c.call = a.flags + 4 * a.nunits + a.S;
    c.abort_w = c.call + 1;
    return c;
}

__device__ __forceinline__ FinalizeArgs tiny_finalize_args(const TinyArgs &a) {
    FinalizeArgs fa{};
    fa.kind = a.kind; fa.branch = a.branch; fa.prior_type = a.prior_type; fa.shared_terms = a.shared_terms;
    fa.T = a.T; fa.D = a.D; fa.P = a.P; fa.M = a.M; fa.Ydim = a.Ydim; fa.Dl = a.Dl; fa.d_begin = a.d_begin; fa.S = a.S;
    fa.Z = a.Z; fa.U = a.U; fa.logvar = a.logvar; fa.loglen = a.loglen; fa.log_Q = a.log_Q; fa.CC = a.CC; fa.DD = a.DD;
    fa.log_Rchols = a.logR; fa.chain_terms = a.chain_terms; fa.hterms = a.hterms; fa.route = 0; fa.kterms = nullptr;
    fa.whitened = 0; fa.trpart = nullptr; fa.ntiles = 0; fa.fsq_from_trpart = 0; fa.chain_nll = a.chain_nll;
    fa.out_terms = a.out_terms; fa.info = a.info; fa.ninfo = a.Dl + a.nunits;
    return fa;
}

// The likelihood gradients of chain s and the transition-prior part of dlog_Q (shared_partials_kernel): functions of the inputs only,
// formed by the chain's first head while it waits for its strips.
template <int NW>
__device__ __noinline__ void tiny_chain_part(const TinyArgs &a_mem, const int s, double *red) {
    const TinyArgs a = a_mem;       // private copy: `a_mem` lives in device memory, and behind a reference the compiler must re-read every
                                    // pointer of it after every store (two dependent trips to L2 per operand instead of one)
    constexpr int NTHR = 64 * NW;
    const int tid = threadIdx.x;
    const int T = a.T, D = a.D, Dl = a.Dl, J = a.Ydim;
    const double *Xs = a.X + (size_t)s * (T + 1) * D;
    const double Tn = (double)T;
        double *cp = a.chain_part + (size_t)s * a.sp_stride;
        for (int item = 0; item < D * J + 2 * J + Dl; ++item) {
            double v[1] = {0.0};
            if (item < D * J + 2 * J) {
                if (a.shared_terms) {
                    const int j = (item < D * J) ? item % J : (item - D * J) % J;
                    const int kind = (item < D * J) ? 0 : ((item < D * J + J) ? 1 : 2), d = item / J;
                    const double R = exp(a.logR[j]);
                    for (int t = tid; t < T; t += NTHR) {
                        double ym = a.DD[j];
                        for (int dd = 0; dd < D; ++dd) ym += Xs[(size_t)(t + 1) * D + dd] * a.CC[(size_t)dd * J + j];
                        const double r = (a.Y[(size_t)t * J + j] - ym) / R;
                        if (kind == 0) v[0] += Xs[(size_t)(t + 1) * D + d] * (r / R);
                        else if (kind == 1) v[0] += r / R;
                        else v[0] += r * r - 1.0;
                    }
                }
            } else {
                // (explicit-U branch: the transition term is -1/2 alpha r^2 + T/2 log alpha with r = delta - mean, all of it in the
                //  unit's dl/dalpha -- tiny_kernel, head -- nothing here)
                const int d = a.d_begin + (item - D * J - 2 * J);
                const double Q = exp(a.log_Q[d]);
                if (a.branch == 1) for (int t = tid; t < T; t += NTHR) {
                    const double dlt = Xs[(size_t)(t + 1) * D + d] - Xs[(size_t)t * D + d];
                    v[0] += 0.5 - 0.5 * dlt * dlt / Q;
                }
            }
            tiny_sum<1, NW>(v, red);
            if (tid == 0) cp[item] = (item < D * J + 2 * J) ? -v[0] / Tn : v[0] / Tn;
        }
}

// Everything of unit u is in memory.  The workgroup that completes a chain forms that chain's sums; the one that completes the
// launch assembles the result.
template <int NW>
__device__ __noinline__ void tiny_unit_done(const TinyArgs &a_mem, const int u, double *lds, const int l_ctl, const int l_red, const int l_vec, const int l_mat) {
    const TinyArgs a = a_mem;       // (private copy: see tiny_chain_part)
    // The LDS offsets arrive BY VALUE rather than as a reference to the kernel's TinyLds: NO pointer to the kernel's private memory crosses
    // a call in this file (round 4's silent corruption was a by-value argument block read by a callee through such a pointer; its
    // writer was never identified on the hardware -- DESIGN.md section 13 has what is known -- so the pattern itself is gone).
    TinyLds L{};
    L.ctl = l_ctl; L.red = l_red; L.vec = l_vec; L.mat = l_mat;
    constexpr int NTHR = 64 * NW;
    const int tid = threadIdx.x;
    const int s = u / a.Dl;
    int *slot = reinterpret_cast<int *>(lds + L.ctl);
    double *red = lds + L.red;
    const TinyCtx cx = tiny_ctx(a, u, s);
    const int T = a.T, D = a.D, P = a.P, M = a.M, Mp = a.Mp, Dl = a.Dl, J = a.Ydim, nst = a.nstrips;
    const int Tp = nst * a.SR;
    const int pstride = tiny_pstride(Mp), qstride = tiny_qstride(Mp), ntl = tiny_ntl(a.NT);
    if (a.grad) {
        // ---- unit totals of the backward pass: dl/dZ (K_fu side + K_uu side), dl/dloglengthscales, dl/dlogvariance part --------
        // (every sum over strips / units / chains below is unrolled by eight: not unrolled, each iteration waited for its own load --
        //  one L2 latency per addend, 20 us for the 40 units of ten chains)
        const int dl = u % Dl, dg = a.d_begin + dl;
        const double *Qu = a.Qp + (size_t)u * nst * qstride;
        double *uo = a.unit_out + (size_t)u * (M * P + P + 2);
        double *cs = lds + L.vec;                    // [Mp] column sums of E
        double *etx = lds + L.mat;                   // [P][Mp]
        for (int idx = tid; idx < (P + 1) * Mp; idx += NTHR) {
            double v = 0.0;
#pragma unroll 8
            for (int st = 0; st < nst; ++st) v += Qu[(size_t)st * qstride + idx];
            if (idx < Mp) cs[idx] = v;
            else etx[idx - Mp] = v;
        }
        __syncthreads();
        double acc[TPP + 1];
#pragma unroll
        for (int p = 0; p <= TPP; ++p) acc[p] = 0.0;
        // (the per-component sums over strips / row blocks go to threads at the TOP of the workgroup -- another wavefront than the
        //  rows of Z whenever M leaves one free -- so that their two rounds of loads run beside the m loop's instead of behind it)
        const int tp = NTHR - 1 - tid;
        if (tp < P) {
            const double len = exp(a.loglen[(size_t)dg * P + tp]), inv2 = 1.0 / (len * len);
            double v = 0.0, k2 = 0.0;
#pragma unroll 8
            for (int st = 0; st < nst; ++st) v += Qu[(size_t)st * qstride + 16 * Mp + tp];       // sum_t r_t x_tp^2
#pragma unroll 8
            for (int rb = 0; rb < a.NT; ++rb) k2 += a.kuu_part[((size_t)u * a.NT + rb) * (TPP + 1) + tp];
#pragma unroll
            for (int p = 0; p < TPP; ++p)
                if (tp == p) acc[p] += v * inv2 + k2;                        // (compile-time indices: the array stays in registers)
        }
        if (tp == 8) {
            double v = 0.0, k2 = 0.0;
#pragma unroll 8
            for (int st = 0; st < nst; ++st) v += Qu[(size_t)st * qstride + 16 * Mp + 8];         // sum E
#pragma unroll 8
            for (int rb = 0; rb < a.NT; ++rb) k2 += a.kuu_part[((size_t)u * a.NT + rb) * (TPP + 1) + TPP];
            acc[TPP] = v + k2;
        }
        double inv2p[TPP];
#pragma unroll
        for (int p = 0; p < TPP; ++p) {
            const double len = (p < P) ? exp(a.loglen[(size_t)dg * P + p]) : 1.0;
            inv2p[p] = 1.0 / (len * len);
        }
        for (int m = tid; m < M; m += NTHR) {
#pragma unroll
            for (int p = 0; p < TPP; ++p)
                if (p < P) {
                    const double inv2 = inv2p[p];
                    const double z = a.Z[(size_t)m * P + p], e = etx[p * Mp + m];
                    uo[m * P + p] = (e - z * cs[m]) * inv2 + a.dz2[((size_t)u * Mp + m) * TPP + p];
                    acc[p] += (-2.0 * e * z + cs[m] * z * z) * inv2;
                }
        }
        tiny_sum<TPP + 1, NW>(acc, red);
#pragma unroll
        for (int p = 0; p < TPP; ++p)
            if (tid == p && p < P) uo[M * P + p] = acc[p];
        if (tid == 0) uo[M * P + P] = acc[TPP];
    }
    TSTAMP(12);
    if (tiny_arrive(cx.cchain, slot) != Dl - 1) return;
    TSTAMP(13);
    // ---- chain s is complete: its likelihood / transition / trace sums (chain_reduce_kernel), prior_x_0 ------------------------
    {
        double v[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i < Dl * nst; i += NTHR) {
            const double *sc = a.Pp + ((size_t)s * Dl * nst + i) * pstride + ntl * 256 + Mp;
            v[0] += sc[2]; v[1] += sc[0]; v[2] += sc[1];
        }
        tiny_sum<3, NW>(v, red);
        if (tid == 0) {
            const double *Xs = a.X + (size_t)s * (T + 1) * D;
            double px0 = 0.0;
            for (int d = 0; d < D; ++d) px0 += Xs[d] * Xs[d];
            double *o = a.chain_terms + (size_t)s * 8;
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = -px0 / 2.0;
        }
    }
    if (a.grad) {
        // ---- dl/dX of the chain (dx_kernel) and the likelihood gradients (shared_partials_kernel) ------------------------------
        TSTAMP(30);
        // (dX holds the input-only part already: the strips formed it while they waited for W.  What the backward pass adds, per
        //  element: -1/T of the sum over the local kernels of dl/dx_comb (rows t < T) and, for the unit's own latent dim, +-1/T
        //  dl/ddelta of the two transitions the state takes part in.  Clamped unconditional loads, four elements per thread in flight:
        //  as first written -- everything in this one loop, loads behind branches -- it was 44 us of dependent trips to L2.)
        double *__restrict__ gX = a.dX + (size_t)s * (T + 1) * D;
        const double *__restrict__ dxc = a.dxc + (size_t)s * Dl * Tp * (P + 1);
        const double iT = 1.0 / (double)T, iS = 1.0 / (double)a.S_total;
        const int row = P + 1, nel = (T + 1) * D, d_begin = a.d_begin;
        auto elem = [&](const int idx) -> double {
            const int t = idx / D, d = idx % D;
            const int tc = (t < T) ? t : T - 1, tm = (t > 0) ? t - 1 : 0;
            const int dl_own = d - d_begin;
            const bool own = dl_own >= 0 && dl_own < Dl;
            const double *__restrict__ du = dxc + (size_t)(own ? dl_own : 0) * Tp * row;
            const double g0 = gX[idx], dn = du[(size_t)tc * row + P], dp = du[(size_t)tm * row + P];
            double dcomb = 0.0;
#pragma unroll 4
            for (int dl = 0; dl < Dl; ++dl) dcomb += dxc[((size_t)dl * Tp + tc) * row + d];
            double g = g0;
            g -= (t < T) ? dcomb * iT : 0.0;
            g += (own && t < T) ? dn * iT : 0.0;
            g -= (own && t > 0) ? dp * iT : 0.0;
            return g * iS;
        };
        for (int base = tid; base < nel; base += 4 * NTHR) {
            double g4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = base + q * NTHR;
                g4[q] = elem(idx < nel ? idx : nel - 1);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = base + q * NTHR;
                if (idx < nel) gX[idx] = g4[q];
            }
        }
    }
    TSTAMP(14);
    if (tiny_arrive(cx.call, slot) != a.S - 1) return;
    TSTAMP(15);
    // ---- the launch is complete: nll assembly, shared-parameter gradients, flags re-armed ----------------------------------------
    {
        const FinalizeArgs fa = tiny_finalize_args(a);
        double sm[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) sm[i] = a.prior_sums[i];                 // formed by head 0 while it waited for its strips
        finalize_assemble<NTHR>(fa, reinterpret_cast<double(*)[10]>(red), sm);
        __syncthreads();
    }
    if (a.grad) {
        const double Tn = (double)T, Sn = (double)a.S_total, wgt = (double)a.S / Sn;
        const int S = a.S, ustride = M * P + P + 2;
        for (int idx = tid; idx < M * P; idx += NTHR) {                       // grad_dz_kernel
            double acc = 0.0;
#pragma unroll 8
            for (int uu = 0; uu < a.nunits; ++uu) acc += a.unit_out[(size_t)uu * ustride + idx];
            double g = -acc / Tn / Sn;
            if (a.shared_terms && a.prior_type == 1) g += wgt * a.Z[idx] / Tn;
            a.dZ[idx] = g;
        }
        for (int idx = tid; idx < D * P; idx += NTHR) {                       // grad_finalize_kernel
            const int d = idx / P, p = idx % P, dl = d - a.d_begin;
            double g = 0.0;
            if (dl >= 0 && dl < Dl) {
                double acc = 0.0;
#pragma unroll 8
                for (int ss = 0; ss < S; ++ss) acc += a.unit_out[(size_t)(ss * Dl + dl) * ustride + M * P + p];
                g = (a.kind != 0) ? 0.0 : -acc / Tn / Sn + wgt * a.loglen[idx] / Tn;
            }
            a.dloglen[idx] = g;
        }
        for (int d = tid; d < D; d += NTHR) {
            const int dl = d - a.d_begin;
            double gv = 0.0, gq = 0.0;
            if (dl >= 0 && dl < Dl) {
                const double alpha = 1.0 / exp(a.log_Q[d]), s2 = exp(a.logvar[d]);
                double v0 = 0.0, v1 = 0.0, v2 = 0.0;
#pragma unroll 8
                for (int ss = 0; ss < S; ++ss) {
                    const size_t bb = (size_t)ss * Dl + dl;
                    v0 += a.unit_out[bb * ustride + M * P + P] - 0.5 * alpha * s2 * Tn;
                    v1 += a.uterms[bb * 8] * (-alpha);
                    v2 += a.chain_part[(size_t)ss * a.sp_stride + D * J + 2 * J + dl];
                }
                gv = -v0 / Tn / Sn + wgt * (a.logvar[d] - (a.kind == 0 ? LOG_PRIOR_VARIANCE_SE : LOG_PRIOR_VARIANCE_LIN)) / Tn;
                gq = -v1 / Tn / Sn + v2 / Sn + wgt * a.log_Q[d] / Tn;
            }
            a.dlogvar[d] = gv;
            a.dlogQ[d] = gq;
        }
        if (a.branch == 0)                                                    // dU (explicit-U branch): -du / T per chain + prior_U (dgp_model.py:134-135)
            for (int idx = tid; idx < M * D; idx += NTHR) {
                const int m = idx / D, d = idx % D, dl = d - a.d_begin;
                double g = 0.0;
                if (dl >= 0 && dl < Dl) {
                    double acc = 0.0;
#pragma unroll 8
                    for (int ss = 0; ss < S; ++ss) acc += a.du_unit[(size_t)(ss * Dl + dl) * Mp + m];
                    g = -acc / Tn / Sn + wgt * a.U[idx] / Tn;
                }
                a.dU[idx] = g;
            }
        for (int idx = tid; idx < D * J + J + J * J; idx += NTHR) {
            double g = 0.0;
            if (a.shared_terms) {
                if (idx < D * J + 2 * J) {
                    double acc = 0.0;
#pragma unroll 8
                    for (int ss = 0; ss < S; ++ss) acc += a.chain_part[(size_t)ss * a.sp_stride + idx];
                    acc /= Sn;
                    if (idx < D * J) g = acc + wgt * a.CC[idx] / Tn;
                    else if (idx < D * J + J) g = acc + wgt * a.DD[idx - D * J] / Tn;
                    else g = acc + wgt * a.logR[idx - D * J - J] / Tn;
                } else g = wgt * a.logR[idx - D * J - J] / Tn;                 // only row 0 of log_Rchols enters the likelihood
            }
            if (idx < D * J) a.dCC[idx] = g;
            else if (idx < D * J + J) a.dDD[idx - D * J] = g;
            else a.dlogR[idx - D * J - J] = g;
        }
    }
    for (int i = tid; i < 4 * a.nunits + a.S + 1; i += NTHR) a.flags[i] = 0;      // the abort word stays as it is (0 on this path)
}

// K_uu side of the backward pass, one 16-row block rb of unit u (one workgroup, after the head's flag N):
//   Psi = 1/2 W N2 W^T (dl/dK_uu),  N2 = N - (H - I),  E_u = Psi o K(Z,Z),  its row sums and E_u Z  ->  rows of dl/dZ, partials of
//   dl/dloglengthscales, dl/dlogvariance.  ZO = [1 | Z] rows in LDS; three 16-row patches of the strip matrix area.
template <int NW>
__device__ __noinline__ void tiny_kuu_rows(const TinyArgs &a_mem, const int u, const int rb, double *Ks, const double *ZO, const double *ilen,
                                           const double var) {
    const TinyArgs a = a_mem;       // (private copy: see tiny_chain_part)
    constexpr int NTHR = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = a.P, M = a.M, Mp = a.Mp, NT = a.NT, LD = Mp + 1;
    const size_t msq = (size_t)Mp * Mp;
    const double *Wu = a.Wg + (size_t)u * msq, *Wtu = a.Wt + (size_t)u * msq, *Nu = a.Nw + (size_t)u * msq;
        double *T1 = Ks;                                                      // [16][LD]
        double *Eu = Ks + (size_t)16 * LD;                                    // [16][LD]
        double *Wr = Ks + (size_t)32 * LD;                                    // [16][LD] rows rb of W
        const double *Hu = a.Hs + (size_t)u * msq;
        for (int e = tid; e < 16 * Mp; e += NTHR) {
            const int m = e / Mp, j = e - m * Mp;
            Wr[(size_t)m * LD + j] = (j >= 16 * rb) ? Wu[(size_t)(16 * rb + m) * Mp + j] : 0.0;
        }
        __syncthreads();
        for (int j = wave; j < NT; j += NW) {                                 // T1 = W(rb, :) N2
            d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
            // every B fragment of the tile's k range is requested before the first product: one L2 round trip per tile
            double nn[TNT][4], nh[TNT][4];
#pragma unroll
            for (int kk = 0; kk < TNT; ++kk)
                if (rb + kk < NT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const size_t idx = (size_t)(16 * (rb + kk) + 4 * t + lk) * Mp + 16 * j + lr;
                        nn[kk][t] = Nu[idx]; nh[kk][t] = (a.branch == 1) ? Hu[idx] : 0.0;      // (explicit U: Nw holds 2 Phi, the Cholesky adjoint's middle factor)
                    }
                }
#pragma unroll
            for (int kk = 0; kk < TNT; ++kk)
                if (rb + kk < NT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double av = Wr[(size_t)lr * LD + 16 * (rb + kk) + 4 * t + lk];
                        if (t & 1) a1 = mfma_f64(av, nn[kk][t] - nh[kk][t], a1);
                        else a0 = mfma_f64(av, nn[kk][t] - nh[kk][t], a0);
                    }
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) T1[(size_t)(lk + 4 * r) * LD + 16 * j + lr] = a0[r] + a1[r];
        }
        __syncthreads();
        for (int j = wave; j < NT; j += NW) {                                 // Psi(rb, j) = 1/2 T1 W^T(:, j),  E_u = Psi o K_uu
            d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
            double wb[TNT][4];
#pragma unroll
            for (int kk = 0; kk < TNT; ++kk)
                if (j + kk < NT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) wb[kk][t] = Wtu[(size_t)(16 * (j + kk) + 4 * t + lk) * Mp + 16 * j + lr];
                }
#pragma unroll
            for (int kk = 0; kk < TNT; ++kk)
                if (j + kk < NT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double av = T1[(size_t)lr * LD + 16 * (j + kk) + 4 * t + lk];
                        if (t & 1) a1 = mfma_f64(av, wb[kk][t], a1);
                        else a0 = mfma_f64(av, wb[kk][t], a0);
                    }
                }
            const int gj = 16 * j + lr;
            double zj[8], zj2 = 0.0;
#pragma unroll
            for (int p = 0; p < 8; ++p) { zj[p] = (p < P) ? ZO[(size_t)gj * 16 + 1 + p] / ilen[p] : 0.0; zj2 += zj[p] * zj[p]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 16 * rb + lk + 4 * r;
                double kv = 0.0;
                if (gi < M && gj < M) {                                        // K(Z,Z) without the jitter (ffvd_grad_oracle.py: Psi * Kuu)
                    double dot = 0.0, zi2 = 0.0;
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const double zi = (p < P) ? ZO[(size_t)gi * 16 + 1 + p] / ilen[p] : 0.0;
                        dot += zi * zj[p]; zi2 += zi * zi;
                    }
                    kv = kernel_value<0>(dot, zi2, zj2, var);
                }
                Eu[(size_t)(lk + 4 * r) * LD + gj] = 0.5 * (a0[r] + a1[r]) * kv;
            }
        }
        __syncthreads();
        if (wave == 0) {
            d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
            for (int k = 0; k < NT; ++k) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double av = Eu[(size_t)lr * LD + 16 * k + 4 * t + lk], bw = ZO[(size_t)(16 * k + 4 * t + lk) * 16 + lr];
                    if (t & 1) a1 = mfma_f64(av, bw, a1);
                    else a0 = mfma_f64(av, bw, a0);
                }
            }
            double sll = 0.0;
            const double len = (lr >= 1 && lr <= P) ? ilen[lr - 1] : 1.0, inv2 = 1.0 / (len * len);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = a0[r] + a1[r];
                const double ru = __shfl(v, lane & 48);
                const int gi = 16 * rb + lk + 4 * r;
                if (lr >= 1 && lr <= P) {
                    const double z = ZO[(size_t)gi * 16 + lr];
                    a.dz2[((size_t)u * Mp + gi) * TPP + lr - 1] = -2.0 * (z * ru - v) * inv2;       // E_u symmetric: both roles of Z
                    sll += 2.0 * (ru * z * z - z * v) * inv2;
                } else if (lr == 0) sll += v;
            }
            sll += __shfl_xor(sll, 16);
            sll += __shfl_xor(sll, 32);
            if (lk == 0 && lr <= P) a.kuu_part[((size_t)u * NT + rb) * (TPP + 1) + (lr == 0 ? TPP : lr - 1)] = sll;
        }
        __syncthreads();
    }

// Without side workgroups the NT row blocks of the K_uu side go to the nst strips (block rb to strip rb mod nst) -- except hn blocks
// [h0, h0 + hn) that the head takes: it idles from the moment it has published N, while a strip still has its phase 2 ahead.
__device__ __forceinline__ void tiny_kuu_split(const int NT, const int nst, int &hn, int &h0) {
    hn = NT - nst;
    if (hn > 3) hn = 3;
    if (hn < 0 || NT < 3) hn = 0;               // (the head's matrix area must hold the three 16-row patches: Mp >= 48)
#ifdef FFVD_TINY_NO_HEAD_HELP
    hn = 0;
#endif

    h0 = nst;
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------
// B fragments of one k step of a row-panel product C(16 rows x Mp) += A(16 x 16 k-block) B(k-block, :), straight from L2 into
// registers one step ahead of their use: b[j][t] = B[(16 k + 4 t + lk)][16 j + lr] for the column tiles j in [jlo, jhi).
__device__ __forceinline__ void tiny_load_b(double (&b)[TNT][4], const double *B, const int Mp, const int k, const int jlo, const int jhi,
                                            const int lr, const int lk) {
#pragma unroll
    for (int j = 0; j < TNT; ++j)
        if (j >= jlo && j < jhi) {
#pragma unroll
            for (int t = 0; t < 4; ++t) b[j][t] = B[(size_t)(16 * k + 4 * t + lk) * Mp + 16 * j + lr];
        }
}
__device__ __forceinline__ void tiny_mma_row(d4 (&acc)[TNT], const double *Arow /* &A[16w + lr][0] in LDS */, const double (&b)[TNT][4],
                                             const int k, const int jlo, const int jhi, const int lk) {
    double av[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) av[t] = Arow[16 * k + 4 * t + lk];
#pragma unroll
    for (int j = 0; j < TNT; ++j)
        if (j >= jlo && j < jhi) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[j] = mfma_f64(av[t], b[j][t], acc[j]);
        }
}
// acc(16 x Mp) = A(16 x Mp, LDS rows) B(Mp x Mp, L2), B's tile (k, j) taken for j in [jlo(k), jhi(k)):
//   MODE 0: all j;  MODE 1: j >= k (B upper triangular by tiles);  MODE 2: j <= k (lower).
template <int MODE>
__device__ __forceinline__ void tiny_row_gemm(d4 (&acc)[TNT], const double *Arow, const double *B, const int Mp, const int NT,
                                              const int lr, const int lk) {
    double b0[TNT][4], b1[TNT][4];
    auto lo = [&](int k) { return MODE == 1 ? k : 0; };
    auto hi = [&](int k) { return MODE == 2 ? k + 1 : NT; };
    tiny_load_b(b0, B, Mp, 0, lo(0), hi(0), lr, lk);
    for (int k = 0; k < NT; k += 2) {
        if (k + 1 < NT) tiny_load_b(b1, B, Mp, k + 1, lo(k + 1), hi(k + 1), lr, lk);
        tiny_mma_row(acc, Arow, b0, k, lo(k), hi(k), lk);
        if (k + 1 < NT) {
            if (k + 2 < NT) tiny_load_b(b0, B, Mp, k + 2, lo(k + 2), hi(k + 2), lr, lk);
            tiny_mma_row(acc, Arow, b1, k + 1, lo(k + 1), hi(k + 1), lk);
        }
    }
}

// The argument block lives in DEVICE memory and every role reads its fields through one pointer (scalar loads): passed by value, the
// 50-field block either sat in SGPRs for the whole kernel (400+ SGPR spills) or -- for a callee that is not inlined -- in a
// private-memory copy that did not survive the spill traffic of the K_uu-side code (wrong chain sums, then memory faults, once a
// head ran that code before closing its unit; found by diffing the scratch block of two builds, tools/dbg_cmp.py).
// BR: the branch as a compile-time constant (1 collapsed U, 0 explicit U) -- the collapsed branch's code is what it was before the
// explicit-U roles were written into the same body (as a run-time switch they cost it ten spilled registers).
template <int NW, int BR>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void tiny_kernel(const TinyArgs *__restrict__ ap) {      // (<= 256 registers: the MFMAs take VGPR accumulators, no AGPR copies)
    const TinyArgs &a = *ap;
    constexpr int NTHR = 64 * NW, SR = 16 * NW;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, D = a.D, P = a.P, M = a.M, Mp = a.Mp, NT = a.NT, Dl = a.Dl, nst = a.nstrips;
    const int LD = Mp + 1, ntl = tiny_ntl(NT), pstride = tiny_pstride(Mp);
    const TinyLds L = tiny_lds(Mp, SR);
    int *slot = reinterpret_cast<int *>(lds + L.ctl);
    int *tab = reinterpret_cast<int *>(lds + L.tab);
    double *ilen = lds + L.tab + 18;                                         // [8] lengthscales (1 where p >= P)
    double *red = lds + L.red;
    double *Am = lds + L.mat;
    // role-major virtual id: heads [0, nunits), strips behind them, K_uu-side workgroups last.  xcd_map: the hardware deals
    // consecutive workgroup ids to the 8 XCDs in turn, each with an L2 of its own -- unit u's workgroups take the ids = u (mod 8), so
    // that everything they hand each other stays in one L2; ids whose unit does not exist leave at once
    int vb = (int)blockIdx.x;
    if (a.xcd_map) {
        const int wpu = 1 + nst + (a.side ? NT : 0), xcd = vb & 7, slot = vb >> 3;
        const int uu = (slot / wpu) * 8 + xcd, role = slot % wpu;
        if (uu >= a.nunits) return;
        vb = (role == 0) ? uu : ((role <= nst) ? a.nunits + uu * nst + (role - 1) : a.nunits * (1 + nst) + uu * NT + (role - 1 - nst));
    }
    const bool head = vb < a.nunits;
    const bool side = vb >= a.nunits * (1 + nst);                           // (only launched with a.side)
    const int u = head ? vb : (side ? (vb - a.nunits * (1 + nst)) / NT : (vb - a.nunits) / nst);
    const int strip = head ? 0 : (side ? (vb - a.nunits * (1 + nst)) % NT : (vb - a.nunits) % nst);
    const int s = u / Dl, dl = u % Dl, dg = a.d_begin + dl;
    const TinyCtx cx = tiny_ctx(a, u, s);
    const double var = exp(a.logvar[dg]);                                   // kernels_multi_output.py:157
    const double Qd = exp(a.log_Q[dg]), alpha = 1.0 / Qd;
    const size_t msq = (size_t)Mp * Mp;
    double *Wu = a.Wg + (size_t)u * msq, *Wtu = a.Wt + (size_t)u * msq;
    TSTAMP(0);
    for (int tau = tid; tau < ntl; tau += NTHR) {
        int ti, tj;
        tiny_tile_ij(tau, ti, tj);
        tab[tau] = ti | (tj << 8);
    }
    if (tid < 8) ilen[tid] = (tid < P) ? exp(a.loglen[(size_t)dg * P + tid]) : 1.0;      // :161
    if (head)                                                                          // the identity rows of the pivot chains (tiny_chain16)
        for (int e = tid; e < 16 * 17; e += NTHR) lds[L.sc + 16 * 17 + e] = ((e / 17) == (e % 17)) ? 1.0 : 0.0;

    if (head) {
        // This iteration's factorisation flags start at zero (a head is the only writer of its words), and the result starts as NaN:
        // a launch that is abandoned on a bounded wait never reaches the workgroup that writes the sums, and a collective caller
        // (ffvd_elbo_allreduce, no retry) must not find the previous iteration's finite values there.
        if (tid == 0) {
            a.info[Dl + u] = 0;
            if (s == 0) a.info[dl] = 0;
        }
        if (u == 0 && tid < 7) a.out_terms[tid] = __longlong_as_double(0x7ff8000000000000LL);
        // ======================================================================================================================
        // head, phase 0:  K = K(Z,Z) + jitter I  ->  L, W = L^-T                              (conditionals_multi_output.py:159-166)
        // ======================================================================================================================
        double (*Dinv)[16][17] = reinterpret_cast<double(*)[16][17]>(lds + L.dinv);
        double (*Sc)[17] = reinterpret_cast<double(*)[17]>(lds + L.sc);
        {
            double *zs = lds + L.dinv, *zz = zs + (size_t)Mp * 9;           // aliases Dinv: dead before the factorisation starts.  Row stride 9:
                                                                             // 16 lanes reading 16 consecutive rows hit 16 different banks
            __syncthreads();
            TSTAMP(16);
            {
                const int p = tid & 7;                                       // (NTHR is a multiple of 8: a thread keeps its component)
                const double len = ilen[p];
                for (int e = tid; e < Mp * 8; e += NTHR) {
                    const int m = e >> 3;
                    zs[m * 9 + p] = (m < M && p < P) ? a.Z[(size_t)m * P + p] / len : 0.0;   // :170
                }
            }
            __syncthreads();
            for (int m = tid; m < Mp; m += NTHR) {
                double sacc = 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) sacc += zs[m * 9 + p] * zs[m * 9 + p];
                zz[m] = sacc;
            }
            __syncthreads();
            TSTAMP(17);
            // lower-triangular tiles only.  A thread keeps its element position (r, cj) inside the tile; the inducing inputs of ITS column
            // in every tile column (z_j, |z_j|^2: NT x 9 doubles) go to registers once, the row's (z_i) once per tile row -- the loop
            // that read both from LDS per element was bound by the LDS pipe, not by the exp (tools/probes/lat_probe.hip: 5.7 us per
            // pass of which the exp is 0.8); 256 threads = one tile at a time, 512 = two neighbours
            {
                const int e = tid & 255, r = e >> 4, cj = e & 15, half = tid >> 8, step = NTHR >> 8;
                double zj[TNT][8], zzj[TNT];
#pragma unroll
                for (int tj = 0; tj < TNT; ++tj) {
                    const int j = (tj < NT) ? 16 * tj + cj : cj;
#pragma unroll
                    for (int p = 0; p < 8; ++p) zj[tj][p] = zs[j * 9 + p];
                    zzj[tj] = zz[j];
                }
#pragma unroll
                for (int ti = 0; ti < TNT; ++ti) {
                    if (ti < NT) {
                        const int i = 16 * ti + r;
                        double zi[8];
#pragma unroll
                        for (int p = 0; p < 8; ++p) zi[p] = zs[i * 9 + p];
                        const double zzi = zz[i];
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj) {
                            if ((tj & (step - 1)) != half) continue;         // (uniform per wavefront)
                            const int j = 16 * tj + cj;
                            double dot = 0.0;
#pragma unroll
                            for (int p = 0; p < 8; ++p) dot += zi[p] * zj[tj][p];
                            double v = kernel_value<0>(dot, zzi, zzj[tj], var);
                            if (i == j) v += a.jitter;                       // :159
                            if (i >= M || j >= M) v = (i == j) ? 1.0 : 0.0;
                            Am[(size_t)i * LD + j] = v;
                        }
                    }
                }
            }
            __syncthreads();
        }
        TSTAMP(1);
        tiny_chol_inv<NW>(Am, LD, NT, Dinv, Sc, a.info + (s == 0 ? dl : Dl + u));      // (every chain's head factorises the same K_uu)
        TSTAMP(2);
        // W (tiles on and above the block diagonal) and W^T (on and below) to L2: the strips read nothing else of them
        for (int j = lane; j < Mp; j += 64) {
            const int tj = j >> 4;
#pragma unroll 4
            for (int i = wave; i < 16 * tj; i += NW) Wu[(size_t)i * Mp + j] = Am[(size_t)i * LD + j];                 // W(i, j), tile row above the diagonal
#pragma unroll 4
            for (int i = 16 * tj + wave; i < 16 * tj + 16; i += NW) {
                Wu[(size_t)i * Mp + j] = Dinv[tj][i & 15][j & 15];
                if (a.grad) Wtu[(size_t)i * Mp + j] = Dinv[tj][j & 15][i & 15];
            }
            if (a.grad) {                                                    // (only the backward pass reads W^T)
#pragma unroll 4
                for (int i = 16 * (tj + 1) + wave; i < Mp; i += NW) Wtu[(size_t)i * Mp + j] = Am[(size_t)j * LD + i];     // W^T(i, j) = W(j, i)
            }
        }
        TSTAMP(26);
#ifdef FFVD_TINY_TEST_STALL
        if (u != 0)
#endif
        tiny_publish(cx.fW, 1);
        TSTAMP(3);
        if (u == 0) {                                                         // the ten parameter-only sums of the nll assembly, while the strips work
            const FinalizeArgs fa = tiny_finalize_args(a);
            double sm[10];
            finalize_priors<NTHR>(fa, reinterpret_cast<double(*)[10]>(red), sm);
#pragma unroll
            for (int i = 0; i < 10; ++i)
                if (tid == i) a.prior_sums[i] = sm[i];
            __syncthreads();
        }
        if (a.grad && dl == 0) tiny_chain_part<NW>(a, s, red);                  // likelihood gradients of the chain: inputs only
        // ======================================================================================================================
        // head, phase 1:  H = I + F^T F / Q, b = delta^T F / Q, log|H|, b H^-1 b^T                           (:246-254)
        // ======================================================================================================================
        if (tiny_wait(cx.cP, nst, cx.abort_w, slot) < 0) {
            if (tid == 0) a.info[Dl + u] = -1;
            return;
        }
        TSTAMP(4);
        constexpr bool BA = BR == 0;
        if (BA && !a.grad) {                                                  // explicit U, forward: no H -- the strips' sums are the unit's terms
            if (tid == 0) { a.hterms[2 * u] = 0.0; a.hterms[2 * u + 1] = 0.0; }
            tiny_unit_done<NW>(a, u, lds, L.ctl, L.red, L.vec, L.mat);
            TSTAMP(8);
            return;
        }
        double *bv = lds + L.vec, *yv = bv + Mp, *wl = yv + Mp;
        const double *Pu = a.Pp + (size_t)u * nst * pstride;
        double *Hu = a.Hs + (size_t)u * msq;
        double trHm = 0.0;                                                    // tr(H - I), this thread's share
        {
            // the strips' partial blocks are contiguous vectors: [ntl tiles of 256 | Mp]; two elements per load, four loads
            // per strip in flight ahead of the adds
            const int total2 = (ntl * 256 + Mp) / 2;
            for (int base = 0; base < total2; base += 4 * NTHR) {
                double2 acc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = make_double2(0.0, 0.0);
                for (int st0 = 0; st0 < nst; st0 += 8) {                      // 32 loads in flight, then their adds in fixed order
                    double2 v[8][4];
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const double2 *src = reinterpret_cast<const double2 *>(Pu + (size_t)(st0 + g) * pstride);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int i2 = base + q * NTHR + tid;
                            v[g][q] = (st0 + g < nst && i2 < total2) ? src[i2] : make_double2(0.0, 0.0);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 8; ++g)
#pragma unroll
                        for (int q = 0; q < 4; ++q) { acc[q].x += v[g][q].x; acc[q].y += v[g][q].y; }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i2 = base + q * NTHR + tid;
                    if (i2 >= total2) continue;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int idx = 2 * i2 + h;
                        const double v = alpha * (h ? acc[q].y : acc[q].x);   // batch_size == Y_N: the factor Y_N / batch is 1 (:246)
                        if (idx >= ntl * 256) { bv[idx - ntl * 256] = v; continue; }          // :248
                        const int tt = tab[idx >> 8], e = idx & 255;
                        const int gi = 16 * (tt & 255) + (e >> 4), gj = 16 * (tt >> 8) + (e & 15);
                        if (!BA) Am[(size_t)gi * LD + gj] = v + ((gi == gj) ? 1.0 : 0.0);      // (explicit U: Am keeps L and W)
                        if (gi == gj) trHm += v;
                        if (a.grad) { Hu[(size_t)gi * Mp + gj] = v; Hu[(size_t)gj * Mp + gi] = v; }
                    }
                }
            }
        }
        __syncthreads();
        TSTAMP(5);
        if (BA) {
            // ==================================================================================================================
            // head, explicit-U backward (conditionals_multi_output.py:6-70 differentiated; oracle/ffvd_grad_oracle.py, nll_grad_explicit_u):
            //   bv = alpha F^T r = dl/du,   Hs = alpha F^T F,   beta = W u,   X = W (dl/dW)^T W = W Hs + beta bv^T,
            //   Lbar = -tril(X),   S = L^T Lbar,   2 Phi = strict lower of S + its transpose + diag(S)   (the Cholesky adjoint's middle
            //   factor: dl/dK_uu = W Phi W^T, formed row block by row block in tiny_kuu_rows),   dl/dalpha = Q (v0 + v1 + T / 2).
            // Am still holds L (lower triangle) and W (tiles above the diagonal, Dinv on it): no second factorisation in this branch.
            // ==================================================================================================================
            double *uv = yv;                                                  // u, zero padded
            for (int m = tid; m < Mp; m += NTHR) {
                uv[m] = (m < M) ? a.U[(size_t)m * D + dg] : 0.0;
                a.du_unit[(size_t)u * Mp + m] = bv[m];
            }
            __syncthreads();
            for (int i = tid; i < Mp; i += NTHR) {                            // beta = W u (W upper triangular by tiles)
                const int ti = i >> 4;
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc0 += Dinv[ti][i & 15][k] * uv[16 * ti + k];
                for (int k = 16 * (ti + 1); k + 1 < Mp; k += 2) { acc0 += Am[(size_t)i * LD + k] * uv[k]; acc1 += Am[(size_t)i * LD + k + 1] * uv[k + 1]; }
                wl[i] = acc0 + acc1;
            }
            __syncthreads();                                                  // (also: every thread's stores of Hs are issued; read back below by this workgroup)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            double *Lb = a.Nm2 + (size_t)u * msq;
            for (int tau = wave; tau < ntl; tau += NW) {                      // Lbar(ti, tj), tj <= ti
                const int ti = tab[tau] & 255, tj = tab[tau] >> 8;
                d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
                for (int k = ti; k < NT; ++k) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double av = (k == ti) ? Dinv[ti][lr][4 * t + lk] : Am[(size_t)(16 * ti + lr) * LD + 16 * k + 4 * t + lk];
                        const double bw = Hu[(size_t)(16 * k + 4 * t + lk) * Mp + 16 * tj + lr];
                        if (t & 1) a1 = mfma_f64(av, bw, a1);
                        else a0 = mfma_f64(av, bw, a0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = 16 * ti + lk + 4 * r, gj = 16 * tj + lr;
                    const double x = (a0[r] + a1[r]) + wl[gi] * bv[gj];
                    Lb[(size_t)gi * Mp + gj] = (gi >= gj) ? -x : 0.0;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            double *Nu = a.Nw + (size_t)u * msq;
            for (int tau = wave; tau < ntl; tau += NW) {                      // S(ti, tj) = sum_{k >= ti} L(k, ti)^T Lbar(k, tj), tj <= ti
                const int ti = tab[tau] & 255, tj = tab[tau] >> 8;
                d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
                for (int k = ti; k < NT; ++k) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double av = Am[(size_t)(16 * k + 4 * t + lk) * LD + 16 * ti + lr];      // L^T(ti, k)[lr][4 t + lk]; L's diagonal tiles are zero above the diagonal
                        const double bw = Lb[(size_t)(16 * k + 4 * t + lk) * Mp + 16 * tj + lr];
                        if (t & 1) a1 = mfma_f64(av, bw, a1);
                        else a0 = mfma_f64(av, bw, a0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = 16 * ti + lk + 4 * r, gj = 16 * tj + lr;
                    if (gi >= gj) {
                        const double v = a0[r] + a1[r];
                        Nu[(size_t)gi * Mp + gj] = v;
                        Nu[(size_t)gj * Mp + gi] = v;
                    }
                }
            }
            {
                double v[2] = {0.0, 0.0};                                    // the strips' transition and trace sums of this unit
                for (int st = tid; st < nst; st += NTHR) {
                    const double *sc = Pu + (size_t)st * pstride + ntl * 256 + Mp;
                    v[0] += sc[0]; v[1] += sc[1];
                }
                tiny_sum<2, NW>(v, red);
                if (tid == 0) {
                    a.hterms[2 * u] = 0.0; a.hterms[2 * u + 1] = 0.0;
                    a.uterms[(size_t)u * 8] = Qd * (v[0] + v[1] + 0.5 * (double)T);      // dl/dalpha = -1/2 sum r^2 - 1/2 sum var + T / (2 alpha)
                }
            }
            (void)trHm;
        } else {
        tiny_chol_inv<NW>(Am, LD, NT, Dinv, Sc, a.info + Dl + u);
        TSTAMP(6);
        for (int j = tid; j < Mp; j += NTHR) {                                // y = L_H^-1 b = W_H^T b
            const int tj = j >> 4;
            double acc0 = 0.0, acc1 = 0.0;
            for (int i = 0; i + 1 < 16 * tj; i += 2) { acc0 += Am[(size_t)i * LD + j] * bv[i]; acc1 += Am[(size_t)(i + 1) * LD + j] * bv[i + 1]; }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc0 += Dinv[tj][i][j & 15] * bv[16 * tj + i];
            yv[j] = acc0 + acc1;
        }
        __syncthreads();
        double quad;
        {
            double v[2] = {0.0, 0.0};
            for (int i = tid; i < Mp; i += NTHR) { v[0] += log(Am[(size_t)i * LD + i]); v[1] += yv[i] * yv[i]; }
            tiny_sum<2, NW>(v, red);
            quad = v[1];
            if (tid == 0) { a.hterms[2 * u] = 2.0 * v[0]; a.hterms[2 * u + 1] = v[1]; }     // logdet (:253), b H^-1 b^T (:254)
        }
        TSTAMP(7);
        if (!a.grad) {
            tiny_unit_done<NW>(a, u, lds, L.ctl, L.red, L.vec, L.mat);
            TSTAMP(8);
            return;
        }
        // ---- backward: H^-1 = W_H W_H^T, w = W_H y, N = I - H^-1 - w w^T, dl/dalpha ------------------------------------------------
        for (int i = tid; i < Mp; i += NTHR) {
            const int ti = i >> 4;
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc0 += Dinv[ti][i & 15][k] * yv[16 * ti + k];
            for (int k = 16 * (ti + 1); k + 1 < Mp; k += 2) { acc0 += Am[(size_t)i * LD + k] * yv[k]; acc1 += Am[(size_t)i * LD + k + 1] * yv[k + 1]; }
            wl[i] = acc0 + acc1;
            a.wv[(size_t)u * Mp + i] = acc0 + acc1;
        }
        __syncthreads();
        double *Nu = a.Nw + (size_t)u * msq;
        double trhinv = 0.0;
        for (int tau = wave; tau < ntl; tau += NW) {
            const int ti = tab[tau] & 255, tj = tab[tau] >> 8;
            d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
            for (int k = ti; k < NT; ++k) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double av = (k == ti) ? Dinv[ti][lr][4 * t + lk] : Am[(size_t)(16 * ti + lr) * LD + 16 * k + 4 * t + lk];
                    const double bw = (k == tj) ? Dinv[tj][lr][4 * t + lk] : Am[(size_t)(16 * tj + lr) * LD + 16 * k + 4 * t + lk];
                    if (t & 1) a1 = mfma_f64(av, bw, a1);
                    else a0 = mfma_f64(av, bw, a0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 16 * ti + lk + 4 * r, gj = 16 * tj + lr;
                const double hinv = a0[r] + a1[r];
                const double nw = ((gi == gj) ? 1.0 : 0.0) - hinv - wl[gi] * wl[gj];
                Nu[(size_t)gi * Mp + gj] = nw;
                if (ti != tj) Nu[(size_t)gj * Mp + gi] = nw;
                if (gi == gj) trhinv += hinv;
            }
        }
        {
            double v[4] = {trhinv, trHm, 0.0, 0.0};                          // tr H^-1, tr(H - I), w^T w, w^T b
            for (int i = tid; i < Mp; i += NTHR) { v[2] += wl[i] * wl[i]; v[3] += wl[i] * bv[i]; }
            tiny_sum<4, NW>(v, red);
            if (tid == 0) {
                // dl/dalpha = -1/2 tr(A^-1 G) + u^T g - 1/2 u^T G u - 1/2 (T sigma^2 - tr(K^-1 G)) in whitened variables
                // (ffvd_grad_oracle.py);  w^T (H - I) w = w^T b - w^T w because H w = b
                const double trAinvG = ((double)Mp - v[0]) / alpha, trKinvG = v[1] / alpha;
                a.uterms[(size_t)u * 8] = -0.5 * trAinvG + v[3] / alpha - 0.5 * (v[3] - v[2]) / alpha - 0.5 * ((double)T * var - trKinvG);
            }
            (void)quad;
        }
        }       // (collapsed branch)
        tiny_publish(cx.fN, 1);
        TSTAMP(8);
        if (a.side) return;
        // No workgroups of their own for the K_uu side (they did not fit): the head is free from here on while its strips run their
        // phase 2 -- it takes up to three of the row blocks (tiny_kuu_split), then counts in like a strip.
        {
            int hn, h0;
            tiny_kuu_split(NT, nst, hn, h0);
            if (hn > 0) {
                double *ZO = lds + L.dinv;                                    // [Mp][16]: the inverted diagonal tiles are dead
                __syncthreads();
                for (int e = tid; e < Mp * 16; e += NTHR) {
                    const int m = e >> 4, n = e & 15;
                    ZO[e] = (n == 0) ? 1.0 : ((m < M && n <= P) ? a.Z[(size_t)m * P + n - 1] : 0.0);
                }
                __syncthreads();
                for (int rb = h0; rb < h0 + hn; ++rb) tiny_kuu_rows<NW>(a, u, rb, Am, ZO, ilen, var);
            }
            if (tiny_arrive(cx.c2, slot) != nst) return;                      // nst strips + this head
            tiny_unit_done<NW>(a, u, lds, L.ctl, L.red, L.vec, L.mat);
        }
        return;
    }

    const int narrive2 = nst + (a.side ? NT : 1);                            // arrivals that complete the backward pass of a unit (side workgroups, or the head)
    if (side) {
        // ======================================================================================================================
        // side(u, rb): one 16-row block of the K_uu side of the backward pass, beside the strips' phase 2
        // ======================================================================================================================
        double *ZO = lds + L.zo;
        for (int e = tid; e < Mp * 16; e += NTHR) {
            const int m = e >> 4, n = e & 15;
            ZO[e] = (n == 0) ? 1.0 : ((m < M && n <= P) ? a.Z[(size_t)m * P + n - 1] : 0.0);
        }
        if (tiny_wait(cx.fN, 1, cx.abort_w, slot) < 0) return;
        TSTAMP(6);
        tiny_kuu_rows<NW>(a, u, strip, lds + L.mat, ZO, ilen, var);
        TSTAMP(9);
        if (tiny_arrive(cx.c2, slot) != narrive2 - 1) return;
        TSTAMP(10);
        tiny_unit_done<NW>(a, u, lds, L.ctl, L.red, L.vec, L.mat);
        TSTAMP(11);
        return;
    }
    // ==========================================================================================================================
    // strip, phase 0:  rows [t0, t0 + SR) of K_fu = K(x_comb, Z) in registers (accumulator layout) and LDS                   (:240)
    // ==========================================================================================================================
    const int t0 = strip * SR;
    double *Ks = lds + L.mat;
    double *xs = lds + L.xo, *xx = xs + (size_t)SR * 9;                      // (row stride 9: no bank conflicts between rows)
    double *zs = lds + L.zo, *zz = zs + (size_t)Mp * 9;
    double *dlt = lds + L.misc, *rowv = dlt + SR, *chn = rowv + SR;          // chn: [2][SR] per-row transition / likelihood terms
    const double *Xs = a.X + (size_t)s * (T + 1) * D;
    constexpr bool BA = BR == 0;                                              // explicit-U branch (regularizer, dgp_model.py:337-359)
    double *uvs = lds + L.vec;                                                // explicit U: the unit's column of U, zero padded (phase 2 reads it as `wl`)
    __syncthreads();
    if (BA)
        for (int m = tid; m < Mp; m += NTHR) uvs[m] = (m < M) ? a.U[(size_t)m * D + dg] : 0.0;
    {
        const int p = tid & 7;
        const double len = ilen[p];
        for (int e = tid; e < Mp * 8; e += NTHR) {
            const int m = e >> 3;
            zs[m * 9 + p] = (m < M && p < P) ? a.Z[(size_t)m * P + p] / len : 0.0;
        }
        for (int e = tid; e < SR * 8; e += NTHR) {
            const int r = e >> 3, t = t0 + r;
            double v = 0.0;
            if (t < T && p < P) v = ((p < D) ? Xs[(size_t)t * D + p] : a.ctrl[(size_t)t * a.C + (p - D)]) / len;
            xs[r * 9 + p] = v;
        }
    }
    for (int r = tid; r < SR; r += NTHR) {
        const int t = t0 + r;
        double dv = 0.0, tq = 0.0, tl = 0.0;
        if (t < T) {
            dv = Xs[(size_t)(t + 1) * D + dg] - Xs[(size_t)t * D + dg];     // :247
            const double q = dv / sqrt(Qd);                                  // dgp_model.py:283-284
            tq = BA ? 0.0 : -0.5 * (q * q);                                  // (explicit U: the residual r = delta - mean, once F is there)
            if (dl == 0 && a.shared_terms)
                for (int j = 0; j < a.Ydim; ++j) {
                    double ym = 0.0;
                    for (int d = 0; d < D; ++d) ym += Xs[(size_t)(t + 1) * D + d] * a.CC[(size_t)d * a.Ydim + j];      // likelihoods.py:76-79
                    ym += a.DD[j];
                    const double R = exp(a.logR[j]);                         // Rchols[0] = first row (dgp_model.py:250)
                    const double rr = (a.Y[(size_t)t * a.Ydim + j] - ym) / R;
                    tl += -0.5 * (rr * rr);
                }
        }
        dlt[r] = dv; chn[r] = tq; chn[SR + r] = tl;
    }
    __syncthreads();
    for (int m = tid; m < Mp; m += NTHR) {
        double sacc = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) sacc += zs[m * 9 + p] * zs[m * 9 + p];
        zz[m] = sacc;
    }
    for (int r = tid; r < SR; r += NTHR) {
        double sacc = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) sacc += xs[r * 9 + p] * xs[r * 9 + p];
        xx[r] = sacc;
    }
    __syncthreads();
    // (backward: each lane parks its K_fu elements in L2 until E = dl/dK_fu o K_fu needs them: 64 registers less in both phases)
    double *kst = a.grad ? a.Kst + (((size_t)u * nst + strip) * NTHR + tid) * 32 : nullptr;
    {
        double xr[4][8], xxr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + lk + 4 * r;
#pragma unroll
            for (int p = 0; p < 8; ++p) xr[r][p] = xs[row * 9 + p];
            xxr[r] = xx[row];
        }
#pragma unroll
        for (int c = 0; c < TNT; ++c) {
            if (c < NT) {
                const int m = 16 * c + lr;
                double zr[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) zr[p] = zs[m * 9 + p];
                const double zzm = zz[m];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * wave + lk + 4 * r;
                    double dot = 0.0;
#pragma unroll
                    for (int p = 0; p < 8; ++p) dot += xr[r][p] * zr[p];
                    double v = kernel_value<0>(dot, xxr[r], zzm, var);
                    if (m >= M || t0 + row >= T) v = 0.0;
                    if (kst) kst[c * 4 + r] = v;
                    Ks[(size_t)row * LD + m] = v;
                }
            }
        }
    }
    wave_lds_order();
    TSTAMP(1);
    if (a.grad) {
        // The part of dl/dX that is a function of the inputs alone -- likelihood (:248-250,264), transition prior (:283-284), prior_x_0
        // (:252) -- for this strip's rows, while it would otherwise wait for W: column dg (the unit's own latent dim; the chain's first
        // unit also takes the columns no unit of this handle owns), unscaled, straight into dX.  The chain's closer adds what comes
        // out of the backward pass (tiny_unit_done); as one loop there it was 22 us of dependent trips to L2 behind the last strip.
        const double iT = 1.0 / (double)T, iQT = iT / Qd;
        const int J = a.Ydim, ncols = 1 + ((dl == 0) ? D - Dl : 0);
        const int rows = SR + ((strip == nst - 1) ? 1 : 0);                   // (row T when T is a whole number of strips)
        double *gXs = a.dX + (size_t)s * (T + 1) * D;
        for (int item = tid; item < rows * ncols; item += NTHR) {
            const int r = item / ncols, ci = item % ncols, t = t0 + r;
            if (t > T) continue;
            const int d = (ci == 0) ? dg : ((ci - 1 < a.d_begin) ? ci - 1 : ci - 1 + Dl);
            double g = 0.0;
            if (ci == 0 && !BA) {                                             // (explicit U: the whole dl/ddelta comes out of phase 2)
                const double x1 = Xs[(size_t)t * D + d];
                if (t < T) g -= (Xs[(size_t)(t + 1) * D + d] - x1) * iQT;     // delta_t = x_{t+1} - x_t
                if (t > 0) g += (x1 - Xs[(size_t)(t - 1) * D + d]) * iQT;
            }
            if (a.shared_terms) {
                if (t > 0)
                    for (int j = 0; j < J; ++j) {
                        double ym = a.DD[j];
                        for (int dd = 0; dd < D; ++dd) ym += Xs[(size_t)t * D + dd] * a.CC[(size_t)dd * J + j];
                        const double R = exp(a.logR[j]);
                        g -= (a.Y[(size_t)(t - 1) * J + j] - ym) * (iT / (R * R)) * a.CC[(size_t)d * J + j];
                    }
                else g += Xs[d] * iT;
            }
            gXs[(size_t)t * D + d] = g;
        }
    }
    // ==========================================================================================================================
    // strip, phase 1:  F = K_fu W (:242), F^T F, F^T delta, sum F^2 (:255), chain-term partials
    // ==========================================================================================================================
    if (tiny_wait(cx.fW, 1, cx.abort_w, slot) < 0) return;
    TSTAMP(2);
    const double *Arow = Ks + (size_t)(16 * wave + lr) * LD;                  // this lane's A-operand row of the wavefront's 16 rows
    d4 facc[TNT];
#pragma unroll
    for (int j = 0; j < TNT; ++j) facc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    tiny_row_gemm<1>(facc, Arow, Wu, Mp, NT, lr, lk);
    wave_lds_order();
    {
        double rs[4] = {0.0, 0.0, 0.0, 0.0}, fm[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < TNT; ++j)
            if (j < NT) {
                const double uj = BA ? uvs[16 * j + lr] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    rs[r] += facc[j][r] * facc[j][r];
                    fm[r] += facc[j][r] * uj;                                  // fmean = A^T u (conditionals_multi_output.py:48)
                    Ks[(size_t)(16 * wave + lk + 4 * r) * LD + 16 * j + lr] = facc[j][r];      // F replaces K_fu in LDS (K_fu stays in registers)
                }
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) { rs[r] += __shfl_xor(rs[r], m); fm[r] += __shfl_xor(fm[r], m); }
            if (lr == 0) {
                const int row = 16 * wave + lk + 4 * r;
                rowv[row] = rs[r];
                if (BA) {
                    // the transition term of the explicit-U branch: logdensity_norm_diag(X[1:], mean + X[:-1], sqrt(Q)) (dgp_model.py:346-351);
                    // from here on `dlt` holds the residual r_t = delta_t - mean_t (F^T r, r u^T and dl/ddelta below)
                    const double res = (t0 + row < T) ? dlt[row] - fm[r] : 0.0;
                    const double q = res / sqrt(Qd);
                    dlt[row] = res;
                    chn[row] = -0.5 * (q * q);
                }
            }
        }
    }
    __syncthreads();
    TSTAMP(3);
    double *Pu = a.Pp + ((size_t)u * nst + strip) * pstride;
    const bool need_mm = !BA || a.grad;                                       // (explicit U, forward only: the head reads the three sums alone)
    if (need_mm)
    for (int tau = wave; tau < ntl; tau += NW) {
        const int ti = tab[tau] & 255, tj = tab[tau] >> 8;
        d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
#pragma unroll 4
        for (int q = 0; q < SR / 4; ++q) {
            const double av = Ks[(size_t)(4 * q + lk) * LD + 16 * ti + lr], bw = Ks[(size_t)(4 * q + lk) * LD + 16 * tj + lr];
            if (q & 1) a1 = mfma_f64(av, bw, a1);
            else a0 = mfma_f64(av, bw, a0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Pu[tau * 256 + (lk + 4 * r) * 16 + lr] = a0[r] + a1[r];
    }
    TSTAMP(16);
    if (need_mm)
    for (int m = tid; m < Mp; m += NTHR) {
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
        for (int r = 0; r < SR; r += 2) { acc0 += Ks[(size_t)r * LD + m] * dlt[r]; acc1 += Ks[(size_t)(r + 1) * LD + m] * dlt[r + 1]; }
        Pu[ntl * 256 + m] = acc0 + acc1;
    }
    TSTAMP(17);
    {
        double v[3] = {0.0, 0.0, 0.0};          // transition, trace, likelihood partial sums of these rows (chain_reduce_kernel)
        for (int r = tid; r < SR; r += NTHR) {
            v[0] += chn[r];
            if (t0 + r < T) v[1] += -0.5 * ((var - rowv[r]) / Qd);           // conditionals_multi_output.py:255
            v[2] += chn[SR + r];
        }
        tiny_sum<3, NW>(v, red);
        if (tid == 0) { double *sc = Pu + ntl * 256 + Mp; sc[0] = v[0]; sc[1] = v[1]; sc[2] = v[2]; }
    }
    TSTAMP(4);
    tiny_arrive(cx.cP, slot);
    TSTAMP(5);
    if (!a.grad) return;
    // ==========================================================================================================================
    // strip, phase 2 (backward):  dl/dK_fu = alpha (F N + delta w^T) W^T,  E = dl/dK_fu o K_fu,  its reductions
    // ==========================================================================================================================
    double *wl = lds + L.vec;
    double *XO = lds + L.xo, *ZO = lds + L.zo;
    // (the operands [1 | Z] and [1 | x_comb] of the reductions, while the head factorises H: the K-build copies are dead)
    for (int e = tid; e < Mp * 16; e += NTHR) {
        const int m = e >> 4, n = e & 15;
        ZO[e] = (n == 0) ? 1.0 : ((m < M && n <= P) ? a.Z[(size_t)m * P + n - 1] : 0.0);
    }
    for (int e = tid; e < SR * 16; e += NTHR) {
        const int r = e >> 4, n = e & 15, t = t0 + r;
        double v = 0.0;
        if (n == 0) v = 1.0;
        else if (t < T && n <= P) v = (n - 1 < D) ? Xs[(size_t)t * D + n - 1] : a.ctrl[(size_t)t * a.C + (n - 1 - D)];
        XO[e] = v;
    }
    // (explicit U: dl/dK_fu = alpha (F + r u^T) W^T -- the collapsed branch's form with N = I, w = u, delta -> r -- needs nothing of the
    //  head: the strip runs straight on; only the K_uu-side row blocks below wait for its flag)
    if (!BA && tiny_wait(cx.fN, 1, cx.abort_w, slot) < 0) return;
    TSTAMP(6);
    if (!BA)
        for (int m = tid; m < Mp; m += NTHR) wl[m] = a.wv[(size_t)u * Mp + m];
    __syncthreads();
    const int Tp = nst * SR, drow = P + 1;
    double *dxu = a.dxc + (size_t)u * Tp * drow;
    if (BA) {
        for (int r = tid; r < SR; r += NTHR)                                  // dl/ddelta_t = -alpha r_t
            if (t0 + r < T) dxu[(size_t)(t0 + r) * drow + P] = -alpha * dlt[r];
#pragma unroll
        for (int j = 0; j < TNT; ++j)
            if (j < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ks[(size_t)(16 * wave + lk + 4 * r) * LD + 16 * j + lr] += dlt[16 * wave + lk + 4 * r] * wl[16 * j + lr];          // R = F + r u^T (own rows)
            }
    } else {
    for (int r = tid; r < SR; r += NTHR) {                                   // dl/ddelta_t = alpha (F w)_t
        double acc0 = 0.0, acc1 = 0.0;
        for (int j = 0; j + 1 < Mp; j += 2) { acc0 += Ks[(size_t)r * LD + j] * wl[j]; acc1 += Ks[(size_t)r * LD + j + 1] * wl[j + 1]; }
        if (t0 + r < T) dxu[(size_t)(t0 + r) * drow + P] = alpha * (acc0 + acc1);
    }
    const double *Nu = a.Nw + (size_t)u * msq;
#pragma unroll
    for (int j = 0; j < TNT; ++j) facc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    tiny_row_gemm<0>(facc, Arow, Nu, Mp, NT, lr, lk);
    __syncthreads();                                                          // every thread's reads of F (dl/ddelta above) are done
#pragma unroll
    for (int j = 0; j < TNT; ++j)
        if (j < NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ks[(size_t)(16 * wave + lk + 4 * r) * LD + 16 * j + lr] = facc[j][r] + dlt[16 * wave + lk + 4 * r] * wl[16 * j + lr];   // R
        }
    }
    wave_lds_order();
#pragma unroll
    for (int j = 0; j < TNT; ++j) facc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    tiny_row_gemm<2>(facc, Arow, Wtu, Mp, NT, lr, lk);                        // W^T is lower triangular by tiles
    wave_lds_order();
#pragma unroll
    for (int j = 0; j < TNT; ++j)
        if (j < NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                facc[j][r] = alpha * facc[j][r] * kst[j * 4 + r];             // E, accumulator layout = B-operand layout with k = row
                Ks[(size_t)(16 * wave + lk + 4 * r) * LD + 16 * j + lr] = facc[j][r];
            }
        }
    wave_lds_order();
    TSTAMP(7);
    // row side: [r | E Z] = E [1 | Z]   (16 rows of this wavefront x 16 columns)
    double *part = rowv;                                                      // [NW][16] per-wavefront partials of sum_t r_t x_tp^2 and sum E
    {
        d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
        for (int k = 0; k < NT; ++k) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double av = Arow[16 * k + 4 * t + lk], bw = ZO[(size_t)(16 * k + 4 * t + lk) * 16 + lr];
                if (t & 1) a1 = mfma_f64(av, bw, a1);
                else a0 = mfma_f64(av, bw, a0);
            }
        }
        double rx2 = 0.0;
        const double len = (lr >= 1 && lr <= P) ? ilen[lr - 1] : 1.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = a0[r] + a1[r];
            const double rt = __shfl(v, lane & 48);                           // lane lr = 0 of this row: the row sum r_t
            const int row = 16 * wave + lk + 4 * r, t = t0 + row;
            const double x = XO[(size_t)row * 16 + lr];
            if (lr >= 1 && lr <= P) {
                if (t < T) dxu[(size_t)t * drow + lr - 1] = -(x * rt - v) / (len * len);      // dl/dx_comb_t,p (_se_chain)
                rx2 += rt * x * x;
            } else if (lr == 0) rx2 += v;                                     // sum of the row sums
        }
        rx2 += __shfl_xor(rx2, 16);
        rx2 += __shfl_xor(rx2, 32);
        __syncthreads();                                                      // rowv (phase 1) is dead; every wavefront is past its E rows
        if (lk == 0) part[wave * 16 + lr] = rx2;
    }
    // column side: [cs ; E^T x] = [1 | x]^T E for this wavefront's rows, then the wavefronts' tiles added in fixed order
    {
        double *CB = Ks;                                                      // [NW][16][Mp]: E is dead in LDS (it stays in registers)
#pragma unroll
        for (int j = 0; j < TNT; ++j)
            if (j < NT) {
                d4 c0 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) c0 = mfma_f64(XO[(size_t)(16 * wave + 4 * q + lk) * 16 + lr], facc[j][q], c0);
#pragma unroll
                for (int r = 0; r < 4; ++r) CB[((size_t)wave * 16 + lk + 4 * r) * Mp + 16 * j + lr] = c0[r];
            }
        __syncthreads();
        double *Qu = a.Qp + ((size_t)u * nst + strip) * tiny_qstride(Mp);
        for (int idx = tid; idx < (P + 1) * Mp; idx += NTHR) {
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v += CB[(size_t)w * 16 * Mp + idx];
            Qu[idx] = v;
        }
        if (tid <= P) {
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v += part[w * 16 + tid];
            if (tid == 0) Qu[16 * Mp + 8] = v;                                // sum E
            else Qu[16 * Mp + tid - 1] = v;                                   // sum_t r_t x_tp^2
        }
        __syncthreads();
    }
    TSTAMP(8);
    // K_uu side, one 16-row block per strip (unless the launch has workgroups of its own for it: tiny_plan)
    if (!a.side) {
        int hn, h0;
        tiny_kuu_split(NT, nst, hn, h0);
        if (BA && tiny_wait(cx.fN, 1, cx.abort_w, slot) < 0) return;         // (2 Phi from the head; the collapsed branch waited for N above)
        for (int rb = strip; rb < NT; rb += nst) {
            if (rb >= h0 && rb < h0 + hn) continue;                           // the head's blocks
            tiny_kuu_rows<NW>(a, u, rb, Ks, ZO, ilen, var);
        }
    }
    TSTAMP(9);
    if (tiny_arrive(cx.c2, slot) != narrive2 - 1) return;
    TSTAMP(10);
    tiny_unit_done<NW>(a, u, lds, L.ctl, L.red, L.vec, L.mat);
    TSTAMP(11);
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
TinyPlan tiny_plan(int kind, int T, int D, int C, int M, int S, int Dl, int grad, int cus) {
    TinyPlan pl{};
    pl.ok = false;
    const int P = D + C;
    if (kind != 0 || P > TINY_PMAX || M < 1 || T < 1 || T > TINY_TMAX || S < 1 || Dl < 1) return pl;
    pl.Mp = round_up(M, 16);
    if (pl.Mp > TINY_MPMAX) return pl;
    pl.NT = pl.Mp / 16;
    pl.nunits = S * Dl;
    for (int nw = 4; nw <= 8; nw += 4) {
        const int SR = 16 * nw, nst = (T + SR - 1) / SR;
        const TinyLds l = tiny_lds(pl.Mp, SR);
        const size_t bytes = (size_t)l.total * sizeof(double);
        if (bytes + 1024 > 160 * 1024) continue;           // (the kernel also has a few hundred bytes of static LDS)
        if (nst > 32) continue;
        if (pl.NT > nst * 8) continue;                     // (the K_uu side deals its NT row blocks over the strips)
        if ((long long)pl.nunits * (1 + nst) > (long long)cus) continue;      // every workgroup resident at once, one per CU
        pl.ok = true; pl.nw = nw; pl.SR = SR; pl.nstrips = nst; pl.lds_bytes = bytes;
        pl.side = (grad && (long long)pl.nunits * (1 + nst + pl.NT) <= (long long)cus) ? 1 : 0;
        return pl;
    }
    return pl;
}

namespace {
struct TinyCarve {
    size_t Wg, Wt, Pp, Hs, Nw, Nm2, wv, hterms, uterms, Qp, dxc, dz2, kuu, uout, cterms, cpart, psums, kst, duu, total;
};
TinyCarve tiny_carve(const TinyPlan &pl, int T, int P, int M, int S, int Dl, int D, int Ydim, int grad) {
    (void)T;
    TinyCarve c{};
    const size_t nu = pl.nunits, msq = (size_t)pl.Mp * pl.Mp, nst = pl.nstrips;
    auto al = [](size_t n) { return (n + 31) / 32 * 32; };
    size_t o = 0;
    c.Wg = o; o += al(nu * msq);
    c.Wt = o; o += al(nu * msq);
    c.Pp = o; o += al(nu * nst * tiny_pstride(pl.Mp));
    c.hterms = o; o += al(nu * 2);
    c.cterms = o; o += al((size_t)S * 8);
    c.psums = o; o += 32;
    c.Hs = o; o += al(nu * msq);                    // (written in forward-only launches too: cheap, keeps the kernel uniform)
    if (grad) {
        c.Nw = o; o += al(nu * msq);
        c.Nm2 = o; o += al(nu * msq);
        c.wv = o; o += al(nu * pl.Mp);
        c.uterms = o; o += al(nu * 8);
        c.Qp = o; o += al(nu * nst * tiny_qstride(pl.Mp));
        c.dxc = o; o += al(nu * nst * pl.SR * (P + 1));
        c.kst = o; o += al(nu * nst * (size_t)pl.nw * 64 * 32);
        c.dz2 = o; o += al(nu * pl.Mp * TINY_PMAX);
        c.kuu = o; o += al(nu * pl.NT * (TINY_PMAX + 1));
        c.uout = o; o += al(nu * ((size_t)M * P + P + 2));
        c.cpart = o; o += al((size_t)S * (D * Ydim + 2 * Ydim + Dl));
        c.duu = o; o += al(nu * pl.Mp);
    }
    c.total = o;
    return c;
}
}  // namespace

size_t tiny_scratch_doubles(const TinyPlan &pl, int T, int P, int M, int S, int Dl, int D, int Ydim, int grad) {
    return tiny_carve(pl, T, P, M, S, Dl, D, Ydim, grad).total;
}
size_t tiny_flag_ints(const TinyPlan &pl, int S) { return (size_t)4 * pl.nunits + S + 8; }

void tiny_bind_scratch(TinyArgs &a, const TinyPlan &pl, double *scratch, int *flags) {
    const TinyCarve c = tiny_carve(pl, a.T, a.P, a.M, a.S, a.Dl, a.D, a.Ydim, a.grad);
    a.Mp = pl.Mp; a.NT = pl.NT; a.SR = pl.SR; a.nstrips = pl.nstrips; a.nunits = pl.nunits;
    a.side = (a.grad && pl.side) ? 1 : 0;
    a.Wg = scratch + c.Wg; a.Wt = scratch + c.Wt; a.Pp = scratch + c.Pp; a.hterms = scratch + c.hterms;
    a.chain_terms = scratch + c.cterms; a.Hs = scratch + c.Hs; a.prior_sums = scratch + c.psums;
    a.sp_stride = a.D * a.Ydim + 2 * a.Ydim + a.Dl;
    if (a.grad) {
        a.Nw = scratch + c.Nw; a.Nm2 = scratch + c.Nm2; a.wv = scratch + c.wv; a.uterms = scratch + c.uterms;
        a.Qp = scratch + c.Qp; a.dxc = scratch + c.dxc; a.Kst = scratch + c.kst; a.dz2 = scratch + c.dz2; a.kuu_part = scratch + c.kuu;
        a.unit_out = scratch + c.uout; a.chain_part = scratch + c.cpart; a.du_unit = scratch + c.duu;
    }
    a.flags = flags;
}

// Private (scratch) memory per lane of the kernel a plan launches, as the loaded code object reports it.  The plan needs EVERY
// workgroup resident at once, and a wave is only resident with its scratch: ffvd_create checks this figure against what the file was
// validated with (TINY_PRIVATE_BYTES_MAX) and against a budget for the whole launch before it lets a handle take the one-launch path.
static const void *tiny_kernel_fn(int nw, int branch) {
    if (nw == 4) return branch ? reinterpret_cast<const void *>(&tiny_kernel<4, 1>) : reinterpret_cast<const void *>(&tiny_kernel<4, 0>);
    return branch ? reinterpret_cast<const void *>(&tiny_kernel<8, 1>) : reinterpret_cast<const void *>(&tiny_kernel<8, 0>);
}
hipError_t tiny_kernel_private_bytes(int nw, int branch, size_t *bytes) {
    hipFuncAttributes fa;
    const void *fn = tiny_kernel_fn(nw, branch);
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return e;
    *bytes = (size_t)fa.localSizeBytes;
    return hipSuccess;
}

hipError_t tiny_ring_create(TinyArgRing &r) {
    hipError_t e = hipHostMalloc((void **)&r.pinned, TinyArgRing::N * sizeof(TinyArgs));
    if (e != hipSuccess) return e;
    for (int i = 0; i < TinyArgRing::N; ++i) {
        e = hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming);
        if (e != hipSuccess) return e;
        r.used[i] = false;
    }
    r.next = 0; r.uploads = 0; r.held_valid = false;
    return hipSuccess;
}
void tiny_ring_destroy(TinyArgRing &r) {
    for (int i = 0; i < TinyArgRing::N; ++i)
        if (r.ev[i]) { hipEventDestroy(r.ev[i]); r.ev[i] = nullptr; }
    if (r.pinned) { hipHostFree(r.pinned); r.pinned = nullptr; }
}

hipError_t launch_tiny(hipStream_t stream, const TinyArgs &a, const TinyPlan &pl, TinyArgs *dev_args, TinyArgRing &ring) {
    static size_t attr_bytes_dev[16][4] = {};          // dynamic LDS each instantiation has been allowed so far, per device
    int dev = 0;
    size_t *attr_bytes = (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? attr_bytes_dev[dev] : nullptr;      // (unknown device: always set)
    const int wpu = 1 + pl.nstrips + (a.side ? pl.NT : 0);
    const int grid = a.xcd_map ? 8 * wpu * ((pl.nunits + 7) / 8) : pl.nunits * wpu;
    const int which = (pl.nw == 4 ? 0 : 1) + (a.branch ? 0 : 2);
    const void *fn = tiny_kernel_fn(pl.nw, a.branch);
    if ((!attr_bytes || pl.lds_bytes > attr_bytes[which]) && pl.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return e;
        if (attr_bytes) attr_bytes[which] = pl.lds_bytes;
    }
    // the argument block travels only when it differs from what the device copy holds (steady state: never)
    if (!ring.held_valid || memcmp(&ring.held, &a, sizeof(TinyArgs)) != 0) {
        const int i = ring.next;
        ring.next = (i + 1) % TinyArgRing::N;
        if (ring.used[i]) {                             // the copy that read this slot N uploads ago: long done, but make sure
            hipError_t e = hipEventSynchronize(ring.ev[i]);
            if (e != hipSuccess) return e;
        }
        memcpy(&ring.pinned[i], &a, sizeof(TinyArgs));
        hipError_t e = hipMemcpyAsync(dev_args, &ring.pinned[i], sizeof(TinyArgs), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        e = hipEventRecord(ring.ev[i], stream);
        if (e != hipSuccess) return e;
        ring.used[i] = true;
        memcpy(&ring.held, &a, sizeof(TinyArgs));
        ring.held_valid = true;
        ++ring.uploads;
    }
    const TinyArgs *da = dev_args;
    if (pl.nw == 4 && a.branch) hipLaunchKernelGGL((tiny_kernel<4, 1>), dim3(grid), dim3(256), pl.lds_bytes, stream, da);
    else if (pl.nw == 4) hipLaunchKernelGGL((tiny_kernel<4, 0>), dim3(grid), dim3(256), pl.lds_bytes, stream, da);
    else if (a.branch) hipLaunchKernelGGL((tiny_kernel<8, 1>), dim3(grid), dim3(512), pl.lds_bytes, stream, da);
    else hipLaunchKernelGGL((tiny_kernel<8, 0>), dim3(grid), dim3(512), pl.lds_bytes, stream, da);
    return hipGetLastError();
}

}  // namespace ffvd
