// Bodies of the kernels a step of the posterior rollouts / of the particle-Gibbs sweep consists of (collect_samples_formal,
// base_model.py:288-314; PG_for_X_speedup, :99-115), as device functions of a VIRTUAL block index: kernels.hip wraps each in the
// __global__ kernel the per-step launches use (block index = blockIdx), loops.hip calls them from ONE persistent launch per call
// whose workgroups walk the virtual blocks of every phase of every step and meet at a grid-wide barrier in between -- the same
// code, so the two forms give the same bits.  (Definitions moved here unchanged from kernels.hip, round 4.)
#pragma once
#include "kernels.h"
#include "dev_common.h"
#include <type_traits>

namespace ffvd {

template <int KIND, int NQ, bool NT = false>
__device__ __forceinline__ void kfu_build_body(const ProjectArgs &a, const int bx, const int by, const int bzz) {
    constexpr bool SMALLP = NQ > 0;
    __shared__ double xs[SMALLP ? 1 : MAXP][SMALLP ? 1 : 64];
    __shared__ __attribute__((aligned(16))) double xr8[SMALLP ? 64 : 1][8];
    __shared__ double xx[64];
    __shared__ double zs[64][(SMALLP ? 8 : MAXP) + 1];
    __shared__ double gsum[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int t0 = bx * 64, m0 = by * 64, bz = bzz;
    const int b = a.b0 + bz, s = b / a.Dl, dl = b % a.Dl;
    // delta_t = x_{t+1,d} - x_{t,d} (:247) of a row is the same for the whole wavefront (a wavefront = 16 rows x 64 columns): every
    // wavefront loads its 16 values once, one per lane, and hands them out by v_readlane
    const double *xdl = a.x + (size_t)s * a.x_chain_stride + (a.d_begin + dl);
    const int P = a.P, Mp = a.Mp;
    const double var = a.hv.variance[dl];
    for (int p = tid >> 6; p < (SMALLP ? 8 : P); p += 4) {
        const int t = t0 + lane;
        double v = 0.0;
        if (t < a.T && p < P) {
            v = (p < a.x_cols) ? a.x[(size_t)s * a.x_chain_stride + (size_t)t * a.x_ld + p]
                               : a.ctrl[(size_t)t * a.C + (p - a.x_cols)];
            if (KIND == 0) v = v / a.hv.len[(size_t)dl * P + p];
            else v = v * var;
        }
        if (SMALLP) xr8[lane][p] = v;
        else xs[p][lane] = v;
        zs[lane][p] = (p < P) ? a.hv.Zs[((size_t)dl * Mp + m0 + lane) * P + p] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        double acc = 0.0;
        if (KIND == 0) {
            if (SMALLP) for (int p = 0; p < 8; ++p) acc += xr8[tid][p] * xr8[tid][p];     // padding adds exact zeros
            else for (int p = 0; p < P; ++p) acc += xs[p][tid] * xs[p][tid];
        }
        xx[tid] = acc;
    }
    __syncthreads();
    const double zzv = a.hv.zz[(size_t)dl * Mp + m0 + lane];
    double *out = a.F + ((size_t)bz * a.Tp + t0) * Mp + m0 + lane;
    const bool mok = (m0 + lane) < a.M;
    double zr[8];                               // this thread's inducing input (first 8 components) in registers
#pragma unroll
    for (int p = 0; p < 8; ++p) zr[p] = (SMALLP || p < P) ? zs[lane][p] : 0.0;
    const int rbase = __builtin_amdgcn_readfirstlane(tid >> 6) * 16;
    const bool want_g = a.gpart != nullptr;
    // NT: K_fu of the big batch (2.1 GB at config 2) is a pure write stream nobody reads before it has left every cache: streaming
    // stores take the build from 0.515 to 0.446 ms (4.2 -> 4.8 TB/s), the Gram kernel behind it loses 0.03 (profiles/r04_ab_kfu_nt.txt);
    // a compile-time switch -- as a run-time flag inside the row loop the same stores gained 0.016 ms
    double gacc = 0.0, dlane = 0.0;             // lane i < 16 of every wavefront: delta of row rbase + i
    if (want_g) {
        const int t = t0 + rbase + (lane & 15);
        if (t < a.T) dlane = xdl[(size_t)(t + 1) * a.x_ld] - xdl[(size_t)t * a.x_ld];
    }
    auto rows = [&](auto edge_tag) {             // interior tiles skip the per-element range selects
        constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int r = rbase + i;
            double dot = 0.0;
            if (SMALLP) {
#pragma unroll
                for (int q = 0; q < (NQ > 0 ? NQ : 1); ++q) {
                    const double2 xv = *reinterpret_cast<const double2 *>(&xr8[r][2 * q]);
                    dot += xv.x * zr[2 * q];
                    dot += xv.y * zr[2 * q + 1];
                }
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    if (p < P) dot += xs[p][r] * zr[p];
                for (int p = 8; p < P; ++p) dot += xs[p][r] * zs[lane][p];
            }
            double v = kernel_value<KIND>(dot, xx[r], zzv, var);
            if (EDGE && (!mok || t0 + r >= a.T)) v = 0.0;
            if (NT) __builtin_nontemporal_store(v, &out[(size_t)r * Mp]);
            else out[(size_t)r * Mp] = v;
            if (want_g) {                       // delta of this row out of lane i: two v_readlane, then ONE vector FMA
                const double dr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dlane), i),
                                                   __builtin_amdgcn_readlane(__double2loint(dlane), i));
                gacc = fma(dr, v, gacc);
            }
        }
    };
    if (t0 + 64 > a.T || m0 + 64 > a.M) rows(std::true_type{});
    else rows(std::false_type{});
    if (want_g) {                               // the four row groups of a column, added in fixed order
        gsum[tid >> 6][lane] = gacc;
        __syncthreads();
        if (tid < 64)
            a.gpart[((size_t)bz * (a.Tp / 64) + bx) * Mp + m0 + tid] = (gsum[0][tid] + gsum[1][tid]) + (gsum[2][tid] + gsum[3][tid]);
    }
    __syncthreads();      // (a persistent caller reuses the LDS tiles for its next block)
}

// Debug build only (-DFFVD_STEP_TRACE, variant `steptrace`, tools/step_trace.py): wall-clock stamps of every workgroup of the last
// skinny launch (start, operands of the first k block there, k loop done, end) -- kernels.hip owns the buffer.
#if defined(FFVD_STEP_TRACE) && defined(FFVD_STEP_TRACE_OWNER)
__device__ long long step_trace_buf[4096 * 8];
#define STEP_STAMP(wg, slot) do { if (threadIdx.x == 0 && (wg) < 4096) step_trace_buf[(wg) * 8 + (slot)] = wall_clock64(); } while (0)
#define STEP_NOTE(wg, slot, v) do { if (threadIdx.x == 0 && (wg) < 4096) step_trace_buf[(wg) * 8 + (slot)] = (v); } while (0)
// first start / last end of each of a step's kernels, for the first 64 steps of the process (0: K build, 1: product, 2: epilogue, 3: step)
__device__ int step_span_ctr;
__device__ unsigned long long step_span[64 * 4 * 2];
// (plain stores: the first workgroup's start, and whichever end stamp lands last -- atomics from 512 workgroups stretch the kernels)
#define STEP_SPAN_BEGIN(k) do { if (threadIdx.x == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0 && step_span_ctr < 64) step_span[(step_span_ctr * 4 + (k)) * 2] = (unsigned long long)wall_clock64(); } while (0)
#define STEP_SPAN_END(k) do { if (threadIdx.x == 0 && step_span_ctr < 64) step_span[(step_span_ctr * 4 + (k)) * 2 + 1] = (unsigned long long)wall_clock64(); } while (0)
#define STEP_SPAN_NEXT() do { if (threadIdx.x == 0) step_span_ctr = step_span_ctr + 1; } while (0)
#else
#define STEP_STAMP(wg, slot) do { } while (0)
#define STEP_NOTE(wg, slot, v) do { } while (0)
#define STEP_SPAN_BEGIN(k) do { } while (0)
#define STEP_SPAN_END(k) do { } while (0)
#define STEP_SPAN_NEXT() do { } while (0)
#endif
// K(x, Z) of a step stored TRANSPOSED, KT[b][m][ldt] (m-major, the step's rows contiguous): what the skinny product below wants as
// its A operand -- an MFMA lane holds one ROW's value, so with row-major K a 16-lane group reads 16 rows 4 KB apart (16 cache lines
// per load, the texture addresser's limit: 1.1 us per k block measured by stamps, profiles/r05_step_trace.txt); m-major the same lanes
// read 128 contiguous bytes.  A lane owns a row here and walks 16 of the tile's inducing points; every value is computed by the
// expression kfu_build_body uses (same operands, same order), so the two layouts hold the same bits.
template <int KIND, int NQ, int MW>       // MW inducing points per wavefront: a workgroup's tile is 64 rows x 4 MW points
__device__ __forceinline__ void kfu_build_t_body(const ProjectArgs &a, const int ldt, const int bx, const int by, const int bzz) {
    constexpr bool SMALLP = NQ > 0;
    __shared__ double xs[SMALLP ? 1 : MAXP][SMALLP ? 1 : 64];
    __shared__ __attribute__((aligned(16))) double xr8[SMALLP ? 64 : 1][8];
    __shared__ double zs[4 * MW][(SMALLP ? 8 : MAXP) + 1];
    __shared__ double zzs[4 * MW];
    const int tid = threadIdx.x, lane = tid & 63;
    STEP_SPAN_BEGIN(0);
    const int t0 = bx * 64, m0 = by * (4 * MW), bz = bzz;
    const int b = a.b0 + bz, dl = b % a.Dl, s = b / a.Dl;
    const int P = a.P, Mp = a.Mp;
    const double var = a.hv.variance[dl];
    for (int p = tid >> 6; p < (SMALLP ? 8 : P); p += 4) {
        const int t = t0 + lane;
        double v = 0.0;
        if (t < a.T && p < P) {
            v = (p < a.x_cols) ? a.x[(size_t)s * a.x_chain_stride + (size_t)t * a.x_ld + p]
                               : a.ctrl[(size_t)t * a.C + (p - a.x_cols)];
            if (KIND == 0) v = v / a.hv.len[(size_t)dl * P + p];
            else v = v * var;
        }
        if (SMALLP) xr8[lane][p] = v;
        else xs[p][lane] = v;
        if (lane < 4 * MW) zs[lane][p] = (p < P) ? a.hv.Zs[((size_t)dl * Mp + m0 + lane) * P + p] : 0.0;
    }
    if (tid < 4 * MW) zzs[tid] = a.hv.zz[(size_t)dl * Mp + m0 + tid];
    __syncthreads();
    double xr[8];                               // this lane's row (first 8 components) in registers
#pragma unroll
    for (int p = 0; p < 8; ++p) xr[p] = SMALLP ? xr8[lane][p] : (p < P ? xs[p][lane] : 0.0);
    double xxv = 0.0;
    if (KIND == 0) {
        if (SMALLP) for (int p = 0; p < 8; ++p) xxv += xr[p] * xr[p];                   // padding adds exact zeros
        else for (int p = 0; p < P; ++p) xxv += xs[p][lane] * xs[p][lane];
    }
    const int mbase = __builtin_amdgcn_readfirstlane(tid >> 6) * MW;
    const bool tok = t0 + lane < a.T;
    double *out = a.F + ((size_t)bz * Mp + m0) * ldt + t0 + lane;
#pragma unroll 4
    for (int i = 0; i < MW; ++i) {
        const int m = mbase + i;
        double dot = 0.0;
        if (SMALLP) {
#pragma unroll
            for (int q = 0; q < (NQ > 0 ? NQ : 1); ++q) {
                dot += xr[2 * q] * zs[m][2 * q];
                dot += xr[2 * q + 1] * zs[m][2 * q + 1];
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (p < P) dot += xr[p] * zs[m][p];
            for (int p = 8; p < P; ++p) dot += xs[p][lane] * zs[m][p];
        }
        double v = kernel_value<KIND>(dot, xxv, zzs[m], var);
        if (!tok || m0 + m >= a.M) v = 0.0;
        out[(size_t)m * ldt] = v;
    }
    STEP_SPAN_END(0);
}

struct SkinnyArgs {
    const double *A; size_t a_stride; int lda;
    const double *B; size_t b_stride; int ldb;      // b_stride = 0: one B for every unit
    int upper, rows, K, N, nb, Tp;
    double *C; size_t c_stride; int ldc;            // or null
    const double *u; size_t u_stride;               // or null
    double *sq, *dot;                               // [nb][N / 16][Tp], or null
    // optional SECOND right-hand side in the same launch (its slabs behind the first one's): C2 = A B2 is not stored, only
    // sq2[nb][N2 / 16][Tp] = sum over the slab of C2^2.  The rollouts' q_sqrt inflation |F q_sqrt|^2 = |K (W q_sqrt)|^2 rides beside
    // F = K W this way: both products read the same K rows and the step loses a dependent launch.
    const double *B2; size_t b2_stride; int ldb2, N2;
    double *sq2;
    int a_trans;                                    // A is stored k-major, AT[b][K][lda] (kfu_build_t_body): coalesced operand loads
    int upper2;                                     // B2[k][n] = 0 for k > n as well (W q_sqrt with an upper-triangular q_sqrt: the reference's L_H^-T)
};

#ifndef FFVD_SKINNY_CHUNK
#define FFVD_SKINNY_CHUNK 1       // k blocks whose operands are in flight together: 2 and 4 measured no faster (profiles/r05_step_trace.txt)
#endif
__device__ __forceinline__ void skinny_body(const SkinnyArgs &a, const int bx, const int by, const int bzz) {
    __shared__ double red[4][2][4][64];
    const int wg_lin = blockIdx.x + (int)gridDim.x * (blockIdx.y + (int)gridDim.y * blockIdx.z);
    STEP_STAMP(wg_lin, 0);
    STEP_SPAN_BEGIN(1);
    const int nslab1 = a.N / 16;
    const bool second = bx >= nslab1;
    const int n0 = (second ? bx - nslab1 : bx) * 16, r0 = by * 32, b = bzz;
    if (r0 >= a.rows) return;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *Ab = a.A + (size_t)b * a.a_stride, *Bb = second ? a.B2 + (size_t)b * a.b2_stride : a.B + (size_t)b * a.b_stride;
    const int ldb = second ? a.ldb2 : a.ldb;
    const int kend = ((second ? a.upper2 : a.upper) && n0 + 16 < a.K) ? n0 + 16 : a.K;
    const int nkb = kend / 16, per = (nkb + 3) / 4;                 // 16-wide k blocks, a quarter of them per wavefront
    const int kb0 = w * per, kb1 = (kb0 + per < nkb) ? kb0 + per : nkb;
    d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
    // Inside a k block the four lane groups take k = 4 lk + s in MFMA s (a sum does not care about its order), so a lane reads
    // four CONSECUTIVE doubles of its A row (two 16-byte loads feed four MFMAs, 128-byte runs per row) instead of four 8-byte ones
    const double *bp = Bb + (size_t)(4 * lk) * ldb + n0 + lr;
    if (a.a_trans) {
        const double *atp = Ab + (size_t)(4 * lk) * a.lda + r0 + lr;        // AT[k][row]: 16 lanes = 128 contiguous bytes
        const size_t la = (size_t)a.lda;
        constexpr int CH = FFVD_SKINNY_CHUNK;
        // operands of up to CH k blocks in flight (the loads are one line per 16 lanes now: the loop is bound by their latency)
        for (int kb = kb0; kb < kb1; kb += CH) {
            double av[CH][8], bv[CH][4];
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (kb + j < kb1) {
                    const double *aq = atp + (size_t)(16 * (kb + j)) * la, *bq = bp + (size_t)(16 * (kb + j)) * ldb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { av[j][e] = aq[e * la]; av[j][4 + e] = aq[e * la + 16]; bv[j][e] = bq[e * (size_t)ldb]; }
                }
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (kb + j < kb1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = mfma_f64(av[j][e], bv[j][e], acc[0]); acc[1] = mfma_f64(av[j][4 + e], bv[j][e], acc[1]);
                    }
                }
        }
    } else {
    const double *ap0 = Ab + (size_t)(r0 + lr) * a.lda + 4 * lk, *ap1 = ap0 + (size_t)16 * a.lda;
    for (int kb = kb0; kb < kb1; ++kb) {       // (the compiler does not unroll this loop; two blocks in flight by hand were no faster)
        const int k0 = 16 * kb;
        const d2 a0l = *reinterpret_cast<const d2 *>(ap0 + k0), a0h = *reinterpret_cast<const d2 *>(ap0 + k0 + 2);
        const d2 a1l = *reinterpret_cast<const d2 *>(ap1 + k0), a1h = *reinterpret_cast<const d2 *>(ap1 + k0 + 2);
        const double *bq = bp + (size_t)k0 * ldb;
        const double b0 = bq[0], b1 = bq[ldb], b2 = bq[2 * (size_t)ldb], b3 = bq[3 * (size_t)ldb];
        acc[0] = mfma_f64(a0l.x, b0, acc[0]); acc[1] = mfma_f64(a1l.x, b0, acc[1]);
        acc[0] = mfma_f64(a0l.y, b1, acc[0]); acc[1] = mfma_f64(a1l.y, b1, acc[1]);
        acc[0] = mfma_f64(a0h.x, b2, acc[0]); acc[1] = mfma_f64(a1h.x, b2, acc[1]);
        acc[0] = mfma_f64(a0h.y, b3, acc[0]); acc[1] = mfma_f64(a1h.y, b3, acc[1]);
    }
    }
    STEP_STAMP(wg_lin, 1);
    STEP_NOTE(wg_lin, 4, bx); STEP_NOTE(wg_lin, 5, kb1 > kb0 ? kb1 - kb0 : 0);
    STEP_NOTE(wg_lin, 6, __builtin_amdgcn_s_getreg((31 << 11) | 4)); STEP_NOTE(wg_lin, 7, __builtin_amdgcn_s_getreg((31 << 11) | 20));   // HW_ID, XCC_ID
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[w][x][q][lane] = acc[x][q];
    __syncthreads();
    STEP_STAMP(wg_lin, 2);
    {   // every wavefront finishes two of the eight 4-row groups; the four partial sums of an element are added in wavefront order
    const int slab = n0 / 16, nslab = second ? a.N2 / 16 : nslab1;
    const double un = (a.u && !second) ? a.u[(size_t)b * a.u_stride + n0 + lr] : 0.0;
    double *sqp = second ? a.sq2 : a.sq, *dotp = second ? nullptr : a.dot, *Cp = second ? nullptr : a.C;
    const int x = w >> 1;
    double cv[2], s2[2], du[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = 2 * (w & 1) + h;
        cv[h] = ((red[0][x][q][lane] + red[1][x][q][lane]) + red[2][x][q][lane]) + red[3][x][q][lane];
        s2[h] = cv[h] * cv[h]; du[h] = cv[h] * un;
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { s2[h] += __shfl_xor(s2[h], m); du[h] += __shfl_xor(du[h], m); }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = 2 * (w & 1) + h, row = r0 + 16 * x + lk + 4 * q;
        if (Cp) Cp[(size_t)b * a.c_stride + (size_t)row * a.ldc + n0 + lr] = cv[h];
        if (lr == 0) {
            const size_t o = ((size_t)b * nslab + slab) * a.Tp + row;
            if (sqp) sqp[o] = s2[h];
            if (dotp) dotp[o] = du[h];
        }
    }
    }
    __syncthreads();      // (a persistent caller reuses `red`)
    STEP_STAMP(wg_lin, 3);
    STEP_SPAN_END(1);
}

__device__ __forceinline__ void conditional_finish_body(const int vb, int kind, const double *x, int N, int P, const double *variance,
                                          const double *rowsq, const double *fmean, int ng, int Tp, int D,
                                          double *mean, double *var, const double *extra /*[D][extra_ng][Tp] or null*/,
                                          int extra_ng) {
    STEP_SPAN_BEGIN(2);
    const int idx = (vb * 256 + (int)threadIdx.x) >> 4, l = threadIdx.x & 15;
    const bool live = idx < N * D;
    const int n = live ? idx / D : 0, d = live ? idx % D : 0;
    double rs = 0.0, fm = 0.0, ex = 0.0;
    for (int g = l; g < ng; g += 16) {
        rs += rowsq[((size_t)d * ng + g) * Tp + n];
        fm += fmean[((size_t)d * ng + g) * Tp + n];
    }
    if (extra)
        for (int g = l; g < extra_ng; g += 16) ex += extra[((size_t)d * extra_ng + g) * Tp + n];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { rs += __shfl_xor(rs, m); fm += __shfl_xor(fm, m); ex += __shfl_xor(ex, m); }
    STEP_SPAN_END(2);                            // (before the 8-byte stores of the lane-0s)
    if (!live || l != 0) return;
    double kd = variance[d];
    if (kind == 1) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) { double v = x[(size_t)n * P + p]; s += (v * v) * variance[d]; }
        kd = s;
    }
    mean[idx] = fm;
    var[idx] = kd - rs;
    if (extra) var[idx] = var[idx] + ex;                            // fvar + reduce_sum(square(LTA), 1)  (:380)
}

struct FinishIn {
    int kind, P, ng, Tp, D, extra_ng;
    const double *variance, *rowsq, *fmean, *extra;     // extra: optional [D][extra_ng][Tp]
};
__device__ __forceinline__ void finish_mean_var16(const FinishIn &f, const double *xrow /* P inputs of row n */, const bool live, const int n,
                                                  const int d, const int l, double &mean, double &var) {
    double rs = 0.0, fm = 0.0, ex = 0.0;
    if (live) {
        for (int g = l; g < f.ng; g += 16) {
            rs += f.rowsq[((size_t)d * f.ng + g) * f.Tp + n];
            fm += f.fmean[((size_t)d * f.ng + g) * f.Tp + n];
        }
        if (f.extra)
            for (int g = l; g < f.extra_ng; g += 16) ex += f.extra[((size_t)d * f.extra_ng + g) * f.Tp + n];
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { rs += __shfl_xor(rs, m); fm += __shfl_xor(fm, m); ex += __shfl_xor(ex, m); }
    double kd = live ? f.variance[d] : 0.0;
    if (live && f.kind == 1) {
        double s = 0.0;
        for (int p = 0; p < f.P; ++p) { const double v = xrow[p]; s += (v * v) * f.variance[d]; }
        kd = s;
    }
    mean = fm;
    var = kd - rs;
    if (f.extra) var = var + ex;
}

// Rollout step with the conditional epilogue folded in.  The input rows are read from x_in and the advanced rows written to x_out
// (the caller alternates two buffers): LinearK's Kdiag reads a whole row, which another lane group of the same row advances.
__device__ __forceinline__ void rollout_finish_update_body(const int vb, const FinishIn &f, const double *log_Q, const double *eps_t,
                                                           const double *ctrl_next, int R, int C, int t, int steps,
                                                           const double *x_in, double *x_out, double *predict_x,
                                                           double *predict_var) {
    const int idx = (vb * 256 + (int)threadIdx.x) >> 4, l = threadIdx.x & 15;
    const int D = f.D, P = f.P;
    const bool live = idx < R * D;
    const int r = live ? idx / D : 0, d = live ? idx % D : 0;
    double m, vv;
    finish_mean_var16(f, x_in + (size_t)r * P, live, r, d, l, m, vv);
    if (!live) return;
    if (l == 0) {
        const double v = vv + exp(log_Q[d]);
        const double xn = (m + x_in[(size_t)r * P + d]) + eps_t[r * D + d] * sqrt(v);
        const size_t o = ((size_t)r * steps + t) * D + d;
        predict_x[o] = xn;
        predict_var[o] = v;
        x_out[(size_t)r * P + d] = xn;
    } else if (d == 0) {        // the control columns of row r: the next step's, or carried over
        for (int c = l - 1; c < C; c += 15) x_out[(size_t)r * P + D + c] = ctrl_next ? ctrl_next[c] : x_in[(size_t)r * P + D + c];
    }
}

constexpr int PG_MAXN = 1024, PG_MAXY = 8;
// One particle-Gibbs step by ONE workgroup (pg_step_kernel: a thread per particle, STRIDED = 0; the persistent sweep: 256 threads,
// a thread walks the particles i, i + 256, ...): the same arithmetic per particle either way.
template <int STRIDED>
__device__ __forceinline__ void pg_step_general_body(double *w, double *cdf, double &wmax_s, const double *mean, const double *var, const double *log_Q,
                                                          const double *eps_t, const double *unif_t, const double *y_t,
                                                          const double *x_ref_next, const double *CC, const double *DD,
                                                          const double *Rch, const double *ctrl_next, int R, int D, int C,
                                                          int Ydim, double *xc, double *cand, double *parts_next,
                                                          int32_t *idx_out) {
    const int N = R + 1, P = D + C, i0 = threadIdx.x, istep = STRIDED ? (int)blockDim.x : PG_MAXN;
    for (int i = i0; i < N; i += istep) {
        for (int p = 0; p < D; ++p) {
            double xn;
            if (i < R) {
                const double v = var[i * D + p] + exp(log_Q[p]);
                xn = (mean[i * D + p] + xc[i * P + p]) + eps_t[i * D + p] * sqrt(v);          // :99-101
            } else xn = x_ref_next[p];                                                         // :111
            cand[(size_t)i * D + p] = xn;
        }
        // logdensity_norm(Y[tt], predict_mean(x), Rchols), likelihoods.py:76-79,114-127
        double a[PG_MAXY], q = 0.0, ld = 0.0;
        for (int j = 0; j < Ydim; ++j) {
            double ym = 0.0;
            for (int d = 0; d < D; ++d) ym += cand[(size_t)i * D + d] * CC[d * Ydim + j];
            ym += DD[j];
            double r = y_t[j] - ym;
            for (int k = 0; k < j; ++k) r -= Rch[j * Ydim + k] * a[k];
            a[j] = r / Rch[j * Ydim + j];
            q += a[j] * a[j];
            ld += log(Rch[j * Ydim + j]);
        }
        w[i] = -0.5 * q + (-ld);
    }
    __syncthreads();
    if (i0 == 0) {
        double m = w[0];
        for (int k = 1; k < N; ++k) m = (w[k] > m) ? w[k] : m;
        wmax_s = m;
    }
    __syncthreads();
    for (int i = i0; i < N; i += istep) w[i] = exp(w[i] - wmax_s);
    __syncthreads();
    if (i0 == 0) {
        double c = 0.0;
        for (int k = 0; k < N; ++k) { c += w[k]; cdf[k] = c; }
    }
    __syncthreads();
    for (int i = i0; i < R; i += istep) {
        const double target = unif_t[i] * cdf[N - 1];
        int lo = 0, hi = N;                      // first k with cdf[k] > target
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > target) hi = mid; else lo = mid + 1;
        }
        const int k = (lo < N) ? lo : N - 1;
        idx_out[i] = k;
        for (int p = 0; p < D; ++p) {
            const double x = cand[(size_t)k * D + p];
            parts_next[(size_t)i * D + p] = x;
            xc[i * P + p] = x;
        }
        if (ctrl_next)
            for (int c = 0; c < C; ++c) xc[i * P + D + c] = ctrl_next[c];
    }
    __syncthreads();      // (a persistent caller reuses w / cdf)
}

// The same step when D <= 8 and (R + 1) D <= PG_FAST_ND (the shapes the reference runs): the step's constants (exp(log_Q), CC, DD,
// R's rows and log diagonal, y_t, the next controls) are read once into LDS, a particle's inputs are all in flight before the first
// use, its candidate stays in registers and LDS (no store -> load trip through memory), loops over D and Ydim are unrolled with
// predicates (no private-memory arrays), the max is a wavefront reduction.  Every expression is the general body's with the same
// operands in the same order (stamps: 8.7 -> us inside the kernel, profiles/r05_step_trace.txt); the CDF stays ONE thread's
// running sum in index order, as the oracle's cumsum.
constexpr int PG_FAST_ND = 2048;
template <int STRIDED>
__device__ __forceinline__ void pg_step_fast_body(double *w, double *cdf, double &wmax_s, const double *mean, const double *var,
                                                  const double *log_Q, const double *eps_t, const double *unif_t, const double *y_t,
                                                  const double *x_ref_next, const double *CC, const double *DD, const double *Rch,
                                                  const double *ctrl_next, int R, int D, int C, int Ydim, double *xc,
                                                  double *parts_next, int32_t *idx_out) {
    __shared__ double cands[PG_FAST_ND];
    __shared__ double eq[8], CCs[64], DDs[8], ys[8], logd[8], Rs[64], ctl[MAXP], xref[8];
    const int N = R + 1, P = D + C, tid = threadIdx.x, nt = blockDim.x, istep = STRIDED ? nt : PG_MAXN;
    STEP_STAMP(4095, 0);
    STEP_SPAN_BEGIN(3);
    double v[8], mu[8], x0[8], e[8], un = 0.0;
    auto load = [&](const int i) {
        if (i < R) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (p < D) { v[p] = var[i * D + p]; mu[p] = mean[i * D + p]; x0[p] = xc[i * P + p]; e[p] = eps_t[i * D + p]; }
            un = unif_t[i];
        }
    };
    int i = tid;
    if (i < N) load(i);
    for (int k = tid; k < D; k += nt) { eq[k] = exp(log_Q[k]); xref[k] = x_ref_next[k]; }
    for (int k = tid; k < D * Ydim; k += nt) CCs[k] = CC[k];
    for (int k = tid; k < Ydim * Ydim; k += nt) Rs[k] = Rch[k];
    for (int k = tid; k < Ydim; k += nt) { DDs[k] = DD[k]; ys[k] = y_t[k]; logd[k] = log(Rch[k * Ydim + k]); }
    if (ctrl_next) for (int k = tid; k < C; k += nt) ctl[k] = ctrl_next[k];
    __syncthreads();
    STEP_STAMP(4095, 6);
    double unl = un;                             // (STRIDED: a thread's particles are resampled in the same order below)
    for (; i < N;) {
        double xn[8];
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (p < D) {
                if (i < R) {
                    const double vv = v[p] + eq[p];
                    xn[p] = (mu[p] + x0[p]) + e[p] * sqrt(vv);                                 // :99-101
                } else xn[p] = xref[p];                                                        // :111
                cands[i * D + p] = xn[p];
            }
        STEP_STAMP(4095, 7);
        // logdensity_norm(Y[tt], predict_mean(x), Rchols), likelihoods.py:76-79,114-127
        double a[PG_MAXY], q = 0.0, ld = 0.0;
#pragma unroll
        for (int j = 0; j < PG_MAXY; ++j)
            if (j < Ydim) {
                double ym = 0.0;
#pragma unroll
                for (int d = 0; d < 8; ++d)
                    if (d < D) ym += xn[d] * CCs[d * Ydim + j];
                ym += DDs[j];
                double r = ys[j] - ym;
#pragma unroll
                for (int k = 0; k < j; ++k) r -= Rs[j * Ydim + k] * a[k];
                a[j] = r / Rs[j * Ydim + j];
                q += a[j] * a[j];
                ld += logd[j];
            }
        w[i] = -0.5 * q + (-ld);
        i += istep;
        if (STRIDED && i < N) load(i);
    }
    STEP_STAMP(4095, 1);
    __syncthreads();
    if (tid < 64) {                              // max over the particles: a wavefront reduction (a max does not care about its order)
        double m = w[0];
        for (int k = tid; k < N; k += 64) m = (w[k] > m) ? w[k] : m;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) { const double o = __shfl_xor(m, sft); m = (o > m) ? o : m; }
        if (tid == 0) wmax_s = m;
    }
    __syncthreads();
    STEP_STAMP(4095, 2);
    for (int k = tid; k < N; k += istep) w[k] = exp(w[k] - wmax_s);
    __syncthreads();
    STEP_STAMP(4095, 3);
    if (tid == 0) {                              // the running sum in index order, eight reads ahead of the dependent adds
        double c = 0.0;                          // (35 cycles per element; the next block's reads in flight during the adds: no faster)
        int k = 0;
        for (; k + 8 <= N; k += 8) {
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t8[u] = w[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { c += t8[u]; cdf[k + u] = c; }
        }
        for (; k < N; ++k) { c += w[k]; cdf[k] = c; }
    }
    __syncthreads();
    STEP_STAMP(4095, 4);
    for (int r = tid; r < R; r += istep) {
        const double target = (STRIDED ? unif_t[r] : unl) * cdf[N - 1];
        int lo = 0, hi = N;                      // first k with cdf[k] > target
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > target) hi = mid; else lo = mid + 1;
        }
        const int k = (lo < N) ? lo : N - 1;
        idx_out[r] = k;
        for (int p = 0; p < D; ++p) {
            const double x = cands[k * D + p];
            parts_next[(size_t)r * D + p] = x;
            xc[r * P + p] = x;
        }
        if (ctrl_next)
            for (int c = 0; c < C; ++c) xc[r * P + D + c] = ctl[c];
    }
    __syncthreads();      // (a persistent caller reuses the LDS blocks)
    STEP_STAMP(4095, 5);
    STEP_SPAN_END(3);
    STEP_SPAN_NEXT();
}

// One particle-Gibbs step by one workgroup; the shapes decide the form (both forms of a caller -- per-step launch, persistent
// sweep -- come through here, so they agree bit for bit).
template <int STRIDED>
__device__ __forceinline__ void pg_step_body(const double *mean, const double *var, const double *log_Q,
                                             const double *eps_t, const double *unif_t, const double *y_t,
                                             const double *x_ref_next, const double *CC, const double *DD,
                                             const double *Rch, const double *ctrl_next, int R, int D, int C,
                                             int Ydim, double *xc, double *cand, double *parts_next, int32_t *idx_out) {
    __shared__ double w[PG_MAXN], cdf[PG_MAXN];
    __shared__ double wmax_s;
    if (D <= 8 && (R + 1) * D <= PG_FAST_ND)
        pg_step_fast_body<STRIDED>(w, cdf, wmax_s, mean, var, log_Q, eps_t, unif_t, y_t, x_ref_next, CC, DD, Rch, ctrl_next, R, D, C, Ydim,
                                   xc, parts_next, idx_out);
    else
        pg_step_general_body<STRIDED>(w, cdf, wmax_s, mean, var, log_Q, eps_t, unif_t, y_t, x_ref_next, CC, DD, Rch, ctrl_next, R, D, C,
                                      Ydim, xc, cand, parts_next, idx_out);
}


// ---- one persistent launch per step loop (loops.hip) ------------------------------------------------------------------------------
struct RolloutLoopArgs {
    ProjectArgs pa;                 // K_fu rows of the current states (pa.x is set per step from xbuf0 / xbuf1)
    SkinnyArgs sk;                  // F = K W with sum C^2 / C.u per slab, optionally |K (W q_sqrt)|^2 as the second right-hand side
    FinishIn f;
    const double *log_Q, *eps, *ctrl;      // eps [steps][R][D]; ctrl [steps][C] (row t + 1 feeds step t + 1) or null
    int R, C, steps;
    double *xbuf0, *xbuf1;          // the R x (D + C) input rows of step t live in xbuf[t & 1]
    double *predict_x, *predict_var;
    unsigned *bar;                  // counter block (loop_words ints), zero before the launch
    int *abort_w;                   // zero before the launch; non-zero afterwards: a wait gave up, the results are incomplete
};
struct PgLoopArgs {
    ProjectArgs pa;                 // pa.x: the R x (D + C) particle rows, advanced in place
    SkinnyArgs sk;
    int kind, R, D, C, Ydim, steps, ngs;
    const double *variance, *rowsq, *fmean;
    double *mean, *var, *cand, *parts;
    int32_t *idx;
    const double *log_Q, *eps, *unif, *Y, *X_ref, *CC, *DD, *Rch, *ctrl;
    unsigned *bar;
    int *abort_w;
};
// ---- the rollout loop with RESIDENT operands (loops.hip, rollout_resident_kernel) ------------------------------------------------
struct RolloutResidentArgs {
    int kind, R, RT, D, C, P, M, Mp, steps, NS;     // RT = row tiles of 16 rollouts, NS = Mp / 16 column slabs per dim
    HyperView hv;
    const double *W;  size_t w_stride;              // [D][Mp][Mp]  W = L^-T (upper triangular), row-major
    const double *WQ;                                // optional [D][Mp][Mp]  W q_sqrt
    int wq_upper;                                    // W q_sqrt is upper triangular like W (q_sqrt = L_H^-T): its rows below a slab's last column are skipped
    const double *ucol;                              // [D][Mp]
    const double *log_Q, *eps, *ctrl, *x_last;      // eps [steps][R][D]; ctrl [steps][C] or null; x_last [D]
    double *Kt;                                      // [D][Mp][16 RT]   K(x_t, Z) of the step, rollouts contiguous
    double *part;                                    // [D][NS][16 RT][4] per-slab sums of a row: sum F^2, F.u, sum (F q)^2
    double *xbuf;                                    // [2][16 RT][D]     the states of step t in xbuf[t & 1]
    double *predict_x, *predict_var;                 // [R][steps][D]
    int *words, *abort_w;                            // counter block (rollout_resident_words ints, zero before the launch), abort = words + 1
    int test_stall;                                  // tests (FFVD_RR_TEST_STALL=1): slab 0 of dim 0 leaves at step 3 -- every wait of the launch must give up
    long long *stamps;                               // optional (FFVD_RR_STAMPS=1, tools): [2][16] wall-clock stamps of step 10, slabs 0 and NS - 1 of dim 0
};
bool rollout_resident_ok(int R, int D, int P, int Mp);
int rollout_resident_words();
int launch_rollout_resident(hipStream_t stream, const RolloutResidentArgs &a);      // 0, or a hipError_t value

int loop_words(int nb);          // ints of the counter block (`bar`; abort word = bar[1]) for nb units, zero before the launch
void launch_rollout_loop(hipStream_t stream, const RolloutLoopArgs &a);
void launch_pg_loop(hipStream_t stream, const PgLoopArgs &a);

}  // namespace ffvd
