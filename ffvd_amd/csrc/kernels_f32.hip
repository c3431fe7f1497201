// fp32-contraction kernels of the collapsed branch (ffvd_config.dtype = FFVD_F32C; BASELINE configs[3]).
//
// What runs in fp32 (SURVEY.md section 7 "Conditioning"): the generation of K_fu (conditionals_multi_output.py:240), the
// product F = K_fu L^-T (:242) and the product F^T F (:246) -- the two T x M x M contractions -- on
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 157.3 TFLOP/s dense peak).  What stays in fp64: K_uu, its Cholesky and
// inverse factor (rounded to fp32 only as the GEMM operand), H = F^T F / Q + I from the moment it leaves the matrix
// cores, Cholesky(H), logdet, the solve (:253-254), delta^T F (:247-248), and the sum of F^2 of the trace term (:255).
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; it owns D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31] in accumulator register r.
// The sum over k is order-free up to rounding, so both kernels choose which k a lane half supplies at each step such
// that one 16- or 8-byte LDS read feeds several MFMAs (see the kernels).
#include "kernels_f32.h"

namespace ffvd {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16v mfma_f32(float a, float b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// K_fu in fp32: out[bz][t][m] = K_d(x_t, Z_m), zero for t >= T or m >= M.  Direct-difference form
// sum_p ((x_p - z_p) / l_p)^2 on inputs scaled in fp64 and rounded once: in fp32 the reference's expanded form
// |x|^2 + |z|^2 - 2 x.z (kernels_multi_output.py:170-182) would lose three digits to cancellation.
// 64 x 64 tile per workgroup; thread = 4 consecutive columns x 4 rows, 16-byte stores.
// ---------------------------------------------------------------------------------------------
// SMALLP (P <= 12): inputs zero-padded to 12 components, the thread's four inducing inputs live in registers and a
// row of x arrives as three 16-byte broadcast reads -- the build then stays HBM-write-bound instead of LDS-bound.
constexpr int KP = 12;
template <int KIND, bool SMALLP>
__global__ __launch_bounds__(256) void kfu_build_f32_kernel(ProjectArgs a, float *out) {
    constexpr int XLD = SMALLP ? KP : MAXP + 1;
    __shared__ __attribute__((aligned(16))) float xs[64][XLD];
    __shared__ __attribute__((aligned(16))) float zs[64][XLD];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 64, m0 = blockIdx.y * 64, bz = blockIdx.z;
    const int b = a.b0 + bz, s = b / a.Dl, dl = b % a.Dl;
    const int P = a.P, Mp = a.Mp;
    const int PP = SMALLP ? KP : P;
    const double var = a.hv.variance[dl];
    for (int idx = tid; idx < 64 * PP; idx += 256) {
        const int r = idx / PP, p = idx % PP;
        const int t = t0 + r;
        double v = 0.0, z = 0.0;
        if (p < P) {
            if (t < a.T) {
                v = (p < a.x_cols) ? a.x[(size_t)s * a.x_chain_stride + (size_t)t * a.x_ld + p]
                                   : a.ctrl[(size_t)t * a.C + (p - a.x_cols)];
                if (KIND == 0) v = v / a.hv.len[(size_t)dl * P + p];
                else v = v * var;                               // LinearK: (X * variance) X2^T  (kernels.py:276)
            }
            z = a.hv.Zs[((size_t)dl * Mp + m0 + r) * P + p];
        }
        xs[r][p] = (float)v;
        zs[r][p] = (float)z;
    }
    __syncthreads();
    const int tr = tid >> 4, cg = tid & 15;
    const float fvar = (float)var;
    float *ob = out + ((size_t)bz * a.Tp + t0) * Mp + m0 + 4 * cg;
    float zr[4][KP];
    if (SMALLP) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p4 = 0; p4 < KP / 4; ++p4) {
                const float4 v = *reinterpret_cast<const float4 *>(&zs[4 * cg + q][4 * p4]);
                zr[q][4 * p4] = v.x; zr[q][4 * p4 + 1] = v.y; zr[q][4 * p4 + 2] = v.z; zr[q][4 * p4 + 3] = v.w;
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = tr + 16 * i;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (SMALLP) {
#pragma unroll
            for (int p4 = 0; p4 < KP / 4; ++p4) {
                const float4 xv4 = *reinterpret_cast<const float4 *>(&xs[r][4 * p4]);
                const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (KIND == 0) { const float d = xv[pp] - zr[q][4 * p4 + pp]; acc[q] = fmaf(d, d, acc[q]); }
                        else acc[q] = fmaf(xv[pp], zr[q][4 * p4 + pp], acc[q]);
                    }
            }
        } else {
            for (int p = 0; p < P; ++p) {
                const float xv = xs[r][p];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float zv = zs[4 * cg + q][p];
                    if (KIND == 0) { const float d = xv - zv; acc[q] = fmaf(d, d, acc[q]); }
                    else acc[q] = fmaf(xv, zv, acc[q]);
                }
            }
        }
        float4 v;
        float *vv = reinterpret_cast<float *>(&v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float k = (KIND == 0) ? fvar * __expf(-0.5f * acc[q]) : acc[q];
            if (t0 + r >= a.T || m0 + 4 * cg + q >= a.M) k = 0.f;
            vv[q] = k;
        }
        *reinterpret_cast<float4 *>(ob + (size_t)r * Mp) = v;
    }
}
void launch_kfu_build_f32(hipStream_t stream, const ProjectArgs &a, float *out) {
    dim3 grid(a.Tp / 64, a.Mp / 64, a.nb);
    if (a.P <= KP) {
        if (a.kind == 0) hipLaunchKernelGGL((kfu_build_f32_kernel<0, true>), grid, dim3(256), 0, stream, a, out);
        else hipLaunchKernelGGL((kfu_build_f32_kernel<1, true>), grid, dim3(256), 0, stream, a, out);
    } else {
        if (a.kind == 0) hipLaunchKernelGGL((kfu_build_f32_kernel<0, false>), grid, dim3(256), 0, stream, a, out);
        else hipLaunchKernelGGL((kfu_build_f32_kernel<1, false>), grid, dim3(256), 0, stream, a, out);
    }
}

// out32[dl][j][k] = (float) ext[dl][k][j]: the L^-T rows of the extended Cholesky slab (upper triangular, ld Mp)
// transposed into L^-1 (row j = column j of L^-T, contiguous along the contraction index k) and rounded to fp32.
__global__ __launch_bounds__(256) void linv_f32_kernel(const double *ext, size_t ext_stride, float *out, int Mp) {
    __shared__ float t[64][65];
    const int dl = blockIdx.z, bk = blockIdx.y * 64, bj = blockIdx.x * 64, tid = threadIdx.x;
    const double *I = ext + (size_t)dl * ext_stride;
    float *O = out + (size_t)dl * Mp * Mp;
    for (int r = tid >> 6; r < 64; r += 4) t[r][tid & 63] = (float)I[(size_t)(bk + r) * Mp + bj + (tid & 63)];
    __syncthreads();
    for (int r = tid >> 6; r < 64; r += 4) O[(size_t)(bj + r) * Mp + bk + (tid & 63)] = t[tid & 63][r];
}
void launch_linv_f32(hipStream_t stream, const double *ext, size_t ext_stride, float *out, int Mp, int Dl) {
    hipLaunchKernelGGL(linv_f32_kernel, dim3(Mp / 64, Mp / 64, Dl), dim3(256), 0, stream, ext, ext_stride, out, Mp);
}

// ---------------------------------------------------------------------------------------------
// F = K_fu L^-T in fp32 (conditionals_multi_output.py:242) as an "NT" product: both operands are contiguous along
// the contraction index k (K_fu row-major, B^T = L^-1 row-major), so a lane fetches four consecutive k of its row
// with ONE 16-byte LDS read and the two lane halves take the two halves of each 8-k group: MFMA step s of group q
// contracts k = 8q + s (lanes 0-31) and k = 8q + 4 + s (lanes 32-63), for A and B alike.
// Workgroup = 128 x 128 tile, 4 wavefronts of 64 x 64 (2 x 2 accumulators of 32 x 32: 64 registers), k-tile 32,
// register-staged double-buffered LDS with rows padded to 144 B (16 lanes x 16 B hit 16 distinct bank quads).
// L^-T is upper triangular: column tile tj only contracts k < (tj + 1) 128 (and a wavefront stops at its own last column
// inside that last block: 1.03 x the triangular flop count); heavy column tiles are dispatched first.
// Epilogue: F (fp32) to HBM; sum of F^2 accumulated in fp64 per tile (the trace term's cancellation stays in fp64).
// ---------------------------------------------------------------------------------------------
#ifndef FFVD_F32_KT
#define FFVD_F32_KT 32               // k-tile of both fp32 products (tuning builds: -DFFVD_F32_KT=16 -DFFVD_F32_OCC=3)
#endif
#ifndef FFVD_F32_OCC
#define FFVD_F32_OCC 2               // workgroups per CU the register budget is sized for
#endif
constexpr int PK = FFVD_F32_KT;      // k-tile
constexpr int P_LD = PK + 4;         // LDS row stride in floats (144 B)

__global__ __launch_bounds__(256, FFVD_F32_OCC) void proj_gemm_f32_kernel(ProjF32Args a) {
    __shared__ __attribute__((aligned(16))) float As[2][128][P_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][128][P_LD];
    __shared__ double red[4];
    const int bz = blockIdx.y;
    const int ntj = a.Mp / 128 + ((a.Mp % 128) ? 1 : 0);
    const int nti = a.Tp / 128 + ((a.Tp % 128) ? 1 : 0);
    const int tj = ntj - 1 - (int)(blockIdx.x / nti), ti = blockIdx.x % nti;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int Mp = a.Mp, Tp = a.Tp;
    const int b = a.b0 + bz, dl = b % a.Dl;
    const float *Kfb = a.Kf + (size_t)bz * a.kf_stride;
    const bool full = a.Bunit != nullptr;                   // full symmetric right operand per unit (backward pass)
    const float *Wb = full ? a.Bunit + (size_t)bz * a.bunit_stride : a.LinvT + (size_t)dl * Mp * Mp;
    const int kend = (!full && (tj + 1) * 128 < Mp) ? (tj + 1) * 128 : Mp;
    const int nchunk = kend / PK;
    // inside the diagonal k-block a wavefront stops at its own last column (W[k][j] = 0 for k > j)
    const int my_chunks = full ? nchunk : ((tj * 128 + wc * 64 + 64 < kend) ? tj * 128 + wc * 64 + 64 : kend) / PK;

    // staging: a row of the k-tile is PK / 4 lanes x 16 bytes; thread moves rows sr + SROWS i, columns sc of each operand
    constexpr int LPR = PK / 4, SROWS = 256 / LPR, NPASS = 128 / SROWS;
    static_assert(PK % 8 == 0 && NPASS >= 1, "k-tile must be a multiple of 8 and at most 128");
    const int sr = tid / LPR, sc = 4 * (tid % LPR);
    f4v ra[NPASS], rb[NPASS];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int row = sr + SROWS * i;
            int ta = ti * 128 + row; ta = ta < Tp ? ta : Tp - 1;          // clamped: rows beyond Tp are never stored
            int jb = tj * 128 + row; jb = jb < Mp ? jb : Mp - 1;
            ra[i] = *reinterpret_cast<const f4v *>(Kfb + (size_t)ta * Mp + c * PK + sc);
            rb[i] = *reinterpret_cast<const f4v *>(Wb + (size_t)jb * Mp + c * PK + sc);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            *reinterpret_cast<f4v *>(&As[buf][sr + SROWS * i][sc]) = ra[i];
            *reinterpret_cast<f4v *>(&Bs[buf][sr + SROWS * i][sc]) = rb[i];
        }
    };
    f16v acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload(c + 1);
        if (c < my_chunks) {
#pragma unroll
        for (int q = 0; q < PK / 8; ++q) {
            const f4v a0 = *reinterpret_cast<const f4v *>(&As[buf][wr * 64 + lr][8 * q + 4 * lh]);
            const f4v a1 = *reinterpret_cast<const f4v *>(&As[buf][wr * 64 + 32 + lr][8 * q + 4 * lh]);
            const f4v b0 = *reinterpret_cast<const f4v *>(&Bs[buf][wc * 64 + lr][8 * q + 4 * lh]);
            const f4v b1 = *reinterpret_cast<const f4v *>(&Bs[buf][wc * 64 + 32 + lr][8 * q + 4 * lh]);
#define FFVD_PSTEP(C)                                          \
            acc[0][0] = mfma_f32(a0.C, b0.C, acc[0][0]);           \
            acc[0][1] = mfma_f32(a0.C, b1.C, acc[0][1]);           \
            acc[1][0] = mfma_f32(a1.C, b0.C, acc[1][0]);           \
            acc[1][1] = mfma_f32(a1.C, b1.C, acc[1][1]);
            FFVD_PSTEP(x) FFVD_PSTEP(y) FFVD_PSTEP(z) FFVD_PSTEP(w)
#undef FFVD_PSTEP
        }
        }
        if (c + 1 < nchunk) lstore(buf ^ 1);
        __syncthreads();
    }
    // epilogue
    float *Fb = a.F + (size_t)bz * a.f_stride;
    const int I0 = ti * 128 + wr * 64, J0 = tj * 128 + wc * 64;
    double ssq = 0.0;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = I0 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh, j = J0 + 32 * y + lr;
                const float f = acc[x][y][r];
                if (i < Tp && j < Mp) {
                    Fb[(size_t)i * Mp + j] = f;
                    ssq = fma((double)f, (double)f, ssq);
                }
            }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) ssq += __shfl_xor(ssq, m);
    if (lane == 0) red[wave] = ssq;
    __syncthreads();
    if (tid == 0 && a.sqpart) a.sqpart[(size_t)b * (nti * ntj) + (size_t)tj * nti + ti] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void to_f32_kernel(const double *in, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
void launch_to_f32(hipStream_t stream, const double *in, float *out, size_t n) {
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(to_f32_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, in, out, n);
}
int proj_f32_ntiles(int Tp, int Mp) { return ((Tp + 127) / 128) * ((Mp + 127) / 128); }
void launch_proj_gemm_f32(hipStream_t stream, const ProjF32Args &a) {
    const int nti = (a.Tp + 127) / 128, ntj = (a.Mp + 127) / 128;
    hipLaunchKernelGGL(proj_gemm_f32_kernel, dim3(nti * ntj, a.nb), dim3(256), 0, stream, a);
}

// out[b] = sum of the per-tile partials (fixed order)
__global__ __launch_bounds__(256) void sum_partials_kernel(const double *part, int n, double *out) {
    __shared__ double scratch[4][1];
    const int b = blockIdx.x;
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += 256) v[0] += part[(size_t)b * n + i];
    block_sum_multi_256<1>(v, scratch);
    if (threadIdx.x == 0) out[b] = v[0];
}
void launch_sum_partials(hipStream_t stream, const double *part, int n, int nb, double *out) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(nb), dim3(256), 0, stream, part, n, out);
}

// ---------------------------------------------------------------------------------------------
// H = F^T F / Q + I from the fp32 F (conditionals_multi_output.py:246) and b = delta^T F / Q (:247-248).
// "TN" product: the contraction index t is the ROW index of F, so both operands sit in LDS as [t][column] and a lane
// reads its operand for a given t with the columns contiguous.  Each 64-wide operand strip is dealt to the two
// 32 x 32 accumulator blocks by column parity -- block x holds columns 2 r + x -- so ONE 8-byte read per lane feeds
// both blocks; the lane halves take t and t + 1 of each step.  Output element (block x, y; register r; lane c, h):
//     i = I0 + 2 ((r & 3) + 8 (r >> 2) + 4 h) + x,   j = J0 + 2 c + y      (two adjacent j per lane: 16-byte stores).
// Workgroup = 128 x 128 lower-triangular tile, 4 wavefronts of 64 x 64, t-tile 32, XCD-aware tile map (all tiles of a
// unit on one XCD's L2).  Diagonal tiles: the quadrant above the diagonal has no work, its wavefront forms
// delta^T F in fp64 instead.  The fp32 accumulators are flushed into the fp64 tile in HBM every `flush` t-tiles
// (flush = 0: once at the end), which bounds the length of any fp32 summation chain.
// ---------------------------------------------------------------------------------------------
constexpr int GK = FFVD_F32_KT;      // t-tile
constexpr int GF_LD = 128;           // LDS row stride in floats

template <bool DIAG>
__device__ __forceinline__ void gram_f32_body(const GramF32Args &a, const int bz, const int ti, const int tj,
                                              float (*As)[GK][GF_LD], float (*Bs)[GK][GF_LD], double (*dls)[GK]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lc = lane & 31, lh = lane >> 5;
    const int Mp = a.Mp;
    const int b = a.b0 + bz, s = b / a.Dl, dl = b % a.Dl, dg = a.d_begin + dl;
    const bool gemv = DIAG && wr == 0 && wc == 1;            // strictly above the diagonal: no tile work
    const float *Fb = a.F + (size_t)bz * a.f_stride;
    const double *Xs = a.X + (size_t)s * (a.T + 1) * a.D;
    const int sr = tid >> 5, sc = 4 * (tid & 31);            // staging: rows sr + 8 i, 16 bytes at column sc
    const int colA = ti * 128 + sc, colB = tj * 128 + sc;
    const bool okA = colA < Mp, okB = colB < Mp;
    constexpr int NPASS = GK / 8;
    static_assert(GK % 8 == 0 && GK <= 256, "t-tile must be a multiple of 8");
    f4v ra[NPASS], rb[NPASS];
    double dreg = 0.0;
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const size_t t = (size_t)c * GK + sr + 8 * i;
            ra[i] = okA ? *reinterpret_cast<const f4v *>(Fb + t * Mp + colA) : (f4v){0.f, 0.f, 0.f, 0.f};
            if (!DIAG) rb[i] = okB ? *reinterpret_cast<const f4v *>(Fb + t * Mp + colB) : (f4v){0.f, 0.f, 0.f, 0.f};
        }
        if (DIAG && tid < GK) {
            const int tt = c * GK + tid;
            dreg = (tt < a.T) ? Xs[(size_t)(tt + 1) * a.D + dg] - Xs[(size_t)tt * a.D + dg] : 0.0;    // :247
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            *reinterpret_cast<f4v *>(&As[buf][sr + 8 * i][sc]) = ra[i];
            if (!DIAG) *reinterpret_cast<f4v *>(&Bs[buf][sr + 8 * i][sc]) = rb[i];
        }
        if (DIAG && tid < GK) dls[buf][tid] = dreg;
    };
    f16v acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    double bs0 = 0.0, bs1 = 0.0;
    const double scale = a.yn_over_batch / exp(a.log_Q[dg]);
    double *Hb = a.H + (size_t)bz * a.h_stride;
    const int I0 = ti * 128 + wr * 64, J0 = tj * 128 + wc * 64;
    const bool active = !gemv && I0 < Mp && J0 < Mp;
    // fp64 tile in HBM <- (first ? 0 : tile) + fp32 accumulators; on the last flush the epilogue of :246 is applied
    auto flush = [&](bool first, bool last) {
        if (!active) return;
        // the 32 tile addresses are loop-invariant; hoisted out of the t loop they would sit in 64 registers for the whole
        // kernel: keep their computation here by making the base opaque to the optimiser
        double *hb = Hb + (size_t)(I0 + 8 * lh) * Mp + J0 + 2 * lc;
        asm volatile("" : "+v"(hb));
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = I0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + x, j = J0 + 2 * lc;
                d2v *p = reinterpret_cast<d2v *>(hb + (size_t)(2 * ((r & 3) + 8 * (r >> 2)) + x) * Mp);
                d2v v = first ? (d2v){0.0, 0.0} : *p;
                v.x += (double)acc[x][0][r];
                v.y += (double)acc[x][1][r];
                if (last) {
                    v.x = v.x * scale + ((i == j) ? 1.0 : 0.0);
                    v.y = v.y * scale + ((i == j + 1) ? 1.0 : 0.0);
                }
                *p = v;
                acc[x][0][r] = 0.f;
                acc[x][1][r] = 0.f;
            }
    };
    const int nchunk = a.rows / GK;
    const int fl = a.flush > 0 ? (a.flush * 32 / GK > 0 ? a.flush * 32 / GK : 1) : nchunk;      // a.flush counts 32-row units
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload(c + 1);
        if (active) {
            const float(*Bp)[GF_LD] = DIAG ? As[buf] : Bs[buf];
#pragma unroll
            for (int st = 0; st < GK / 2; ++st) {
                const f2v av = *reinterpret_cast<const f2v *>(&As[buf][2 * st + lh][wr * 64 + 2 * lc]);
                const f2v bv = *reinterpret_cast<const f2v *>(&Bp[2 * st + lh][wc * 64 + 2 * lc]);
                acc[0][0] = mfma_f32(av.x, bv.x, acc[0][0]);
                acc[0][1] = mfma_f32(av.x, bv.y, acc[0][1]);
                acc[1][0] = mfma_f32(av.y, bv.x, acc[1][0]);
                acc[1][1] = mfma_f32(av.y, bv.y, acc[1][1]);
            }
        }
        if (gemv) {
#pragma unroll 8
            for (int r = 0; r < GK; ++r) {
                const f2v f = *reinterpret_cast<const f2v *>(&As[buf][r][2 * lane]);
                const double d = dls[buf][r];
                bs0 = fma(d, (double)f.x, bs0);
                bs1 = fma(d, (double)f.y, bs1);
            }
        }
        if (c + 1 < nchunk) lstore(buf ^ 1);
        __syncthreads();
        if ((c + 1) % fl == 0 && c + 1 < nchunk) flush(c + 1 == fl, false);
    }
    flush(nchunk <= fl, true);
    if (gemv && a.with_row) {
        const int col = ti * 128 + 2 * lane;
        if (col < Mp) {
            d2v v = (d2v){bs0 * scale, bs1 * scale};
            *reinterpret_cast<d2v *>(Hb + (size_t)a.brow * Mp + col) = v;
        }
    }
}

__global__ __launch_bounds__(256, FFVD_F32_OCC) void gram_f32_kernel(GramF32Args a) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][GF_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][GF_LD];
    __shared__ double dls[2][GK];
    const int id = blockIdx.x;
    const int xcd = id & 7, loc = id >> 3;
    const int bz = (loc / a.ntiles) * 8 + xcd;
    if (bz >= a.nb) return;
    const int tile = loc % a.ntiles;
    int ti = 0;                       // tile = ti (ti + 1) / 2 + tj,  tj <= ti
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    const int tj = tile - ti * (ti + 1) / 2;
    if (ti == tj) gram_f32_body<true>(a, bz, ti, tj, As, Bs, dls);
    else gram_f32_body<false>(a, bz, ti, tj, As, Bs, dls);
}
void launch_gram_f32(hipStream_t stream, GramF32Args a) {
    a.ntiles = gram_ntiles(a.Mp);
    if (a.brow <= 0) a.brow = a.Mp;
    const int groups = (a.nb + 7) / 8;
    hipLaunchKernelGGL(gram_f32_kernel, dim3(groups * 8 * a.ntiles), dim3(256), 0, stream, a);
}

}  // namespace ffvd
