// Hand-written HIP kernels of the FFVD ELBO hot path for gfx950 (MI355X, CDNA4), fp64.
//
// Reference arithmetic (TensorFlow op call sites):
//   K1/K2  kernels_multi_output.py:163-182,246-247 (SE), kernels.py:270-281 (LinearK)
//   K3/K4  conditionals_multi_output.py:159-166  chol(K_uu + jitter I), L^{-T}
//   K5b    conditionals_multi_output.py:242      F = K_fu L^{-T}
//   K6/K7  conditionals_multi_output.py:246-248  H = F^T F / Q + I,  b = delta^T F / Q
//   K9/K10 conditionals_multi_output.py:253-254  logdet H, b H^{-1} b^T
//   K8     conditionals_multi_output.py:255, :41 per-point explained variance
//   K11    conditionals_multi_output.py:48       GP mean A^T u
//   K12/13 likelihoods.py:76-111, dgp_model.py:105-143,248-297,326-359
//
// Design notes (DESIGN.md has the full story):
//   * every M x M matrix is padded to Mp = multiple of 64 with an identity block, T to a multiple of 64 with
//     zero rows of K_fu, so no kernel has edge handling along M or T;
//   * dense contractions run on v_mfma_f64_16x16x4_f64 (one f64 A and B element per lane, 4 accumulators per lane);
//   * K_fu is produced tile-wise in LDS and consumed by the MFMA in the same workgroup; it never reaches HBM;
//   * Cholesky is a blocked right-looking factorisation whose "extra rows" carry a right-hand side through the
//     same panel/trailing kernels: identity rows become L^{-T}, the row delta^T F / Q becomes L_H^{-1} b.
#include "kernels.h"
#include "dev_common.h"
#define FFVD_STEP_TRACE_OWNER
#include "step_bodies.h"
#if defined(FFVD_STEP_TRACE)
extern "C" int ffvd_debug_step_spans(unsigned long long *out) {      // [64 steps][4 kernels][first workgroup's start, a late end]
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ffvd::step_span), sizeof(unsigned long long) * 64 * 4 * 2);
}
extern "C" int ffvd_debug_step_trace(long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ffvd::step_trace_buf), sizeof(long long) * 4096 * 8);
}
#endif
#include <type_traits>
#include <cstdlib>

namespace ffvd {

__device__ __forceinline__ double block_sum_256(double v, double *scratch /*[256]*/) {
    const int tid = threadIdx.x;
    scratch[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) scratch[tid] += scratch[tid + s];
        __syncthreads();
    }
    double r = scratch[0];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------------------------
__global__ void fill_kernel(double *p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
// Diagnostic (schedule tests, FFVD_DEBUG_SIDE_DELAY_US / FFVD_DEBUG_MAIN_DELAY_US): one wavefront that occupies its stream for `us`
// microseconds of the 100 MHz wall clock and touches no memory.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void launch_spin(hipStream_t stream, int us) {
    if (us > 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, (long long)us * 100);
}
void launch_fill(hipStream_t stream, double *p, size_t n, double v) {
    if (n == 0) return;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, stream, p, n, v);
}

// ---------------------------------------------------------------------------------------------
// hyper-parameter preparation: variance = exp(logvariance) (kernels_multi_output.py:157),
// lengthscales = exp(loglengthscales) (:161), Zs = Z / lengthscales (:170), zz = reduce_sum(square(Zs)) (:171)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_hypers_kernel(int kind, const double *Z, int M, int Mp, int P, int d_begin,
                                                          const double *logvar, const double *loglen,
                                                          double *variance, double *len, double *Zs, double *zz,
                                                          int32_t *info, int ninfo) {
    __shared__ double ls[MAXP];
    const int dl = blockIdx.x, dg = d_begin + dl, tid = threadIdx.x;
    if (info && dl == 0)                             // re-arm the factorisation flags of this iteration
        for (int i = tid; i < ninfo; i += 256) info[i] = 0;
    if (tid == 0) variance[dl] = exp(logvar[dg]);
    if (tid < P) {
        double l = (kind == 0) ? exp(loglen[(size_t)dg * P + tid]) : 1.0;
        ls[tid] = l;
        len[(size_t)dl * P + tid] = l;
    }
    __syncthreads();
    for (int m = tid; m < Mp; m += 256) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) {
            double v = (m < M) ? Z[(size_t)m * P + p] / ls[p] : 0.0;
            Zs[((size_t)dl * Mp + m) * P + p] = v;
            s += v * v;
        }
        zz[(size_t)dl * Mp + m] = s;
    }
}
void launch_prep_hypers(hipStream_t stream, int kind, const double *Z, int M, int Mp, int P, int Dl, int d_begin,
                        const double *logvar, const double *loglen, double *variance, double *len,
                        double *Zs, double *zz, int32_t *info, int ninfo) {
    hipLaunchKernelGGL(prep_hypers_kernel, dim3(Dl), dim3(256), 0, stream, kind, Z, M, Mp, P, d_begin, logvar,
                       loglen, variance, len, Zs, zz, info, ninfo);
}

// One workgroup = KUU_ROWS rows x 256 consecutive columns.  The 256 rows z_j are staged once through LDS with coalesced loads
// (row stride P | 1 doubles: conflict-free reads whatever P) and reused for every row i, whose z_i is a scalar operand.  The
// first version (one output per thread, 256 per workgroup, operands from global) was bound by the dispatch of its tiny
// workgroups: 147 us for the 16 matrices of config 5 (32768 workgroups), 12 us for the 4 of config 2.
// The row operands z_i (and |z_i|^2) go through LDS too and the inner product runs over a compile-time PM >= P with zero padding
// (same terms in the same order, then zeros): with a run-time P the p loop was not unrolled and read z_i[p] from global memory
// with a wait per iteration -- 19 us for the 4 matrices of config 2, 37-46 us for the 16 of config 5 (P = 17), for a few
// microseconds of arithmetic.
constexpr int KUU_ROWS = 16;
template <int PM>
__global__ __launch_bounds__(256) void kuu_build_kernel(int kind, HyperView hv, int M, int Mp, int P, double jitter,
                                                        double *A, double *Kcopy, int zt_rows, double *zero, int nzero) {
    extern __shared__ double zs_lds[];                    // [256][P | 1] | zi [KUU_ROWS][PM] | zzi [KUU_ROWS]
    const int dl = blockIdx.z, tid = threadIdx.x;
    // the progress words of the dataflow Cholesky that follows on the same stream (launch_kuu_build, flow_words): one launch less
    if (zero && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = tid; i < nzero; i += 256) zero[i] = 0.0;
    const int ncc = (Mp + 255) / 256;                     // column chunks per row
    const int i0 = (int)(blockIdx.x / ncc) * KUU_ROWS, j0 = (int)(blockIdx.x % ncc) * 256;
    const int j = j0 + tid;
    double *slab = A + (size_t)dl * 2 * Mp * Mp;
    if (blockIdx.y == 1) {   // extra rows: identity, becomes L^{-T}
        if (zt_rows) {       // ... or (LinearK through its rank) ONE 64-row block holding Z^T, which becomes C = Z^T L^-T: rows >= P zero
            if (j < Mp && i0 < NB)
                for (int r = 0; r < KUU_ROWS; ++r) {
                    const int i = i0 + r;
                    slab[(size_t)Mp * Mp + (size_t)i * Mp + j] = (i < P) ? hv.Zs[((size_t)dl * Mp + j) * P + i] : 0.0;
                }
            return;
        }
        if (j < Mp)
            for (int r = 0; r < KUU_ROWS && i0 + r < Mp; ++r) slab[(size_t)Mp * Mp + (size_t)(i0 + r) * Mp + j] = (i0 + r == j) ? 1.0 : 0.0;
        return;
    }
    const int ld = P | 1;
    {
        const int nrow = (j0 + 256 <= Mp) ? 256 : Mp - j0;
        const double *zsrc = hv.Zs + ((size_t)dl * Mp + j0) * P;
        int r = tid / P, c = tid % P;                     // element e = tid + 256 k  <->  (r, c), advanced without divisions
        const int dr = 256 / P, dc = 256 % P;
        for (int e = tid; e < nrow * P; e += 256) {
            zs_lds[r * ld + c] = zsrc[e];
            r += dr; c += dc;
            if (c >= P) { c -= P; ++r; }
        }
    }
    double *zi_lds = zs_lds + (size_t)256 * ld, *zzi_lds = zi_lds + KUU_ROWS * PM;
    for (int e = tid; e < KUU_ROWS * PM; e += 256) {
        const int rr = e / PM, pp = e % PM, i = i0 + rr;
        zi_lds[e] = (pp < P && i < Mp) ? hv.Zs[((size_t)dl * Mp + i) * P + pp] : 0.0;
    }
    if (tid < KUU_ROWS) zzi_lds[tid] = (kind == 0 && i0 + tid < Mp) ? hv.zz[(size_t)dl * Mp + i0 + tid] : 0.0;
    __syncthreads();
    if (j >= Mp) return;
    double zl[PM];
#pragma unroll
    for (int p = 0; p < PM; ++p) zl[p] = (p < P) ? zs_lds[(size_t)tid * ld + p] : 0.0;
    const double var = hv.variance[dl];
    const double zzj = (kind == 0 && j < M) ? hv.zz[(size_t)dl * Mp + j] : 0.0;
    for (int r = 0; r < KUU_ROWS && i0 + r < Mp; ++r) {
        const int i = i0 + r;
        double v;
        if (i >= M || j >= M) {
            v = (i == j) ? 1.0 : 0.0;
        } else {
            const double *zi = zi_lds + r * PM;
            double dot = 0.0;
            if (kind == 0) {
#pragma unroll
                for (int p = 0; p < PM; ++p) dot += zi[p] * zl[p];
                v = kernel_value<0>(dot, zzi_lds[r], zzj, var);
            } else {
#pragma unroll
                for (int p = 0; p < PM; ++p) dot += (zi[p] * var) * zl[p];
                v = dot;
            }
            if (i == j) v += jitter;   // conditionals_multi_output.py:108,159
        }
        const size_t idx = (size_t)i * Mp + j;
        slab[idx] = v;
        if (Kcopy) Kcopy[(size_t)dl * Mp * Mp + idx] = v;     // the factorisation overwrites `slab` in place
    }
}
void launch_kuu_build(hipStream_t stream, int kind, HyperView hv, int M, int Mp, int P, int Dl, double jitter, double *A,
                      double *Kcopy, bool zt_rows, double *flow_words) {
    // (what potrf_flow_clear zeroes for Dl matrices)
    const int nzero = flow_words ? (int)((((size_t)Dl * 64 + 4 + 3) / 4 * 4) / 2) : 0;
    const int ncc = (Mp + 255) / 256, nrg = (Mp + KUU_ROWS - 1) / KUU_ROWS;
    dim3 grid((unsigned)(ncc * nrg), 2, Dl);
    if (P <= 8)
        hipLaunchKernelGGL(kuu_build_kernel<8>, grid, dim3(256), ((size_t)256 * (P | 1) + KUU_ROWS * 9) * sizeof(double), stream, kind, hv, M,
                           Mp, P, jitter, A, Kcopy, zt_rows ? 1 : 0, flow_words, nzero);
    else
        hipLaunchKernelGGL(kuu_build_kernel<MAXP>, grid, dim3(256), ((size_t)256 * (P | 1) + KUU_ROWS * (MAXP + 1)) * sizeof(double), stream,
                           kind, hv, M, Mp, P, jitter, A, Kcopy, zt_rows ? 1 : 0, flow_words, nzero);
}

// out[dl][i][j] = in[dl][j][i] for Dl square Mp x Mp matrices (L^-T -> L^-1)
__global__ __launch_bounds__(256) void transpose_kernel(const double *in, size_t in_stride, double *out, size_t out_stride,
                                                        int Mp) {
    __shared__ double t[64][65];
    const int dl = blockIdx.z, bi = blockIdx.y * 64, bj = blockIdx.x * 64, tid = threadIdx.x;
    const double *I = in + (size_t)dl * in_stride;
    double *O = out + (size_t)dl * out_stride;
    for (int r = tid >> 6; r < 64; r += 4) t[r][tid & 63] = I[(size_t)(bi + r) * Mp + bj + (tid & 63)];
    __syncthreads();
    for (int r = tid >> 6; r < 64; r += 4) O[(size_t)(bj + r) * Mp + bi + (tid & 63)] = t[tid & 63][r];
}
void launch_transpose(hipStream_t stream, const double *in, size_t in_stride, double *out, size_t out_stride, int Mp,
                      int Dl) {
    hipLaunchKernelGGL(transpose_kernel, dim3(Mp / 64, Mp / 64, Dl), dim3(256), 0, stream, in, in_stride, out,
                       out_stride, Mp);
}

// K_fu materialised (only the K_uu + K_uf K_fu / Q route needs it in HBM): out[bz][t][m] = K_d(x_t, Z_m),
// zero for t >= T or m >= M.  64 x 64 tile per workgroup; thread (tid & 63) owns a column, 16 rows.
// SMALLP (P <= 8): x / l is kept row-major and zero-padded to 8 components in LDS, so a row is four 16-byte
// broadcast reads and the dot product eight unconditional FMAs (with a run-time `p < P` test the compiler emits one
// LDS round trip and one scalar branch per component: 0.55 instead of 0.40 ms for the 2 GiB of config 2).
// NQ > 0: the SMALLP form with 2 NQ components (P = 5: six FMAs and three reads per element instead of eight and four -- the
// build is bound by its fp64 VALU work as much as by the write); NQ = 0: the general form.
template <int KIND, int NQ, bool NT>
__global__ __launch_bounds__(256) void kfu_build_kernel(ProjectArgs a) {
    kfu_build_body<KIND, NQ, NT>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}


__global__ __launch_bounds__(256) void brow_finish_kernel(const double *gpart, int nblk, int Mp, int Dl, int d_begin, int b0,
                                                          const double *log_Q, double yn_over_batch, double *H, size_t h_stride, int brow) {
    const int m = blockIdx.x * 256 + threadIdx.x, bz = blockIdx.y;
    if (m >= Mp) return;
    const double *g = gpart + (size_t)bz * nblk * Mp + m;
    double acc = 0.0;
    for (int k0 = 0; k0 < nblk; k0 += 8) {      // eight independent loads in flight, one fixed summation order
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (k0 + k < nblk) ? g[(size_t)(k0 + k) * Mp] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    const int dg = d_begin + (b0 + bz) % Dl;
    H[(size_t)bz * h_stride + (size_t)brow * Mp + m] = acc * (yn_over_batch / exp(log_Q[dg]));
}
void launch_brow_finish(hipStream_t stream, const double *gpart, int nblk, int Mp, int Dl, int d_begin, int b0, int nb,
                        const double *log_Q, double yn_over_batch, double *H, size_t h_stride, int brow) {
    hipLaunchKernelGGL(brow_finish_kernel, dim3((Mp + 255) / 256, nb), dim3(256), 0, stream, gpart, nblk, Mp, Dl, d_begin, b0, log_Q,
                       yn_over_batch, H, h_stride, brow);
}
template <bool NT>
static void launch_kfu_build_nt(hipStream_t stream, const ProjectArgs &a) {
    dim3 grid(a.Tp / 64, a.Mp / 64, a.nb);
    if (a.P <= 8) {
        const int nq = (a.P + 1) / 2;
        if (a.kind == 0) {
            if (nq <= 2) hipLaunchKernelGGL((kfu_build_kernel<0, 2, NT>), grid, dim3(256), 0, stream, a);
            else if (nq == 3) hipLaunchKernelGGL((kfu_build_kernel<0, 3, NT>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((kfu_build_kernel<0, 4, NT>), grid, dim3(256), 0, stream, a);
        } else hipLaunchKernelGGL((kfu_build_kernel<1, 4, NT>), grid, dim3(256), 0, stream, a);
    } else {
        if (a.kind == 0) hipLaunchKernelGGL((kfu_build_kernel<0, 0, NT>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((kfu_build_kernel<1, 0, NT>), grid, dim3(256), 0, stream, a);
    }
}
template <int KIND, int NQ, int MW>
__global__ __launch_bounds__(256) void kfu_build_t_kernel(ProjectArgs a, int ldt) {
    kfu_build_t_body<KIND, NQ, MW>(a, ldt, blockIdx.x, blockIdx.y, blockIdx.z);
}
// K(x, Z) of a step, transposed: a.F receives KT[nb][Mp][ldt] (ldt >= a.Tp) -- the skinny product's coalesced A operand.  A step has
// few rows: tiles of 64 rows x 16 inducing points (4 per wavefront) put its 128 x 512 x 4 values on 256 workgroups instead of 64.
void launch_kfu_build_t(hipStream_t stream, const ProjectArgs &a, int ldt) {
    constexpr int MW = 4;
    dim3 grid(a.Tp / 64, a.Mp / (4 * MW), a.nb);
    if (a.P <= 8) {
        const int nq = (a.P + 1) / 2;
        if (a.kind == 0) {
            if (nq <= 2) hipLaunchKernelGGL((kfu_build_t_kernel<0, 2, MW>), grid, dim3(256), 0, stream, a, ldt);
            else if (nq == 3) hipLaunchKernelGGL((kfu_build_t_kernel<0, 3, MW>), grid, dim3(256), 0, stream, a, ldt);
            else hipLaunchKernelGGL((kfu_build_t_kernel<0, 4, MW>), grid, dim3(256), 0, stream, a, ldt);
        } else hipLaunchKernelGGL((kfu_build_t_kernel<1, 4, MW>), grid, dim3(256), 0, stream, a, ldt);
    } else {
        if (a.kind == 0) hipLaunchKernelGGL((kfu_build_t_kernel<0, 0, MW>), grid, dim3(256), 0, stream, a, ldt);
        else hipLaunchKernelGGL((kfu_build_t_kernel<1, 0, MW>), grid, dim3(256), 0, stream, a, ldt);
    }
}
void launch_kfu_build(hipStream_t stream, const ProjectArgs &a, int streaming) {
    static const bool no_nt = getenv("FFVD_KFU_NO_NT") != nullptr;          // A/B switch (read once)
    // streaming stores for outputs of 1 GB and more (nothing of them survives in a cache until the Gram kernel reads it);
    // `streaming` 0 / 1 decides for the caller (a pass of a pipelined iteration is part of a larger output)
    const bool nt = streaming >= 0 ? streaming != 0 : ((size_t)a.nb * a.Tp * a.Mp * sizeof(double) >= ((size_t)1 << 30) && !no_nt);
    if (nt) launch_kfu_build_nt<true>(stream, a);
    else launch_kfu_build_nt<false>(stream, a);
}

// Operator-API kernel matrix (one kernel, arbitrary N, N2; no padding).
__global__ __launch_bounds__(256) void kernel_matrix_kernel(int kind, const double *X, int N, const double *X2, int N2,
                                                            int P, double logvar, const double *loglen, double jitter,
                                                            int same, double *out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)N * N2) return;
    const int i = (int)(idx / N2), j = (int)(idx % N2);
    const double var = exp(logvar);
    const double *xi = X + (size_t)i * P, *zj = X2 + (size_t)j * P;
    double v;
    if (kind == 0) {
        double dot = 0.0, xx = 0.0, zz = 0.0;
        for (int p = 0; p < P; ++p) {
            double l = exp(loglen[p]);
            double a = xi[p] / l, b = zj[p] / l;
            dot += a * b; xx += a * a; zz += b * b;
        }
        v = kernel_value<0>(dot, xx, zz, var);
    } else {
        double dot = 0.0;
        for (int p = 0; p < P; ++p) dot += (xi[p] * var) * zj[p];
        v = dot;
    }
    if (same && i == j) v += jitter;
    out[idx] = v;
}
void launch_kernel_matrix(hipStream_t stream, int kind, const double *X, int N, const double *X2, int N2, int P,
                          double logvar, const double *loglen_dev, double jitter, int same, double *out) {
    size_t n = (size_t)N * N2;
    if (n == 0) return;
    hipLaunchKernelGGL(kernel_matrix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, kind, X, N, X2,
                       N2, P, logvar, loglen_dev, jitter, same, out);
}
__global__ void kernel_diag_kernel(int kind, const double *X, int N, int P, double logvar, double *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double var = exp(logvar);
    if (kind == 0) { out[i] = var; return; }                     // kernels_multi_output.py:199-200
    double s = 0.0;
    for (int p = 0; p < P; ++p) { double x = X[(size_t)i * P + p]; s += (x * x) * var; }   // kernels.py:278-281
    out[i] = s;
}
void launch_kernel_diag(hipStream_t stream, int kind, const double *X, int N, int P, double logvar, double *out) {
    if (N == 0) return;
    hipLaunchKernelGGL(kernel_diag_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, kind, X, N, P, logvar, out);
}

// ---------------------------------------------------------------------------------------------
// Extended blocked Cholesky (right-looking, NB = 64), two launches per block step:
//   panel(k):  rows below the diagonal block  R <- R * L_kk^{-T}   (forward substitution, lane = row,
//              L_kk broadcast from LDS; one wavefront per 64 rows)
//   trail(k):  C(i,j) -= P_i P_j^T for the remaining tiles (one workgroup per 64x64 tile, MFMA); the workgroup
//              of tile (k+1,k+1) goes on to factorise that diagonal block (look-ahead inside the launch)
// The 64x64 diagonal factorisation runs in ONE workgroup (lane = row, 16 row entries per lane in registers, pivot
// column broadcast through LDS), and no other workgroup reads a diagonal block before a finished launch has
// published its factor.
// ---------------------------------------------------------------------------------------------
#ifdef FFVD_DF_TRACE
__device__ long long df_trace_buf[64 * 64];
__device__ long long gram_trace_buf[128 * 10 * 4];      // start, end, HW_ID, blockIdx of every Gram tile (<= 128 units of 10 tiles)
#define DF_STAMP0(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) df_trace_buf[63 * 64 + (slot)] = wall_clock64(); } while (0)
#else
#define DF_STAMP0(slot) do { } while (0)
#endif

// 64x64 Cholesky by ONE wavefront with no workgroup barriers: lane = row, the whole row in registers
// (a[c] = A[lane][c]), left-looking: column j <- a[j] - sum_{i<j} L[:,i] L[j][i].  Row j of L reaches every lane
// as broadcast LDS reads (16 bytes = two entries per instruction) of the copy Lr that the wavefront extends by one
// column per pivot.  Software pipeline: while the sqrt chain of pivot j is in flight, the wavefront already sums
// the terms i <= j-2 of column j+1 (all of them final and visible); when pivot j is done only two terms are
// missing -- L[j+1][j-1] from LDS and the newest one, L[j+1][j], by v_readlane.  A single wavefront issues in
// order, so without this interleaving every pivot would pay its full dependent latency.
// Entries above the diagonal carry don't-care values.
// Out: Lr[r][c] = L[r][c] for c <= r (LDS); invd[j] = 1 / L[j][j] (LDS).  Returns 0 or 1 + first bad pivot.
// PIPE = false drops the software pipeline (plain left-looking sums, the newest term by v_readlane): about 100
// VGPRs fewer, for launches whose many workgroups care about occupancy more than about one tile's latency.
constexpr int LR_LD = NB + 2;
#ifndef DF_FACTOR_4W
#define DF_FACTOR_4W 1            // the 64-pivot diagonal factor by all four wavefronts (chol64_mfma_4w); 0: wavefront 0 alone (A/B builds)
#endif
#ifndef DF_FACTOR_ROWS
#define DF_FACTOR_ROWS 1          // chol64_mfma_4w: the tiles below a diagonal tile go through its pivot chain (0: tile solves on the matrix cores, A/B builds)
#endif
constexpr int DV_LD = 17;       // row stride of the 16 x 16 inverse / scratch tiles (doubles)
template <bool PIPE>
__device__ __forceinline__ int chol64_1w(double (&a)[NB], double (*Lr)[LR_LD], double *invd, const int lane) {
    int bad = 0;
    if (!PIPE) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double s0 = a[j], s1 = 0.0;
            if (j >= 2) wave_lds_sync();                     // column j-2 (and older) of Lr is visible
#pragma unroll
            for (int i = 0; i + 1 < j - 1 + (j & 1); i += 2) {          // pairs (i, i+1) with i + 1 <= j - 2
                const double2 lj = *reinterpret_cast<const double2 *>(&Lr[j][i]);
                s0 -= a[i] * lj.x;
                s1 -= a[i + 1] * lj.y;
            }
            if (j >= 2 && !(j & 1)) s0 -= a[j - 2] * Lr[j][j - 2];
            if (j >= 1) s1 -= a[j - 1] * readlane_f64(a[j - 1], j);
            const double v = s0 + s1;
            const double ajj = readlane_f64(v, j);
            if (!(ajj > 0.0) && bad == 0) bad = j + 1;
            double piv, y;
            pivot_sqrt(ajj, piv, y);
            a[j] = v * y;
            Lr[lane][j] = a[j];
            if (lane == 0) invd[j] = y;
        }
        return bad;
    }
    double nxt = 0.0;                                    // sum_{i <= j-3} L[:,i] L[j][i], formed during pivot j-1
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double v = a[j] - nxt;                           // the two newest terms come straight from the registers
        if (j >= 2) v -= a[j - 2] * readlane_f64(a[j - 2], j);
        if (j >= 1) v -= a[j - 1] * readlane_f64(a[j - 1], j);
        const double ajj = readlane_f64(v, j);
        if (!(ajj > 0.0) && bad == 0) bad = j + 1;
        double piv, y;
        pivot_sqrt(ajj, piv, y);
        if (j + 1 < NB) {                                // terms i = 0 .. j-2 of column j+1, independent of pivot j
            if (j >= 2) wave_lds_sync();                 // columns <= j-2 of Lr (stored >= 1 pivot ago) are visible
            double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
            for (int i = 0; i + 1 <= j - 2; i += 2) {
                const double2 l2 = *reinterpret_cast<const double2 *>(&Lr[j + 1][i]);
                if (i & 2) { q2 += a[i] * l2.x; q3 += a[i + 1] * l2.y; }
                else { q0 += a[i] * l2.x; q1 += a[i + 1] * l2.y; }
            }
            if (j >= 2 && !(j & 1)) q0 += a[j - 2] * Lr[j + 1][j - 2];        // j-1 terms: odd count when j is even
            nxt = (q0 + q1) + (q2 + q3);
        }
        a[j] = v * y;                                    // lane j: ajj * y = sqrt(ajj) to an ulp
        Lr[lane][j] = a[j];
        if (lane == 0) invd[j] = y;
    }
    return bad;
}

// 64x64 Cholesky by ONE wavefront, blocked 4 x 4 in 16 x 16 tiles, left-looking at tile level: everything but the
// pivots runs on the matrix cores.  For tile column s:
//   S_s = T_ss - sum_{k<s} L_sk L_sk^T                        (s tile products, accumulator layout -> LDS -> lane = row)
//   16-pivot chain on S_s in registers (right-looking, the multipliers by v_readlane); lanes 16..31 carry the rows of the
//     identity through the same updates and come out as L_ss^-T, which is at once the operand of the tile solves below and the
//     inverted diagonal sub-block the panel kernels multiply by (dinv_b, no separate substitution pass)
//   L_is^T = L_ss^-1 (T_is - sum_{k<s} L_ik L_sk^T)^T  for i > s, kept TRANSPOSED in the accumulator layout (= the B-operand
//     layout of the next product, as in potrf_panel_kernel), one residual refinement against L_ss.
// The single-wavefront chain above (chol64_1w) issues 2016 dependent-free but in-order FMA + LDS-broadcast pairs for its
// left-looking sums and is bound by that issue stream (13 us); here those sums are 64 + 48 MFMAs and the four chains have at
// most 15 terms per pivot.  No workgroup barrier inside: LDS hand-offs are within the wavefront.
// In: Ts (lower triangle valid).  Out: Lr = L (lower triangle of every tile row, exact zeros above the diagonal inside the
// diagonal tiles), dinv_b[(16 s + r) * 16 + c] = (L_ss^-1)[r][c] in global memory.  Sc: two 16 x DV_LD scratch tiles.
// Returns 0 or 1 + first bad pivot.
__device__ __forceinline__ int chol64_mfma_1w(double (*Ts)[NB + 1], double (*Lr)[LR_LD], double (*Sc)[16][DV_LD],
                                              double *dinv_b, const int lane) {
    const int lr = lane & 15, lk = lane >> 4;
    int bad = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int s0 = 16 * s;
        DF_STAMP0(4 * s + 0);
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < s; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double v = Lr[s0 + lr][16 * k + 4 * t + lk];      // A[m][kk] = B[kk][n]^T: the same tile
                acc = mfma_f64(v, v, acc);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) Sc[0][lk + 4 * r][lr] = Ts[s0 + lk + 4 * r][s0 + lr] - acc[r];
        wave_lds_order();
        DF_STAMP0(4 * s + 1);
        double a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = (lane < 16) ? Sc[0][lr][c] : ((lane < 32 && lr == c) ? 1.0 : 0.0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double ajj = readlane_f64(a[j], j);
            if (!(ajj > 0.0) && bad == 0) bad = s0 + j + 1;
            double piv, y;
            pivot_sqrt(ajj, piv, y);
            a[j] *= y;
#pragma unroll
            for (int c = j + 1; c < 16; ++c) a[c] = fma(-a[j], readlane_f64(a[j], c), a[c]);
        }
        DF_STAMP0(4 * s + 2);
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c) Lr[s0 + lr][s0 + c] = (c <= lr) ? a[c] : 0.0;
        } else if (lane < 32) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                Sc[1][lr][c] = a[c];                                    // X = L_ss^-T
                dinv_b[(s0 + c) * 16 + lr] = a[c];                      // (L_ss^-1)[c][lr] = X[lr][c]
            }
        }
        wave_lds_order();
        DF_STAMP0(4 * s + 3);
#pragma unroll
        for (int i = s + 1; i < 4; ++i) {
            const int i0 = 16 * i;
            d4 ac2 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < s; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    ac2 = mfma_f64(Lr[s0 + lr][16 * k + 4 * t + lk], Lr[i0 + lr][16 * k + 4 * t + lk], ac2);
            d4 Rt;
#pragma unroll
            for (int r = 0; r < 4; ++r) Rt[r] = Ts[i0 + lr][s0 + lk + 4 * r] - ac2[r];      // (T')^T[m][n] = T'[n][m]
            d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < 4; ++t) x = mfma_f64(Sc[1][lk + 4 * t][lr], Rt[t], x);       // A = L_ss^-1[m][kk] = X[kk][m]
            d4 res = Rt;
#pragma unroll
            for (int t = 0; t < 4; ++t) res = mfma_f64(-Lr[s0 + lr][s0 + 4 * t + lk], x[t], res);
#pragma unroll
            for (int t = 0; t < 4; ++t) x = mfma_f64(Sc[1][lk + 4 * t][lr], res[t], x);
#pragma unroll
            for (int r = 0; r < 4; ++r) Lr[i0 + lr][s0 + lk + 4 * r] = x[r];                // L_is[n][m]
        }
        wave_lds_order();
    }
    DF_STAMP0(16);
    return bad;
}

// The same factor by the FOUR wavefronts of the workgroup (round 4): wavefront 0 keeps the critical path -- tile (s+1, s), the sums
// of diagonal tile s + 1 and its 16-pivot chain -- and the tiles (s+2, s), (s+3, s) of column step s, which chol64_mfma_1w solved
// behind every chain, go to wavefronts 1 and 2 beside it (tiny_chol_inv's schedule on the 4 x 4 tiles of a 64-block): 15.5 -> about
// 11.5 us per diagonal block, i.e. per block column of every factorisation's latency chain.  Sc: three 16 x DV_LD scratch tiles
// (S; L_ss^-T of the even and of the odd steps: the helpers of step s read it while wavefront 0 writes the next one).
// All 256 threads call; returns wavefront 0's verdict on every thread of wavefront 0 (others: 0).
__device__ __forceinline__ void chol64_tile_solve(double (*Ts)[NB + 1], double (*Lr)[LR_LD], const double (*X)[DV_LD], const int s,
                                                  const int i, const int lr, const int lk) {
    const int s0 = 16 * s, i0 = 16 * i;
    d4 ac2 = (d4){0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < s; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            ac2 = mfma_f64(Lr[s0 + lr][16 * k + 4 * t + lk], Lr[i0 + lr][16 * k + 4 * t + lk], ac2);
    d4 Rt;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rt[r] = Ts[i0 + lr][s0 + lk + 4 * r] - ac2[r];      // (T')^T[m][n] = T'[n][m]
    d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) x = mfma_f64(X[lk + 4 * t][lr], Rt[t], x);           // A = L_ss^-1[m][kk] = X[kk][m]
    d4 res = Rt;
#pragma unroll
    for (int t = 0; t < 4; ++t) res = mfma_f64(-Lr[s0 + lr][s0 + 4 * t + lk], x[t], res);
#pragma unroll
    for (int t = 0; t < 4; ++t) x = mfma_f64(X[lk + 4 * t][lr], res[t], x);
#pragma unroll
    for (int r = 0; r < 4; ++r) Lr[i0 + lr][s0 + lk + 4 * r] = x[r];                // L_is[n][m]
}
// sums + 16-pivot chain of diagonal tile s (one wavefront): L_ss into Lr, L_ss^-T into Xo and (transposed) into dinv_b.
// Id: an identity tile in LDS (the rows lanes 16-31 start from).  The chain as in tiny.hip's tiny_chain16 (tools/probes/lat_probe.hip:
// 1.92 -> 1.54 us): per-lane base pointers instead of an exec-masked load per element, no scalar test per pivot -- a non-positive or
// NaN pivot leaves NaN on L's diagonal from there on, looked for once behind the chain.
__device__ __forceinline__ int chol64_diag_tile(double (*Ts)[NB + 1], double (*Lr)[LR_LD], double (*Sw)[DV_LD], double (*Xo)[DV_LD],
                                                const double (*Id)[DV_LD], double *dinv_b, const int s, const int lane) {
    const int lr = lane & 15, lk = lane >> 4, s0 = 16 * s;
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = acc;
    for (int k = 0; k < s; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double v = Lr[s0 + lr][16 * k + 4 * t + lk];
            if (t & 1) acc1 = mfma_f64(v, v, acc1);
            else acc = mfma_f64(v, v, acc);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) Sw[lk + 4 * r][lr] = Ts[s0 + lk + 4 * r][s0 + lr] - (acc[r] + acc1[r]);
    wave_lds_order();
    double a[16];
    const double *src = (lane < 16) ? &Sw[lr][0] : &Id[lr][0];
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
        double *base = (lane < 16) ? &Lr[s0 + lr][s0] : &Xo[lr][0];
#pragma unroll
        for (int c = 0; c < 16; ++c) base[c] = (lane >= 16 || c <= lr) ? a[c] : 0.0;
        if (lane >= 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c) dinv_b[(s0 + c) * 16 + lr] = a[c];              // (L_ss^-1)[c][lr] = X[lr][c]
        }
    }
    const unsigned long long m = __ballot((lane < 16) & !(diag > 0.0));
    return m ? s0 + (int)__builtin_ctzll(m) + 1 : 0;
}
#if DF_FACTOR_ROWS
// Round 4, second form: the tiles BELOW diagonal tile s ride through its 16-pivot chain.  The chain is right-looking -- pivot j scales
// column j of every row and subtracts its multiple of row j's multipliers from the columns right of it -- so a lane that starts from a
// row of T'_is = T_is - sum_{k<s} L_ik L_sk^T comes out holding that row of L_is, exactly as the unblocked factorisation would produce
// it.  Lanes 0-15: the diagonal tile, 16-31: the identity (-> L_ss^-T for the panel solves), 32-47 / 48-63: tiles (s+1, s), (s+2, s);
// step 0 has a third tile below, which a second wavefront takes through the same chain (it repeats the pivots for itself).  What was
// a tile solve on the matrix cores behind every chain (three dependent groups of four MFMAs with an LDS round trip, then the sums of
// the next diagonal tile) is gone from the latency chain: the wavefronts form the left-looking sums of the step's tiles side by
// side, one barrier, chain.  tools/df_trace.py: the factor of a 64-block 14.5 -> about 10.5 us.
__device__ __forceinline__ int chol64_mfma_4w(double (*Ts)[NB + 1], double (*Lr)[LR_LD], double (*Sc)[16][DV_LD], double *dinv_b) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < 256) Sc[3][threadIdx.x >> 4][threadIdx.x & 15] = ((threadIdx.x >> 4) == (threadIdx.x & 15)) ? 1.0 : 0.0;
    __syncthreads();
    int bad = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int s0 = 16 * s;
        if (s > 0) {
            const int i = s + wave;
            if (i < 4) {                                     // T'_is in row-per-lane reach: Sc[wave]
                const int i0 = 16 * i;
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = acc;
                for (int k = 0; k < s; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double av = Lr[i0 + lr][16 * k + 4 * t + lk], bv = Lr[s0 + lr][16 * k + 4 * t + lk];
                        if (t & 1) acc1 = mfma_f64(av, bv, acc1);
                        else acc = mfma_f64(av, bv, acc);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) Sc[wave][lk + 4 * r][lr] = Ts[i0 + lk + 4 * r][s0 + lr] - (acc[r] + acc1[r]);
            }
            __syncthreads();
        }
        if (wave == 0 || (s == 0 && wave == 1)) {
            __builtin_amdgcn_s_setprio(3);
            // which tile this lane's row belongs to: 0 the diagonal one, -1 the identity, t > 0: tile (s + t, s); -2: nothing (a copy of the
            // diagonal rows, not stored)
            int tl;
            if (wave == 0) tl = (lk == 0) ? 0 : (lk == 1) ? -1 : ((s + lk - 1 < 4) ? lk - 1 : -2);
            else tl = (lk == 1) ? 3 : -2;
            const double *src;
            if (tl == -1) src = &Sc[3][lr][0];
            else if (s == 0) src = &Ts[16 * (tl > 0 ? tl : 0) + lr][0];
            else src = &Sc[tl > 0 ? tl : 0][lr][0];
            double a[16];
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
            if (tl >= 0) {
                double *base = &Lr[s0 + 16 * tl + lr][s0];
#pragma unroll
                for (int c = 0; c < 16; ++c) base[c] = (tl > 0 || c <= lr) ? a[c] : 0.0;
            } else if (tl == -1) {
#pragma unroll
                for (int c = 0; c < 16; ++c) dinv_b[(s0 + c) * 16 + lr] = a[c];              // (L_ss^-1)[c][lr] = (L_ss^-T)[lr][c]
            }
            if (wave == 0) {
                const unsigned long long m = __ballot((lane < 16) & !(diag > 0.0));
                if (m && !bad) bad = s0 + (int)__builtin_ctzll(m) + 1;
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }
    return bad;
}
#else
__device__ __forceinline__ int chol64_mfma_4w(double (*Ts)[NB + 1], double (*Lr)[LR_LD], double (*Sc)[16][DV_LD], double *dinv_b) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Sc[3]: the identity tile of the pivot chains (the four scratch tiles are the panel code's inverse blocks at other times)
    if (threadIdx.x < 256) Sc[3][threadIdx.x >> 4][threadIdx.x & 15] = ((threadIdx.x >> 4) == (threadIdx.x & 15)) ? 1.0 : 0.0;
    __syncthreads();
    int bad = 0;
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        bad = chol64_diag_tile(Ts, Lr, Sc[0], Sc[1], Sc[3], dinv_b, 0, lane);
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double (*X)[DV_LD] = Sc[1 + (s & 1)];
        if (wave == 0) {
            __builtin_amdgcn_s_setprio(3);
            chol64_tile_solve(Ts, Lr, X, s, s + 1, lr, lk);
            wave_lds_order();
            const int b2 = chol64_diag_tile(Ts, Lr, Sc[0], Sc[1 + ((s + 1) & 1)], Sc[3], dinv_b, s + 1, lane);
            if (b2 && !bad) bad = b2;
            __builtin_amdgcn_s_setprio(0);
        } else if (s + 1 + wave < 4) chol64_tile_solve(Ts, Lr, X, s, s + 1 + wave, lr, lk);
        __syncthreads();
    }
    return bad;
}
#endif

// Factorise the diagonal block held in LDS tile `Ts` (row-major, stride NB+1): wavefront 0 runs chol64_1w, which
// leaves L in the LDS tile `Lr`; then all 256 threads publish L in place (lower triangle of the global block) and
// the four wavefronts invert the four 16x16 diagonal sub-blocks of L (lane = column, 16-step forward substitution)
// into dinv_b[4][16][16] -- what the panel kernel's blocked substitution multiplies by.
// (Sc != nullptr: the blocked matrix-core factor chol64_mfma_1w, which needs Lr and Ts in DIFFERENT memory and two scratch
// tiles, and leaves the inverted sub-blocks behind itself.)
template <bool PIPE>
__device__ __forceinline__ void diag_block_finish(double (*Ts)[NB + 1], double (*Lr)[LR_LD], double *invd, double *S, int n,
                                                   int k0, int32_t *info_b, double *dinv_b, double (*Sc)[16][DV_LD] = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (Sc) {
#if DF_FACTOR_4W
        {
            const int bad = chol64_mfma_4w(Ts, Lr, Sc, dinv_b);      // (ends with the workgroup synchronised)
            if (w == 0 && bad && lane == 0 && *info_b == 0) *info_b = k0 + bad;
        }
#else
        if (w == 0) {
            __builtin_amdgcn_s_setprio(3);
            const int bad = chol64_mfma_1w(Ts, Lr, Sc, dinv_b, lane);
            __builtin_amdgcn_s_setprio(0);
            if (bad && lane == 0 && *info_b == 0) *info_b = k0 + bad;
        }
        __syncthreads();
#endif
        for (int r = tid >> 6; r < NB; r += 4)
            if (lane <= r) S[(size_t)(k0 + r) * n + k0 + lane] = Lr[r][lane];         // coalesced rows of L
        return;
    }
    if (w == 0) {
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = Ts[lane][c];
        // the pivot chain is one wavefront's dependent latency: it goes first whenever a wavefront of another workgroup
        // shares its SIMD
        __builtin_amdgcn_s_setprio(3);
        const int bad = chol64_1w<PIPE>(a, Lr, invd, lane);
        __builtin_amdgcn_s_setprio(0);
        if (bad && lane == 0 && *info_b == 0) *info_b = k0 + bad;
    }
    __syncthreads();
    for (int r = tid >> 6; r < NB; r += 4)
        if (lane <= r) S[(size_t)(k0 + r) * n + k0 + lane] = Lr[r][lane];         // coalesced rows of L
    if (lane < 16) {
        const int o = 16 * w;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double acc = (lane == r) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < r; ++i) acc -= Lr[o + r][o + i] * x[i];
            x[r] = acc * invd[o + r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dinv_b[(w * 16 + r) * 16 + lane] = x[r];
    }
}

__global__ __launch_bounds__(256) void potrf_diag_kernel(double *A, int n, int k, size_t slab_stride, int32_t *info,
                                                         double *dinv) {
    __shared__ double Ts[NB][NB + 1];
    __shared__ double Lr[NB][LR_LD];
    __shared__ double invd[NB];
    const int b = blockIdx.x, tid = threadIdx.x;
    double *S = A + (size_t)b * slab_stride;
    const int k0 = k * NB;
    for (int r = tid >> 6; r < NB; r += 4) Ts[r][tid & 63] = S[(size_t)(k0 + r) * n + k0 + (tid & 63)];
    __syncthreads();
    diag_block_finish<true>(Ts, Lr, invd, S, n, k0, info + b, dinv + (size_t)b * DINV_STRIDE);
}

// tile bookkeeping shared by the panel and trailing kernels
// Extra-row blocks: the first `nid` blocks are identity-structured (block e is still zero in block-columns < e,
// so only e <= k is live at step k: `nlive` of them), the blocks after them are always live.
__device__ __forceinline__ int extra_block(int e, int nlive, int nid) { return (e < nlive) ? e : nid + (e - nlive); }
__device__ __forceinline__ int chunk_row0(int chunk, int k, int nmain, int n, int nlive, int nid) {
    return (chunk < nmain) ? (k + 1 + chunk) * NB : n + extra_block(chunk - nmain, nlive, nid) * NB;
}

constexpr int TR_LD = NB + 2;      // LDS row stride of the staged panel blocks (doubles)

// R <- R * L_kk^{-T} for one 64-row chunk, as a blocked substitution on the matrix cores.  With L_kk cut into 16x16
// blocks and X = R L_kk^{-T}:   X_s^T = L_ss^{-1} (R_s^T - sum_{m<s} L_sm X_m^T).
// Wavefront w owns rows 16w..16w+15 of the chunk and keeps the four TRANSPOSED 16x16 blocks R_s^T in the MFMA
// accumulator layout, which is exactly the B-operand layout of the next product, so the whole chain
// (4 multiplications by the inverted diagonal blocks from diag_block_finish + 6 block updates) runs register to
// register with the A operands (-L_ts, L_ss^{-1}) read from LDS: 40 MFMAs per wavefront instead of a 64-step scalar
// forward substitution.
__global__ __launch_bounds__(256) void potrf_panel_kernel(double *A, int n, int k, int nmain, size_t slab_stride,
                                                          int nlive, int nid, const double *dinv) {
    __shared__ double Ls[NB][TR_LD];       // -L_kk (only blocks below the block diagonal are read)
    __shared__ double Dv[4][16][DV_LD];    // inverses of the four 16x16 diagonal blocks of L_kk
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *S = A + (size_t)b * slab_stride;
    const int k0 = k * NB;
    const int row0 = chunk_row0(blockIdx.x, k, nmain, n, nlive, nid);
    // this lane's 16 entries: R[row0 + 16w + lr][k0 + 16s + 4r + lk]  (32-byte runs per row)
    double *Rl = S + (size_t)(row0 + 16 * w + lr) * n + k0 + lk;
    d4 Rt[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rt[s4][r] = Rl[16 * s4 + 4 * r];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (tid >> 5) + 8 * i, c2 = 2 * (tid & 31);
        const double2 v = *reinterpret_cast<const double2 *>(S + (size_t)(k0 + r) * n + k0 + c2);
        Ls[r][c2] = (c2 <= r) ? -v.x : 0.0;              // the block's upper triangle holds don't-care values
        Ls[r][c2 + 1] = (c2 + 1 <= r) ? -v.y : 0.0;
    }
    {
        const double *dv = dinv + (size_t)b * DINV_STRIDE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            Dv[e >> 8][(e >> 4) & 15][e & 15] = dv[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = mfma_f64(Dv[s4][lr][4 * ks + lk], Rt[s4][ks], x);
        // one step of iterative refinement: multiplying by an explicit inverse is not backward stable, a
        // substitution is; the residual B - L_ss X brings the product back to substitution accuracy
        d4 res = Rt[s4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) res = mfma_f64(Ls[16 * s4 + lr][16 * s4 + 4 * ks + lk], x[ks], res);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = mfma_f64(Dv[s4][lr][4 * ks + lk], res[ks], x);
        Rt[s4] = x;
#pragma unroll
        for (int t = s4 + 1; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Rt[t] = mfma_f64(Ls[16 * t + lr][16 * s4 + 4 * ks + lk], x[ks], Rt[t]);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rl[16 * s4 + 4 * r] = Rt[s4][r];
}

// Trailing update of step k: one workgroup (4 wavefronts, a 32x32 quadrant each) per 64x64 tile,
// C(i,j) -= P_i * P_j^T with P = solved panel, both panel blocks staged through LDS with wide coalesced loads.
// Tiles: main lower triangle (k < j <= i < nb) then extra-row tiles (e, j) for j in (k, nb).
// Tile 0 is the next diagonal block (k+1,k+1): wavefront 0 of its workgroup factorises it right away, so the
// 64-pivot chain of step k+1 overlaps with the rest of step k's trailing update (look-ahead inside one launch).
template <bool PIPE>
__global__ __launch_bounds__(256) void potrf_trail_kernel(double *A, int n, int k, int nmain_tiles, int n1,
                                                          size_t slab_stride, int32_t *info, int nlive, int nid,
                                                          double *dinv) {
    // One 34.8 KB LDS buffer: the panel blocks are staged in two 32-column halves (Pi half | Pj half), so that three
    // workgroups fit on a CU; the look-ahead tile later reuses the same memory as its 64 x 66 factor tile.
    constexpr int HLD = 34;                               // row stride of a staged half (doubles)
    __shared__ double sm[2 * NB * HLD];
    __shared__ double invd[NB];
    static_assert(2 * NB * HLD >= NB * LR_LD, "the staging buffer must hold the factor tile");
    double *Pi_h = sm, *Pj_h = sm + NB * HLD;
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    double *S = A + (size_t)b * slab_stride;
    const int k0 = k * NB;
    int rowblk0, colblk;   // first row / first col of the tile
    if (tile < nmain_tiles) {
        int i = 0;          // unrank lower triangle: tile = i(i+1)/2 + j
        while ((i + 1) * (i + 2) / 2 <= tile) ++i;
        int j = tile - i * (i + 1) / 2;
        rowblk0 = (k + 1 + i) * NB;
        colblk = (k + 1 + j) * NB;
    } else {
        int t = tile - nmain_tiles;
        int e = t / n1, j = t % n1;
        rowblk0 = n + extra_block(e, nlive, nid) * NB;
        colblk = (k + 1 + j) * NB;
    }
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qr = wave >> 1, qc = wave & 1;          // 32x32 quadrant of the tile
    const bool same = (rowblk0 == colblk);
    const double *Pi = S + (size_t)rowblk0 * n + k0;
    const double *Pj = S + (size_t)colblk * n + k0;
    // global -> registers for both halves at once: thread t moves 16 bytes of rows (t >> 4) + 16 i, columns
    // 32 h + 2 (t & 15) of each operand
    const int sr = tid >> 4, sc = 2 * (tid & 15);
    double2 vi[2][4], vj[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vi[h][i] = *reinterpret_cast<const double2 *>(Pi + (size_t)(sr + 16 * i) * n + 32 * h + sc);
            if (!same) vj[h][i] = *reinterpret_cast<const double2 *>(Pj + (size_t)(sr + 16 * i) * n + 32 * h + sc);
        }
    const double *Bh = same ? Pi_h : Pj_h;
    // the tile itself is requested now, so that its latency hides behind the staging and the MFMAs
    double *Ct = S + (size_t)rowblk0 * n + colblk;
    d4 cold[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cold[x][y][q] = Ct[(size_t)(qr * 32 + 16 * x + lk + 4 * q) * n + qc * 32 + 16 * y + lr];
    d4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();                        // everyone is done reading the first half
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Pi_h[(sr + 16 * i) * HLD + sc] = vi[h][i].x; Pi_h[(sr + 16 * i) * HLD + sc + 1] = vi[h][i].y;
            if (!same) { Pj_h[(sr + 16 * i) * HLD + sc] = vj[h][i].x; Pj_h[(sr + 16 * i) * HLD + sc + 1] = vj[h][i].y; }
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < NB / 8; ++ks) {
            double af[2], bf[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                af[x] = Pi_h[(qr * 32 + 16 * x + lr) * HLD + 4 * ks + lk];     // A[row][k]
                bf[x] = Bh[(qc * 32 + 16 * x + lr) * HLD + 4 * ks + lk];       // B[k][col] = P_j[col][k]
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = mfma_f64(af[x], bf[y], acc[x][y]);
        }
    }
    if (tile != 0) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t off = (size_t)(qr * 32 + 16 * x + lk + 4 * q) * n + qc * 32 + 16 * y + lr;
                    Ct[off] = cold[x][y][q] - acc[x][y][q];
                }
        return;
    }
    // tile 0 = next diagonal block: assemble the updated block in LDS, factorise, publish.  The factor tile Lr
    // (stride 66) reuses the memory of the input tile Ts (stride 65): wavefront 0 has the whole input in registers
    // before it writes the first column of L.
    __syncthreads();
    double(*Ts)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(sm);
    double(*Lr)[LR_LD] = reinterpret_cast<double(*)[LR_LD]>(sm);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = qr * 32 + 16 * x + lk + 4 * q, cc = qc * 32 + 16 * y + lr;
                Ts[r][cc] = cold[x][y][q] - acc[x][y][q];
            }
    __syncthreads();
    diag_block_finish<PIPE>(Ts, Lr, invd, S, n, k0 + NB, info + b, dinv + (size_t)b * DINV_STRIDE);
}

// ---------------------------------------------------------------------------------------------
// Left-looking block column (the default since round 2; north_star: "blocked left-looking Cholesky with wavefront-level
// trsm").  After potrf_diag_kernel has factorised block (0,0): ONE launch per block column j, one workgroup per
// 64-row block r below the diagonal (main rows j+1..nb-1 and the live extra-row blocks), no dependency between the
// workgroups of a launch:
//     X(r,j) = (A(r,j) - sum_{k<j} X(r,k) L(j,k)^T) L_jj^-T        gathered from the finished block columns, then a
//                                                                  blocked substitution on the matrix cores
// and the workgroup of row j+1 goes on (look-ahead): it also sums X(j+1,k) X(j+1,k)^T over k <= j, factorises
// S = A(j+1,j+1) - sum in its first wavefront and publishes L_{j+1,j+1} for the next launch.
// Each block of the matrix is written ONCE (the right-looking variant re-reads and re-writes the whole trailing matrix at
// every step: 1.4 GB of tile traffic per factorisation of the 128 A-matrices of config 2 against 0.64 GB here) and a
// step is one launch instead of two.  Both operand tiles of step k sit in LDS whole (stride 66: conflict-free fragment
// reads); the global loads of step k+1 are issued before the MFMAs of step k.
// A first draft let EVERY workgroup factorise S_jj itself (no look-ahead, no diagonal launch): each workgroup then holds
// its slot for the 13 us pivot chain and the 5632 workgroup executions of one 128-matrix factorisation took 1.14 ms
// against the right-looking 0.63 ms.
// ---------------------------------------------------------------------------------------------
constexpr int LL_LD = NB + 2;
template <bool PIPE>
__global__ __launch_bounds__(256) void potrf_ll_kernel(double *A, int n, int j, int nmain, int nchunks, size_t slab_stride,
                                                       int32_t *info, int nlive, int nid, int batch, double *dinv) {
    __shared__ double sm0[NB * LL_LD];      // X(r,k) tiles; then T = A(r,j) - sum and the solved X(r,j)
    __shared__ double sm1[NB * LL_LD];      // L(j,k) tiles; then -L_jj; then (look-ahead) S_{j+1,j+1} and its factor
    __shared__ double invd[NB];
    __shared__ double Dv[4][16][DV_LD];
    // chunk-major order: the look-ahead workgroups (chunk 0) of ALL matrices are dispatched first -- their pivot chain is
    // the tail of the launch.  XCD-aware: the batch is padded to a multiple of 8, so all workgroups of one matrix share
    // blockIdx % 8, i.e. one XCD's L2 (they all read row block j).
    const int bpad = (int)(gridDim.x / nchunks);
    const int chunk = blockIdx.x / bpad, b = blockIdx.x % bpad;
    if (b >= batch) return;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qr = wave >> 1, qc = wave & 1;
    double *S = A + (size_t)b * slab_stride;
    const int j0 = j * NB;
    const bool look = (chunk == 0 && nmain > 0);        // row block j + 1: also factorises the next diagonal block
    int row0, k0 = 0;                       // first row of this workgroup's block; first block column with non-zero X(r,k)
    if (chunk < nmain) row0 = (j + 1 + chunk) * NB;
    else {
        const int e = extra_block(chunk - nmain, nlive, nid);
        row0 = n + e * NB;
        if (e < nid) k0 = e;                // identity-structured rows: block e is zero left of block column e
    }
    double (*Xs)[LL_LD] = reinterpret_cast<double(*)[LL_LD]>(sm0);
    double (*Ls)[LL_LD] = reinterpret_cast<double(*)[LL_LD]>(sm1);
    // the output tile(s) are requested first: their latency hides behind the whole k loop
    d4 cold_t[2][2], cold_d[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t rr = (size_t)(qr * 32 + 16 * x + lk + 4 * q), cc = (size_t)(qc * 32 + 16 * y + lr);
                cold_t[x][y][q] = S[(row0 + rr) * n + j0 + cc];
                cold_d[x][y][q] = look ? S[(row0 + rr) * n + row0 + cc] : 0.0;
            }
    d4 acc_t[2][2], acc_d[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) { acc_t[x][y] = (d4){0.0, 0.0, 0.0, 0.0}; acc_d[x][y] = (d4){0.0, 0.0, 0.0, 0.0}; }
    // staging: thread moves 16 bytes of rows (tid >> 5) + 8 i, columns 2 (tid & 31) of each operand
    const int sr = tid >> 5, sc = 2 * (tid & 31);
    d2 vx[8], vl[8];
    auto gload = [&](int k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            vl[i] = *reinterpret_cast<const d2 *>(S + (size_t)(j0 + sr + 8 * i) * n + k * NB + sc);
            vx[i] = *reinterpret_cast<const d2 *>(S + (size_t)(row0 + sr + 8 * i) * n + k * NB + sc);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Ls[sr + 8 * i][sc] = vl[i].x; Ls[sr + 8 * i][sc + 1] = vl[i].y;
            Xs[sr + 8 * i][sc] = vx[i].x; Xs[sr + 8 * i][sc + 1] = vx[i].y;
        }
    };
    const bool do_d = look && !(qr == 0 && qc == 1);    // the quadrant above the diagonal of S is never read
    if (k0 < j) gload(k0);
    for (int k = k0; k < j; ++k) {
        if (k > k0) __syncthreads();                // everyone is done reading the previous tiles
        lstore();
        __syncthreads();
        if (k + 1 < j) gload(k + 1);                // in flight behind this step's MFMAs
#pragma unroll 4
        for (int ks = 0; ks < NB / 4; ++ks) {
            double bl[2], ax[2], bx[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                bl[x] = Ls[qc * 32 + 16 * x + lr][4 * ks + lk];        // B[k][col] = L(j,k)[col][k]
                ax[x] = Xs[qr * 32 + 16 * x + lr][4 * ks + lk];
                bx[x] = do_d ? Xs[qc * 32 + 16 * x + lr][4 * ks + lk] : 0.0;
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc_t[x][y] = mfma_f64(ax[x], bl[y], acc_t[x][y]);
            if (do_d) {
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc_d[x][y] = mfma_f64(ax[x], bx[y], acc_d[x][y]);
            }
        }
    }
    if (k0 < j) __syncthreads();
    // T into sm0; -L_jj (exactly zero above the diagonal: the refinement step reads the diagonal blocks) into sm1; the
    // inverted 16 x 16 diagonal sub-blocks of L_jj, left in the scratch block by the workgroup that factorised it
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Xs[qr * 32 + 16 * x + lk + 4 * q][qc * 32 + 16 * y + lr] = cold_t[x][y][q] - acc_t[x][y][q];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = sr + 8 * i;
        const d2 v = *reinterpret_cast<const d2 *>(S + (size_t)(j0 + r) * n + j0 + sc);
        Ls[r][sc] = (sc <= r) ? -v.x : 0.0;
        Ls[r][sc + 1] = (sc + 1 <= r) ? -v.y : 0.0;
    }
    {
        // slot j & 1: the look-ahead workgroup of THIS launch writes the other slot while late workgroups still read this one
        const double *dv = dinv + (size_t)b * DINV_STRIDE + (j & 1) * (DINV_STRIDE / 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            Dv[e >> 8][(e >> 4) & 15][e & 15] = dv[e];
        }
    }
    __syncthreads();
    // X(r,j) = T L_jj^-T: wavefront w owns rows 16w..16w+15, the four transposed 16 x 16 blocks in the MFMA accumulator
    // layout (potrf_panel_kernel has the derivation): 4 multiplications by the inverted diagonal blocks, one residual
    // refinement each, 6 block updates
    d4 Rt[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rt[s4][r] = Xs[16 * wave + lr][16 * s4 + 4 * r + lk];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = mfma_f64(Dv[s4][lr][4 * ks + lk], Rt[s4][ks], x);
        d4 res = Rt[s4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) res = mfma_f64(Ls[16 * s4 + lr][16 * s4 + 4 * ks + lk], x[ks], res);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = mfma_f64(Dv[s4][lr][4 * ks + lk], res[ks], x);
        Rt[s4] = x;
#pragma unroll
        for (int t = s4 + 1; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Rt[t] = mfma_f64(Ls[16 * t + lr][16 * s4 + 4 * ks + lk], x[ks], Rt[t]);
    }
    double *Rl = S + (size_t)(row0 + 16 * wave + lr) * n + j0 + lk;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rl[16 * s4 + 4 * r] = Rt[s4][r];
    if (!look) return;
    // look-ahead: S = A(j+1,j+1) - sum_{k<=j} X(j+1,k) X(j+1,k)^T; the newest term comes from the tile just solved
    // (each wavefront rewrites its own 16 rows of sm0, which it alone has read)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[16 * wave + lr][16 * s4 + 4 * r + lk] = Rt[s4][r];
    __syncthreads();
    if (do_d) {
#pragma unroll 4
        for (int ks = 0; ks < NB / 4; ++ks) {
            double ax[2], bx[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                ax[x] = Xs[qr * 32 + 16 * x + lr][4 * ks + lk];
                bx[x] = Xs[qc * 32 + 16 * x + lr][4 * ks + lk];
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc_d[x][y] = mfma_f64(ax[x], bx[y], acc_d[x][y]);
        }
    }
    double (*Ts)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(sm1);     // -L_jj is no longer needed (all past the solve:
    double (*Lr)[LR_LD] = reinterpret_cast<double(*)[LR_LD]>(sm1);        //  the barrier above)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Ts[qr * 32 + 16 * x + lk + 4 * q][qc * 32 + 16 * y + lr] = cold_d[x][y][q] - acc_d[x][y][q];
    __syncthreads();
    diag_block_finish<PIPE>(Ts, Lr, invd, S, n, j0 + NB, info + b, dinv + (size_t)b * DINV_STRIDE + ((j + 1) & 1) * (DINV_STRIDE / 2));
}

// Inverses of the four 16 x 16 diagonal sub-blocks of the GIVEN factor block L_jj into scratch slot j & 1 (what
// diag_block_finish leaves behind when it has just factorised the block): lets the block-column kernel run as a plain
// triangular solve against a factor that already exists (launch_trsm_ext).
__global__ __launch_bounds__(256) void diag_inv_kernel(const double *A, int n, int j, size_t slab_stride, double *dinv) {
    __shared__ double Lr[NB][NB + 1];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *S = A + (size_t)b * slab_stride;
    const int j0 = j * NB;
    for (int r = tid >> 6; r < NB; r += 4) Lr[r][lane] = S[(size_t)(j0 + r) * n + j0 + lane];
    __syncthreads();
    if (lane < 16) {
        const int o = 16 * w;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double acc = (lane == r) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < r; ++i) acc -= Lr[o + r][o + i] * x[i];
            x[r] = acc / Lr[o + r][o + r];
        }
        double *dv = dinv + (size_t)b * DINV_STRIDE + (j & 1) * (DINV_STRIDE / 2);
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[(w * 16 + r) * 16 + lane] = x[r];
    }
}

// R <- R L^-T for `extra_rows` rows stored below an n x n lower-triangular factor L in every slab (ld n): the
// block-column kernel without its factorisation part, one launch pair per block column.
void launch_trsm_ext(hipStream_t stream, double *A, int n, int extra_rows, int batch, size_t slab_stride, double *dinv) {
    const int nb = n / NB, ntail = extra_rows / NB;
    if (ntail == 0) return;
    const int groups = (batch + 7) / 8;
    for (int j = 0; j < nb; ++j) {
        hipLaunchKernelGGL(diag_inv_kernel, dim3(batch), dim3(256), 0, stream, A, n, j, slab_stride, dinv);
        hipLaunchKernelGGL(potrf_ll_kernel<false>, dim3(groups * 8 * ntail), dim3(256), 0, stream, A, n, j, 0, ntail,
                           slab_stride, (int32_t *)nullptr, 0, 0, batch, dinv);
    }
}

// ---------------------------------------------------------------------------------------------
// Dataflow variant: the whole left-looking factorisation of a batch in ONE launch.  One workgroup per 64-row block row of
// every matrix walks its row from left to right,
//     for j < r:  X(r,j) = (A(r,j) - sum_{k<j} X(r,k) L(j,k)^T) L_jj^-T        then (main rows)  L_rr = chol(A(r,r) - sum_k X(r,k) X(r,k)^T),
// and learns from one progress word per (matrix, block row) when the row above it is ready:  prog[j] = j once the panels
// X(j,0..j-1) are in memory, j + 1 once L_jj and its inverted 16 x 16 diagonal sub-blocks are.  What the per-column launches
// of potrf_ll_kernel serialised -- the gather of column j (j block products) in front of every diagonal factor -- now runs
// beside the pivot chain of the block above: the chain of dependent work per block column is one substitution, one block
// product and the 64-pivot factor.
// Hand-off (MI355X guide, inter-workgroup visibility): producer = plain stores, every storing wavefront waits for them, a
// workgroup barrier, ONE lane's agent-scope release, its wait, a relaxed agent-scope store of the word; consumer = ONE lane
// polls the word relaxed, ONE agent-scope acquire and its wait, a workgroup barrier, then plain loads.  Results never depend on
// where or when a workgroup runs.  Progress does: a row waits for rows of the same matrix with a lower blockIdx only, which
// the dispatcher starts first; every spin is bounded all the same (wall clock; a stuck launch sets the abort word, every
// waiting workgroup leaves, info[b] = -1 says so).
// Block order (group, row, matrix-in-group): G a multiple of 8, so all rows of a matrix share blockIdx % 8 = one XCD's L2.
// ---------------------------------------------------------------------------------------------
#ifndef DF_MFMA_FACTOR
#define DF_MFMA_FACTOR 1
#endif
constexpr int DF_PS = 64;                         // progress words per matrix (main block rows: n <= 4096)
static_assert(DF_PS == 64, "launch_kuu_build zeroes 64 progress words per matrix");
constexpr int DF_DINV = 4 * 16 * 16;              // doubles of inverted diagonal sub-blocks per (matrix, block column)
constexpr long long DF_SPIN_TICKS = 100000000LL;  // bound of one wait: 1 s of the 100 MHz wall clock

struct DfArgs {
    double *A;
    int n, nb, next, nid, batch, G;
    size_t slab_stride;
    int32_t *info;
    double *dinv;      // [batch][nb][DF_DINV]
    int *prog;         // [batch][DF_PS], zeroed before the launch
    int *abort_w;      // one word, zeroed before the launch
    double *xt;        // optional: the identity-structured extra rows (R L^-T = L^-T) also land TRANSPOSED here, i.e. L^-1 as
    size_t xt_stride;  //   an n x n lower-triangular matrix (ld n) per slab; blocks above its diagonal are not written
    int vec_tail;      // the extra-row blocks behind the identity rows carry ONE live row each (their first): df_vector_row
    double *kinv;      // optional (identity rows = all of L^-T): (A)^-1 = L^-T L^-1, full symmetric n x n per slab (ld n), formed by
    size_t kinv_stride;//   the identity-row workgroups once their rows are complete (df_inverse_tiles)
    int defer_ext;     // block order: identity-structured rows of all groups behind the main rows of all groups (potrf_df_kernel)
    int kinv_help;     // (with kinv, every workgroup resident at once) the main-row workgroups form half of the inverse's tiles: df_inverse_tiles
    int kacc;          // (with kinv) the inverse's tiles are accumulated column by column as the rows of L^-T arrive: df_inverse_column
    int fine;          // small batches (every block row on a CU of its own): a main row also announces every COLUMN it has solved
                       // (word 2 nb + row), and a gather waits term by term -- see df_column
    const double *lt;  // optional: the identity-structured extra rows of slab b start as L_d^T (d = b % lt_dl) instead of what
    size_t lt_stride;  //   memory holds: block (e, j) = L_d(j, e)^T, read from the lower-triangular n x n factor L_d (ld n)
    int lt_dl;
};

// Debug build only (-DFFVD_DF_TRACE, tools/df_trace.py): wall-clock stamps of matrix 0's block rows, in a buffer of their own.
#ifdef FFVD_DF_TRACE
#define DF_STAMP(row, slot) do { if ((row) >= 0 && threadIdx.x == 0) df_trace_buf[(row) * 64 + (slot)] = wall_clock64(); } while (0)
extern "C" int ffvd_debug_df_trace(long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(df_trace_buf), sizeof(long long) * 64 * 64);
}
extern "C" int ffvd_debug_gram_trace(long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gram_trace_buf), sizeof(long long) * 128 * 10 * 4);
}
#else
#define DF_STAMP(row, slot) do { } while (0)
#endif

// Thread 0 waits until *flag >= need, acquires, and hands the value it saw (or -1: gave up) to the workgroup.
__device__ __forceinline__ int df_wait(int *flag, int need, int *abort_w, int *slot) {
    if (threadIdx.x == 0) {
        int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < need) {
            const long long t0 = wall_clock64();
            for (;;) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= need) break;
                if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { v = -1; break; }
                if (wall_clock64() - t0 > DF_SPIN_TICKS) {
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
    return *slot;
}

// ONE lane, behind the workgroup barrier that follows every storing wavefront's own wait for its stores.
__device__ __forceinline__ void df_publish(int *flag, int value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the compiler may drop the fence's own wait (guide: compiler hazard)
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a later store to the same word (another wavefront) must not overtake
}

// One block column of one block row: X(r,j) = (A(r,j) - sum_{k0<=k<j} X(r,k) L(j,k)^T) L_jj^-T.  LAST (the column left of
// the diagonal of a main row) also loads A(r,r), sums X(r,k) X(r,k)^T over k < j into acc_d beside the gather, and leaves the
// solved tile in sm0 for the newest term.  Returns false when a wait gave up.
// EARLY (main rows in the fine mode, columns left of the LAST one): the tile just solved adds its X X^T to acc_d right away, in the
// time this row would otherwise wait for the next diagonal block, and the LAST column's gather carries L(j,k) terms only -- a late
// row's last gather is 3.6 us of matrix-pipe time per term with both sums in it, 30 us in front of row 7's factor.  Same terms
// in the same order: the result does not change by a bit.
template <bool LAST, bool EARLY = false>
__device__ __forceinline__ bool df_column(const DfArgs &a, double *S, int *pg, const double *dvb, const int row0, const int j,
                                          const int k0, double (*Xs)[LL_LD], double (*Ls)[LL_LD], double (*Dv)[16][DV_LD],
                                          int *wslot, int &wc, d4 (&cold_d)[2][2], d4 (&acc_d)[2][2], const int trow,
                                          double *xt_row = nullptr, const double *lt_src = nullptr, const int lt_e = 0) {
    const int n = a.n;
    DF_STAMP(trow, 5 * (j & 7) + 0);
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qr = wave >> 1, qc = wave & 1;
    const int sr = tid >> 5, sc = 2 * (tid & 31);
    const int j0 = j * NB;
    const bool tri = !(qr == 0 && qc == 1);                // the quadrant above the diagonal of S_rr is never read
    const bool do_d = LAST && tri && a.fine != 1;          // (fine = 1: the earlier columns have added their terms already)
    // this row's own tiles (nobody else writes them): requested first, their latency hides behind the waits and the k loop
    d4 cold_t[2][2], acc_t[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t rr = (size_t)(row0 + qr * 32 + 16 * x + lk + 4 * q), cc = (size_t)(qc * 32 + 16 * y + lr);
                if (!LAST && lt_src) {       // block (e, j) of L^T: element [il][cl] = L[j0 + cl][e NB + il], zero below its diagonal
                    const int il = qr * 32 + 16 * x + lk + 4 * q, cl = (int)cc;
                    const double v = lt_src[(size_t)(j0 + cl) * n + lt_e * NB + il];
                    cold_t[x][y][q] = (j > lt_e || cl >= il) ? v : 0.0;
                } else cold_t[x][y][q] = S[rr * n + j0 + cc];
                if (LAST) cold_d[x][y][q] = S[rr * n + row0 + cc];
            }
            acc_t[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
        }
    int seen = -2;
    if (j > k0) {
        // fine: term k of the gather needs tile (j, k) of block row j only -- in memory as soon as that row has solved its column k,
        // long before it has announced all of them (the last column of a late row otherwise starts its r - 1 terms, 5 us each, only
        // when the row above has finished its own: 30 us in front of row 7's factor where rows 1-4 need 7, tools/df_trace.py)
        int *pc = pg + 2 * a.nb + j;
        int have = 0;                                                        // columns of row j known to be in memory
        if (a.fine) {
            have = df_wait(pc, k0 + 1, a.abort_w, &wslot[wc++ & 1]);
            if (have < 0) return false;
        } else {
            seen = df_wait(pg + j, j, a.abort_w, &wslot[wc++ & 1]);          // the panels of block row j are in memory
            if (seen < 0) return false;
        }
        DF_STAMP(trow, 5 * (j & 7) + 1);
        d2 vx[8], vl[8];
        auto gload = [&](int k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vl[i] = *reinterpret_cast<const d2 *>(S + (size_t)(j0 + sr + 8 * i) * n + k * NB + sc);
                vx[i] = *reinterpret_cast<const d2 *>(S + (size_t)(row0 + sr + 8 * i) * n + k * NB + sc);
            }
        };
        gload(k0);
        for (int k = k0; k < j; ++k) {
            if (k > k0) __syncthreads();                    // everyone is done reading the previous tiles
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                Ls[sr + 8 * i][sc] = vl[i].x; Ls[sr + 8 * i][sc + 1] = vl[i].y;
                Xs[sr + 8 * i][sc] = vx[i].x; Xs[sr + 8 * i][sc + 1] = vx[i].y;
            }
            __syncthreads();
            if (k + 1 < j) {
                if (a.fine && have < k + 2) {
                    have = df_wait(pc, k + 2, a.abort_w, &wslot[wc++ & 1]);
                    if (have < 0) return false;
                }
                gload(k + 1);                               // in flight behind this step's MFMAs
            }
#pragma unroll 4
            for (int ks = 0; ks < NB / 4; ++ks) {
                double bl[2], ax[2], bx[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    bl[x] = Ls[qc * 32 + 16 * x + lr][4 * ks + lk];            // B[k][col] = L(j,k)[col][k]
                    ax[x] = Xs[qr * 32 + 16 * x + lr][4 * ks + lk];
                    bx[x] = do_d ? Xs[qc * 32 + 16 * x + lr][4 * ks + lk] : 0.0;
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc_t[x][y] = mfma_f64(ax[x], bl[y], acc_t[x][y]);
                if (do_d) {
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y) acc_d[x][y] = mfma_f64(ax[x], bx[y], acc_d[x][y]);
                }
            }
        }
        __syncthreads();
    }
    DF_STAMP(trow, 5 * (j & 7) + 2);
    // T into sm0
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Xs[qr * 32 + 16 * x + lk + 4 * q][qc * 32 + 16 * y + lr] = cold_t[x][y][q] - acc_t[x][y][q];
    if (seen < j + 1) {                                                       // L_jj and its inverted sub-blocks are in memory
        seen = df_wait(pg + j, j + 1, a.abort_w, &wslot[wc++ & 1]);
        if (seen < 0) return false;
    }
    DF_STAMP(trow, 5 * (j & 7) + 3);
    // -L_jj (exactly zero above the diagonal: the refinement step reads the diagonal blocks) into sm1
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = sr + 8 * i;
        const d2 v = *reinterpret_cast<const d2 *>(S + (size_t)(j0 + r) * n + j0 + sc);
        Ls[r][sc] = (sc <= r) ? -v.x : 0.0;
        Ls[r][sc + 1] = (sc + 1 <= r) ? -v.y : 0.0;
    }
    {
        const double *dv = dvb + (size_t)j * DF_DINV;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            Dv[e >> 8][(e >> 4) & 15][e & 15] = dv[e];
        }
    }
    __syncthreads();
    DF_STAMP(trow, 52 + (j & 7));
    // X(r,j) = T L_jj^-T (potrf_panel_kernel has the derivation)
    d4 Rt[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rt[s4][r] = Xs[16 * wave + lr][16 * s4 + 4 * r + lk];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = mfma_f64(Dv[s4][lr][4 * ks + lk], Rt[s4][ks], x);
        d4 res = Rt[s4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) res = mfma_f64(Ls[16 * s4 + lr][16 * s4 + 4 * ks + lk], x[ks], res);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = mfma_f64(Dv[s4][lr][4 * ks + lk], res[ks], x);
        Rt[s4] = x;
#pragma unroll
        for (int t = s4 + 1; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Rt[t] = mfma_f64(Ls[16 * t + lr][16 * s4 + 4 * ks + lk], x[ks], Rt[t]);
    }
    double *Rl = S + (size_t)(row0 + 16 * wave + lr) * n + j0 + lk;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rl[16 * s4 + 4 * r] = Rt[s4][r];
    if (!LAST && xt_row) {                                 // block (j, e) of L^-1 = X(e,j)^T: 128-byte runs along lr
        double *Tl = xt_row + (size_t)(j0 + lk) * n + 16 * wave + lr;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int r = 0; r < 4; ++r) Tl[(size_t)(16 * s4 + 4 * r) * n] = Rt[s4][r];
    }
    if (LAST) {
        // the newest term of S_rr comes from the tile just solved (each wavefront rewrites its own 16 rows of sm0, which it
        // alone has read)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xs[16 * wave + lr][16 * s4 + 4 * r + lk] = Rt[s4][r];
        // X(r,j) is in sm0 for everyone; the global stores stay in flight (the caller waits for them in front of the barrier
        // that publishes them): a barrier that orders the LDS traffic only
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (EARLY) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xs[16 * wave + lr][16 * s4 + 4 * r + lk] = Rt[s4][r];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 64) df_publish(pg + 2 * a.nb + row0 / NB, j + 1);           // column j of this row is in memory -- say so
        if (tri) {
#pragma unroll 4
            for (int ks = 0; ks < NB / 4; ++ks) {
                double ax[2], bx[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    ax[x] = Xs[qr * 32 + 16 * x + lr][4 * ks + lk];
                    bx[x] = Xs[qc * 32 + 16 * x + lr][4 * ks + lk];
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc_d[x][y] = mfma_f64(ax[x], bx[y], acc_d[x][y]);
            }
        }
        __syncthreads();                                   // sm0 / sm1 are free for the next column
    } else {
        if (a.fine && row0 < n) {                          // (a main row without the early sums: not used by the kernel below)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 64) df_publish(pg + 2 * a.nb + row0 / NB, j + 1);
        } else __syncthreads();                            // sm0 / sm1 are free for the next column
    }
    DF_STAMP(trow, 5 * (j & 7) + 4);
    return true;
}

// An extra-row block of which only the FIRST row is live (the row b of the ELBO: b <- b L^-T, of which the caller reads
// |b L^-T|^2): a vector walks the block columns instead of a 64-row panel -- per column a few 64 x 64 matrix-vector products
// (each wavefront a quarter of every term's inner range, partial sums through LDS) and a 64-step forward substitution in one
// wavefront -- 1/64 of the panel's work and operand traffic.  It keeps up with the diagonal blocks however late its workgroup
// starts (about 5 us per column), so the launch ends a substitution after the last diagonal factor instead of a 7-term panel
// gather (30 us at 128 matrices).  The other 63 rows of the block are not touched.
__device__ __forceinline__ void df_vector_row(const DfArgs &a, double *S, int *pg, const double *dvb, const int row0, const int b,
                                              double *xv /* [n] */, double (*Ls)[LL_LD], double *part /* [4][64] */,
                                              double *invd /* [64] */, int *wslot) {
    const int n = a.n, nb = a.nb, tid = threadIdx.x, c = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sr = tid >> 5, sc = 2 * (tid & 31);
    double *v = S + (size_t)row0 * n;
    int wc = 0;
    for (int j = 0; j < nb; ++j) {
        const int j0 = j * NB;
        const double vj = v[j0 + c];                        // this row is nobody else's
        double acc = 0.0;
        int seen = -2;
        if (j > 0) {
            seen = df_wait(pg + j, j, a.abort_w, &wslot[wc++ & 1]);          // the panels of block row j are in memory
            if (seen < 0) { if (tid == 0 && a.info) a.info[b] = -1; return; }
            const double *Lrow = S + (size_t)(j0 + c) * n + 16 * q;           // row c of block row j, this wavefront's quarter
            for (int k = 0; k < j; ++k) {
                const double *xp = xv + k * NB + 16 * q;
#pragma unroll
                for (int m = 0; m < 16; m += 2) {
                    const d2 l = *reinterpret_cast<const d2 *>(Lrow + k * NB + m);
                    acc = fma(xp[m], l.x, acc);
                    acc = fma(xp[m + 1], l.y, acc);
                }
            }
        }
        part[q * 64 + c] = acc;
        if (seen < j + 1) {                                                   // L_jj and its inverted sub-blocks are in memory
            seen = df_wait(pg + j, j + 1, a.abort_w, &wslot[wc++ & 1]);
            if (seen < 0) { if (tid == 0 && a.info) a.info[b] = -1; return; }
        } else __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = sr + 8 * i;
            const d2 l = *reinterpret_cast<const d2 *>(S + (size_t)(j0 + r) * n + j0 + sc);
            Ls[r][sc] = l.x; Ls[r][sc + 1] = l.y;
        }
        if (tid < 64) invd[tid] = dvb[(size_t)j * DF_DINV + ((tid >> 4) * 16 + (tid & 15)) * 16 + (tid & 15)];   // 1 / L_jj[c][c]
        __syncthreads();
        if (q == 0) {
            double t = vj - ((part[c] + part[64 + c]) + (part[128 + c] + part[192 + c]));
            double x = 0.0;
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                const double xm = readlane_f64(t, m) * invd[m];               // lane m's t is final at step m
                if (c == m) x = xm;
                if (c > m) t = fma(-xm, Ls[c][m], t);
            }
            v[j0 + c] = x;
            xv[j0 + c] = x;
        }
        __syncthreads();
    }
}

// Round 5 (lead 2): the inverse's tiles accumulated AS THE COLUMNS ARRIVE.  Tile (e, f) of A^-1 = W W^T, W = L^-T, is
// sum_{j >= e} W(e,j) W(f,j)^T, and term j exists as soon as block column j of rows e and f is solved -- df_inverse_tiles forms all
// of it behind the last column (up to 20 tile products: 60-70 us behind the chain of a K_uu matrix).  Here identity-row workgroup e,
// right behind its column j: announces it (word 3 nb + e = columns done), and for every f <= e waits for row f's column j, carries the
// tile's accumulator over from the previous column THROUGH MEMORY (the same lanes wrote it), runs the same 16 MFMA steps on it and
// puts it back; the last column also writes the mirror image.  The same products on the same accumulators in the same order as
// df_inverse_tiles: the same bits.  What is left behind the last column is one product per tile of the workgroup's row.
__device__ __forceinline__ bool df_inverse_column(const DfArgs &a, double *S, int *pg, const int b, const int e, const int j,
                                                  double (*Xs)[LL_LD], double (*Ls)[LL_LD], int *wslot, int &wc) {
    const int n = a.n, nb = a.nb;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qr = wave >> 1, qc = wave & 1;
    const int sr = tid >> 5, sc = 2 * (tid & 31);
    int *pc = pg + 3 * nb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wavefront's stores of W(e,j) (and of the previous column's tiles) have left it
    __syncthreads();
    if (tid == 0) df_publish(pc + e, j + 1);
    double *Kb = a.kinv + (size_t)b * a.kinv_stride;
    const double *We = S + (size_t)(n + e * NB) * n + (size_t)j * NB;
    const bool last = j + 1 == nb;
    for (int f = 0; f <= e; ++f) {
        if (f < e) {
            const int seen = df_wait(pc + f, j + 1, a.abort_w, &wslot[wc++ & 1]);
            if (seen < 0) return false;
        }
        const double *Wf = S + (size_t)(n + f * NB) * n + (size_t)j * NB;
        d2 vx[8], vl[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            vx[i] = *reinterpret_cast<const d2 *>(We + (size_t)(sr + 8 * i) * n + sc);
            vl[i] = *reinterpret_cast<const d2 *>(Wf + (size_t)(sr + 8 * i) * n + sc);
        }
        d4 acc[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t i = (size_t)(e * NB + qr * 32 + 16 * x + lk + 4 * q), jj = (size_t)(f * NB + qc * 32 + 16 * y + lr);
                    acc[x][y][q] = (j > e) ? Kb[i * n + jj] : 0.0;
                }
        __syncthreads();                                // everyone is done reading the previous tiles
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Xs[sr + 8 * i][sc] = vx[i].x; Xs[sr + 8 * i][sc + 1] = vx[i].y;
            Ls[sr + 8 * i][sc] = vl[i].x; Ls[sr + 8 * i][sc + 1] = vl[i].y;
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < NB / 4; ++ks) {
            double ax[2], bl[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                ax[x] = Xs[qr * 32 + 16 * x + lr][4 * ks + lk];
                bl[x] = Ls[qc * 32 + 16 * x + lr][4 * ks + lk];
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = mfma_f64(ax[x], bl[y], acc[x][y]);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t i = (size_t)(e * NB + qr * 32 + 16 * x + lk + 4 * q), jj = (size_t)(f * NB + qc * 32 + 16 * y + lr);
                    Kb[i * n + jj] = acc[x][y][q];
                    if (last && f < e) Kb[jj * n + i] = acc[x][y][q];
                }
    }
    __syncthreads();                                    // (the next column's staging reuses the tiles)
    return true;
}

// Identity-row workgroup e, its row W(e, e..) = (L^-T)(e, .) complete: announce it, then form the tiles (e, f), f <= e, of
// A^-1 = L^-T L^-1 = W W^T:  (e,f) = sum_{j >= e} W(e,j) W(f,j)^T  (rows f < e belong to workgroups dispatched earlier).  The
// same staging loop as the panel gather; both triangles are written.  A separate product launch after the factorisation
// costs the K_uu chain 0.13-0.17 ms beside the K_fu build (its workgroups queue behind that kernel's), these tiles 0.04.
// DfArgs::kinv_help (every workgroup of the launch resident at once): main-row workgroup e, done with its own block row, takes
// the tiles (e, f) with e + f odd off identity-row workgroup e -- the most loaded one had 20 tile products behind the chain (60 us;
// half of them now), and the main rows had left the chip by then.
__device__ __forceinline__ bool df_inverse_tiles(const DfArgs &a, double *S, int *pg, const int b, const int e,
                                                 double (*Xs)[LL_LD], double (*Ls)[LL_LD], int *wslot, const bool helper = false) {
    const int n = a.n, nb = a.nb;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qr = wave >> 1, qc = wave & 1;
    const int sr = tid >> 5, sc = 2 * (tid & 31);
    int wc = 0;
    if (!helper) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wavefront's stores of the row have left it
        __syncthreads();
        if (tid == 0) df_publish(pg + nb + e, 1);
    } else if (e > 0) {                                         // (tile (0, 0) is even: row 0's helper has nothing to do)
        const int seen = df_wait(pg + nb + e, 1, a.abort_w, &wslot[wc++ & 1]);
        if (seen < 0) return false;
    }
    double *Kb = a.kinv + (size_t)b * a.kinv_stride;
    const double *We = S + (size_t)(n + e * NB) * n;
    for (int f = 0; f <= e; ++f) {
        if (a.kinv_help && (((e + f) & 1) != 0) != helper) continue;
        if (f < e) {
            const int seen = df_wait(pg + nb + f, 1, a.abort_w, &wslot[wc++ & 1]);
            if (seen < 0) return false;
        }
        const double *Wf = S + (size_t)(n + f * NB) * n;
        d4 acc[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
        d2 vx[8], vl[8];
        auto gload = [&](int j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vx[i] = *reinterpret_cast<const d2 *>(We + (size_t)(sr + 8 * i) * n + j * NB + sc);
                vl[i] = *reinterpret_cast<const d2 *>(Wf + (size_t)(sr + 8 * i) * n + j * NB + sc);
            }
        };
        gload(e);
        for (int j = e; j < nb; ++j) {
            __syncthreads();                                // everyone is done reading the previous tiles
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                Xs[sr + 8 * i][sc] = vx[i].x; Xs[sr + 8 * i][sc + 1] = vx[i].y;
                Ls[sr + 8 * i][sc] = vl[i].x; Ls[sr + 8 * i][sc + 1] = vl[i].y;
            }
            __syncthreads();
            if (j + 1 < nb) gload(j + 1);
#pragma unroll 4
            for (int ks = 0; ks < NB / 4; ++ks) {
                double ax[2], bl[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    ax[x] = Xs[qr * 32 + 16 * x + lr][4 * ks + lk];
                    bl[x] = Ls[qc * 32 + 16 * x + lr][4 * ks + lk];
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = mfma_f64(ax[x], bl[y], acc[x][y]);
            }
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t i = (size_t)(e * NB + qr * 32 + 16 * x + lk + 4 * q), jj = (size_t)(f * NB + qc * 32 + 16 * y + lr);
                    Kb[i * n + jj] = acc[x][y][q];
                    if (f < e) Kb[jj * n + i] = acc[x][y][q];
                }
    }
    return true;
}

template <bool PIPE>
__global__ __launch_bounds__(256, 2) void potrf_df_kernel(DfArgs a) {
    __shared__ double sm0[NB * LL_LD];      // X(r,k) tiles; then T = A(r,j) - sum and the solved X(r,j)
    __shared__ double sm1[NB * LL_LD];      // L(j,k) tiles; then -L_jj; at the end S_rr and its factor
    __shared__ double invd[NB];
    __shared__ double Dv[4][16][DV_LD];
    __shared__ int wslot[2];
    const int n = a.n, nb = a.nb;
    int grp, rem, ri;
    if (a.defer_ext) {
        // several groups with identity-structured rows: the main rows (and the tail rows) of EVERY group first, then the
        // identity-structured rows of every group -- those wait for diagonal blocks only, and behind the main rows of all matrices
        // they find them finished and hold their slot for a few microseconds per column instead of for the length of the chain
        const int rows_a = nb + a.next - a.nid, per_a = rows_a * a.G, n_a = ((a.batch + a.G - 1) / a.G) * per_a;
        if ((int)blockIdx.x < n_a) {
            grp = blockIdx.x / per_a; rem = blockIdx.x % per_a;
            const int rr = rem / a.G;
            ri = (rr < nb) ? rr : rr + a.nid;
        } else {
            const int id2 = blockIdx.x - n_a, per_b = a.nid * a.G;
            grp = id2 / per_b; rem = id2 % per_b;
            ri = nb + rem / a.G;
        }
    } else {
        const int per_group = (nb + a.next) * a.G;
        grp = blockIdx.x / per_group; rem = blockIdx.x % per_group;
        ri = rem / a.G;
    }
    const int b = grp * a.G + rem % a.G;
    if (b >= a.batch) return;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qr = wave >> 1, qc = wave & 1;
    double *S = a.A + (size_t)b * a.slab_stride;
    int *pg = a.prog + (size_t)b * DF_PS;
    double *dvb = a.dinv + (size_t)b * nb * DF_DINV;
    const bool main_row = ri < nb;
    int row0, k0 = 0, ncols;                // first row of the block row; first block column with non-zero X(r,k); columns to solve
    if (main_row) { row0 = ri * NB; ncols = ri; }
    else {
        const int e = ri - nb;
        row0 = n + e * NB; ncols = nb;
        if (e < a.nid) k0 = e;              // identity-structured rows: block e is zero left of block column e
    }
    double (*Xs)[LL_LD] = reinterpret_cast<double(*)[LL_LD]>(sm0);
    double (*Ls)[LL_LD] = reinterpret_cast<double(*)[LL_LD]>(sm1);
    if (!main_row && a.vec_tail && ri - nb >= a.nid) {
        df_vector_row(a, S, pg, dvb, row0, b, sm0, Ls, &Dv[0][0][0], invd, wslot);
        return;
    }
    int wc = 0;
    const int trow = (b == 0) ? ri : -1;
    DF_STAMP(trow, 51);
    // S_rr = A(r,r) - sum_{k<r} X(r,k) X(r,k)^T (main rows)
    d4 cold_d[2][2], acc_d[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc_d[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    {
        const int nplain = main_row ? ncols - 1 : ncols;
        double *xt_row = (a.xt && !main_row && ri - nb < a.nid) ? a.xt + (size_t)b * a.xt_stride + (size_t)(ri - nb) * NB : nullptr;
        const double *lt_src = (a.lt && !main_row && ri - nb < a.nid) ? a.lt + (size_t)(b % a.lt_dl) * a.lt_stride : nullptr;
        if (main_row && a.fine == 1) {
            for (int j = 0; j < nplain; ++j) {
                if (!df_column<false, true>(a, S, pg, dvb, row0, j, 0, Xs, Ls, Dv, wslot, wc, cold_d, acc_d, trow)) {
                    if (tid == 0 && a.info) a.info[b] = -1;
                    return;
                }
            }
        } else {
            const bool kacc = a.kacc && !main_row && ri - nb < a.nid;
            for (int j = k0; j < nplain; ++j) {
                d4 unused_c[2][2], unused_a[2][2];
                if (!df_column<false>(a, S, pg, dvb, row0, j, k0, Xs, Ls, Dv, wslot, wc, unused_c, unused_a, trow, xt_row, lt_src,
                                      ri - nb)) {
                    if (tid == 0 && a.info) a.info[b] = -1;
                    return;
                }
                if (kacc && !df_inverse_column(a, S, pg, b, ri - nb, j, Xs, Ls, wslot, wc)) {
                    if (tid == 0 && a.info) a.info[b] = -1;
                    return;
                }
            }
        }
    }
    if (!main_row) {
        if (a.kinv && !a.kacc && ri - nb < a.nid && !df_inverse_tiles(a, S, pg, b, ri - nb, Xs, Ls, wslot)) {
            if (tid == 0 && a.info) a.info[b] = -1;
        }
        return;
    }
    if (ncols > 0) {
        if (!df_column<true>(a, S, pg, dvb, row0, ncols - 1, k0, Xs, Ls, Dv, wslot, wc, cold_d, acc_d, trow)) {
            if (tid == 0 && a.info) a.info[b] = -1;
            return;
        }
        if (!(qr == 0 && qc == 1)) {
#pragma unroll 4
            for (int ks = 0; ks < NB / 4; ++ks) {
                double ax[2], bx[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    ax[x] = Xs[qr * 32 + 16 * x + lr][4 * ks + lk];
                    bx[x] = Xs[qc * 32 + 16 * x + lr][4 * ks + lk];
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc_d[x][y] = mfma_f64(ax[x], bx[y], acc_d[x][y]);
            }
        }
    } else {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cold_d[x][y][q] = S[(size_t)(row0 + qr * 32 + 16 * x + lk + 4 * q) * n + row0 + qc * 32 + 16 * y + lr];
    }
    double (*Ts)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(sm1);     // -L_jj is no longer needed (all past the solve)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Ts[qr * 32 + 16 * x + lk + 4 * q][qc * 32 + 16 * y + lr] = cold_d[x][y][q] - acc_d[x][y][q];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's panel stores have left it (published below)
    __syncthreads();                                      // ... and everyone is done with X(r,j) in sm0: the factor goes there
    double (*Lr)[LR_LD] = reinterpret_cast<double(*)[LR_LD]>(sm0);
    DF_STAMP(trow, 48);
    // the panels of this row are published by wavefront 1 while wavefront 0 runs the pivot chain
    if (ncols > 0 && tid == 64) {
        df_publish(pg + ri, ri);
        if (a.fine) __hip_atomic_store(pg + 2 * nb + ri, ri, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (behind the release above)
    }
    diag_block_finish<PIPE>(Ts, Lr, invd, S, n, row0, a.info + b, dvb + (size_t)ri * DF_DINV, DF_MFMA_FACTOR ? Dv : nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DF_STAMP(trow, 49);
#ifdef FFVD_DF_TEST_STALL
    if (b == 0 && ri == 0) return;      // test build (tests/test_gpu_ops.py): matrix 0 never announces its first diagonal block
#endif
    if (tid == 0) df_publish(pg + ri, ri + 1);
    DF_STAMP(trow, 50);
    if (a.kinv_help) {
        __syncthreads();                                      // (LDS of the diagonal factor is free again)
        if (!df_inverse_tiles(a, S, pg, b, ri, Xs, Ls, wslot, true)) {
            if (tid == 0 && a.info) a.info[b] = -1;
        }
    }
}

size_t potrf_scratch_doubles(int n, int batch) {
    const size_t nb = (size_t)(n / NB), bt = (size_t)(batch > 0 ? batch : 1);
    const size_t flow = ((bt * DF_PS + 4) * sizeof(int) + 15) / 16 * 2 + bt * nb * DF_DINV;
    const size_t step = bt * DINV_STRIDE;
    return flow > step ? flow : step;
}

// Zero the progress words of a dataflow launch for `batch` matrices (they sit at the start of the scratch block, padded to 16
// bytes).  launch_potrf_ext does this itself unless told that the caller already has (words_zeroed).
void potrf_flow_clear(hipStream_t stream, double *scratch, int batch) {
    const size_t words = ((size_t)batch * DF_PS + 4 + 3) / 4 * 4;
    // our own fill kernel, not hipMemsetAsync: in front of a 16-matrix factorisation the runtime's memset started 40 us after
    // the kernel before it had finished (tools/prof_timeline.sh, SYNC_S=4)
    const size_t n = words / 2;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, scratch, n, 0.0);
}

static void launch_potrf_flow(hipStream_t stream, double *A, int n, int extra_rows, int identity_rows, int batch,
                              size_t slab_stride, int32_t *info, double *scratch, double *linv_t, size_t linv_t_stride,
                              bool words_zeroed, bool tail_is_vector, double *kinv, size_t kinv_stride, const double *lt_rows,
                              size_t lt_stride, int lt_dl, bool kinv_help) {
    DfArgs a{};
    a.lt = lt_rows; a.lt_stride = lt_stride; a.lt_dl = lt_dl > 0 ? lt_dl : 1;
    a.xt = linv_t; a.xt_stride = linv_t_stride;
    a.vec_tail = tail_is_vector ? 1 : 0;
    if (kinv && identity_rows == n && 2 * (n / NB) <= DF_PS) { a.kinv = kinv; a.kinv_stride = kinv_stride; }
    a.A = A; a.n = n; a.nb = n / NB; a.next = extra_rows / NB; a.nid = identity_rows / NB; a.batch = batch;
    a.slab_stride = slab_stride; a.info = info;
    // the polled words sit at the start of the scratch block, padded to 16 bytes, and are zeroed before every launch
    const size_t words = ((size_t)batch * DF_PS + 4 + 3) / 4 * 4;
    a.prog = reinterpret_cast<int *>(scratch);
    a.abort_w = a.prog + (size_t)batch * DF_PS;
    a.dinv = scratch + words / 2;
    if (!words_zeroed) potrf_flow_clear(stream, scratch, batch);
    const int R = a.nb + a.next, bp = (batch + 7) / 8 * 8;
    // all matrices in lockstep while the grid is a few rounds of the chip (rows leave early and make room); beyond that,
    // groups of about two chip-fulls of block rows, so that the rows of one matrix run together (they share L(j,k) in L2)
    a.G = ((size_t)bp * R <= 2048) ? bp : ((1024 / R + 7) / 8 * 8 < 8 ? 8 : (1024 / R + 7) / 8 * 8);
    if (a.G > bp) a.G = bp;
    const int groups = (batch + a.G - 1) / a.G;
    static const int defer_mode = [] { const char *e = getenv("FFVD_DF_DEFER"); return e ? atoi(e) : -1; }();
    a.defer_ext = (a.nid > 0 && groups > 1 && !a.xt && !a.kinv && defer_mode != 0) ? 1 : 0;
    // Few matrices (every block row finds a CU of its own): 16 KB of unused dynamic LDS keep a second workgroup off the CU --
    // a pivot chain that shares its SIMD with another row's MFMA loop takes up to twice as long (tools/df_trace.py)
    static const int pad_mode = [] { const char *e = getenv("FFVD_DF_PAD"); return e ? atoi(e) : -1; }();
    // (... and up to 640 block rows without identity-structured rows -- 64 matrices of a 16-chain rank: 1.87 vs 1.945 ms per iteration with
    //  one row per compute unit; 288 and 432 rows no difference, 864 none, 1152 rows 2 % slower: profiles/r04_ab_df_pad.txt)
    const bool one_per_cu = (size_t)batch * R <= 256 || (a.nid == 0 && (size_t)batch * R <= 640);
    static const int kh_mode = [] { const char *e = getenv("FFVD_DF_KINV_HELP"); return e ? atoi(e) : -1; }();
    // column-wise accumulation of the inverse's tiles (df_inverse_column): opt-in, FFVD_DF_KACC=1 -- built for the few-chain schedules,
    // where the chain's tail is on the iteration's critical path, bit-identical, and measured SLOWER there (per-rank iteration at
    // 1 / 2 / 4 / 8 chains 0.458 / 0.540 / 0.718 / 1.026 ms against 0.428 / 0.508 / 0.685 / 1.003, profiles/r05_ab_kacc.txt): a tile
    // update through memory is a wait, two tile loads, the accumulator's round trip, 64 MFMAs and a store in sequence (4-8 us), row 7
    // has eight of them behind the last column, nobody helps it (the main rows' help needs finished rows), and the identity rows now
    // wait for each other at every column.  Needs a fourth set of progress words.
    const int kacc_mode = [] { const char *e = getenv("FFVD_DF_KACC"); return e ? atoi(e) : -1; }();      // (read per launch: tests switch it)
    a.kacc = (a.kinv && a.nid == a.nb && 4 * a.nb <= DF_PS && kacc_mode > 0) ? 1 : 0;
    a.kinv_help = (kinv_help && a.kinv && !a.kacc && a.nid == a.nb && (size_t)batch * R <= 256 && kh_mode != 0) ? 1 : 0;
    const bool alone = (pad_mode >= 0) ? (pad_mode != 0) : one_per_cu;
    const int fine_mode = [] { const char *e = getenv("FFVD_DF_FINE"); return e ? atoi(e) : -1; }();      // (read per launch: tests switch it)
    a.fine = ((fine_mode >= 0 ? fine_mode != 0 : (alone && (size_t)batch * R <= 256)) && 3 * a.nb <= DF_PS) ? (fine_mode == 2 ? 2 : 1) : 0;    // (2: A/B, no early sums)
    hipLaunchKernelGGL(potrf_df_kernel<true>, dim3((unsigned)((size_t)groups * R * a.G)), dim3(256), alone ? 16384 : 0, stream, a);
}

// Which blocked variant factorises a batch: the dataflow kernel (one launch, rows handing blocks to each other) wherever the
// factorisation sits on the critical path; the right-looking launches for the few matrices of the K_uu chain, whose small
// workgroups slip in beside the Gram kernel when the chain runs on the side stream (the 76.8 KB / 256-VGPR row workgroups of the
// other two variants would wait for that kernel to drain).  The per-column left-looking launches (round 2's first variant)
// stay selectable.  FFVD_CHOL=flow|left|right forces one (read once).
static int chol_mode() {                    // 0 = auto, 1 = left-looking launches, 2 = right-looking launches, 3 = dataflow
    static const int v = [] {
        const char *e = getenv("FFVD_CHOL");
        if (!e) return 0;
        return (e[0] == 'l') ? 1 : ((e[0] == 'r') ? 2 : ((e[0] == 'f') ? 3 : 0));
    }();
    return v;
}
// Per-thread override (0 = none): the caller re-runs a batch whose dataflow launch gave up (a bounded wait fired) with one of
// the launch-per-step variants, which have no inter-workgroup waits and therefore cannot stall (abi.hip, stall recovery).
static thread_local int g_chol_override = 0;
void potrf_override_variant(int variant) { g_chol_override = variant; }
int potrf_override_current() { return g_chol_override; }
static int chol_variant(int batch, int nb, int hint);
bool potrf_flow_selected(int n, int batch, int hint) { return chol_variant(batch, n / NB, hint) == 3; }
bool potrf_flow_forms_inverse(int n, int batch, int hint) { return potrf_flow_selected(n, batch, hint) && 2 * (n / NB) <= DF_PS; }
static int chol_variant(int batch, int nb, int hint) {
    int m = g_chol_override ? g_chol_override : chol_mode();
    if (m == 0) m = (hint == CHOL_FLOW || batch >= 32) ? 3 : 2;
    if (m == 3 && nb > DF_PS) m = 1;
    return m;
}

void launch_potrf_ext(hipStream_t stream, double *A, int n, int extra_rows, int identity_rows, int batch,
                      size_t slab_stride, int32_t *info, double *dinv, int hint, double *linv_t, size_t linv_t_stride,
                      bool words_zeroed, bool tail_is_vector, double *kinv, size_t kinv_stride, const double *lt_rows,
                      size_t lt_stride, int lt_dl, bool kinv_help) {
    const int nb = n / NB;
    const int nid = identity_rows / NB, ntail = extra_rows / NB - nid;
    const int variant = chol_variant(batch, nb, hint);
    if (variant == 3) {
        launch_potrf_flow(stream, A, n, extra_rows, identity_rows, batch, slab_stride, info, dinv, linv_t, linv_t_stride, words_zeroed,
                          tail_is_vector, kinv, kinv_stride, lt_rows, lt_stride, lt_dl, kinv_help);
        return;
    }
    if (variant == 1) {
        const int groups = (batch + 7) / 8;
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(batch), dim3(256), 0, stream, A, n, 0, slab_stride, info, dinv);
        for (int j = 0; j < nb; ++j) {
            const int nmain = nb - j - 1;
            const int nlive = (j + 1 < nid) ? j + 1 : nid;
            const int nchunks = nmain + nlive + ntail;
            if (nchunks == 0) continue;
            // few workgroups: the launch lasts as long as the look-ahead workgroup's 64-pivot chain -> pipelined factor variant
            if ((size_t)nchunks * batch <= 512)
                hipLaunchKernelGGL(potrf_ll_kernel<true>, dim3(groups * 8 * nchunks), dim3(256), 0, stream, A, n, j, nmain,
                                   nchunks, slab_stride, info, nlive, nid, batch, dinv);
            else
                hipLaunchKernelGGL(potrf_ll_kernel<false>, dim3(groups * 8 * nchunks), dim3(256), 0, stream, A, n, j, nmain,
                                   nchunks, slab_stride, info, nlive, nid, batch, dinv);
        }
        return;
    }
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(batch), dim3(256), 0, stream, A, n, 0, slab_stride, info, dinv);
    for (int k = 0; k < nb; ++k) {
        const int nmain = nb - k - 1;
        const int nlive = (k + 1 < nid) ? k + 1 : nid;
        const int nextra = nlive + ntail;
        const int nchunks = nmain + nextra;
        if (nchunks > 0)
            hipLaunchKernelGGL(potrf_panel_kernel, dim3(nchunks, batch), dim3(256), 0, stream, A, n, k, nmain,
                               slab_stride, nlive, nid, dinv);
        const int n1 = nmain;
        if (n1 > 0) {
            const int nmain_tiles = n1 * (n1 + 1) / 2;
            const int ntiles = nmain_tiles + nextra * n1;
            // few workgroups: the launch lasts as long as tile 0's look-ahead factorisation -> pipelined variant
            // (256 VGPRs, one workgroup per CU); many: the leaner variant keeps two workgroups per CU
            if ((size_t)ntiles * batch <= 512)
                hipLaunchKernelGGL(potrf_trail_kernel<true>, dim3(ntiles, batch), dim3(256), 0, stream, A, n, k,
                                   nmain_tiles, n1, slab_stride, info, nlive, nid, dinv);
            else
                hipLaunchKernelGGL(potrf_trail_kernel<false>, dim3(ntiles, batch), dim3(256), 0, stream, A, n, k,
                                   nmain_tiles, n1, slab_stride, info, nlive, nid, dinv);
        }
    }
}

// rows [row0, row0 + n) of every slab <- identity (n x n), used to re-arm the extra rows that become L^-T
__global__ void set_identity_kernel(double *A, size_t slab_stride, int row0, int n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx % n);
    A[(size_t)blockIdx.y * slab_stride + (size_t)(row0 + i) * n + j] = (i == j) ? 1.0 : 0.0;
}
void launch_set_identity(hipStream_t stream, double *A, size_t slab_stride, int row0, int n, int batch) {
    hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256), batch), dim3(256), 0, stream, A,
                       slab_stride, row0, n);
}

// rows [row0, row0 + n) of slab b <- L_d^T (upper triangular, zeros left of the diagonal), d = b % Dl, from the lower-triangular
// n x n factor L_d (ld n).  Training forward of the Gram route: with L^T instead of I in the extension rows, the factorisation of
// A = K + K_uf K_fu / Q leaves  L^T L_A^-T = L_H^-T  there, L_H = L^-1 L_A being the Cholesky factor of H = L^-1 A L^-T -- the
// whitened quantities of the backward pass without forming H.  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void set_lt_rows_kernel(const double *L, size_t l_stride, int Dl, double *A, size_t slab_stride,
                                                          int row0, int n) {
    __shared__ double tile[64][65];
    const int nt = n / 64, tr = blockIdx.x / nt, tc = blockIdx.x % nt, b = blockIdx.y;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double *Ab = A + (size_t)b * slab_stride + (size_t)(row0 + tr * 64) * n + tc * 64;
    if (tc < tr) {
        for (int r = ty; r < 64; r += 4) Ab[(size_t)r * n + tx] = 0.0;
        return;
    }
    // out[r][c] = L[tc * 64 + c][tr * 64 + r]
    const double *Lb = L + (size_t)(b % Dl) * l_stride + (size_t)(tc * 64) * n + tr * 64;
    for (int c = ty; c < 64; c += 4) tile[c][tx] = Lb[(size_t)c * n + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) Ab[(size_t)r * n + tx] = (tc > tr || tx >= r) ? tile[tx][r] : 0.0;
}
void launch_set_lt_rows(hipStream_t stream, const double *L, size_t l_stride, int Dl, double *A, size_t slab_stride, int row0,
                        int n, int batch) {
    const int nt = n / 64;
    hipLaunchKernelGGL(set_lt_rows_kernel, dim3(nt * nt, batch), dim3(256), 0, stream, L, l_stride, Dl, A, slab_stride, row0, n);
}

// ---------------------------------------------------------------------------------------------
// Projection:  F = K_fu * L^{-T}  (conditionals_multi_output.py:240-242), K_fu generated on the fly.
// One workgroup (8 wavefronts) = 64 rows of one (chain, latent dim) x one column group of <= 512 columns.
// ---------------------------------------------------------------------------------------------
constexpr int KC = 32;          // k-chunk (columns of K_fu produced per barrier)
constexpr int KS_LD = STRIP + 16;   // LDS row stride of the K chunk: 80 doubles => lanes l and l+16 hit different bank halves

template <int KIND, int NW>
__global__ __launch_bounds__(NW * 64) void project_kernel(ProjectArgs a) {
    constexpr int NT = NW * 64;          // threads
    constexpr int TPW = 32 / NW;         // 16-column tiles per wavefront
    __shared__ double Ks[2][KC][KS_LD];
    __shared__ double xs[MAXP][STRIP];
    __shared__ double xx[STRIP];
    __shared__ double zs[2][KC][MAXP];
    __shared__ double zzs[2][KC];
    __shared__ double part[2][NW][STRIP];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int t0 = blockIdx.x * STRIP;
    const int g = blockIdx.y;
    const int bz = blockIdx.z;                 // index inside this pass
    const int b = a.b0 + bz;                   // global batch index
    const int s = b / a.Dl, dl = b % a.Dl, dg = a.d_begin + dl;
    const int P = a.P, Mp = a.Mp;
    const double var = a.hv.variance[dl];
    const double *Zsd = a.hv.Zs + (size_t)dl * Mp * P;
    const double *zzd = a.hv.zz + (size_t)dl * Mp;
    const double *Wd = a.W + (size_t)dl * a.w_stride;

    // ---- x rows of this strip, divided by the lengthscales (kernels_multi_output.py:170) ----
    for (int p = tid >> 6; p < P; p += NW) {
        const int t = t0 + lane;
        double v = 0.0;
        if (t < a.T) {
            v = (p < a.x_cols) ? a.x[(size_t)s * a.x_chain_stride + (size_t)t * a.x_ld + p]
                               : a.ctrl[(size_t)t * a.C + (p - a.x_cols)];
            if (KIND == 0) v = v / a.hv.len[(size_t)dl * P + p];
            else v = v * var;                                    // kernels.py:276  (X * variance) @ X2^T
        }
        xs[p][lane] = v;
    }
    const int kend = (g + 1) * 512 < Mp ? (g + 1) * 512 : Mp;
    const int nchunk = kend / KC;
    auto load_z = [&](int c, int buf) {
        if (tid < KC * P) {
            const int kk = tid / P, p = tid % P;
            zs[buf][kk][p] = Zsd[(size_t)(c * KC + kk) * P + p];
        }
        if (tid < KC) zzs[buf][tid] = zzd[c * KC + tid];
    };
    load_z(0, 0);
    __syncthreads();
    if (tid < STRIP) {
        double acc = 0.0;
        if (KIND == 0) for (int p = 0; p < P; ++p) acc += xs[p][tid] * xs[p][tid];
        xx[tid] = acc;
    }
    __syncthreads();
    auto gen = [&](int c, int buf) {
        const int row = lane;
#pragma unroll
        for (int i = 0; i < (KC * STRIP) / NT; ++i) {
            const int kk = (tid >> 6) + NW * i;
            const int kcol = c * KC + kk;
            double dot = 0.0;
            for (int p = 0; p < P; ++p) dot += xs[p][row] * zs[buf][kk][p];
            double v = kernel_value<KIND>(dot, xx[row], zzs[buf][kk], var);
            if (t0 + row >= a.T || kcol >= a.M) v = 0.0;
            Ks[buf][kk][row] = v;
        }
    };
    gen(0, 0);
    if (nchunk > 1) load_z(1, 1);
    __syncthreads();

    // ---- column tiles owned by this wavefront (snake order balances the triangular work) ----
    int c0[TPW];
    bool tv[TPW];
#pragma unroll
    for (int r = 0; r < TPW; ++r) {
        const int lt = r * NW + ((r & 1) ? NW - 1 - wave : wave);
        c0[r] = (g * 32 + lt) * 16;
        tv[r] = c0[r] < kend;
    }
    d4 acc[4][TPW];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < TPW; ++r) acc[rt][r] = (d4){0.0, 0.0, 0.0, 0.0};

    // B fragments (rows of L^{-T}) are prefetched one k-step ahead straight from L2 into registers.
    // L^{-T} is upper triangular: a 16-column tile starting at c0 only needs rows k < c0 + 16.
    // uniform row base (scalar registers) + 32-bit per-lane offset: lets the load use the saddr form
    int boff[TPW];
#pragma unroll
    for (int r = 0; r < TPW; ++r) boff[r] = lk * Mp + c0[r] + lr;
    auto loadB = [&](int kglob, double(&bv)[TPW]) {
        const double *Wk = Wd + (size_t)kglob * Mp;
#pragma unroll
        for (int r = 0; r < TPW; ++r)
            bv[r] = (tv[r] && kglob < c0[r] + 16) ? Wk[boff[r]] : 0.0;
    };
    // SIMD partners (waves w and w+4) alternate roles inside a chunk: one generates the next K_fu chunk on the
    // VALU while the other feeds the matrix pipe, instead of all eight waves doing the same phase in lockstep.
    const bool gen_first = ((wave >> 2) & 1) == 0;
    double bcur[TPW], bnxt[TPW];
    loadB(0, bcur);
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 2 < nchunk) load_z(c + 2, buf);        // zs[buf] was consumed by gen(c) before the last barrier
        if (gen_first && c + 1 < nchunk) gen(c + 1, buf ^ 1);   // zs[buf^1] was loaded one iteration ago
        double af[4], afn[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) af[rt] = Ks[buf][lk][16 * rt + lr];
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            const int kglob = c * KC + 4 * ks;
            if (kglob + 4 < kend) loadB(kglob + 4, bnxt);
            if (ks + 1 < KC / 4) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) afn[rt] = Ks[buf][4 * (ks + 1) + lk][16 * rt + lr];
            }
#pragma unroll
            for (int r = 0; r < TPW; ++r) {
                if (tv[r] && kglob < c0[r] + 16) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) acc[rt][r] = mfma_f64(af[rt], bcur[r], acc[rt][r]);
                }
            }
#pragma unroll
            for (int r = 0; r < TPW; ++r) bcur[r] = bnxt[r];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) af[rt] = afn[rt];
        }
        if (!gen_first && c + 1 < nchunk) gen(c + 1, buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    if (a.F) {
        double *Fb = a.F + ((size_t)bz * a.Tp + t0) * Mp;
#pragma unroll
        for (int r = 0; r < TPW; ++r) {
            if (!tv[r]) continue;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    Fb[(size_t)(16 * rt + lk + 4 * q) * Mp + c0[r] + lr] = acc[rt][r][q];
        }
    }
    if (a.rowsq || a.fmean) {
        double uc[TPW];
#pragma unroll
        for (int r = 0; r < TPW; ++r) {
            const int col = c0[r] + lr;
            uc[r] = (a.fmean && tv[r] && col < a.M) ? a.U[(size_t)col * a.u_ld + dg] : 0.0;
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double sq = 0.0, fm = 0.0;
#pragma unroll
                for (int r = 0; r < TPW; ++r) {
                    const double v = tv[r] ? acc[rt][r][q] : 0.0;
                    sq += v * v;
                    fm += v * uc[r];
                }
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    sq += __shfl_xor(sq, m);
                    fm += __shfl_xor(fm, m);
                }
                if (lr == 0) {
                    part[0][wave][16 * rt + lk + 4 * q] = sq;
                    part[1][wave][16 * rt + lk + 4 * q] = fm;
                }
            }
        __syncthreads();
        if (tid < STRIP) {
            double sq = 0.0, fm = 0.0;
            for (int w = 0; w < NW; ++w) { sq += part[0][w][tid]; fm += part[1][w][tid]; }
            const size_t o = ((size_t)b * a.ng + g) * a.Tp + t0 + tid;
            if (a.rowsq) a.rowsq[o] = sq;
            if (a.fmean) a.fmean[o] = fm;
        }
    }
}

void launch_project(hipStream_t stream, const ProjectArgs &a) {
    dim3 grid(a.Tp / STRIP, a.ng, a.nb);
    // 16 wavefronts per workgroup (4 per SIMD, 2 column tiles each) measured 4.1-4.2 ms against 4.6-4.7 ms for
    // 8 wavefronts with 4 tiles each on config 2: one fp64 MFMA wavefront fills at most half a SIMD's matrix pipe.
    if (a.kind == 0) hipLaunchKernelGGL((project_kernel<0, 16>), grid, dim3(1024), 0, stream, a);
    else hipLaunchKernelGGL((project_kernel<1, 16>), grid, dim3(1024), 0, stream, a);
}

// ---------------------------------------------------------------------------------------------
// LinearK, explicit-U branch, forward only: the projection through the kernel's rank.  K_fu = sigma^2 X Z^T (kernels.py:276) has
// rank P = D + C, so F = K_fu L^-T = sigma^2 X C with C = Z^T L^-T (P x Mp), and what the branch reads of F
// (conditionals_multi_output.py:44-52, dgp_model.py:346-351) are two forms per row:
//     fmean_t = F_t u = sigma^2 x_t . v,   v = C u (P);       sum_j F_tj^2 = sigma^4 x_t^T G x_t,   G = C C^T (P x P).
// T M^2 -> T P^2 flops per unit (BASELINE configs[4]: 0.50 ms of projection -> two launches of a few microseconds).  Same values as
// the M-wide route up to summation order; C, v, G are bounded by construction (C C^T = Z^T K^-1 Z), so nothing cancels.
// C comes out of the K_uu chain itself: its launch carries ONE 64-row extension block holding Z^T (kuu_build_kernel, zt_rows) instead
// of the M identity rows that become L^-T -- extension rows X end as X L^-T.  linear_gv: partial sums of G and v per (dim, 64 columns);
// linear_rows: one thread per row t, every workgroup first adds the partials.
// ---------------------------------------------------------------------------------------------
constexpr int LRP = 32;             // bound on P for this path (register arrays; <= 64 rows of one extra block); larger P takes project_kernel
__global__ __launch_bounds__(256) void linear_gv_kernel(const double *Crows, size_t c_stride, const double *U, int u_ld, int d_begin,
                                                        int M, int Mp, int P, double *part /*[Dl][nblk][P*P + P]*/) {
    __shared__ double cs[LRP][64];                       // C[:, this block of 64 columns]
    const int jb = blockIdx.x, dl = blockIdx.y, tid = threadIdx.x, nblk = gridDim.x;
    const double *C = Crows + (size_t)dl * c_stride;     // row p at C + p * Mp: C = Z^T L^-T out of the K_uu chain's launch
    for (int e = tid; e < P * 64; e += 256) cs[e >> 6][e & 63] = C[(size_t)(e >> 6) * Mp + jb * 64 + (e & 63)];
    __syncthreads();
    double *o = part + ((size_t)dl * nblk + jb) * ((size_t)P * P + P);
    for (int e = tid; e < P * P + P; e += 256) {
        double v = 0.0;
        if (e < P * P) {
            const double *a = cs[e / P], *b = cs[e % P];
#pragma unroll 8
            for (int c = 0; c < 64; ++c) v = fma(a[c], b[c], v);
        } else {
            const double *a = cs[e - P * P];
            for (int c = 0; c < 64; ++c) {
                const int jj = jb * 64 + c;
                const double u = (jj < M) ? U[(size_t)jj * u_ld + d_begin + dl] : 0.0;
                v = fma(a[c], u, v);
            }
        }
        o[e] = v;
    }
}
__global__ __launch_bounds__(256) void linear_rows_kernel(ProjectArgs a, const double *part, int nblk) {
    extern __shared__ double gv[];                       // G [P][P] | v [P]
    const int bz = blockIdx.y, b = a.b0 + bz, s = b / a.Dl, dl = b % a.Dl, tid = threadIdx.x, P = a.P;
    const int ne = P * P + P;
    for (int e = tid; e < ne; e += 256) {
        double v = 0.0;
        for (int k = 0; k < nblk; ++k) v += part[((size_t)dl * nblk + k) * ne + e];
        gv[e] = v;
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + tid;
    if (t >= a.Tp) return;
    double rs = 0.0, fm = 0.0;
    if (t < a.T) {
        double x[LRP];
        const double *xr = a.x + (size_t)s * a.x_chain_stride + (size_t)t * a.x_ld;
#pragma unroll
        for (int p = 0; p < LRP; ++p) {
            x[p] = 0.0;
            if (p < a.x_cols) x[p] = xr[p];
            else if (p < P) x[p] = a.ctrl[(size_t)t * a.C + (p - a.x_cols)];
        }
        double q = 0.0, m = 0.0;
#pragma unroll
        for (int p = 0; p < LRP; ++p) {
            if (p < P) {
                double y = 0.0;
#pragma unroll
                for (int r = 0; r < LRP; ++r)
                    if (r < P) y = fma(gv[p * P + r], x[r], y);
                q = fma(x[p], y, q);
                m = fma(x[p], gv[P * P + p], m);
            }
        }
        const double var = a.hv.variance[dl];
        rs = var * var * q;
        fm = var * m;
    }
    const size_t o = (size_t)b * a.Tp + t;              // one column group (ng = 1 for this path)
    if (a.rowsq) a.rowsq[o] = rs;
    if (a.fmean) a.fmean[o] = fm;
}
bool linear_lowrank_supported(int kind, int P) { return kind == 1 && P <= LRP; }
size_t linear_lowrank_doubles(int Mp, int Dl, int P) { return (size_t)Dl * (Mp / 64) * ((size_t)P * P + P); }
void launch_linear_lowrank(hipStream_t stream, const ProjectArgs &a, double *part) {
    const int nblk = a.Mp / 64;
    if (a.b0 == 0)          // G, v depend on the dim only: once per iteration (the first pass)
        hipLaunchKernelGGL(linear_gv_kernel, dim3(nblk, a.Dl), dim3(256), 0, stream, a.W, a.w_stride, a.U, a.u_ld, a.d_begin, a.M, a.Mp,
                           a.P, part);
    hipLaunchKernelGGL(linear_rows_kernel, dim3((a.Tp + 255) / 256, a.nb), dim3(256), ((size_t)a.P * a.P + a.P) * sizeof(double), stream, a,
                       part, nblk);
}

// ---------------------------------------------------------------------------------------------
// Gram:  C = A^T A over the rows of A (T x Mp), lower-triangular tiles only, batched.
//   mode GRAM_F   : H = F^T F * (Y_N / (batch Q_d)) + I            (conditionals_multi_output.py:246)
//   mode GRAM_KFU : A_d = K_uf K_fu * (Y_N / (batch Q_d)) + (K_uu + jitter I)     (collapsed bound in the
//                   K_uu + K_uf K_fu / Q form) plus the partial sums of tr(K^-1 K_uf K_fu)
//   mode GRAM_PLAIN: C = A^T A (used for K^-1 = L^-T L^-1)
//   with_row != 0 : extra row Mp = delta^T A * (Y_N / (batch Q_d))                 (conditionals_multi_output.py:247-248)
// Workgroup = 128 x 128 output tile of the lower triangle, 8 wavefronts of 64 x 32 (8 accumulator tiles = 64
// VGPRs), two workgroups per CU = four resident wavefronts per SIMD -- one fp64 MFMA wavefront can use at most
// about half of a SIMD's matrix pipe (tools/mfma_probe2), so the pipe only saturates with several MFMA-ready
// wavefronts per SIMD, and a second workgroup covers the first one's barrier / staging bubbles.
// (Measured alternatives at M = 512, T = 4096, 128 units: 4 wavefronts of 64x64 3.45 ms; 16 wavefronts of 32x32,
//  one workgroup per CU 3.36 ms; 128x64 tiles, 8 wavefronts of 32x32 3.27 ms; this layout 2.98 ms.)
// ---------------------------------------------------------------------------------------------
#ifndef GRAM_GLDS_OFFDIAG
#define GRAM_GLDS_OFFDIAG 0         // 1: off-diagonal tiles staged by LDS-DMA too -- measured neutral (2.24 vs 2.25 ms, profiles/r04_ab_gram_offdiag_glds.txt)
#endif
#ifndef GRAM_COMBO
#define GRAM_COMBO 2                // 0: none, 1: three 64 x 32-sub-block workgroups per four diagonal tiles (round 3), 2: pair combos (round 4)
#endif
constexpr int GT = 16;              // rows of A per LDS chunk
constexpr int G_LD = 128 + 16;      // LDS row stride (doubles): lanes l and l+16 land in different bank halves

// 64x32 sub-block (row half, column quarter) of each wavefront.  Diagonal tiles only need the 6 sub-blocks that
// touch the lower triangle; they go to wavefronts 0..5 (SIMDs 0,1 get two, SIMDs 2,3 one); 6 and 7 do the gemv.
__device__ __constant__ unsigned char GRAM_DIAG_WR[8] = {1, 1, 1, 1, 0, 0, 0, 0};
__device__ __constant__ unsigned char GRAM_DIAG_WC[8] = {0, 1, 2, 3, 0, 1, 2, 3};

template <int N>
__device__ __forceinline__ bool gram_tail_exchange(const GramArgs &a, const int tail_id, const int half, const bool active,
                                                   d4 (&acc)[N], double &bs0, double &bs1, int *tail_slot);

template <int MODE, bool DIAG, bool GLDS = false>
__device__ __forceinline__ void gram_body(const GramArgs a, const int bz, const int ti, const int tj, const int tile,
                                          const int kpart, const int ksplit, double (*As)[GT][G_LD],
                                          double (*Bs)[GT][G_LD], double (*dls)[GT], double *red, const int tail_id = -1,
                                          int *tail_slot = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = DIAG ? GRAM_DIAG_WR[wave] : (wave >> 2), wc = DIAG ? GRAM_DIAG_WC[wave] : (wave & 3);
    const int lr = lane & 15, lk = lane >> 4;
    const int Mp = a.Mp;
    const int b = a.b0 + bz, s = b / a.Dl, dl = b % a.Dl, dg = a.d_begin + dl;
    const int I0 = ti * 128 + wr * 64, J0 = tj * 128 + wc * 32;
    const bool active = (I0 < Mp) && (J0 < Mp) && (J0 < I0 + 64);
    const bool gemv = DIAG && a.with_row && tid >= 384;          // wavefronts 6, 7: no sub-block in diagonal tiles

    const double *Ab = a.A + (size_t)bz * a.a_stride;
    const double *Xs = (a.with_row && !(MODE == GRAM_PLAIN && a.rvec)) ? a.X + (size_t)s * (a.T + 1) * a.D : nullptr;
    // global -> register staging: thread (rowl, lane) moves 16 bytes of rows rowl and rowl + 8 per operand.
    // Out-of-range columns (odd number of 64-blocks) are read from a clamped valid address and zeroed.
    const int colA = ti * 128 + 2 * lane, colB = tj * 128 + 2 * lane;
    const bool okA = colA < Mp, okB = colB < Mp;
    const int colAc = okA ? colA : 0, colBc = okB ? colB : 0;
    const int rowl = tid >> 6;   // 0..7

    // The loaded registers are only touched again in lstore() (masking included), so the global loads of chunk
    // c + 1 stay in flight behind the whole MFMA phase of chunk c.
    double2 ra[2], rb[2];
    double d1 = 0.0, d0 = 0.0;
    bool dok = false;
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t t = (size_t)c * GT + rowl + 8 * i;
            ra[i] = *reinterpret_cast<const double2 *>(Ab + t * Mp + colAc);
            if (!DIAG) rb[i] = *reinterpret_cast<const double2 *>(Ab + t * Mp + colBc);
        }
        if (DIAG && a.with_row) {
            const int tt = c * GT + (tid & (GT - 1));
            if (MODE == GRAM_PLAIN && a.rvec) {                // a caller-supplied vector (residuals, backward pass)
                d1 = a.rvec[(size_t)bz * a.rows + tt];
                d0 = 0.0;
                dok = true;
            } else {
                const int tc = tt < a.T ? tt : a.T - 1;
                d1 = Xs[(size_t)(tc + 1) * a.D + dg];
                d0 = Xs[(size_t)tc * a.D + dg];
                dok = tt < a.T;
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double2 va = ra[i];
            va.x = okA ? va.x : 0.0;
            va.y = okA ? va.y : 0.0;
            *reinterpret_cast<double2 *>(&As[buf][rowl + 8 * i][2 * lane]) = va;
            if (!DIAG) {
                double2 vb = rb[i];
                vb.x = okB ? vb.x : 0.0;
                vb.y = okB ? vb.y : 0.0;
                *reinterpret_cast<double2 *>(&Bs[buf][rowl + 8 * i][2 * lane]) = vb;
            }
        }
        if (DIAG && a.with_row && tid < GT) dls[buf][tid] = dok ? d1 - d0 : 0.0;     // :247
    };

    d4 acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    double bsum = 0.0;

    const int nchunk_all = a.rows / GT;
    const int nrange = (tail_id >= 0) ? 2 : ksplit;                   // a tail workgroup is one of two row halves (kpart = the half)
    const int per = (nchunk_all + nrange - 1) / nrange;
    const int cbeg = kpart * per;
    const int nchunk = (cbeg + per <= nchunk_all) ? cbeg + per : nchunk_all;      // this range: chunks [cbeg, nchunk)
    // GLDS (off-diagonal tiles of launches whose panels are whole: Mp a multiple of 128): the panels go global -> LDS by DMA, as in the
    // pair combos (gram_pair_role has the notes), no staging registers and no ds_write pass
    const unsigned voffA = (unsigned)((rowl * Mp + ti * 128 + 2 * lane) * (int)sizeof(double));
    const unsigned voffB = (unsigned)((rowl * Mp + tj * 128 + 2 * lane) * (int)sizeof(double));
    auto glds = [&](const double *base, const unsigned voff, const void *lds_row) {
        typedef __attribute__((address_space(3))) void lvoid;
        const unsigned dst = (unsigned)(uintptr_t)(lvoid *)lds_row;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
    };
    auto dma = [&](int c, int buf) {
        const double *base0 = Ab + (size_t)c * GT * Mp, *base1 = base0 + (size_t)8 * Mp;
        glds(base0, voffA, &As[buf][wave][0]);
        glds(base0, voffB, &Bs[buf][wave][0]);
        glds(base1, voffA, &As[buf][wave + 8][0]);
        glds(base1, voffB, &Bs[buf][wave + 8][0]);
    };
    if (GLDS) {
        dma(cbeg, cbeg & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        gload(cbeg);
        lstore(cbeg & 1);
    }
    __syncthreads();
    for (int c = cbeg; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) { if (GLDS) dma(c + 1, buf ^ 1); else gload(c + 1); }
        if (active) {
            const double(*Bp)[G_LD] = DIAG ? As[buf] : Bs[buf];
            double af[4], bf[2], afn[4], bfn[2];
#pragma unroll
            for (int x = 0; x < 4; ++x) af[x] = As[buf][lk][wr * 64 + 16 * x + lr];
#pragma unroll
            for (int y = 0; y < 2; ++y) bf[y] = Bp[lk][wc * 32 + 16 * y + lr];
#pragma unroll
            for (int ks = 0; ks < GT / 4; ++ks) {
                if (ks + 1 < GT / 4) {       // fragments of the next k-step are requested before this step's MFMAs
#pragma unroll
                    for (int x = 0; x < 4; ++x) afn[x] = As[buf][4 * (ks + 1) + lk][wr * 64 + 16 * x + lr];
#pragma unroll
                    for (int y = 0; y < 2; ++y) bfn[y] = Bp[4 * (ks + 1) + lk][wc * 32 + 16 * y + lr];
                }
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = mfma_f64(af[x], bf[y], acc[x][y]);
#pragma unroll
                for (int x = 0; x < 4; ++x) af[x] = afn[x];
#pragma unroll
                for (int y = 0; y < 2; ++y) bf[y] = bfn[y];
            }
        }
        if (gemv) {
#pragma unroll
            for (int r = 0; r < GT; ++r) bsum += As[buf][r][tid - 384] * dls[buf][r];
        }
        if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next chunk has landed (the DMAs are invisible to hipcc's counting)
        else if (c + 1 < nchunk) lstore(buf ^ 1);
        __syncthreads();
    }

    if (tail_id >= 0) {
        double unused = 0.0;
        if (!gram_tail_exchange(a, tail_id, kpart, active, reinterpret_cast<d4(&)[8]>(acc), bsum, unused, tail_slot)) return;
    }
    if (ksplit > 1) {            // raw partial sums of this row range; gram_combine finishes the job
        double *Pb = a.part + ((size_t)kpart * a.nb + bz) * ((size_t)(Mp + 1) * Mp);
        if (active) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        Pb[(size_t)(I0 + 16 * x + lk + 4 * q) * Mp + J0 + 16 * y + lr] = acc[x][y][q];
        }
        if (gemv) {
            const int col = ti * 128 + tid - 384;
            if (col < Mp) Pb[(size_t)Mp * Mp + col] = bsum;
        }
        return;
    }
    const double scale = (MODE == GRAM_PLAIN) ? 1.0 : a.yn_over_batch / exp(a.log_Q[dg]);
    double *Hb = a.H + (size_t)bz * a.h_stride;
    const double *Kadd = (MODE == GRAM_KFU || MODE == GRAM_KFU_RAW) ? a.Kadd + (size_t)dl * a.kadd_stride : nullptr;
    const double *Kinv = (MODE == GRAM_KFU) ? a.Kinv + (size_t)dl * a.kinv_stride : nullptr;
    double *Rb = (MODE == GRAM_KFU_RAW) ? a.part + (size_t)bz * ((size_t)(Mp + 1) * Mp) : nullptr;
    double *Cb2 = ((MODE == GRAM_KFU_RAW || MODE == GRAM_KFU) && a.Hcopy) ? a.Hcopy + (size_t)bz * a.hcopy_stride : nullptr;
    double trp = 0.0;
    if (active) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = I0 + 16 * x + lk + 4 * q, j = J0 + 16 * y + lr;
                    const double g = acc[x][y][q];
                    double v;
                    if (MODE == GRAM_F) v = g * scale + ((i == j) ? 1.0 : 0.0);
                    else if (MODE == GRAM_KFU_RAW) {
                        v = g * scale + Kadd[(size_t)i * Mp + j];
                        Rb[(size_t)i * Mp + j] = g;
                        if (Cb2 && j <= i) {                    // the copy holds the lower triangle (its readers sum over the chains first and mirror the sum: a mirror here was 64 cache lines per store)
                            Cb2[(size_t)i * Mp + j] = v;
                        }
                    } else if (MODE == GRAM_KFU) {
                        v = g * scale + Kadd[(size_t)i * Mp + j];
                        if (Cb2 && j <= i) {                    // second copy, lower triangle (the backward pass keeps A)
                            Cb2[(size_t)i * Mp + j] = v;
                        }
                        const double w = (i > j) ? 2.0 : ((i == j) ? 1.0 : 0.0);
                        trp += w * (Kinv[(size_t)i * Mp + j] * g);
                    } else v = g;
                    Hb[(size_t)i * Mp + j] = v;
                }
    }
    if (gemv) {
        const int col = ti * 128 + tid - 384;
        if (col < Mp) Hb[(size_t)a.brow * Mp + col] = bsum * scale;
    }
    if (MODE == GRAM_KFU) {      // deterministic workgroup reduction of the trace partial
        red[tid] = trp;
        __syncthreads();
        for (int st = 256; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) a.trpart[(size_t)b * a.ntiles + tile] = red[0];
    }
}

// Row half of a tail workgroup (GramArgs::tail_wg): leave the accumulators in memory, count in, and -- if the other half has
// counted in before -- add its values and carry on (returns true); else done (returns false).  Nobody waits.  The blocks are
// stored WRITE-THROUGH (sc1, relaxed agent-scope atomic stores) so that no release fence is needed -- an agent-scope release
// writes back the XCD L2's dirty lines, which with 135 KB per half and the chip streaming K_fu made a half last twice its time;
// the reader's ONE acquire invalidates its CU's L1 and then loads plainly (MI355X guide, inter-workgroup visibility: sc1 payload
// stores, every storing wavefront's vmcnt(0), the workgroup's barrier, one lane's relaxed agent-scope add).
// TARGET-SPECIFIC hand-off: the partial block is published with relaxed agent-scope stores + s_waitcnt vmcnt(0) + barrier +
// a relaxed fetch_add, with NO release fence (an agent-scope release writes back the XCD L2's dirty lines and doubled the time of
// a half).  That is sound on gfx942 / gfx950 only: stores are counted in vmcnt, the atomic stores are emitted
// with sc1 (write-through to memory-side coherence) and a CU's L1 is not shared across workgroups' hand-off.  Targets that track
// stores in vscnt (gfx10+) or another compiler lowering of the atomic stores would break it silently -- so the build refuses them.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "gram_tail_exchange relies on gfx942/gfx950 store semantics (vmcnt-counted sc1 write-through stores); build for gfx950"
#endif
template <int N>
__device__ __forceinline__ bool gram_tail_exchange(const GramArgs &a, const int tail_id, const int half, const bool active,
                                                   d4 (&acc)[N], double &bs0, double &bs1, int *tail_slot) {
    static_assert((N * 4 + 2) * 512 <= GRAM_TAIL_DOUBLES, "tail block too small");
    typedef __attribute__((address_space(1))) double gdouble;
    const int tid = threadIdx.x;
    gdouble *Pm = (gdouble *)(a.tail_part + ((size_t)tail_id * 2 + half) * GRAM_TAIL_DOUBLES);
    const double *Po = a.tail_part + ((size_t)tail_id * 2 + (half ^ 1)) * GRAM_TAIL_DOUBLES;
    if (active) {
#pragma unroll
        for (int x = 0; x < N; ++x)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __hip_atomic_store(Pm + (size_t)(x * 4 + q) * 512 + tid, acc[x][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(Pm + (size_t)(4 * N) * 512 + tid, bs0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(Pm + (size_t)(4 * N + 1) * 512 + tid, bs1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wavefront's block has left it
    __syncthreads();
    if (tid == 0) {
        const int seen = __hip_atomic_fetch_add(a.tail_cnt + tail_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == 1) {                                         // the other half is complete and in memory
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(a.tail_cnt + tail_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
        }
        *tail_slot = seen;
    }
    __syncthreads();
    if (*tail_slot != 1) return false;
    // (a + b is the same number whichever half is `a`: the result does not depend on the order of arrival)
    if (active) {
#pragma unroll
        for (int x = 0; x < N; ++x)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[x][q] += Po[(size_t)(x * 4 + q) * 512 + tid];
    }
    bs0 += Po[(size_t)(4 * N) * 512 + tid];
    bs1 += Po[(size_t)(4 * N + 1) * 512 + tid];
    return true;
}

// "Combo" workgroups: the 24 sub-blocks (64 x 32) of the four diagonal tiles of a group of four column panels b .. b + 3, dealt
// to three workgroups of eight -- every wavefront a full sub-block, every SIMD two matrix wavefronts, as in an off-diagonal tile:
//   type 0 (panels b, b+1 in LDS):     tile (b,b) complete (6)                + rows 0-63 of tile (b+1,b+1) (2)
//   type 1 (panels b+1, b+2):          rows 64-127 of tile (b+1,b+1) (4)      + rows 64-127 of tile (b+2,b+2) (4)
//   type 2 (panels b+2, b+3):          rows 0-63 of tile (b+2,b+2) (2)        + tile (b+3,b+3) complete (6)
// (gram_body<.., true> gives a diagonal tile's six sub-blocks to six wavefronts: SIMDs 0 and 1 carry two matrix wavefronts, SIMDs
// 2 and 3 one, the tile lasts as long as an off-diagonal one -- 1085 vs 1091 us, tools/gram_rounds.py -- and what SIMDs 2 and 3
// have to spare nobody can use.)  Combos form no delta^T A row -- there is no idle wavefront for it, and vector FMAs beside the
// matrix work cost 0.47 ms at config 2: launch_gram uses them only for launches with with_row = 0 (the Gram route sums
// delta^T K_fu in the K_fu build, kfu_build_kernel / brow_finish_kernel).
// entry = buffer (0 = first panel, 1 = second) << 3 | row half << 2 | column quarter
__device__ __constant__ unsigned char GRAM_COMBO_ROLE[3][8] = {
    {0 << 3 | 1 << 2 | 0, 0 << 3 | 1 << 2 | 1, 0 << 3 | 1 << 2 | 2, 0 << 3 | 1 << 2 | 3, 0 << 3 | 0 << 2 | 0, 0 << 3 | 0 << 2 | 1, 1 << 3 | 0 << 2 | 0, 1 << 3 | 0 << 2 | 1},
    {0 << 3 | 1 << 2 | 0, 0 << 3 | 1 << 2 | 1, 0 << 3 | 1 << 2 | 2, 0 << 3 | 1 << 2 | 3, 1 << 3 | 1 << 2 | 0, 1 << 3 | 1 << 2 | 1, 1 << 3 | 1 << 2 | 2, 1 << 3 | 1 << 2 | 3},
    {0 << 3 | 0 << 2 | 0, 0 << 3 | 0 << 2 | 1, 1 << 3 | 1 << 2 | 0, 1 << 3 | 1 << 2 | 1, 1 << 3 | 1 << 2 | 2, 1 << 3 | 1 << 2 | 3, 1 << 3 | 0 << 2 | 0, 1 << 3 | 0 << 2 | 1}};

template <int MODE>
__device__ __forceinline__ void gram_combo_body(const GramArgs a, const int bz, const int pbase, const int type, const int tail_id,
                                                const int half, double (*As)[GT][G_LD], double (*Bs)[GT][G_LD],
                                                double *red, int *tail_slot) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int Mp = a.Mp;
    const int b = a.b0 + bz, s = b / a.Dl, dl = b % a.Dl, dg = a.d_begin + dl;
    const int pa = pbase + type, pb = pbase + type + 1;                  // the two panels this workgroup stages
    const int role = __builtin_amdgcn_readfirstlane((int)GRAM_COMBO_ROLE[type][wave]);
    const int rbuf = role >> 3, rh = (role >> 2) & 1, rq = role & 3;
    const int panel = rbuf ? pb : pa;
    const int I0 = panel * 128 + rh * 64, J0 = panel * 128 + rq * 32;   // Mp % 512 == 0: every sub-block is real
    const int r0 = rh * 64, c0 = rq * 32;                               // offsets inside the staged panel

    const double *Ab = a.A + (size_t)bz * a.a_stride;
    const int colA = pa * 128 + 2 * lane, colB = pb * 128 + 2 * lane;
    const int rowl = tid >> 6;   // 0..7
    double2 ra0, ra1, rb0, rb1;
    auto gload = [&](int c) {
        const double *row0 = Ab + ((size_t)c * GT + rowl) * Mp, *row1 = row0 + (size_t)8 * Mp;
        ra0 = *reinterpret_cast<const double2 *>(row0 + colA);
        rb0 = *reinterpret_cast<const double2 *>(row0 + colB);
        ra1 = *reinterpret_cast<const double2 *>(row1 + colA);
        rb1 = *reinterpret_cast<const double2 *>(row1 + colB);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<double2 *>(&As[buf][rowl][2 * lane]) = ra0;
        *reinterpret_cast<double2 *>(&Bs[buf][rowl][2 * lane]) = rb0;
        *reinterpret_cast<double2 *>(&As[buf][rowl + 8][2 * lane]) = ra1;
        *reinterpret_cast<double2 *>(&Bs[buf][rowl + 8][2 * lane]) = rb1;
    };

    d4 acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    const int nchunk_all = a.rows / GT;
    const int nrange = (tail_id >= 0) ? 2 : 1;
    const int per = (nchunk_all + nrange - 1) / nrange;
    const int cbeg = half * per;
    const int nchunk = (cbeg + per <= nchunk_all) ? cbeg + per : nchunk_all;
    gload(cbeg);
    lstore(cbeg & 1);
    __syncthreads();
    for (int c = cbeg; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload(c + 1);
        {
            const double(*Sp)[G_LD] = rbuf ? Bs[buf] : As[buf];          // this wavefront's panel holds both of its operands
            double af[4], bf[2], afn[4], bfn[2];
#pragma unroll
            for (int x = 0; x < 4; ++x) af[x] = Sp[lk][r0 + 16 * x + lr];
#pragma unroll
            for (int y = 0; y < 2; ++y) bf[y] = Sp[lk][c0 + 16 * y + lr];
#pragma unroll
            for (int ks = 0; ks < GT / 4; ++ks) {
                if (ks + 1 < GT / 4) {       // fragments of the next k-step are requested before this step's MFMAs
#pragma unroll
                    for (int x = 0; x < 4; ++x) afn[x] = Sp[4 * (ks + 1) + lk][r0 + 16 * x + lr];
#pragma unroll
                    for (int y = 0; y < 2; ++y) bfn[y] = Sp[4 * (ks + 1) + lk][c0 + 16 * y + lr];
                }
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = mfma_f64(af[x], bf[y], acc[x][y]);
#pragma unroll
                for (int x = 0; x < 4; ++x) af[x] = afn[x];
#pragma unroll
                for (int y = 0; y < 2; ++y) bf[y] = bfn[y];
            }
        }
        if (c + 1 < nchunk) lstore(buf ^ 1);
        __syncthreads();
    }
    double bs0 = 0.0, bs1 = 0.0;           // (no delta^T A row here: launch_gram gives combos only to launches without one)
    if (tail_id >= 0 && !gram_tail_exchange(a, tail_id, half, true, reinterpret_cast<d4(&)[8]>(acc), bs0, bs1, tail_slot)) return;

    const double scale = (MODE == GRAM_PLAIN) ? 1.0 : a.yn_over_batch / exp(a.log_Q[dg]);
    double *Hb = a.H + (size_t)bz * a.h_stride;
    const double *Kadd = (MODE == GRAM_KFU || MODE == GRAM_KFU_RAW) ? a.Kadd + (size_t)dl * a.kadd_stride : nullptr;
    const double *Kinv = (MODE == GRAM_KFU) ? a.Kinv + (size_t)dl * a.kinv_stride : nullptr;
    double *Rb = (MODE == GRAM_KFU_RAW) ? a.part + (size_t)bz * ((size_t)(Mp + 1) * Mp) : nullptr;
    double *Cb2 = ((MODE == GRAM_KFU_RAW || MODE == GRAM_KFU) && a.Hcopy) ? a.Hcopy + (size_t)bz * a.hcopy_stride : nullptr;
    double trp = 0.0;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = I0 + 16 * x + lk + 4 * q, j = J0 + 16 * y + lr;
                const double g = acc[x][y][q];
                double v;
                if (MODE == GRAM_F) v = g * scale + ((i == j) ? 1.0 : 0.0);
                else if (MODE == GRAM_KFU_RAW) {
                    v = g * scale + Kadd[(size_t)i * Mp + j];
                    Rb[(size_t)i * Mp + j] = g;
                    if (Cb2 && j <= i) {                    // the copy holds the lower triangle (its readers sum over the chains first and mirror the sum: a mirror here was 64 cache lines per store)
                        Cb2[(size_t)i * Mp + j] = v;
                    }
                } else if (MODE == GRAM_KFU) {
                    v = g * scale + Kadd[(size_t)i * Mp + j];
                    if (Cb2 && j <= i) {
                        Cb2[(size_t)i * Mp + j] = v;
                    }
                    const double w = (i > j) ? 2.0 : ((i == j) ? 1.0 : 0.0);
                    trp += w * (Kinv[(size_t)i * Mp + j] * g);
                } else v = g;
                Hb[(size_t)i * Mp + j] = v;
            }
    if (MODE == GRAM_KFU) {      // deterministic workgroup reduction of the trace partial; slot of the first panel's diagonal tile
        red[tid] = trp;
        __syncthreads();
        for (int st = 256; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) {
            a.trpart[(size_t)b * a.ntiles + pa * (pa + 1) / 2 + pa] = red[0];
            if (type == 2) a.trpart[(size_t)b * a.ntiles + pb * (pb + 1) / 2 + pb] = 0.0;     // (three workgroups, four slots)
        }
    }
}


// "Pair" combos (round 4): 16-granular triangle.  A diagonal 128-tile has 8 x 9 / 2 = 36 MFMA tiles on or below its diagonal; row
// block i (16 rows) owns i + 1 of them, so the row blocks i and 7 - i together own NINE whatever i -- four wavefronts per diagonal
// tile, each with nine MFMAs per k-step, and no MFMA tile above the diagonal is executed.  A workgroup takes the diagonal tiles of
// two neighbouring panels (both staged as before); it lasts 9 / 8 of an off-diagonal tile, and the four panels of M = 512 cost
// 2 x 1.125 workgroup-times instead of 3: executed flops 1.03 x the triangle's instead of 1.125 x.  Operands: on a diagonal tile
// the A fragment of row block i IS the B fragment of column block i, so role R reads the 8 - R column fragments 0 .. 7 - R and
// nothing else (8, 7, 6, 5 fragment reads for 9 MFMAs; the 64 x 32 sub-blocks read 6 for 8).  Earlier attempts at the triangle
// (1 x 4 row roles with 5 reads per 4 MFMAs, evenly dealt tiles) lost to their read / MFMA ratio.
// The unwritten above-diagonal MFMA tiles INSIDE the diagonal 64-blocks are mirrored from below in the epilogue (bit for bit what
// the sub-blocks used to compute there: the same products in the same order), so the buffers look as they always did.
template <int R>
__device__ __forceinline__ void gram_pair_ksteps(const double (*Sp)[G_LD], const int lr, const int lk, d4 (&acc)[9]) {
    // Registers: nine accumulator tiles are 72 of the 128 VGPRs, so the next k-step's fragments cannot have an array of their own
    // (the 64 x 32 sub-blocks keep 6 + 6): column fragment c is re-loaded into its own register right after its last MFMA of this
    // step, and only the two row operands -- f[R], f[7 - R], needed by every MFMA of the step -- are held in copies.
    constexpr int NF = 8 - R;
    double f[NF];
#pragma unroll
    for (int c = 0; c < NF; ++c) f[c] = Sp[lk][16 * c + lr];
    double aR = f[R], aS = f[7 - R];
#pragma unroll
    for (int ks = 0; ks < GT / 4; ++ks) {
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            acc[R + 1 + c] = mfma_f64(aS, f[c], acc[R + 1 + c]);             // row block 7 - R, columns 0 .. 7 - R
            if (c <= R) acc[c] = mfma_f64(aR, f[c], acc[c]);                 // row block R, columns 0 .. R
            if (ks + 1 < GT / 4) f[c] = Sp[4 * (ks + 1) + lk][16 * c + lr];
            __builtin_amdgcn_sched_barrier(0);          // (keep the re-load behind the MFMAs that free its register: hoisted loads need eight more)
        }
        if (ks + 1 < GT / 4) { aR = f[R]; aS = f[7 - R]; }
    }
}

// (the whole body is instantiated per role: with the role switch inside the chunk loop the nine accumulator tiles crossed a four-way
//  merge every chunk and the register allocator kept them in scratch)
template <int MODE, int R>
__device__ __forceinline__ void gram_pair_role(const GramArgs a, const int bz, const int pa, const int tail_id,
                                               const int half /* row range: tail half or split-K part */, const int ksplit,
                                               double (*As)[GT][G_LD], double (*Bs)[GT][G_LD], double *red, int *tail_slot) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int Mp = a.Mp;
    const int b = a.b0 + bz, dl = b % a.Dl, dg = a.d_begin + dl;
    const int pb = pa + 1;                                               // the two panels this workgroup stages
    const int rbuf = wave >> 2;                                          // wavefronts w and w + 4 share a SIMD: the same role R = w & 3 on both panels
    const int P0 = (rbuf ? pb : pa) * 128;

    const double *Ab = a.A + (size_t)bz * a.a_stride;
    const int rowl = tid >> 6;   // 0..7
    // Staging: LDS-DMA (global_load_lds_dwordx4: one wavefront-instruction = one 1 KiB row of a panel, lane l -> bytes 16 l of the
    // row), no staging registers -- nine accumulator tiles + eight fragments leave no room for the sixteen the register-staged
    // bodies hold across the MFMA phase.  Uniform 64-bit bases per chunk (SGPRs) + ONE 32-bit byte offset per thread.  Written as asm: hipcc waits vmcnt(0) in front of the first ds_read that follows a DMA it
    // knows about (it cannot tell the two buffers apart), i.e. before the MFMA phase the DMA is meant to run behind; the loads
    // it does not count are waited for by hand in front of the barrier.
    const unsigned voff = (unsigned)((rowl * Mp + pa * 128 + 2 * lane) * (int)sizeof(double));
    typedef __attribute__((address_space(3))) void lvoid;
    auto glds = [&](const double *base, const void *lds_row) {
        const unsigned dst = (unsigned)(uintptr_t)(lvoid *)lds_row;
        unsigned keep;
        // (no immediate offset: the instruction adds it to the LDS address as well as to the global one)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
    };
    auto gload = [&](int c, int buf) {
        const double *base0 = Ab + (size_t)c * GT * Mp, *base1 = base0 + (size_t)8 * Mp;
        glds(base0, &As[buf][wave][0]);
        glds(base0 + 128, &Bs[buf][wave][0]);
        glds(base1, &As[buf][wave + 8][0]);
        glds(base1 + 128, &Bs[buf][wave + 8][0]);
    };

    d4 acc[9];
#pragma unroll
    for (int x = 0; x < 9; ++x) acc[x] = (d4){0.0, 0.0, 0.0, 0.0};
    const int nchunk_all = a.rows / GT;
    const int nrange = (tail_id >= 0) ? 2 : ksplit;
    const int per = (nchunk_all + nrange - 1) / nrange;
    const int cbeg = half * per;
    const int nchunk = (cbeg + per <= nchunk_all) ? cbeg + per : nchunk_all;
    gload(cbeg, cbeg & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = cbeg; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload(c + 1, buf ^ 1);                       // the buffer nobody reads until the barrier below
        {
            const double(*Sp)[G_LD] = rbuf ? Bs[buf] : As[buf];          // this wavefront's panel holds both of its operands
            gram_pair_ksteps<R>(Sp, lr, lk, acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the next chunk has landed (the DMAs are invisible to hipcc's counting)
        __syncthreads();
    }
    double bs0 = 0.0, bs1 = 0.0;           // (no delta^T A row here: launch_gram gives combos only to launches without one)
    if (tail_id >= 0 && !gram_tail_exchange(a, tail_id, half, true, acc, bs0, bs1, tail_slot)) return;
    if (ksplit > 1) {            // raw partial sums of this row range; gram_combine finishes the job (it reads whole 64 x 32 sub-blocks:
                                 // the tiles above the diagonal inside the diagonal 64-blocks are mirrored here as well)
        double *Pb = a.part + ((size_t)half * a.nb + bz) * ((size_t)(Mp + 1) * Mp);
#pragma unroll
        for (int x = 0; x < 9; ++x) {
            const int rb = (x <= R) ? R : 7 - R, cb = (x <= R) ? x : x - R - 1;
            const bool mirror = cb < rb && (rb >> 2) == (cb >> 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = P0 + 16 * rb + lk + 4 * q, j = P0 + 16 * cb + lr;
                Pb[(size_t)i * Mp + j] = acc[x][q];
                if (mirror) Pb[(size_t)j * Mp + i] = acc[x][q];
            }
        }
        return;
    }

    const double scale = (MODE == GRAM_PLAIN) ? 1.0 : a.yn_over_batch / exp(a.log_Q[dg]);
    double *Hb = a.H + (size_t)bz * a.h_stride;
    const double *Kadd = (MODE == GRAM_KFU || MODE == GRAM_KFU_RAW) ? a.Kadd + (size_t)dl * a.kadd_stride : nullptr;
    const double *Kinv = (MODE == GRAM_KFU) ? a.Kinv + (size_t)dl * a.kinv_stride : nullptr;
    double *Rb = (MODE == GRAM_KFU_RAW) ? a.part + (size_t)bz * ((size_t)(Mp + 1) * Mp) : nullptr;
    double *Cb2 = ((MODE == GRAM_KFU_RAW || MODE == GRAM_KFU) && a.Hcopy) ? a.Hcopy + (size_t)bz * a.hcopy_stride : nullptr;
    double trp = 0.0;
#pragma unroll
    for (int x = 0; x < 9; ++x) {
        const int rb = (x <= R) ? R : 7 - R, cb = (x <= R) ? x : x - R - 1;        // (accumulator x: row block, column block)
        const bool mirror = cb < rb && (rb >> 2) == (cb >> 2);                     // below the diagonal of a diagonal 64-block
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = P0 + 16 * rb + lk + 4 * q, j = P0 + 16 * cb + lr;
            const double g = acc[x][q];
            double v;
            if (MODE == GRAM_F) v = g * scale + ((i == j) ? 1.0 : 0.0);
            else if (MODE == GRAM_KFU_RAW || MODE == GRAM_KFU) {
                v = g * scale + Kadd[(size_t)i * Mp + j];
                if (MODE == GRAM_KFU_RAW) Rb[(size_t)i * Mp + j] = g;
                if (Cb2 && j <= i) {                    // the copy holds the lower triangle (its readers sum over the chains first and mirror the sum: a mirror here was 64 cache lines per store)
                    Cb2[(size_t)i * Mp + j] = v;
                }
                if (MODE == GRAM_KFU) {
                    const double w = (i > j) ? 2.0 : ((i == j) ? 1.0 : 0.0);
                    trp += w * (Kinv[(size_t)i * Mp + j] * g);
                }
            } else v = g;
            Hb[(size_t)i * Mp + j] = v;
            if (mirror) {
                if (MODE == GRAM_KFU_RAW) Rb[(size_t)j * Mp + i] = g;
                Hb[(size_t)j * Mp + i] = (MODE == GRAM_KFU_RAW || MODE == GRAM_KFU) ? g * scale + Kadd[(size_t)j * Mp + i] : v;
            }
        }
    }
    if (MODE == GRAM_KFU) {      // deterministic workgroup reduction of the trace partial; slot of the first panel's diagonal tile
        red[tid] = trp;
        __syncthreads();
        for (int st = 256; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) {
            a.trpart[(size_t)b * a.ntiles + pa * (pa + 1) / 2 + pa] = red[0];
            a.trpart[(size_t)b * a.ntiles + pb * (pb + 1) / 2 + pb] = 0.0;          // (one workgroup, two slots)
        }
    }
}
template <int MODE>
__device__ __forceinline__ void gram_pair_body(const GramArgs a, const int bz, const int pa, const int tail_id,
                                               const int half, const int ksplit, double (*As)[GT][G_LD], double (*Bs)[GT][G_LD],
                                               double *red, int *tail_slot) {
    const int R = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 3;
    if (R == 0) gram_pair_role<MODE, 0>(a, bz, pa, tail_id, half, ksplit, As, Bs, red, tail_slot);
    else if (R == 1) gram_pair_role<MODE, 1>(a, bz, pa, tail_id, half, ksplit, As, Bs, red, tail_slot);
    else if (R == 2) gram_pair_role<MODE, 2>(a, bz, pa, tail_id, half, ksplit, As, Bs, red, tail_slot);
    else gram_pair_role<MODE, 3>(a, bz, pa, tail_id, half, ksplit, As, Bs, red, tail_slot);
}

template <int MODE>
__global__ __launch_bounds__(512, 4) void gram_kernel(GramArgs a) {
    __shared__ double As[2][GT][G_LD];
    __shared__ double Bs[2][GT][G_LD];
    __shared__ double dls[2][GT];
    __shared__ double red[512];
    // XCD-aware mapping: all tiles of one (chain, dim) share blockIdx % 8, i.e. one XCD's L2 (speed only)
    __shared__ int tail_slot;
    const int id = blockIdx.x;
    const int xcd = id & 7;
    int loc = id >> 3;
    const int ksplit = (a.ksplit > 1 && a.part) ? a.ksplit : 1;
    const int per_unit = a.wg_per_unit;                  // ntiles * ksplit, or (combos) off-diagonal tiles + 3 per group of 4 panels
    // tail split: the last tail_wg / 8 workgroup slots of every XCD's list appear twice, first all their first row halves, then
    // all their second ones (workgroups that run together then read the same rows of K_fu)
    int tail_id = -1, tail_half = 0;
    if (a.tail_wg > 0) {
        const int ntl = a.tail_wg / 8, lfull = ((a.nb + 7) / 8) * per_unit - ntl;
        if (loc >= lfull) {
            const int q = loc - lfull;
            tail_id = (q % ntl) * 8 + xcd;
            tail_half = q / ntl;
            loc = lfull + q % ntl;
        }
    }
    // (pair combos last 9 / 8 of an off-diagonal tile; dealing them FIRST in the launch, one per CU while the chip fills, changed
    //  nothing against this unit-major order: 3.12 vs 3.12 ms, profiles/r04_ab_gram_pair.txt)
    int unit_l = loc / per_unit, wsel = loc % per_unit;
    int bz = unit_l * 8 + xcd;
    // Fewer units than XCDs (1-4 units of a one- or two-chain rank, split-K): a unit's row ranges are dealt to 8 / nb XCDs instead of all to
    // one -- the ranges read disjoint rows of K_fu, so nothing is shared that an L2 could keep; half of the chip idled otherwise
    // (one chain: 220 us of tile pass on four XCDs)
    const int spread = (ksplit > 1 && a.combo && a.nb <= 4 && 8 % a.nb == 0 && ksplit % (8 / a.nb) == 0) ? 8 / a.nb : 1;
    if (spread > 1) {
        const int sub = per_unit / spread;                   // workgroups of a unit on this XCD
        if (loc >= sub) return;
        bz = xcd % a.nb;
        const int rsub = xcd / a.nb, kper = ksplit / spread;
        wsel = (loc / kper) * ksplit + (loc % kper) * spread + rsub;     // (tile w = loc / kper, row range (loc % kper) * spread + rsub)
    }
    if (bz >= a.nb) return;
    int tile, kpart;
    if (a.combo) {
        const int n128 = a.Mp / 128, noff = n128 * (n128 - 1) / 2, w = wsel / ksplit;      // (split-K: the row ranges of a workgroup are neighbours in the list)
        const int cpart = (tail_id >= 0) ? tail_half : wsel % ksplit;
        if (w >= noff) {                                 // a combo workgroup: pair combos take two panels, the older ones three per group of four
#ifdef FFVD_DF_TRACE
            const long long tc0 = wall_clock64();
#endif
#if GRAM_COMBO == 2
            gram_pair_body<MODE>(a, bz, 2 * (w - noff), tail_id, cpart, ksplit, As, Bs, red, &tail_slot);
#else
            gram_combo_body<MODE>(a, bz, 4 * ((w - noff) / 3), (w - noff) % 3, tail_id, tail_half, As, Bs, red, &tail_slot);
#endif
#ifdef FFVD_DF_TRACE
            if (threadIdx.x == 0 && bz < 128 && per_unit <= 10 && MODE == GRAM_KFU) {
                long long *g = gram_trace_buf + (size_t)(bz * 10 + w) * 4;
                g[0] = tc0; g[1] = wall_clock64(); g[2] = 1; g[3] = (long long)blockIdx.x;
            }
#endif
            return;
        }
        int ti = 1;                                      // off-diagonal tile w = ti (ti - 1) / 2 + tj,  tj < ti
        while (ti * (ti + 1) / 2 <= w) ++ti;
        tile = ti * (ti + 1) / 2 + (w - ti * (ti - 1) / 2);
        kpart = cpart;
    } else {
        tile = (loc % per_unit) / ksplit;
        kpart = (tail_id >= 0) ? tail_half : loc % ksplit;
    }
    int ti = 0;                       // tile = ti (ti + 1) / 2 + tj,  tj <= ti
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    const int tj = tile - ti * (ti + 1) / 2;
#ifdef FFVD_DF_TRACE
    const long long tw0 = wall_clock64();     // debug build (tools/gram_trace.py): start and end of the tiles of the first 16 units
#endif
    if (ti == tj) gram_body<MODE, true>(a, bz, ti, tj, tile, kpart, ksplit, As, Bs, dls, red, tail_id, &tail_slot);
#if GRAM_GLDS_OFFDIAG
    else if (a.Mp % 128 == 0) gram_body<MODE, false, true>(a, bz, ti, tj, tile, kpart, ksplit, As, Bs, dls, red, tail_id, &tail_slot);
#endif
    else gram_body<MODE, false>(a, bz, ti, tj, tile, kpart, ksplit, As, Bs, dls, red, tail_id, &tail_slot);
#ifdef FFVD_DF_TRACE
    if (threadIdx.x == 0 && bz < 16 && ksplit == 1 && a.ntiles <= 10 && MODE == GRAM_KFU) {
        df_trace_buf[2048 + (bz * 10 + tile) * 2] = tw0;
        df_trace_buf[2048 + (bz * 10 + tile) * 2 + 1] = wall_clock64();
    }
    if (threadIdx.x == 0 && bz < 128 && ksplit == 1 && a.ntiles <= 10 && MODE == GRAM_KFU) {     // every workgroup of the launch (tools/gram_rounds.py)
        long long *g = gram_trace_buf + (size_t)(bz * 10 + (a.combo ? wsel : tile)) * 4;   // (a tail workgroup: the half that ends last)
        g[0] = tw0; g[1] = wall_clock64(); g[2] = (ti == tj) ? 1 : 0; g[3] = (long long)blockIdx.x;
    }
#endif
}

// Second pass of a split-K Gram launch: one workgroup per (unit, tile) adds the `ksplit` partial tiles in fixed
// order and applies the epilogue of gram_body (scaling, + I / + K_uu, trace partial, the delta^T A row).
template <int MODE>
__global__ __launch_bounds__(1024) void gram_combine_kernel(GramArgs a) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const int bz = blockIdx.y, tile = blockIdx.x;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    const int tj = tile - ti * (ti + 1) / 2;
    const int Mp = a.Mp;
    const int b = a.b0 + bz, dl = b % a.Dl, dg = a.d_begin + dl;
    const size_t pstride = (size_t)(Mp + 1) * Mp;
    const double *P0 = a.part + (size_t)bz * pstride;
    const size_t ks_stride = (size_t)a.nb * pstride;
    const double scale = (MODE == GRAM_PLAIN) ? 1.0 : a.yn_over_batch / exp(a.log_Q[dg]);
    double *Hb = a.H + (size_t)bz * a.h_stride;
    const double *Kadd = (MODE == GRAM_KFU) ? a.Kadd + (size_t)dl * a.kadd_stride : nullptr;
    const double *Kinv = (MODE == GRAM_KFU) ? a.Kinv + (size_t)dl * a.kinv_stride : nullptr;
    double trp = 0.0;
    const bool want_tr = (MODE == GRAM_KFU) && a.trace_mode != 1, want_out = a.trace_mode != 2;
    // one pair of columns per thread and 64 x 32 sub-block (the sub-blocks gram_body writes): 16-byte accesses, the
    // eight sub-blocks unrolled so that all their loads are in flight together
    const int ri = tid >> 4, cj = 2 * (tid & 15);
#pragma unroll
    for (int sb = 0; sb < 8; ++sb) {
        const int wr = sb >> 2, wc = sb & 3;
        const int I0 = ti * 128 + wr * 64, J0 = tj * 128 + wc * 32;
        if (!((I0 < Mp) && (J0 < Mp) && (J0 < I0 + 64))) continue;
        const int i = I0 + ri, j = J0 + cj;
        const size_t off = (size_t)i * Mp + j;
        double2 g = {0.0, 0.0};
        const int nks = (a.raw_summed && a.trace_mode == 2) ? 1 : a.ksplit;
        for (int ks = 0; ks < nks; ++ks) {
            const double2 q = *reinterpret_cast<const double2 *>(P0 + (size_t)ks * ks_stride + off);
            g.x += q.x; g.y += q.y;
        }
        if (a.raw_summed && a.trace_mode == 1) *reinterpret_cast<double2 *>(const_cast<double *>(P0) + off) = g;      // (this thread alone reads and writes the element)
        double2 v = {0.0, 0.0};
        if (MODE == GRAM_F) {
            v.x = g.x * scale + ((i == j) ? 1.0 : 0.0);
            v.y = g.y * scale + ((i == j + 1) ? 1.0 : 0.0);
        } else if (MODE == GRAM_KFU) {
            if (want_out) {
                const double2 ka = *reinterpret_cast<const double2 *>(Kadd + off);
                v.x = g.x * scale + ka.x;
                v.y = g.y * scale + ka.y;
            }
            if (want_tr) {
                const double2 ki = *reinterpret_cast<const double2 *>(Kinv + off);
                const double w0 = (i > j) ? 2.0 : ((i == j) ? 1.0 : 0.0);
                const double w1 = (i > j + 1) ? 2.0 : ((i == j + 1) ? 1.0 : 0.0);
                trp += w0 * (ki.x * g.x);
                trp += w1 * (ki.y * g.y);
            }
        } else v = g;
        if (want_out) *reinterpret_cast<double2 *>(Hb + off) = v;
    }
    if (want_out && a.with_row && ti == tj && tid < 128) {
        const int col = ti * 128 + tid;
        if (col < Mp) {
            double bs = 0.0;
            for (int ks = 0; ks < a.ksplit; ++ks) bs += P0[(size_t)ks * ks_stride + (size_t)Mp * Mp + col];
            Hb[(size_t)a.brow * Mp + col] = bs * scale;
        }
    }
    if (want_tr) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) trp += __shfl_xor(trp, m);      // fixed order: reproducible
        if ((tid & 63) == 0) red[tid >> 6] = trp;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < 16; ++w) t += red[w];
            a.trpart[(size_t)b * a.ntiles + tile] = t;
        }
    }
}

int gram_ntiles(int Mp) {
    const int n128 = (Mp / NB + 1) / 2;
    return n128 * (n128 + 1) / 2;
}

static bool gram_uses_combos(int Mp, int ksplit, int with_row);
static int gram_wg_per_unit(int Mp, int ksplit, int with_row);
// Few tiles cannot fill the 512 workgroup slots (256 CUs x 2), and 1-2 tiles per slot balance badly (640 tiles take
// 1.56 x the time of 512).  Measured at M = 512, T = 4096: 160 tiles 0.60 ms unsplit / 0.48 ms in 3 row ranges (one
// full round) / 0.55-0.62 ms in 2, 4, 6, 8; 320 tiles 1.10 -> 0.86 ms in 3-4 ranges; 640 tiles 1.67 -> 1.52 ms in 2.
int gram_ksplit(int Mp, int nb, int rows, int with_row, bool fill_slots) {
    if (const char *e = getenv("FFVD_GSPLIT")) return atoi(e) > 0 ? atoi(e) : 1;       // tuning override
    const int nchunk = rows / GT;
    if (fill_slots && gram_uses_combos(Mp, 1, with_row) && GRAM_COMBO == 2) {
        // Nothing beside the launch (the K_uu chain runs behind it): the ranges that give every slot of the chip one workgroup --
        // tools/gsplit_s1.sh: 4 chains 320 us in four ranges against 382 in three, 2 chains 157 against 202, 1 chain 150 in eight
        // Measured per-rank iterations, side chain behind the pass vs beside it in three ranges (tools/sync_step.py, same box,
        // profiles/r04_ab_side_late.txt): 8 chains 1.07 vs 1.20 ms (two ranges), 4 chains 0.688 vs 0.705 (four; 0.750 vs 0.734 before the
        // main-row workgroups of the chain's launch helped with its inverse), 1-2 chains 0.44 / 0.53 vs 0.54 / 0.56 (eight ranges)
        const int n = nb * gram_wg_per_unit(Mp, 1, with_row);
        // (other chain counts, whole rounds of the 512 slots: 6 / 12 / 24 chains = three rounds in 8 / 4 / 2 ranges 0.887 / 1.43 / 2.47 vs
        //  0.911 / 1.47 / 2.60 ms; 20 chains = five rounds in four ranges 2.275 vs 2.176: not beyond three rounds)
        if (n > 0 && n % 512 == 0) return 0;                                         // (whole rounds unsplit: the caller's unsplit schedule)
        for (int ks = 2; ks <= 8; ++ks)
            if (n > 0 && (n * ks) % 512 == 0 && n * ks <= 1536 && nchunk / ks >= 8) return ks;
        if (n > 0 && n <= 64 && nchunk / 8 >= 8) return 8;
        return 0;                                                                   // (the caller keeps the first-half schedule)
    }
    if (gram_uses_combos(Mp, 1, with_row) && GRAM_COMBO == 2) {
        // pair combos: 8 workgroups per unit at M = 512.  Measured per-rank iteration times at config 2's shape, 1 .. 24 chains and
        // 1, 2, 3, 4, 6, 8 row ranges (tools/gram_split_sweep.sh, profiles/r04_gram_split.txt): three ranges are the best or within
        // 1 % of it everywhere (few units: 0.55 ms against 0.63 in 2, 4 or 8) -- except when the unsplit launch is exactly whole
        // rounds of the chip's 512 slots (16 chains: 1.97 unsplit, 2.00 in three)
        const int n = nb * gram_wg_per_unit(Mp, 1, with_row);
        if (n <= 0 || n >= 1024 || n % 512 == 0) return 1;
        int ks = 3;
        while (ks > 1 && nchunk / ks < 8) --ks;
        return ks;
    }
    const int n = nb * gram_ntiles(Mp);
    if (n <= 0 || n >= 1024) return 1;
    // one full round; three row ranges up to 640 tiles (re-measured with the 1024-thread combine pass: 160 tiles 1.05 ms per
    // iteration in 3 ranges / 1.08-1.09 in 2, 4; 320 tiles 1.49 in 3 / 1.53 in 2 / 1.58 in 4; 640 tiles 2.46 in 3 / 2.53 in 2)
    int ks = (n <= 256) ? 512 / n : ((n <= 704) ? 3 : (1280 + n / 2) / n);
    if (ks > 8) ks = 8;
    while (ks > 1 && nchunk / ks < 8) --ks;          // keep at least 128 rows per range
    return ks;
}
size_t gram_part_doubles(int Mp, int nb, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * nb * (size_t)(Mp + 1) * Mp : 0;
}

#ifndef GRAM_TAIL_SPLIT
#define GRAM_TAIL_SPLIT 1
#endif
// workgroups per unit of an unsplit launch: with Mp a multiple of 512 the diagonal tiles of every four panels become three combos
// (and only when the kernel has no delta^T A row to form: the combos have no idle wavefront for it)
static bool gram_uses_combos(int Mp, int ksplit, int with_row) {
    if (GRAM_COMBO == 2) return Mp % 256 == 0 && !with_row;          // pair combos: also the diagonal workgroups of a split-K launch
    return GRAM_COMBO && ksplit <= 1 && Mp % 512 == 0 && !with_row;
}
static int gram_wg_per_unit(int Mp, int ksplit, int with_row) {
    const int n128 = (Mp / NB + 1) / 2;
    if (gram_uses_combos(Mp, ksplit, with_row))
        return (n128 * (n128 - 1) / 2 + (GRAM_COMBO == 2 ? n128 / 2 : 3 * (n128 / 4))) * (ksplit > 1 ? ksplit : 1);
    return gram_ntiles(Mp) * (ksplit > 1 ? ksplit : 1);
}
// Which workgroups of an unsplit launch are cut in two row halves: those of the last, partial round, when their halves still fit
// into half of the chip's slots (two per CU) -- then the halves take the slots that free up FIRST at the end of the last full
// round and run beside its stragglers.  tools/gram_rounds.py (128 units): a round's tiles end 0.5-0.7 ms apart (the workgroup
// that arrived first on a CU keeps the matrix pipe), a workgroup alone on its CU reaches 57-65 % of what two reach together,
// and a 2-way split of a whole half round (256 tiles -> 512 halves) ended LATER than no split because the last halves start
// when the last slot frees up; with combos the remainder is 128 workgroups -> 256 halves -> the 256 slots that free up first.
int gram_tail_wg(int Mp, int nb, int ksplit, int with_row) {
    if (!GRAM_TAIL_SPLIT || ksplit > 1) return 0;
    static const int slots = [] {
        int dev = 0, cus = 256;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
        return 2 * cus;
    }();
    const int n = ((nb + 7) / 8) * 8 * gram_wg_per_unit(Mp, ksplit, with_row);
    if (n < 2 * slots) return 0;                                 // small launches have their own split-K path
    const int r = n % slots;
    return (r > 0 && 4 * r <= slots && r % 8 == 0) ? r : 0;
}
size_t gram_tail_doubles(int tail_wg) {
    return tail_wg > 0 ? (size_t)tail_wg * 2 * GRAM_TAIL_DOUBLES + ((size_t)tail_wg + 1) / 2 : 0;
}

// phase: 0 = everything, 1 = the tile pass only, 2 = the combine pass only (split-K launches: the tile pass writes
// raw partials and needs neither K_uu nor K^-1, so the caller may run it before those exist and combine afterwards)
void launch_gram(hipStream_t stream, GramArgs a, int phase) {
    a.ntiles = gram_ntiles(a.Mp);
    if (a.brow <= 0) a.brow = a.Mp;
    if (!a.part || a.ksplit < 1) a.ksplit = 1;
    const int groups = (a.nb + 7) / 8;
    a.combo = gram_uses_combos(a.Mp, a.ksplit, a.with_row) ? 1 : 0;
    a.wg_per_unit = gram_wg_per_unit(a.Mp, a.ksplit, a.with_row);
    if (a.ksplit > 1 || !a.tail_part || a.tail_wg != gram_tail_wg(a.Mp, a.nb, a.ksplit, a.with_row)) a.tail_wg = 0;
    if (a.tail_wg > 0) a.tail_cnt = reinterpret_cast<int *>(a.tail_part + (size_t)a.tail_wg * 2 * GRAM_TAIL_DOUBLES);
    const dim3 grid(groups * 8 * a.wg_per_unit + a.tail_wg);
    if (phase != 2 && phase != 3 && phase != 4) {
        if (a.mode == GRAM_F) hipLaunchKernelGGL(gram_kernel<GRAM_F>, grid, dim3(512), 0, stream, a);
        else if (a.mode == GRAM_KFU) hipLaunchKernelGGL(gram_kernel<GRAM_KFU>, grid, dim3(512), 0, stream, a);
        else if (a.mode == GRAM_KFU_RAW) hipLaunchKernelGGL(gram_kernel<GRAM_KFU_RAW>, grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL(gram_kernel<GRAM_PLAIN>, grid, dim3(512), 0, stream, a);
    }
    if (phase == 3) a.trace_mode = 2;
    if ((a.ksplit > 1 && phase != 1) || phase == 3 || phase == 4) {
        const dim3 cgrid(a.ntiles, a.nb);
        if (a.mode == GRAM_KFU_RAW) a.mode = GRAM_KFU;
        if (a.mode == GRAM_F) hipLaunchKernelGGL(gram_combine_kernel<GRAM_F>, cgrid, dim3(1024), 0, stream, a);
        else if (a.mode == GRAM_KFU) hipLaunchKernelGGL(gram_combine_kernel<GRAM_KFU>, cgrid, dim3(1024), 0, stream, a);
        else hipLaunchKernelGGL(gram_combine_kernel<GRAM_PLAIN>, cgrid, dim3(1024), 0, stream, a);
    }
}

// ---------------------------------------------------------------------------------------------
// logdet(H) = 2 sum log diag(L_H)  (tf.linalg.logdet, :253);  b H^{-1} b^T = |L_H^{-1} b|^2 (:254)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void h_finish_kernel(const double *H, int Mp, size_t h_stride, double *hterms, int yrow) {
    __shared__ double scratch[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *Hb = H + (size_t)b * h_stride;
    double ld = 0.0, qd = 0.0;
    for (int i = tid; i < Mp; i += 256) {
        ld += log(Hb[(size_t)i * Mp + i]);
        const double y = Hb[(size_t)yrow * Mp + i];
        qd += y * y;
    }
    ld = block_sum_256(ld, scratch);
    qd = block_sum_256(qd, scratch);
    if (tid == 0) { hterms[2 * b] = 2.0 * ld; hterms[2 * b + 1] = qd; }
}
void launch_h_finish(hipStream_t stream, const double *H, int Mp, size_t h_stride, int nb, double *hterms, int yrow) {
    hipLaunchKernelGGL(h_finish_kernel, dim3(nb), dim3(256), 0, stream, H, Mp, h_stride, hterms, yrow > 0 ? yrow : Mp);
}

// ---------------------------------------------------------------------------------------------
// Per-chain streaming reductions (likelihoods.py:76-111; dgp_model.py:250-252,283-284,346-351)
// chain_terms[s] = { lik quadratic sum, transition quadratic sum, trace sum, prior_x_0 }
// ---------------------------------------------------------------------------------------------
constexpr int CR_SPLIT = 8;     // workgroups per chain (four times as many from 8 latent dims on: a row's terms are a loop over the dims)
static int chain_reduce_split(int Dl) { return Dl >= 8 ? 4 * CR_SPLIT : CR_SPLIT; }
__global__ __launch_bounds__(256) void chain_reduce_kernel(ReduceArgs a, double *partial /*[S][nsplit][4]*/, const int nsplit) {
    __shared__ double scratch[256];
    const int s = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int T = a.T, D = a.D;
    const double *Xs = a.X + (size_t)s * (T + 1) * D;
    double lik = 0.0, xq = 0.0, tr = 0.0;
    for (int t = part * 256 + tid; t < T; t += 256 * nsplit) {
        if (a.shared_terms) {
            for (int j = 0; j < a.Ydim; ++j) {
                double ym = 0.0;
                for (int d = 0; d < D; ++d) ym += Xs[(size_t)(t + 1) * D + d] * a.CC[(size_t)d * a.Ydim + j];   // :76-79
                ym += a.DD[j];
                const double R = exp(a.log_Rchols[j]);          // Rchols[0] = first row (dgp_model.py:250)
                const double r = (a.Y[(size_t)t * a.Ydim + j] - ym) / R;
                lik += -0.5 * (r * r);
            }
        }
        double xsq = 0.0;
        if (a.kind == 1) {                                       // LinearK.Kdiag (kernels.py:278-281), per unit variance
            const double *xr = a.xk + (size_t)s * a.xk_chain_stride + (size_t)t * a.xk_ld;
            for (int p = 0; p < a.xk_cols; ++p) xsq += xr[p] * xr[p];
            for (int p = 0; p < a.C; ++p) { double x = a.ctrl[(size_t)t * a.C + p]; xsq += x * x; }
        }
        for (int dl = 0; dl < a.Dl; ++dl) {
            const int dg = a.d_begin + dl;
            const double Q = exp(a.log_Q[dg]);
            const double sq = sqrt(Q);                           // Q ** 0.5 (dgp_model.py:284,351)
            const size_t bb = ((size_t)s * a.Dl + dl) * a.ng;
            double rs = 0.0, fm = 0.0;
            for (int g = 0; g < a.ng; ++g) {
                if (a.rowsq) rs += a.rowsq[(bb + g) * a.Tp + t];
                if (a.branch == 0) fm += a.fmean[(bb + g) * a.Tp + t];
            }
            const double kdiag = (a.kind == 0) ? a.variance[dl] : xsq * a.variance[dl];
            tr += -0.5 * ((kdiag - rs) / Q);                     // conditionals_multi_output.py:255 / dgp_model.py:348
            const double x1 = Xs[(size_t)(t + 1) * D + dg], x0 = Xs[(size_t)t * D + dg];
            double r;
            if (a.branch == 1) r = (x1 - x0) / sq;               // dgp_model.py:283-284
            else r = (x1 - (fm + x0)) / sq;                      // dgp_model.py:346,351
            xq += -0.5 * (r * r);
        }
    }
    lik = block_sum_256(lik, scratch);
    xq = block_sum_256(xq, scratch);
    tr = block_sum_256(tr, scratch);
    if (tid == 0) {
        double *o = partial + ((size_t)s * nsplit + part) * 4;
        o[0] = lik; o[1] = xq; o[2] = tr;
    }
}
__global__ void chain_combine_kernel(ReduceArgs a, const double *partial, const int nsplit) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.S) return;
    double lik = 0.0, xq = 0.0, tr = 0.0;
    for (int p = 0; p < nsplit; ++p) {
        const double *o = partial + ((size_t)s * nsplit + p) * 4;
        lik += o[0]; xq += o[1]; tr += o[2];
    }
    const double *Xs = a.X + (size_t)s * (a.T + 1) * a.D;
    double px0 = 0.0;
    for (int d = 0; d < a.D; ++d) px0 += Xs[d] * Xs[d];
    double *o = a.chain_terms + (size_t)s * 8;
    o[0] = lik; o[1] = xq; o[2] = tr; o[3] = a.skip_x0 ? 0.0 : -px0 / 2.0;     // prior_x_0 dgp_model.py:252
    bool bad = false;
    for (int i = 0; i < a.ninfo; ++i) bad = bad || a.info[i] != 0;
    if (bad) o[0] = o[1] = o[2] = __longlong_as_double(0x7ff8000000000000LL);
}
void launch_chain_reduce(hipStream_t stream, const ReduceArgs &a, double *partial) {
    const int nsplit = chain_reduce_split(a.Dl);
    hipLaunchKernelGGL(chain_reduce_kernel, dim3(a.S, nsplit), dim3(256), 0, stream, a, partial, nsplit);
    hipLaunchKernelGGL(chain_combine_kernel, dim3((a.S + 63) / 64), dim3(64), 0, stream, a, partial, nsplit);
}

// conditional() outputs (conditionals_multi_output.py:41,48,120): mean N x D, var N x D
// One output (n, d) per group of 16 lanes: the lanes share the partial-sum groups (the skinny products of the step loops leave
// Mp / 16 of them per array) and add them by a shuffle tree -- a thread walking 96 dependent loads per output took 18 us.
__global__ __launch_bounds__(256) void conditional_finish_kernel(int kind, const double *x, int N, int P, const double *variance,
                                          const double *rowsq, const double *fmean, int ng, int Tp, int D,
                                          double *mean, double *var, const double *extra /*[D][extra_ng][Tp] or null*/,
                                          int extra_ng) {
    conditional_finish_body(blockIdx.x, kind, x, N, P, variance, rowsq, fmean, ng, Tp, D, mean, var, extra, extra_ng);
}
void launch_conditional_finish(hipStream_t stream, int kind, const double *x, int N, int P, const double *variance,
                               const double *rowsq, const double *fmean, int ng, int Tp, int D, double *mean,
                               double *var, const double *extra, int extra_ng) {
    if (N * D == 0) return;
    hipLaunchKernelGGL(conditional_finish_kernel, dim3((N * D * 16 + 255) / 256), dim3(256), 0, stream, kind, x, N, P,
                       variance, rowsq, fmean, ng, Tp, D, mean, var, extra, extra_ng);
}

// Skinny product for the step loops (rollouts, particle Gibbs: at most a few hundred rows per latent dim and step):
//   C[b] (rows x N) = A[b] (rows x K) * B (K x N),   optionally  sq[b][slab][r] = sum_{n in slab} C[r][n]^2,
//                                                                dot[b][slab][r] = sum_{n in slab} C[r][n] u[b][n].
// One workgroup = 32 rows x 16 columns, its four wavefronts a quarter of the k range each (operands straight from L2, no
// LDS staging: 64 MFMAs per wavefront at K = 512), partial tiles added through LDS in fixed order.  N / 16 x rows / 32 x nb
// workgroups instead of the handful of 128 x 128 tiles the projection GEMM cuts such a product into: 50 -> 7 us per step at 32 rows.
// `upper`: B[k][n] = 0 for k > n (L^-T), the k range of a slab ends at its last column.
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyArgs a) {
    skinny_body(a, blockIdx.x, blockIdx.y, blockIdx.z);
}
// Same product from a 1-D grid for nb <= 8 units: workgroup id i runs on XCD i % 8 (round-robin dispatch), unit u owns the XCDs
// x with x % nb == u and deals its (slab, row group) pairs over them, so a unit's A rows and B slabs are fetched into one or two L2s
// instead of all eight.  The arithmetic of a workgroup does not depend on where it runs.
__global__ __launch_bounds__(256) void skinny_gemm_xcd_kernel(SkinnyArgs a, int nbx, int nby) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int u = xcd % a.nb, j = xcd / a.nb, cnt = (8 - u + a.nb - 1) / a.nb;
    const int widx = k * cnt + j;
    if (widx >= nbx * nby) return;
    // a CU takes slots k and k + 32 of its XCD (two workgroups per CU, 32 CUs; HW_ID stamps, profiles/r05_step_trace.txt): every other
    // block of 32 slots walks the slabs backwards, so that a long triangular slab (up to K / 16 k blocks) shares its CU's MFMA pipes
    // with a short one.  The choice depends on the row group only: (slab, row group) pairs are still covered once each.
    const int by = widx / nbx, rev = ((by * nbx) / (32 * cnt)) & 1;
    const int bx = widx % nbx;
    skinny_body(a, rev ? nbx - 1 - bx : bx, by, u);
}
// rows <= Tp (a multiple of 32) rows of A exist; N, K multiples of 16.
void launch_skinny_gemm(hipStream_t stream, const double *A, size_t a_stride, int lda, const double *B, size_t b_stride, int ldb,
                        int upper, int rows, int K, int N, int nb, int Tp, double *C, size_t c_stride, int ldc,
                        const double *u, size_t u_stride, double *sq, double *dot, const double *B2, size_t b2_stride, int ldb2,
                        int N2, double *sq2, int a_trans, int upper2) {
    if (rows <= 0 || nb <= 0) return;
    SkinnyArgs a{A, a_stride, lda, B, b_stride, ldb, upper, rows, K, N, nb, Tp, C, c_stride, ldc, u, u_stride, sq, dot,
                 B2, b2_stride, ldb2, B2 ? N2 : 0, sq2, a_trans, B2 ? upper2 : 0};
    const int nbx = N / 16 + (B2 ? N2 / 16 : 0), nby = (rows + 31) / 32;
    static const bool flat = !(getenv("FFVD_SKINNY_GRID3D") && atoi(getenv("FFVD_SKINNY_GRID3D")));
    if (flat && nb <= 8) {
        const int slots = (nbx * nby + 8 / nb - 1) / (8 / nb);       // per XCD: the unit with the fewest XCDs has 8 / nb of them
        hipLaunchKernelGGL(skinny_gemm_xcd_kernel, dim3(8 * slots), dim3(256), 0, stream, a, nbx, nby);
    } else
        hipLaunchKernelGGL(skinny_gemm_kernel, dim3(nbx, nby, nb), dim3(256), 0, stream, a);
}

// out[b][i] = sum_j W[b][i][j] * y[b][j]   (posterior mean of the whitened inducing outputs: L_H^-T (L_H^-1 b))
__global__ __launch_bounds__(256) void matvec_kernel(const double *W, size_t w_stride, const double *y, size_t y_stride,
                                                     int Mp, double *out, int out_ld, int out_bs, int M, int w_mod) {
    __shared__ double ys[2048];
    const int b = blockIdx.y, tid = threadIdx.x;
    const double *Wb = W + (size_t)(w_mod > 0 ? b % w_mod : b) * w_stride, *yb = y + (size_t)b * y_stride;
    for (int j0 = 0; j0 < Mp; j0 += 2048) {     // Mp <= 2048 per tile of y
        for (int j = tid; j < 2048 && j0 + j < Mp; j += 256) ys[j] = yb[j0 + j];
        __syncthreads();
        const int i = blockIdx.x * 256 + tid;
        if (i < M) {
            double acc = (j0 == 0) ? 0.0 : out[(size_t)i * out_ld + (size_t)b * out_bs];
            const int jn = (Mp - j0 < 2048) ? Mp - j0 : 2048;
            for (int j = 0; j < jn; ++j) acc += Wb[(size_t)i * Mp + j0 + j] * ys[j];
            out[(size_t)i * out_ld + (size_t)b * out_bs] = acc;
        }
        __syncthreads();
    }
}
void launch_matvec(hipStream_t stream, const double *W, size_t w_stride, const double *y, size_t y_stride, int Mp,
                   double *out, int out_ld, int out_bs, int M, int batch, int w_mod) {
    hipLaunchKernelGGL(matvec_kernel, dim3((M + 255) / 256, batch), dim3(256), 0, stream, W, w_stride, y, y_stride, Mp, out,
                       out_ld, out_bs, M, w_mod);
}

// extra[b][n] = sum_j ( sum_m F[b][n][m] * Qs[m][j] )^2   -- the q_sqrt variance inflation of
// base_conditional_after_kernel_precalculation (conditionals_multi_output.py:371-380) with LTA^T = F q_sqrt.
__global__ __launch_bounds__(256) void qsqrt_inflation_kernel(const double *F, size_t f_stride, int Tp, int Mp, int M,
                                                              const double *Qs /*M x M*/, double *extra) {
    __shared__ double fr[2048];
    __shared__ double scratch[256];
    const int n = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const double *Fr = F + (size_t)b * f_stride + (size_t)n * Mp;
    for (int m = tid; m < M; m += 256) fr[m] = Fr[m];
    __syncthreads();
    double acc = 0.0;
    for (int j = tid; j < M; j += 256) {
        double v = 0.0;
        for (int m = 0; m < M; ++m) v += fr[m] * Qs[(size_t)m * M + j];
        acc += v * v;
    }
    acc = block_sum_256(acc, scratch);
    if (tid == 0) extra[(size_t)b * Tp + n] = acc;
}
void launch_qsqrt_inflation(hipStream_t stream, const double *F, size_t f_stride, int Tp, int Mp, int M, const double *Qs,
                            double *extra, int N, int batch) {
    if (N == 0) return;
    hipLaunchKernelGGL(qsqrt_inflation_kernel, dim3(N, batch), dim3(256), 0, stream, F, f_stride, Tp, Mp, M, Qs, extra);
}

// Operator-API elementwise kernels (likelihoods.py:76-79, 89-111; utils.py:11)
__global__ void predict_mean_kernel(const double *X, int N, int D, const double *CC, const double *DD, int J, double *out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * J) return;
    const int n = idx / J, j = idx % J;
    double v = 0.0;
    for (int d = 0; d < D; ++d) v += X[(size_t)n * D + d] * CC[(size_t)d * J + j];     // tf.matmul(X_end, CC)
    out[idx] = v + DD[j];                                                              // + DD
}
// mode 0: logdensity_norm_diag (N outputs), mode 1: logdensity_norm_diag_nonvec (N x J outputs)
__global__ void logdensity_kernel(int mode, const double *y, const double *ymean, const double *R, int N, int J, double *out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 1) {
        if (idx >= N * J) return;
        const int j = idx % J;
        const double r = (y[idx] - ymean[idx]) / R[j];
        out[idx] = -0.5 * (r * r) + (-log(R[j]));
        return;
    }
    if (idx >= N) return;
    double e = 0.0, lr = 0.0;
    for (int j = 0; j < J; ++j) {
        const double r = (y[(size_t)idx * J + j] - ymean[(size_t)idx * J + j]) / R[j];
        e += r * r;
        lr += log(R[j]);
    }
    out[idx] = -0.5 * e + (-lr);
}
__global__ void get_rand_kernel(const double *mean, const double *var, const double *eps, size_t n, double *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mean[i] + eps[i] * sqrt(var[i]);
}
void launch_predict_mean(hipStream_t stream, const double *X, int N, int D, const double *CC, const double *DD, int J,
                         double *out) {
    if (N * J == 0) return;
    hipLaunchKernelGGL(predict_mean_kernel, dim3((N * J + 255) / 256), dim3(256), 0, stream, X, N, D, CC, DD, J, out);
}
void launch_logdensity(hipStream_t stream, int mode, const double *y, const double *ymean, const double *R, int N, int J,
                       double *out) {
    const int n = mode == 1 ? N * J : N;
    if (n == 0) return;
    hipLaunchKernelGGL(logdensity_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, mode, y, ymean, R, N, J, out);
}
void launch_get_rand(hipStream_t stream, const double *mean, const double *var, const double *eps, size_t n, double *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(get_rand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mean, var, eps, n, out);
}

// ---------------------------------------------------------------------------------------------
// Priors + nll assembly (dgp_model.py:105-143, 259-297, 326-334)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a) {
    __shared__ double scratch[4][10];
    if (a.prior_sums) {               // formed earlier in the iteration by the same code (prior_sums_kernel): the same bits
        double sm[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) sm[i] = a.prior_sums[i];
        finalize_assemble<256>(a, scratch, sm);
    } else finalize_body<256>(a, scratch);
}
__global__ __launch_bounds__(256) void prior_sums_kernel(FinalizeArgs a, double *out) {
    __shared__ double scratch[4][10];
    double sm[10];
    finalize_priors<256>(a, scratch, sm);
#pragma unroll
    for (int i = 0; i < 10; ++i)
        if ((int)threadIdx.x == i) out[i] = sm[i];
}
void launch_prior_sums(hipStream_t stream, const FinalizeArgs &a, double *out) {
    hipLaunchKernelGGL(prior_sums_kernel, dim3(1), dim3(256), 0, stream, a, out);
}
void launch_finalize(hipStream_t stream, const FinalizeArgs &a) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, stream, a);
}

// One step of the posterior rollout (collect_samples_formal, base_model.py:304-314) for R rollouts side by side:
//   x_next = x + f_mu + eps * sqrt(f_var + Q);  predict_x[r][t] = x_next;  predict_var[r][t] = f_var + Q;
// and the GP input row of the next step, xc[r] = [x_next, control_inputs[t + 1]].
// The conditional() epilogue inside a step-loop kernel, bit for bit what conditional_finish_kernel computes: 16 lanes per (row n,
// dim d) sum the partials strided, an xor-butterfly combines them; every lane returns the sums (call with ALL lanes of the 16 active).
__global__ __launch_bounds__(256) void rollout_finish_update_kernel(FinishIn f, const double *log_Q, const double *eps_t,
                                                                    const double *ctrl_next, int R, int C, int t, int steps,
                                                                    const double *x_in, double *x_out, double *predict_x,
                                                                    double *predict_var) {
    rollout_finish_update_body(blockIdx.x, f, log_Q, eps_t, ctrl_next, R, C, t, steps, x_in, x_out, predict_x, predict_var);
}
void launch_rollout_finish_update(hipStream_t stream, int kind, const double *variance, const double *rowsq, const double *fmean,
                                  int ng, int Tp, const double *extra, int extra_ng, const double *log_Q, const double *eps_t,
                                  const double *ctrl_next, int R, int D, int C, int t, int steps, const double *x_in, double *x_out,
                                  double *predict_x, double *predict_var) {
    if (R * D == 0) return;
    FinishIn f{kind, D + C, ng, Tp, D, extra_ng, variance, rowsq, fmean, extra};
    hipLaunchKernelGGL(rollout_finish_update_kernel, dim3((R * D * 16 + 255) / 256), dim3(256), 0, stream, f, log_Q, eps_t, ctrl_next, R,
                       C, t, steps, x_in, x_out, predict_x, predict_var);
}

__global__ void rollout_update_kernel(const double *mean, const double *var, const double *log_Q, const double *eps_t,
                                      const double *ctrl_next, int R, int D, int C, int t, int steps, double *xc,
                                      double *predict_x, double *predict_var) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = D + C;
    if (idx >= R * P) return;
    const int r = idx / P, p = idx % P;
    if (p < D) {
        const double v = var[r * D + p] + exp(log_Q[p]);
        const double xn = (mean[r * D + p] + xc[idx]) + eps_t[r * D + p] * sqrt(v);
        const size_t o = ((size_t)r * steps + t) * D + p;
        predict_x[o] = xn;
        predict_var[o] = v;
        xc[idx] = xn;
    } else if (ctrl_next) {
        xc[idx] = ctrl_next[p - D];
    }
}
void launch_rollout_update(hipStream_t stream, const double *mean, const double *var, const double *log_Q,
                           const double *eps_t, const double *ctrl_next, int R, int D, int C, int t, int steps, double *xc,
                           double *predict_x, double *predict_var) {
    const int n = R * (D + C);
    if (n == 0) return;
    hipLaunchKernelGGL(rollout_update_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, mean, var, log_Q, eps_t,
                       ctrl_next, R, D, C, t, steps, xc, predict_x, predict_var);
}

// One workgroup: thread i < R propagates particle i and forms its log weight, thread R the reference's; the softmax
// CDF is summed sequentially in index order (as the oracle's cumsum) by one thread; then every thread i < R finds its
// ancestor by binary search and gathers.
__global__ __launch_bounds__(PG_MAXN) void pg_step_kernel(const double *mean, const double *var, const double *log_Q,
                                                          const double *eps_t, const double *unif_t, const double *y_t,
                                                          const double *x_ref_next, const double *CC, const double *DD,
                                                          const double *Rch, const double *ctrl_next, int R, int D, int C,
                                                          int Ydim, double *xc, double *cand, double *parts_next,
                                                          int32_t *idx_out) {
    pg_step_body<0>(mean, var, log_Q, eps_t, unif_t, y_t, x_ref_next, CC, DD, Rch, ctrl_next, R, D, C, Ydim, xc, cand, parts_next, idx_out);
}
void launch_pg_step(hipStream_t stream, const double *mean, const double *var, const double *log_Q, const double *eps_t,
                    const double *unif_t, const double *y_t, const double *x_ref_next, const double *CC, const double *DD,
                    const double *Rch, const double *ctrl_next, int R, int D, int C, int Ydim, double *xc, double *cand,
                    double *parts_next, int32_t *idx_out) {
    int threads = 64;
    while (threads < R + 1) threads <<= 1;
    hipLaunchKernelGGL(pg_step_kernel, dim3(1), dim3(threads), 0, stream, mean, var, log_Q, eps_t, unif_t, y_t, x_ref_next,
                       CC, DD, Rch, ctrl_next, R, D, C, Ydim, xc, cand, parts_next, idx_out);
}

}  // namespace ffvd
