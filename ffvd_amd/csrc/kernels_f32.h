// Launchers of the fp32-contraction path (ffvd_config.dtype = FFVD_F32C), see kernels_f32.hip.
#pragma once
#include "kernels.h"

namespace ffvd {

// out[bz][t][m] (fp32, ld Mp) = K_d(x_t, Z_m); uses x, ctrl, hv, T, Tp, M, Mp, P, b0, nb, Dl of `a`.
void launch_kfu_build_f32(hipStream_t stream, const ProjectArgs &a, float *out);
// out[dl] (Mp x Mp fp32, row-major) = transpose of the L^-T rows `ext[dl]` (fp64, ld Mp) = L^-1
void launch_linv_f32(hipStream_t stream, const double *ext, size_t ext_stride, float *out, int Mp, int Dl);

struct ProjF32Args {
    const float *Kf;        // [nb][Tp][Mp] K_fu
    size_t kf_stride;
    const float *LinvT;     // [Dl][Mp][Mp] L^-1 (row j = column j of L^-T)
    float *F;               // [nb][Tp][Mp] F = K_fu L^-T
    size_t f_stride;
    double *sqpart;         // [nbatch_total][proj_f32_ntiles] per-tile sums of F^2 (fp64); NULL = not wanted
    int Tp, Mp, Dl, b0, nb;
    // backward pass: the right operand is a FULL symmetric Mp x Mp matrix per UNIT (Gamma, rounded to fp32) instead of the
    // per-dim triangular L^-1:  F = K_fu Gamma  (the fp32 half of dl/dK_fu = 2 K_fu Gamma + delta (alpha u)^T)
    const float *Bunit;     // [nb][Mp][Mp] or NULL
    size_t bunit_stride;
};
// out[i] = (float)in[i]
void launch_to_f32(hipStream_t stream, const double *in, float *out, size_t n);
int proj_f32_ntiles(int Tp, int Mp);
void launch_proj_gemm_f32(hipStream_t stream, const ProjF32Args &a);
void launch_sum_partials(hipStream_t stream, const double *part, int n, int nb, double *out);

struct GramF32Args {
    const float *F;         // [nb][rows][Mp]
    size_t f_stride;
    int rows;               // Tp (multiple of 32)
    int with_row, brow;     // extra row `brow` (0 = Mp) = delta^T F * (Y_N / (batch Q_d))
    const double *X;        // [S][T+1][D]
    const double *log_Q;    // [D]
    int T, D, Mp, Dl, d_begin, b0, nb;
    double yn_over_batch;
    double *H;              // [nb] fp64 slabs (ld Mp), lower-triangular tiles
    size_t h_stride;
    int flush;              // fp32 accumulators are added into the fp64 tile every `flush` t-tiles of 32 rows (0 = at the end)
    int ntiles;             // filled by the launcher
};
void launch_gram_f32(hipStream_t stream, GramF32Args a);

}  // namespace ffvd
