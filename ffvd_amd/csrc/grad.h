// Launchers of the backward pass (gradient of the collapsed-U nll); see grad.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffvd {

enum { ATB_PLAIN = 0, ATB_GAMMA = 1, ATB_BWD_E = 2 };
struct AtbArgs {
    int mode;
    const double *A; size_t a_stride; int lda, nA;     // A: rows x lda, first nA columns used (= output rows)
    int a_rowmajor;                                     // BWD_E only: A is stored [output row][k] (nA x lda) instead
    const double *B; size_t b_stride; int ldb, nB;     // B: rows x ldb, first nB columns used (= output columns)
    int b_per_dim;                                      // 1: B is indexed by latent dim (bz % Dl) instead of unit
    int rows;                                           // rows summed over (multiple of 16)
    double *C; size_t c_stride; int ldc;
    int nb, b0, Dl, d_begin;
    const double *log_Q;
    const double *u; size_t u_stride;                   // GAMMA / BWD_E: u = A^-1 c per unit
    const double *X; int T, D;                          // BWD_E: latent trajectories (delta_t)
    const double *Kf; size_t kf_stride; int ldkf;       // BWD_E: K_fu (T x M) per unit
    const double *Kinv, *Kcopy; size_t k_stride; int ldk;   // GAMMA: per latent dim
    double *part;                                       // GAMMA: [nb][ntiles] partial sums of sum_ij (A^-1)_ij K_ij
    int small_tiles;                                    // PLAIN: 64 x 64 tiles, four workgroups per CU (M^3-sized batched products)
    int ntile;                                          // filled by launch_atb: tiles per unit
    int a_per_dim;                                      // 1: A is indexed by latent dim (bz % Dl) instead of unit
    // triangular operands (exact: skipped terms are zeros).  bit 0: A lower as stored ([k][i] = 0 for k < i): start at
    // 128 ti; bit 1: B lower ([k][j] = 0 for k < j): start at 128 tj; bit 2: A upper ([k][i] = 0 for k > i): end at
    // 128 (ti + 1); bit 3: B upper: end at 128 (tj + 1).  k_lower (below) = bits 0 and 1.
    int krange;
    int sym;                                            // GAMMA / PLAIN: the product is symmetric: only tiles tj <= ti are
                                                        //    launched, off-diagonal ones are also written mirrored
    int k_lower;                                        // 1: A and B are lower triangular as stored ([k][i] = 0 for i > k): the
                                                        //    sum for tile (ti, tj) starts at row 128 max(ti, tj) (exact: skips zeros)
    // BWD_E, explicit-U branch / LinearK: residuals instead of delta ([nb][nA]), u indexed by latent dim, and no
    // Hadamard product with K_fu (a linear kernel's chain rule multiplies by x and z, not by K)
    const double *rvec;
    int u_per_dim, no_hadamard;
};
void launch_atb(hipStream_t stream, const AtbArgs &a);
int atb_ntiles(int nA, int nB);
int atb_ntiles_sym(int n);      // tiles of a symmetric product launched with AtbArgs::sym
int atb_ntiles_sym64(int n);    // ... with AtbArgs::small_tiles (64 x 64 tiles)

void launch_uku(hipStream_t stream, const double *u, size_t u_stride, const double *K, size_t k_stride, int Mp, int Dl,
                int nb, double *out);
void launch_utu(hipStream_t stream, const double *u, size_t u_stride, int Mp, int nb, double *out);      // out[b] = |u_b|^2
void launch_chain_sum(hipStream_t stream, const double *in, size_t in_stride, int S, int Dl, size_t n, double *out,
                      size_t out_stride);
void launch_psi_e(hipStream_t stream, const double *gsum, const double *kgk, const double *Kcopy, int M, int Mp, int Dl,
                  double jitter, double *Eout, int kind = 0);
// xsq_unit[s * Dl + dl] = sum_t |[X_s[t], ctrl[t]]|^2 (LinearK's Kdiag sums, collapsed branch)
void launch_xsq_unit(hipStream_t stream, const double *X, const double *ctrl, int T, int D, int C, int S, int Dl, double *xsq_unit);
void launch_symmetrize(hipStream_t stream, double *A, int Mp, int batch);
void launch_sub_identity(hipStream_t stream, const double *x, double sc, int Mp, int Dl, double *out);
void launch_axpby(hipStream_t stream, const double *x, const double *z, double a, double bcoef, const double *log_Q,
                  int d_begin, int scale_mode, size_t n, int Dl, double *out);

struct EReduceArgs {
    const double *E; size_t e_stride;       // [nb] slabs of Tp x Mp
    const double *Kf;                       // optional, same layout as E: also produce kfu[t] = sum_m Kf_tm u_m
    const double *u; size_t u_stride;
    int x_is_z;                             // 1: the rows are the inducing inputs themselves (K_uu side)
    int u_per_dim;                          // 1: u is indexed by latent dim (explicit-U branch: beta), else by unit
    int kind;                               // FFVD_KERNEL_*: e_finish applies the SE or the LinearK chain rule
    const double *variance;                 // [Dl] exp(logvariance) (LinearK)
    const double *x; size_t x_chain_stride; int x_ld, x_cols;
    const double *ctrl; int C;
    const double *Z;                        // M x P (unscaled)
    const double *len;                      // [Dl][P]
    int T, Tp, M, Mp, P, Dl, b0, nb, nblk;  // nblk = Tp / 64
    double *rsum, *ez, *kfu;                // [nb][Tp], [nb][Tp][P], [nb][Tp]
    double *cs_part, *etx_part, *rx2_part;  // [nb][nblk][Mp], [nb][nblk][Mp][P], [nb][nblk][P]
    // fp32-contraction backward (dtype FFVD_F32C): E is not materialised in fp64; it is formed on the fly from the fp32 product
    // R = K_fu Gamma and the fp32 K_fu,  E_tm = (2 R_tm + alpha delta_t u_m) K_tm,  every sum accumulated in fp64
    const float *R32, *Kf32;                // [nb] slabs of Tp x Mp (same stride e_stride); R32 != NULL selects this form
    const double *Xd;                       // [S][T+1][D] latent trajectories (delta_t = x_{t+1,d} - x_{t,d})
    const double *log_Q;                    // [D]
    int D, d_begin;
};
void launch_e_reduce(hipStream_t stream, const EReduceArgs &a);
void launch_e_finish(hipStream_t stream, const EReduceArgs &a, double *dz_unit, double *dll_unit, double *dls_unit);

// Backward product with the E reductions fused into its epilogue (P <= 6: slots 0 rsum, 1..P ez, 7 kfu): E = (2 K_fu Gamma + delta (alpha u)^T) o K_fu
// is formed tile by tile in the accumulators and never reaches HBM; every reduction of it runs on the matrix cores.
struct BwdFusedArgs {
    const double *Kf; size_t kf_stride;         // [nb] Tp x Mp, row-major (K_fu)
    const double *Gamma; size_t g_stride;       // [nb] Mp x Mp ([Dl] when per_dim)
    const double *u; size_t u_stride;           // [nb] Mp ([Dl] when per_dim)
    int per_dim;                                // explicit-U branch: Gamma and u are indexed by latent dim, not by unit
    const double *rvec;                         // explicit-U branch: [nb][Tp] residuals used in place of delta
    int linear;                                 // LinearK: E = 2 K_fu Gamma + row (alpha u)^T WITHOUT the Hadamard product with K_fu
    const double *X; const double *ctrl;        // chains S x (T+1) x D; control inputs T x C
    const double *Z;                            // M x P
    const double *log_Q;
    int T, Tp, D, C, M, Mp, P, Dl, d_begin, b0, nb;
    double *rp;                                 // [ntj][nb][Tp][8] row partials per column tile: 0 rsum, 1..P ez, 7 kfu
    double *cs_part, *etx_part;                 // [nb][Tp/64][Mp] and [..][P]: column partials per 64-row block
    double *rsum, *ez, *kfu, *rx2_part;         // outputs of the row combine: [nb][Tp], [nb][Tp][P], [nb][Tp], [nb][Tp/64][P]
};
void launch_bwd_fused(hipStream_t stream, const BwdFusedArgs &a);
size_t bwd_fused_rp_doubles(int Mp, int Tp, int nb);

// Reference-route projection as a GEMM:  F = K_fu W  with W = L^-T upper triangular (conditionals_multi_output.py:242),
// K_fu read from the kfu_build output, k-range cut at the tile's last column (and per wavefront inside the diagonal
// block), row sums of F^2 (the trace term, :255) as per-column-tile partials.
struct ProjGemmArgs {
    const double *Kf; size_t kf_stride;        // [nb] Tp x Mp row-major
    const double *W; size_t w_stride;          // [Dl] Mp x Mp (row k, column j), zero below the diagonal
    double *F; size_t f_stride;                // [nb] Tp x Mp
    double *rowsq;                             // [nb][ntj][Tp], ntj = ceil(Mp / 128)
    int Tp, Mp, Dl, b0, nb;
    // explicit-U branch: F itself is not needed (F == nullptr), but its product with the inducing outputs is:
    const double *u; size_t u_stride;          // [Dl] Mp, u_d = U[:, d] (zero padded) or nullptr
    double *fmean;                             // [nb][ntj][Tp] partials of F u (conditionals_multi_output.py:48)
    // collapsed branch, reference route: delta^T F summed where F is made (per 128-row tile), so that the Gram kernel forms no row
    // and may use its fully loaded diagonal workgroups (kernels.h, launch_brow_finish adds the tiles' partials)
    double *gpart;                             // [nb][ceil(Tp / 128)][Mp] or nullptr
    const double *X;                           // chains S x (T+1) x D (delta_t = x_{t+1,d} - x_{t,d}), needed with gpart
    int T, D, d_begin;
};
void launch_proj_gemm(hipStream_t stream, const ProjGemmArgs &a);

struct DxArgs {
    int kind;                               // FFVD_KERNEL_*
    const double *variance;                 // [Dl] (LinearK: d/dx of K_fu and of Kdiag = variance |x|^2)
    const double *X, *Y, *CC, *DD, *log_Rchols, *log_Q, *len;
    const double *rsum, *ez, *kfu;
    int S, S_total, T, Tp, D, P, Ydim, Dl, d_begin, shared_terms;
    double *dX;
    // T-shard handles (cfg.T_total > 0): the handle holds rows [t_begin, t_begin + T) of a job with T_norm transitions -- every 1 / T
    // of dgp_model.py:261-297 is the job's (0 = use T), and X[0] is the job's x_0 (prior_x_0) on the first shard only
    int T_norm, skip_x0;
};
void launch_dx(hipStream_t stream, const DxArgs &a);
void launch_shared_partials(hipStream_t stream, const DxArgs &a, double *out, int stride);

struct GradFinalArgs {
    int T, D, P, M, Mp, Ydim, Dl, d_begin, S, S_total, shared_terms, prior_type;
    const double *Z, *logvar, *loglen, *log_Q, *CC, *DD, *log_Rchols;
    const double *dz_unit, *dll_unit, *dls_unit;      // K_fu side per unit
    const double *dz_kuu, *dll_kuu, *dls_kuu;          // K_uu side per latent dim
    const double *gam_part; int ngam;                  // tr(A^-1 K) partials per unit
    const double *trpart; int ntr;                     // tr(K^-1 G) partials per unit (forward)
    const double *hterms, *uku;                        // forward {logdet, quad}; u^T K u per unit
    const double *shared_part; int sp_stride;          // per-chain likelihood / transition partials
    double *dZ, *dlogvar, *dloglen, *dlogQ, *dCC, *dDD, *dlogR;
    // explicit-U branch (branch_a != 0): dl/dalpha per unit replaces the collapsed formula, the transition part of the
    // shared partials is not added (it lives in dalpha), and dU is assembled from du_dim = W^T g_r (per dim)
    int branch_a;
    const double *dalpha_unit, *du_dim, *U;
    double *dU;
    int kind;                                          // FFVD_KERNEL_*
    const double *xsq_unit;                            // LinearK: [nb] sum_t |x_comb_t|^2 (Kdiag = variance |x|^2 enters the trace term)
    // T-shard handles: T_norm = the job's transitions (0 = T).  Every shard finishes the SAME all-reduced Gram matrices, so the terms
    // that do not sum over this shard's own rows -- the K_uu side, the G-dependent part of dl/dalpha, the priors -- are identical on
    // every shard: replicated_skip != 0 (all shards but the first) leaves them out, and the sum over shards counts them once.
    int T_norm, replicated_skip;
};
void launch_grad_finalize(hipStream_t stream, const GradFinalArgs &a);

// ---- explicit-U branch helpers (see grad.hip) ----
void launch_ucols(hipStream_t stream, const double *U, int M, int Mp, int D, int d_begin, int Dl, double *ucol);
void launch_resid_a(hipStream_t stream, int kind, const double *X, const double *ctrl, int C, const double *fmean,
                    const double *rowsq, const double *variance, const double *log_Q, int T, int Tp, int D, int Dl, int d_begin,
                    int ng, int nb, double *r, double *dalpha_unit, double *xsq_unit);
void launch_scale_kinv(hipStream_t stream, const double *Kinv, const double *log_Q, int Mp, int Dl, int d_begin, double *out);
void launch_dw_a(hipStream_t stream, const double *T1, const double *grs, size_t grs_stride, const double *ucol,
                 const double *log_Q, int Mp, int Dl, int d_begin, double *dW);
void launch_tril_neg(hipStream_t stream, const double *P, int Mp, int Dl, double *out);
void launch_tril_copy(hipStream_t stream, const double *L, size_t l_stride, int Mp, int Dl, double *out);
void launch_phi(hipStream_t stream, const double *S, int Mp, int Dl, double *Phi);
void launch_epsi_a(hipStream_t stream, int kind, const double *dK, const double *Kcopy, int M, int Mp, int Dl, double jitter,
                   double *E);

}  // namespace ffvd
