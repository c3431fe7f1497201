// Device-side launchers of the FFVD ELBO engine (gfx950 / CDNA4, fp64).
// Every function only ENQUEUES work on `stream`; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffvd {

constexpr int NB = 64;        // universal block size: M is padded to a multiple of NB (identity padding)
constexpr int DINV_STRIDE = 2 * 4 * 16 * 16;   // Cholesky scratch per matrix: inverses of the four 16x16 diagonal sub-blocks of the current
                                               // block step, two slots (the look-ahead of step j writes slot (j+1)&1 while step j reads slot j&1)
constexpr int STRIP = 64;     // rows of K_fu handled by one workgroup of the projection kernel
// Centre of the prior on logvariance (Layer.prior_hyper, dgp_model.py:123-130).  SquaredExponential (:127) writes
// tf.cast(tf.math.log(0.05), tf.float64): the logarithm is taken in float32 and then widened, i.e. -2.995732307434082
// (the fp64 logarithm is -2.995732273553991); LinearK (:130) writes np.log(0.05), the fp64 value.
constexpr double LOG_PRIOR_VARIANCE_SE = -2.995732307434082;     // double(float32 log(float32 0.05)), exactly representable
constexpr double LOG_PRIOR_VARIANCE_LIN = -2.995732273553991;    // np.log(0.05)
constexpr int MAXP = 32;      // largest GP input dimension P = D + C supported by the LDS layout

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Per-latent-dim kernel hyper-parameters prepared on device (prep_hypers).
struct HyperView {
    const double *variance;   // [Dl]     exp(logvariance)
    const double *len;        // [Dl][P]  exp(loglengthscales) (SE) or 1 (LINEAR)
    const double *Zs;         // [Dl][Mp][P] inducing inputs divided by lengthscales (rows >= M are 0)
    const double *zz;         // [Dl][Mp]    |Zs_m|^2
};

// hyp <- exp of the log-parameters; Zs, zz as above.  logvar/loglen are indexed by GLOBAL dim (d_begin + dl).
void launch_prep_hypers(hipStream_t stream, int kind, const double *Z, int M, int Mp, int P, int Dl, int d_begin,
                        const double *logvar, const double *loglen, double *variance, double *len,
                        double *Zs, double *zz, int32_t *info = nullptr, int ninfo = 0);   // info: flags to zero

// A[dl] (ld = Mp, rows 0..Mp-1) = K_dl(Z, Z) + jitter*I with identity padding; rows Mp..2Mp-1 = I (the
// "extra rows" that the extended Cholesky turns into L^{-T}).  A has Dl slabs of 2*Mp*Mp doubles.
void launch_kuu_build(hipStream_t stream, int kind, HyperView hv, int M, int Mp, int P, int Dl, double jitter, double *A,
                      double *Kcopy /* optional [Dl][Mp*Mp] copy that survives the factorisation */,
                      bool zt_rows = false /* LINEAR, P <= 64: rows Mp..Mp+63 = Z^T (rows >= P zero) instead of the identity: they
                                              become C = Z^T L^-T (launch_potrf_ext with 64 extra rows, none identity-structured) */,
                      double *flow_words = nullptr /* the scratch block of the dataflow Cholesky that follows on the SAME stream for
                                              these Dl matrices: its progress words are zeroed here (then: words_zeroed = true) */);
void launch_transpose(hipStream_t stream, const double *in, size_t in_stride, double *out, size_t out_stride, int Mp,
                      int Dl);

// General kernel matrix for the operator API: out[N x N2] = K(X, X2) for ONE kernel (dl = 0 of hv is not used;
// lengthscale scaling is applied on the fly).  diag_jitter added where row == col if `same`.
void launch_kernel_matrix(hipStream_t stream, int kind, const double *X, int N, const double *X2, int N2, int P,
                          double logvar, const double *loglen_dev, double jitter, int same, double *out);
void launch_kernel_diag(hipStream_t stream, int kind, const double *X, int N, int P, double logvar, double *out);

// Extended blocked Cholesky, in place, batched:  each slab holds an n x n SPD matrix (n multiple of NB, lower
// triangle referenced) followed by `extra_rows` further rows R (same ld = n).  On exit the lower triangle holds
// L and the extra rows hold R * L^{-T}.  identity_extra != 0 declares that R is the n x n identity on entry
// (so R L^{-T} = L^{-T} is upper triangular and zero blocks are skipped).  info[b] = 0 or 1 + first bad pivot.
// identity_rows: how many of the LEADING extra rows form an identity on entry (multiple of NB, 0 = none).
// dinv: device scratch of potrf_scratch_doubles(n, batch) doubles (inverted 16x16 diagonal sub-blocks of the block steps and,
// for the one-launch dataflow variant, its progress words).  hint = CHOL_FLOW asks for the dataflow variant whatever the batch
// (the factorisation is on the caller's critical path); CHOL_AUTO picks by batch size.
enum { CHOL_AUTO = 0, CHOL_FLOW = 3 };
// force a variant for the launches this THREAD enqueues from now on: 0 = back to the normal choice, 1 = left-looking launches per
// block column, 2 = right-looking launches (neither has inter-workgroup waits); used to re-run a batch whose dataflow launch gave up
enum { CHOL_FORCE_NONE = 0, CHOL_FORCE_LEFT = 1, CHOL_FORCE_RIGHT = 2 };
void launch_spin(hipStream_t stream, int us);
void potrf_override_variant(int variant);
int potrf_override_current();
size_t potrf_scratch_doubles(int n, int batch);
// linv_t (optional, honoured by the dataflow variant only -- ask potrf_flow_selected): the identity-structured extra rows
// are ALSO written transposed, L^-1 as an n x n lower-triangular matrix (ld n) per slab of linv_t_stride doubles; the blocks
// above its diagonal are left alone (keep them zero).
bool potrf_flow_selected(int n, int batch, int hint);
void launch_potrf_ext(hipStream_t stream, double *A, int n, int extra_rows, int identity_rows, int batch,
                      size_t slab_stride, int32_t *info, double *dinv, int hint = CHOL_AUTO, double *linv_t = nullptr,
                      size_t linv_t_stride = 0, bool words_zeroed = false, bool tail_is_vector = false, double *kinv = nullptr,
                      size_t kinv_stride = 0, const double *lt_rows = nullptr, size_t lt_stride = 0, int lt_dl = 1,
                      bool kinv_help = false);      // (with kinv, dataflow variant, whole launch resident: the main-row workgroups stay and form half of the inverse's tiles --
                                                    //  for a chain that runs beside little else; beside the full batch's K_fu build they were in its way: 0.062 vs 0.053 ms)
// lt_rows (dataflow variant only -- ask potrf_flow_selected; the others read the rows from memory, arm them with
// launch_set_lt_rows): the identity-structured extra rows of slab b START as L_d^T, d = b % lt_dl, read straight from the
// lower-triangular factor L_d (n x n, ld n, slabs of lt_stride doubles) -- the rows in memory are not read, only written (and
// their blocks left of the diagonal not at all: keep them zero).
// kinv (dataflow variant with identity_rows == n, i.e. all of L^-T, and n <= 2048 -- ask potrf_flow_forms_inverse): the launch also
// leaves A^-1 = L^-T L^-1 (n x n, ld n, both triangles) in every slab of kinv.
bool potrf_flow_forms_inverse(int n, int batch, int hint);
// tail_is_vector (dataflow variant only; ignored by the others, which treat every extra row alike): of each 64-row block of
// extra rows BEHIND the identity rows only the first row is live (the ELBO's row b); the other 63 are neither read nor written.
// The dataflow variant polls progress words at the start of `dinv`; they must be zero when its kernel starts.  The launcher
// clears them itself (one memset in front of the kernel) unless the caller has done so earlier with potrf_flow_clear on a
// stream whose order reaches the launch, and says so (words_zeroed).
void potrf_flow_clear(hipStream_t stream, double *dinv, int batch);
void launch_set_identity(hipStream_t stream, double *A, size_t slab_stride, int row0, int n, int batch);
// rows [row0, row0 + n) of slab b <- L_d^T, d = b % Dl (upper triangular; L_d = lower-triangular n x n factor, ld n, n % 64 == 0)
void launch_set_lt_rows(hipStream_t stream, const double *L, size_t l_stride, int Dl, double *A, size_t slab_stride, int row0,
                        int n, int batch);
// R <- R L^-T for `extra_rows` rows (multiple of NB) stored below a GIVEN lower-triangular factor L (n x n, ld n) in
// every slab: the wavefront-level blocked substitution of the Cholesky panel step on its own.
void launch_trsm_ext(hipStream_t stream, double *A, int n, int extra_rows, int batch, size_t slab_stride, double *dinv);

struct ProjectArgs {
    int kind;
    // GP inputs x_t = [ x[chain][t][0:x_cols] | ctrl[t][0:C] ], P = x_cols + C  (dgp_model.py:269)
    const double *x;        // rows of the first x_cols input columns
    size_t x_chain_stride;  // doubles between chains ((T+1)*D for Layer.X)
    int x_ld, x_cols;
    const double *ctrl;     // [T][C] control inputs (may be null when C == 0)
    int T, Tp, C, P, M, Mp, Dl, d_begin;
    HyperView hv;
    const double *W;        // [Dl] slabs: W[dl] = L^{-T} (Mp x Mp, upper), slab stride w_stride doubles
    size_t w_stride;
    const double *U;        // [M][u_ld] whitened inducing outputs (branch A), column d_begin + dl; or null
    int u_ld;
    int b0, nb;             // batches [b0, b0+nb): b = s*Dl + dl
    double *F;              // [nb][Tp][Mp] or null
    double *rowsq;          // [nbatch_total][ng][Tp] or null   (sum_j F[t][j]^2 per column group)
    double *fmean;          // [nbatch_total][ng][Tp] or null   (sum_j F[t][j] * U[j][d])
    int ng;                 // column groups = ceil(Mp / 512)
    // kfu_build only, optional: per 64-row block the partial sums of delta^T K_fu (delta_t = x_{t+1,d} - x_{t,d} of the unit's own
    // latent dim; conditionals_multi_output.py:247-248), [nb][Tp / 64][Mp]; launch_brow_finish adds the blocks in fixed order.
    // (Round 3: the Gram kernel used to form this row beside its matrix work -- in idle wavefronts of the diagonal tiles; with
    // the diagonal tiles dealt to fully loaded workgroups there are none, and the vector FMAs cost it 0.47 ms.)
    double *gpart;
};
// H[bz] row `brow` (ld Mp) = (yn_over_batch / Q_d) * sum over the 64-row blocks of gpart (fixed order)
void launch_brow_finish(hipStream_t stream, const double *gpart, int nblk, int Mp, int Dl, int d_begin, int b0, int nb,
                        const double *log_Q, double yn_over_batch, double *H, size_t h_stride, int brow);
// F = K_fu * L^{-T} with K_fu generated on the fly (never stored).
void launch_project(hipStream_t stream, const ProjectArgs &a);
// LinearK, explicit-U branch, forward only (kernels.hip, "the projection through the kernel's rank"): fills a.rowsq / a.fmean with ONE
// column group per unit ([nbatch][Tp]) from x, ctrl, hv, U and a.W = the C rows the K_uu chain left behind (launch_kuu_build zt_rows;
// slab stride a.w_stride) -- no F, no K_fu.  `part`: linear_lowrank_doubles(Mp, Dl, P) doubles.
bool linear_lowrank_supported(int kind, int P);
size_t linear_lowrank_doubles(int Mp, int Dl, int P);
void launch_linear_lowrank(hipStream_t stream, const ProjectArgs &a, double *part);
// a.F[bz][t][m] = K_fu itself (route K_uu + K_uf K_fu / Q); uses x, ctrl, hv, T, Tp, M, Mp, b0, nb of `a`.
void launch_kfu_build_t(hipStream_t stream, const ProjectArgs &a, int ldt);      // a.F <- KT[nb][Mp][ldt]: a step's K(x, Z), m-major
void launch_kfu_build(hipStream_t stream, const ProjectArgs &a, int streaming = -1 /* -1: by output size, 0 / 1: cacheable / streaming stores */);

// GRAM_KFU_RAW: as GRAM_KFU, but the trace partials are left to a later trace-only pass (phase 3) over the raw tiles
// this launch also writes into `part` (ksplit = 1 layout) -- K^-1 is not read
enum { GRAM_F = 0, GRAM_KFU = 1, GRAM_PLAIN = 2, GRAM_KFU_RAW = 3 };
struct GramArgs {
    int mode;               // GRAM_*
    const double *A;        // [nb] slabs of rows x Mp (F, K_fu or L^-1), slab stride a_stride doubles
    size_t a_stride;
    int rows;               // rows of A summed over (multiple of 16; Tp for F / K_fu)
    int with_row;           // 1: also write the extra row `brow` = scale * delta^T A (needs X, T, D)
    int brow;               // row index of that extra row in the output slab (0 = default Mp)
    const double *X;        // [S][T+1][D]
    const double *rvec;     // optional [nb][rows]: the extra row is rvec^T A instead of delta^T A
    const double *log_Q;    // [D] (global dim index)
    int T, D, Mp, Dl, d_begin;
    int b0, nb;             // batches [b0, b0 + nb): b = s*Dl + dl
    double yn_over_batch;   // Y_N / batch_size (== 1 for the full batch)
    double *H;              // [nb] output slabs (ld Mp), lower-triangular tiles; slab stride h_stride
    size_t h_stride;
    const double *Kadd;     // GRAM_KFU: [Dl] K_uu + jitter I (ld Mp), slab stride kadd_stride
    size_t kadd_stride;
    const double *Kinv;     // GRAM_KFU: [Dl] (K_uu + jitter I)^-1 (ld Mp), slab stride kinv_stride
    size_t kinv_stride;
    double *trpart;         // GRAM_KFU: [nbatch_total][ntiles] partial sums of tr(K^-1 K_uf K_fu)
    int ntiles;             // filled by launch_gram
    // split-K for launches with few tiles: the rows are cut into `ksplit` ranges, every range writes its raw partial
    // sums into its own slab of `part` ([ksplit][nb] slabs of (Mp + 1) x Mp, row Mp = the delta^T A partials), and
    // gram_combine adds them in fixed order and applies the epilogue.  ksplit <= 1 or part == nullptr: off.
    int ksplit;
    double *part;
    double *Hcopy;          // GRAM_KFU / GRAM_KFU_RAW (unsplit launch), optional: second copy of the result's lower triangle (slab stride hcopy_stride, ld Mp) --
    size_t hcopy_stride;    // the backward pass keeps A = K_uu + K_uf K_fu / Q, which the factorisation overwrites
    // combine pass only: 0 = epilogue + trace partials, 1 = epilogue without the trace (K^-1 not read), 2 = trace partials
    // only (H not written) -- lets the combine run before K^-1 exists and the trace follow on another stream
    int trace_mode;
    int raw_summed;     // split-K combine: the pass without the trace (trace_mode 1) leaves the summed raw tile in partial 0, the trace pass (2) reads only that -- the caller orders the two passes
    // Launches without a delta^T A row, Mp a multiple of 256: (1) the diagonal tiles of every two neighbouring column panels are
    // ONE workgroup ("pair combo", gram_pair_role, round 4: row blocks i and 7 - i of a diagonal tile on one wavefront, nine MFMA
    // tiles, nothing above the diagonal executed; also the diagonal workgroups of split-K launches).  Round 3's form -- three
    // workgroups of eight 64 x 32 sub-blocks per four diagonal tiles, gram_combo_body -- is kept behind GRAM_COMBO=1 for A/B
    // builds.  wg_per_unit = workgroups per unit (filled by launch_gram);
    // (2) unsplit launches: the workgroups of the LAST, partial round of the launch are cut into two row halves that run side by side on the
    // CUs that free up first; the half that finishes second adds the other's accumulators (two addends: the sum does not
    // depend on which one that is) and runs the epilogue.  tail_wg = workgroups cut (multiple of 8, 0 = off), tail_part =
    // [tail_wg][2] blocks of GRAM_TAIL_DOUBLES followed by [tail_wg] arrival counters (zero between launches: the second
    // arrival re-arms its counter).
    int combo, wg_per_unit;
    int tail_wg;
    double *tail_part;
    int *tail_cnt;
};
int gram_ntiles(int Mp);
// how many row ranges launch_gram should use for `nb` units (1 = no split), and the doubles `part` then needs
int gram_ksplit(int Mp, int nb, int rows, int with_row, bool fill_slots = false);      // fill_slots: nothing runs beside the launch -- the row ranges that make it whole rounds of the chip's 512 slots
size_t gram_part_doubles(int Mp, int nb, int ksplit);
// phase 0: everything; 1: tile pass only; 2: combine pass only; 3: trace-only combine pass (split-K launches);
// 4: combine pass (epilogue + trace) over `part` whatever ksplit is (T-shards: the all-reduced raw tiles, ksplit = 1)
constexpr int GRAM_TAIL_DOUBLES = 38 * 512;       // per thread: up to 36 accumulator values (the pair combos' nine tiles) + two partial sums of the delta^T A row
// workgroups of the last partial round of an unsplit launch over nb units that gram_kernel cuts in two (0 = none), and the
// scratch a handle needs for them (doubles, counters included)
int gram_tail_wg(int Mp, int nb, int ksplit, int with_row);
size_t gram_tail_doubles(int tail_wg);
void launch_gram(hipStream_t stream, GramArgs a, int phase = 0);

// hterms[b] = { logdet(H) = 2 sum log diag(L_H),  b^T H^{-1} b = |row Mp|^2 } after launch_potrf_ext on H.
void launch_h_finish(hipStream_t stream, const double *H, int Mp, size_t h_stride, int nb, double *hterms /*[nb][2]*/,
                     int yrow = 0 /* row holding L^-1 b; 0 = Mp */);

struct ReduceArgs {
    int kind, branch;
    const double *X, *ctrl, *Y;    // X [S][T+1][D]; ctrl [T][C]; Y [T][Ydim]
    const double *log_Q, *CC, *DD, *log_Rchols, *variance /*[Dl]*/;
    int T, Tp, D, C, Ydim, Dl, d_begin, S, ng, shared_terms;
    // rows whose LinearK.Kdiag enters the trace term: [ xk[chain][t][0:xk_cols] | ctrl[t][0:C] ]
    const double *xk;
    size_t xk_chain_stride;
    int xk_ld, xk_cols;
    const double *rowsq, *fmean;   // [S*Dl][ng][Tp]
    double *chain_terms;           // [S][8] partial sums per chain
    int skip_x0;                   // T-shards other than the first: X[0] is not the job's x_0, leave prior_x_0 out
    const int32_t *info;           // optional (T-shard local phase, where the chain sums are part of the exchange buffer): factorisation flags
    int ninfo;                     //   already final on this stream; any non-zero one turns the chain sums into NaN
};
// partial: scratch of S * 8 * 4 doubles
void launch_chain_reduce(hipStream_t stream, const ReduceArgs &a, double *partial);

struct FinalizeArgs {
    int kind, branch, prior_type, shared_terms;
    int T, D, P, M, Ydim, Dl, d_begin, S;
    const double *Z, *U, *logvar, *loglen, *log_Q, *CC, *DD, *log_Rchols;
    const double *chain_terms;     // [S][8]
    const double *hterms;          // [S*Dl][2] (branch B) or null
    int route;                     // 0: F = K_fu L^-T route, 1: K_uu + K_uf K_fu / Q route
    const double *kterms;          // route 1: [Dl][2], kterms[2 dl] = log det (K_uu + jitter I)
    int whitened;                  // route 1, training: hterms come from H = L^-1 A L^-T, whose log det is already
                                   //   log|A| - log|K_uu + jitter I|
    const double *trpart;          // route 1: [S*Dl][ntiles] partial sums of tr(K^-1 K_uf K_fu)
    int ntiles;
    int fsq_from_trpart;           // route 0, fp32-contraction path: trpart[S*Dl][ntiles] = sum_t |F_t|^2 (no row sums)
    double *chain_nll;             // [S]
    double *out_terms;             // [8]
    const double *prior_sums;      // optional [10]: the parameter-only sums of the assembly (finalize_priors), formed earlier in the iteration by
                                   //   launch_prior_sums -- the finalize launch at the iteration's tail then only assembles
    const int32_t *info;           // optional: the iteration's factorisation flags ([ninfo]: K_uu per local dim, then one per unit).  Any
    int ninfo;                     //   non-zero flag (bad pivot, or -1: a dataflow launch gave up on a bounded wait and left finite garbage)
                                   //   turns the seven sums into NaN, so that an all-reduce carries the failure to EVERY rank
};
void launch_finalize(hipStream_t stream, const FinalizeArgs &a);
// the ten parameter-only sums of the nll assembly (priors, log R, log sqrt Q) into out[10]: a function of the parameters alone
void launch_prior_sums(hipStream_t stream, const FinalizeArgs &a, double *out);

// conditional() epilogue: mean[n][d] = sum_g fmean, var[n][d] = Kdiag(x_n) - sum_g rowsq  (N x D outputs)
void launch_conditional_finish(hipStream_t stream, int kind, const double *x, int N, int P, const double *variance,
                               const double *rowsq, const double *fmean, int ng, int Tp, int D, double *mean,
                               double *var, const double *extra /* optional [D][extra_ng][Tp], its groups added to var */,
                               int extra_ng = 1);
// C[b] (rows x N) = A[b] (rows x K) B for a FEW rows (step loops), with the per-16-column partial sums of C^2 and C u:
// sq / dot [nb][N / 16][Tp] (feed launch_conditional_finish with ng = N / 16).  kernels.hip has the details.
void launch_skinny_gemm(hipStream_t stream, const double *A, size_t a_stride, int lda, const double *B, size_t b_stride, int ldb,
                        int upper, int rows, int K, int N, int nb, int Tp, double *C, size_t c_stride, int ldc,
                        const double *u, size_t u_stride, double *sq, double *dot,
                        // optional second right-hand side B2 (K x N2, dense) in the same launch: only sq2[nb][N2 / 16][Tp] is formed
                        const double *B2 = nullptr, size_t b2_stride = 0, int ldb2 = 0, int N2 = 0, double *sq2 = nullptr,
                        int a_trans = 0 /* A is k-major AT[nb][K][lda], as launch_kfu_build_t writes it */,
                        int upper2 = 0 /* B2 is upper triangular too */);
// out[i * out_ld + b * out_bs] = sum_j W[b][i][j] y[b][j]
void launch_matvec(hipStream_t stream, const double *W, size_t w_stride, const double *y, size_t y_stride, int Mp,
                   double *out, int out_ld, int out_bs, int M, int batch, int w_mod = 0);   // w_mod > 0: W slab index = batch index % w_mod
void launch_qsqrt_inflation(hipStream_t stream, const double *F, size_t f_stride, int Tp, int Mp, int M, const double *Qs,
                            double *extra, int N, int batch);

// operator-API elementwise kernels
void launch_predict_mean(hipStream_t stream, const double *X, int N, int D, const double *CC, const double *DD, int J,
                         double *out);
void launch_logdensity(hipStream_t stream, int mode, const double *y, const double *ymean, const double *R, int N, int J,
                       double *out);
void launch_get_rand(hipStream_t stream, const double *mean, const double *var, const double *eps, size_t n, double *out);

// small utilities
void launch_fill(hipStream_t stream, double *p, size_t n, double v);

// rollout step: x_next = x + f_mu + eps sqrt(f_var + Q) (base_model.py:304-314); xc: R x (D + C) GP input rows,
// advanced in place; ctrl_next: the C control inputs of the NEXT step or null.
void launch_rollout_update(hipStream_t stream, const double *mean, const double *var, const double *log_Q,
                           const double *eps_t, const double *ctrl_next, int R, int D, int C, int t, int steps, double *xc,
                           double *predict_x, double *predict_var);

// One particle-Gibbs step for R free particles (base_model.py:99-115, intent; oracle/ffvd_pg_oracle.py): propagate,
// weight against y_t, draw R ancestors among the R new particles + the reference state, gather.  xc: R x (D + C) GP input
// rows, advanced in place; cand: (R + 1) x D scratch; parts_next: R x D; idx_out: R.
void launch_pg_step(hipStream_t stream, const double *mean, const double *var, const double *log_Q, const double *eps_t,
                    const double *unif_t, const double *y_t, const double *x_ref_next, const double *CC, const double *DD,
                    const double *Rch, const double *ctrl_next, int R, int D, int C, int Ydim, double *xc, double *cand,
                    double *parts_next, int32_t *idx_out);
// rollout step with the conditional epilogue folded in (same arithmetic as launch_conditional_finish + launch_rollout_update)
void launch_rollout_finish_update(hipStream_t stream, int kind, const double *variance, const double *rowsq, const double *fmean,
                                  int ng, int Tp, const double *extra, int extra_ng, const double *log_Q, const double *eps_t,
                                  const double *ctrl_next, int R, int D, int C, int t, int steps, const double *x_in, double *x_out,
                                  double *predict_x, double *predict_var);     // x_in != x_out: the caller alternates two R x (D + C) buffers

// N sums over a 256-thread workgroup at once: wavefront shuffles, then the four wavefront partials in fixed order.
// Every thread returns with the totals in v[].
template <int N>
__device__ __forceinline__ void block_sum_multi_256(double (&v)[N], double (*scratch)[N] /*[4][N]*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v[i] += __shfl_xor(v[i], m);
    }
    __syncthreads();                                   // a previous use of scratch is over
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) scratch[wave][i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ((scratch[0][i] + scratch[1][i]) + scratch[2][i]) + scratch[3][i];
}

}  // namespace ffvd
