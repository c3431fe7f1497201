/*
 * ffvd_abi.h -- C ABI of the MI355X-native FFVD ELBO engine (libffvd_hip.so).
 *
 * The reference (xuhuifan/FFVD) has no FFI: its "model/ELBO callable" is the
 * TensorFlow tensor `DGPSSM.nll` (vfegpssm/dgp_model.py:248-297) evaluated by
 * `session.run` (vfegpssm/base_model.py:952-989).  This header is the boundary
 * a maintainer binds instead (ctypes stub in INTEGRATION.md): plain pointers
 * and sizes, row-major contiguous fp64 arrays, no torch / numpy types.
 *
 * Conventions
 *   - status: 0 = OK, <0 usage/runtime error, >0 numerical error (FFVD_ENOTPD);
 *     the message is retrievable with ffvd_last_error(); nothing aborts or throws.
 *   - a handle = one device + one HIP stream + all device workspace (allocated in
 *     ffvd_create, nothing is allocated on the hot path).  Not thread-safe;
 *     distinct handles are independent.
 *   - pointer arguments are HOST pointers unless the call has an `on_device`
 *     flag / FFVD_PARAMS_ON_DEVICE, in which case they are device pointers used
 *     in place (zero copy).  Caller owns every buffer it passes in.
 *   - calls that write host outputs synchronise the handle's stream; the
 *     `_async` forms only enqueue.
 */
#ifndef FFVD_ABI_H
#define FFVD_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFVD_OK        0
#define FFVD_EINVAL   (-1)   /* bad shape / null pointer / unsupported option */
#define FFVD_ENOMEM   (-2)   /* device or host allocation failed */
#define FFVD_EDEVICE  (-3)   /* HIP runtime error (message in ffvd_last_error) */
#define FFVD_ENOTPD     1    /* Cholesky met a non-positive pivot (which matrix/pivot: ffvd_last_error) */

#define FFVD_F64  0   /* everything in fp64 (the reference's dtype: every variable is tf.float64, dgp_model.py:64-69) */
/* fp32 CONTRACTIONS (BASELINE configs[3]; SURVEY 7 "Conditioning"): K_fu (conditionals_multi_output.py:240) and the two
 * T x M x M products F = K_fu L^-T (:242), F^T F (:246) in fp32 on v_mfma_f32_32x32x2_f32; K_uu, every M x M
 * factorisation and solve (:159-166, :253-254), H from the moment it leaves the matrix cores, delta^T F (:247-248) and
 * the sum of F^2 in the trace term (:255) in fp64.  All inputs and outputs of the ABI stay fp64.  Collapsed branch,
 * FFVD_ROUTE_REFERENCE (the Gram route's error is eps * cond(K_uu): unusable in fp32).
 * Measured against the fp64 oracle: every term and the nll within 1e-5 absolute (worst case 5.5e-6 with M = 1100 inducing
 * points for T = 1400; 2e-7 relative on the nll at T=16384, M=2048) -- tests/test_gpu_f32c.py.
 * With grad = 1 the backward pass forms its T x M x M product K_fu Gamma in fp32 as well (Gamma rounded once; dl/dK_fu is never
 * stored, the reductions run in fp64): gradients within 1e-3 (X) / 1e-2 (Z, kernel hyper-parameters) of the largest entry of each
 * array, measured 9e-5 / 3.5e-3 -- what one fp32 product of K_fu with Gamma ~ alpha |L^-T|^2 gives; config 4 trains with it. */
#define FFVD_F32C 1

#define FFVD_KERNEL_SE      0   /* kernels_multi_output.py:140-247 SquaredExponential (ARD) */
#define FFVD_KERNEL_LINEAR  1   /* kernels.py:250-281 LinearK, one scalar variance per latent dim */

#define FFVD_BRANCH_A 0   /* explicit U:  dgp_model.py:289-297 (regularizer :337-359, conditional) */
#define FFVD_BRANCH_B 1   /* collapsed U: dgp_model.py:267-288 (kernel_pre_cal + collapse_after_kernel_precalculation) */

#define FFVD_PRIOR_UNIFORM 0    /* Layer.prior_Z dgp_model.py:106-107 */
#define FFVD_PRIOR_NORMAL  1    /* dgp_model.py:108-109 */

/* How the collapsed bound (branch B) is evaluated.  Both are the same algebra:
 *   REFERENCE: F = K_fu L^-T, H = F^T F / Q + I, exactly the reference's op order (conditionals_multi_output.py:242-255)
 *   GRAM     : log|H| = log|K_uu + K_uf K_fu / Q| - log|K_uu|, b^T H^-1 b = g^T (K_uu + K_uf K_fu / Q)^-1 g / Q^2,
 *              sum_t |F_t|^2 = tr(K_uu^-1 K_uf K_fu); about half the flops, K_fu L^-T is never formed.
 *              Rounding differs: nll agrees with REFERENCE to ~1e-9 relative on the benchmark workloads.   */
#define FFVD_ROUTE_REFERENCE 0
#define FFVD_ROUTE_GRAM      1

#define FFVD_PARAMS_ON_DEVICE 1u

/* indices into the 8-double term vector written by ffvd_elbo*()               */
#define FFVD_TERM_PART_PRIOR   0   /* nll_part_prior            dgp_model.py:286/296 */
#define FFVD_TERM_LOG_LIK      1   /* nll_log_likelihood        dgp_model.py:264     */
#define FFVD_TERM_X_PRIOR_Q    2   /* x_t_prior_Q               dgp_model.py:283/294 */
#define FFVD_TERM_TRACE        3   /* nll_reg_trace_inverse_Q_B dgp_model.py:275/292 */
#define FFVD_TERM_LATER1       4   /* later_term1 (branch B)    dgp_model.py:275     */
#define FFVD_TERM_LATER2       5   /* later_term2 (branch B)    dgp_model.py:275     */
#define FFVD_TERM_NLL          6   /* nll                       dgp_model.py:288/297 */
#define FFVD_TERM_COUNT        7   /* number of chains the sums cover (for the mean after an all-reduce) */

typedef struct ffvd_handle ffvd_handle;

typedef struct ffvd_config {
    int32_t T;            /* transitions = len(Y_train); X has T+1 rows            */
    int32_t D;            /* latent dim x_dims[-1] (columns of X)                  */
    int32_t C;            /* control-input dim; GP input dim P = D + C (models.py:51) */
    int32_t M;            /* inducing points                                        */
    int32_t S_local;      /* latent trajectories (posterior samples/chains) this handle evaluates */
    int32_t Ydim;         /* observation dim                                        */
    int32_t d_begin;      /* first latent dim this handle evaluates (shard D: BASELINE config 5) */
    int32_t d_count;      /* number of latent dims evaluated; 0 = all D             */
    int32_t shared_terms; /* 1: also add the terms not tied to a latent dim (likelihood, prior_Z, prior_x_0, hyper prior) */
    int32_t dtype;        /* FFVD_F64 or FFVD_F32C                                  */
    int32_t kernel_kind;  /* FFVD_KERNEL_*                                          */
    int32_t branch;       /* FFVD_BRANCH_*                                          */
    int32_t prior_type;   /* FFVD_PRIOR_*                                           */
    int32_t device_id;    /* HIP device ordinal                                     */
    int32_t chains_per_pass; /* chains whose T x M projections are resident at once; 0 = auto */
    int32_t route;        /* FFVD_ROUTE_*  (branch B only)                          */
    int32_t grad;         /* 1: also allocate the backward-pass workspace (ffvd_elbo_grad) */
    int32_t T_total;      /* > 0: this handle holds a T-SHARD -- rows [t_begin, t_begin + T) of a job with T_total
                           * transitions (its X has T + 1 rows starting at global row t_begin); see ffvd_elbo_tshard */
    int32_t t_begin;      /* first global transition of the shard (T_total > 0)     */
    int32_t reserved;
    double  jitter;       /* 1e-5: conditionals_multi_output.py:108,159             */
} ffvd_config;

/* All arrays fp64, row-major, contiguous. */
typedef struct ffvd_params {
    const double *X;               /* S_local x (T+1) x D   Layer.X        dgp_model.py:56-64 */
    const double *Z;               /* M x P                 Layer.Z        dgp_model.py:67    */
    const double *U;               /* M x D (branch A; NULL allowed in B)  dgp_model.py:66    */
    const double *logvariance;     /* D                     kernels_multi_output.py:156       */
    const double *loglengthscales; /* D x P (SE; NULL for LINEAR)  kernels_multi_output.py:160 */
    const double *log_Q;           /* D                     dgp_model.py:182                  */
    const double *CC;              /* D x Ydim              likelihoods.py:19                 */
    const double *DD;              /* Ydim                  likelihoods.py:23                 */
    const double *log_Rchols;      /* Ydim x Ydim           likelihoods.py:54                 */
} ffvd_params;

/* ---- lifetime ----------------------------------------------------------- */
int  ffvd_create(const ffvd_config *cfg, ffvd_handle **out);
int  ffvd_destroy(ffvd_handle *h);
/* message of the most recent failure on `h` (or of the last failed ffvd_create / handle-less op when h == NULL) */
const char *ffvd_last_error(const ffvd_handle *h);
int  ffvd_sync(ffvd_handle *h);
/* How often a synchronous call on this handle (ffvd_elbo, ffvd_elbo_grad, ffvd_adam_step, ffvd_sghmc_step) re-ran its iteration
 * because the one-launch Cholesky gave up on a bounded wait (info = -1): the iteration is then enqueued once more, in process,
 * with the launch-per-column Cholesky, the call returns FFVD_OK and ffvd_last_error holds a warning; only a second failure is
 * FFVD_EDEVICE.  Collective calls do not retry (the other ranks have moved on); instead the failure is made COLLECTIVE: the
 * finalize kernel turns the rank's seven partial sums into NaN whenever one of its factorisation flags is non-zero (bad pivot or
 * abandoned launch), so after the all-reduce EVERY rank sees non-finite sums, returns an error (FFVD_EDEVICE on the rank that
 * stalled, FFVD_ENOTPD elsewhere) and leaves its parameters untouched.  ffvd_tshard_finish (no collective after it) retries. */
int  ffvd_stall_recoveries(const ffvd_handle *h);
/* A stall is remembered: after a recovery the handle's synchronous calls stay on the schedule that has no inter-workgroup waits
 * (multi-kernel iteration, launch-per-column Cholesky) for this many further calls -- 16 after the first recovery -- and then try
 * the fast path again; a probe that stalls again doubles the hold (up to 1024 calls), a clean one resets it.  A co-tenant that keeps
 * compute units busy therefore costs one bounded wait per hold instead of one per call (the reference's only failure mode is an
 * error surfaced at session.run, dgp_model.py:320-324; here the call returns FFVD_OK either way).  ffvd_schedule_name names the
 * state while it lasts.  Collective calls neither retry nor consult the hold. */
int  ffvd_stall_hold(const ffvd_handle *h);
/* Non-zero when the handle evaluates its iteration (and, with grad = 1, the backward pass) as ONE kernel launch: the collapsed-U
 * branch with SquaredExponential kernels in fp64 at the reference's own experiment size (FFVD_Main.py:356-369: M <= 128,
 * P = D + C <= 8, every role's workgroup resident at once; ffvd_amd/csrc/tiny.hip).  The value is the wavefronts per workgroup
 * (4 or 8).  The arithmetic is the reference's op order (F = K_fu L^-T, H = I + F^T F / Q, conditionals_multi_output.py:230-257)
 * whatever cfg.route says.  FFVD_NO_TINY=1 in the environment keeps the multi-kernel schedule. */
int  ffvd_single_launch(const ffvd_handle *h);
/* One line naming the launch schedule the handle's iteration runs (decided from the configuration in ONE place, plan_schedule in
 * abi.hip; "INVALID" if its consistency check ever fails -- the ELBO entry points then return FFVD_EINVAL instead of a wrong
 * number).  Diagnostics: FFVD_DEBUG_SIDE_DELAY_US / FFVD_DEBUG_MAIN_DELAY_US = n (read at ffvd_create) put a spin kernel of n
 * microseconds at the head of the side / main stream at every fork; results must be bit-identical with and without. */
const char *ffvd_schedule_name(const ffvd_handle *h);
/* bytes of device workspace owned by the handle */
int64_t ffvd_workspace_bytes(const ffvd_handle *h);

/* ---- data and parameters (resident copies owned by the handle) ------------ */
/* Y: T x Ydim observations (base_model.py:14); control_inputs: at least T x C rows, the first T are used (dgp_model.py:255). */
int  ffvd_set_data(ffvd_handle *h, const double *Y, const double *control_inputs, int on_device);
int  ffvd_set_params(ffvd_handle *h, const ffvd_params *p, int on_device);

/* ---- the hot path ---------------------------------------------------------- */
/*
 * One ELBO iteration = DGPSSM.nll and its component tensors (dgp_model.py:248-297) for the
 * S_local trajectories of this handle.  p == NULL: use the resident parameters.  Otherwise p is
 * uploaded first (host pointers) or used in place (flags & FFVD_PARAMS_ON_DEVICE).
 * out_terms[0..6] = SUMS over the local chains of the per-chain terms, out_terms[7] = chain count
 * (so that an all-reduce(sum) over ranks followed by a division yields the mean);
 * out_nll = out_terms[6] / out_terms[7] (mean nll over the local chains).
 */
int  ffvd_elbo(ffvd_handle *h, const ffvd_params *p, uint32_t flags, double out_terms[8], double *out_nll);
/* enqueue only; the 8 doubles are written to device memory `out_terms_dev` (e.g. the buffer a
 * collective library all-reduces).  ffvd_sync() or a later synchronous call reports numerical errors. */
int  ffvd_elbo_async(ffvd_handle *h, double *out_terms_dev);
/* Gradient of the mean-over-chains nll w.r.t. every parameter (what the reference gets from tf.gradients(nll, vars),
 * base_model.py:148, and AdamOptimizer.minimize(nll), dgp_model.py:303-305).  Needs a handle created with
 * grad = 1: either branch with either kernel (collapsed branch: either route in fp64; FFVD_F32C with the SE kernel only)
 * (LinearK: the loglengthscales gradient is identically zero -- the kernel has none).  Host output pointers with
 * the shapes of ffvd_params; any of them may be NULL.  S_total = number of chains of the whole job (the divisor of
 * the mean).  Sharded jobs: every output is this handle's ADDITIVE share of the whole-job gradient -- entries of
 * dims it does not own are zero, prior gradients are weighted S_local / S_total (and the shared ones only added
 * where shared_terms = 1) -- so summing the shared-parameter gradients over ranks gives the whole-job gradient; X
 * gradients are complete per rank for chain shards and must be summed too for latent-dim shards. */
typedef struct ffvd_grads {
    double *X;               /* S_local x (T+1) x D */
    double *Z;               /* M x P               */
    double *logvariance;     /* D                   */
    double *loglengthscales; /* D x P               */
    double *log_Q;           /* D                   */
    double *CC;              /* D x Ydim            */
    double *DD;              /* Ydim                */
    double *log_Rchols;      /* Ydim x Ydim         */
    double *U;               /* M x D: explicit-U branch; zeros in the collapsed branch (U integrated out) */
} ffvd_grads;
int  ffvd_elbo_grad(ffvd_handle *h, const ffvd_params *p, uint32_t flags, int S_total, double out_terms[8],
                    double *out_nll, const ffvd_grads *g);
/* ---- optimiser / sampler steps -------------------------------------------------
 * ffvd_adam_step: one train_hypers iteration (base_model.py:944-950) entirely on the device: forward + backward on
 * the resident parameters, then tf.compat.v1.train.AdamOptimizer's update (dgp_model.py:303-305; TensorFlow is not
 * vendored by the reference -- the published rule is  lr_t = lr sqrt(1-b2^t)/(1-b1^t),  m = b1 m + (1-b1) g,
 * v = b2 v + (1-b2) g^2,  theta -= lr_t m / (sqrt(v) + eps);  TF defaults b1 = 0.9, b2 = 0.999, eps = 1e-8;
 * lr = 0.003 * 0.95^(1/1000), base_model.py:190) applied in one fused launch to the arrays selected by train_mask.
 * out_terms / out_nll describe the parameters BEFORE the update.  Needs grad = 1, all latent dims on the handle and a handle
 * that is not one rank of a multi-rank communicator (those use ffvd_adam_step_allreduce below);
 * a failed factorisation returns FFVD_ENOTPD and leaves the parameters untouched.  The handle keeps m, v and t;
 * ffvd_optimizer_reset zeroes them.  ffvd_get_params copies the resident parameters to host arrays (NULL = skip). */
#define FFVD_TRAIN_X 1u
#define FFVD_TRAIN_Z 2u
#define FFVD_TRAIN_LOGVARIANCE 4u
#define FFVD_TRAIN_LOGLENGTHSCALES 8u
#define FFVD_TRAIN_LOG_Q 16u
#define FFVD_TRAIN_CC 32u
#define FFVD_TRAIN_DD 64u
#define FFVD_TRAIN_LOG_RCHOLS 128u
#define FFVD_TRAIN_U 256u            /* explicit-U branch only; ignored where U is integrated out */
#define FFVD_TRAIN_ALL 511u
int  ffvd_adam_step(ffvd_handle *h, double lr, double beta1, double beta2, double eps, uint32_t train_mask,
                    double out_terms[8], double *out_nll);
int  ffvd_optimizer_reset(ffvd_handle *h);
int  ffvd_get_params(ffvd_handle *h, const ffvd_params *out_host);
/* overwrite some of the bound parameter arrays from host memory (NULL members are kept): what feeding a stored
 * SG-HMC window sample through feed_dict does in train_hypers (base_model.py:948-949). */
int  ffvd_update_params(ffvd_handle *h, const ffvd_params *p_host);
/* One burn_in_op (burn_in != 0) or sample_op (0) of BaseModel.generate_update_step (base_model.py:143-179) on the
 * device: forward + backward on the resident parameters, then the SG-HMC update of the arrays in sample_mask
 * (FFVD_TRAIN_* bits; X is never sampled, dgp_model.py:213-244) with X_N = T + 1 (dgp_model.py:203).  The handle keeps
 * xi, g, g2 (ones) and p (zeros) per array.  noise: host arrays of standard-normal draws (base_model.py:169) for
 * the sampled arrays, other members ignored.  out_terms / out_nll: the nll before the update. */
int  ffvd_sghmc_step(ffvd_handle *h, double epsilon, double mdecay, uint32_t sample_mask, int burn_in,
                     const ffvd_params *noise_host, double out_terms[8], double *out_nll);
/* ---- sharded training step, device resident (the multi-GPU counterpart of adam.minimize(nll), dgp_model.py:303-305 /
 * train_hypers, base_model.py:944-950, and of burn_in_op / sample_op, base_model.py:143-179) ---------------------------------
 * Every rank runs forward + backward on its shard with S_total (the job's chain count) as the divisor of the mean, which
 * leaves its ADDITIVE share of the job's gradient in ONE device block  [8 term sums | dZ | dlogvariance dloglengthscales
 * dlog_Q dCC dDD dlog_Rchols | dU | dX]  (every segment on a 256-byte boundary; the gradient arrays of ffvd_elbo_grad ARE
 * these segments, nothing is packed).  ONE ncclAllReduce(sum) of the block -- without dX for chain shards, whose rows of X
 * are their own; with it for latent-dim shards, where every rank holds a partial sum for all chains -- then the fused
 * update reads the reduced block.  No gradient crosses PCIe.  out_terms = whole-job sums (out_terms[7] = chains counted),
 * out_nll = their mean, both for the parameters BEFORE the update.  A failed factorisation on any rank makes the sums
 * non-finite on every rank: all return FFVD_ENOTPD and all leave their parameters untouched, so the replicas stay equal.
 * The Adam moments / SG-HMC state live on the handle as for ffvd_adam_step / ffvd_sghmc_step; SG-HMC noise must be the same
 * on every rank (the shared parameters are replicated).  rccl_comm = NULL: the handle's own communicator (ffvd_comm_init).
 * Once a handle belongs to a communicator of more than one rank, plain ffvd_adam_step / ffvd_sghmc_step refuse it
 * (FFVD_EINVAL): they would silently train on the local share only.
 * Three-step form for hosts that carry the exchange themselves (tests: two ranks on ONE GPU over gloo; MPI):
 *   ffvd_train_local(h, S_total)           forward + backward + the 8 sums into the block (enqueue only)
 *   ffvd_train_exchange_count / _ptr       doubles that take part in the exchange / device pointer of the block
 *   ffvd_train_exchange_get / _set         copy the block to / from the host (synchronise)
 *   ffvd_adam_apply / ffvd_sghmc_apply     info + finiteness check, then the update from the block                      */
int  ffvd_adam_step_allreduce(ffvd_handle *h, void *rccl_comm, int S_total, double lr, double beta1, double beta2, double eps,
                              uint32_t train_mask, double out_terms[8], double *out_nll);
int  ffvd_sghmc_step_allreduce(ffvd_handle *h, void *rccl_comm, int S_total, double epsilon, double mdecay,
                               uint32_t sample_mask, int burn_in, const ffvd_params *noise_host, double out_terms[8],
                               double *out_nll);
int  ffvd_train_local(ffvd_handle *h, int S_total);
int64_t ffvd_train_exchange_count(const ffvd_handle *h);
void *ffvd_train_exchange_ptr(ffvd_handle *h);
int  ffvd_train_exchange_get(ffvd_handle *h, double *host_out);
int  ffvd_train_exchange_set(ffvd_handle *h, const double *host_in);
int  ffvd_adam_apply(ffvd_handle *h, double lr, double beta1, double beta2, double eps, uint32_t train_mask,
                     double out_terms[8], double *out_nll);
int  ffvd_sghmc_apply(ffvd_handle *h, double epsilon, double mdecay, uint32_t sample_mask, int burn_in,
                      const ffvd_params *noise_host, double out_terms[8], double *out_nll);

/* ---- multi-GPU: one process per GPU, ONE exchange step ------------------------------------------
 * The reference is single-process (no collective call sites); this build shards chains or latent
 * dims over ranks and all-reduces the 8 partial sums with RCCL over xGMI.  librccl is bound at run time (the copy already
 * mapped in the process, else $FFVD_RCCL_LIB, else the system library); without it these calls return FFVD_EDEVICE.
 *   ffvd_comm_unique_id   rank 0 creates the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other ranks by
 *                         any means the host has (file, socket, MPI, a torch.distributed store ...).
 *   ffvd_comm_init        every rank: ncclCommInitRank on the handle's device; the handle owns the communicator
 *                         (ffvd_comm_destroy / ffvd_destroy release it).  Collective: all ranks must call it.
 *   ffvd_elbo_allreduce   one ELBO iteration of this rank's shard + ncclAllReduce(sum, 8 doubles) on the handle's stream
 *                         + one copy back.  rccl_comm: a caller-owned ncclComm_t, or NULL = the handle's own.  out_terms =
 *                         whole-job sums, out_terms[7] = whole-job chain count (only handles with shared_terms = 1 count
 *                         their chains, so latent-dim shards count each chain once), out_nll = out_terms[6] / out_terms[7].  A failed factorisation on this
 *                         rank -> FFVD_ENOTPD with the matrix/pivot; on another rank -> FFVD_ENOTPD (NaN in the sums).
 *   ffvd_elbo_allreduce_async   enqueue only; sums land in out_terms_dev (NULL = the handle's result block).
 *   ffvd_allreduce_sum_async    in-place ncclAllReduce(sum) of `count` doubles at device pointer buf_dev on the handle's
 *                         stream (shared-parameter gradients of a sharded training step; Gram tiles of a T-shard). */
#define FFVD_COMM_ID_BYTES 128
int  ffvd_comm_unique_id(void *id_out /* FFVD_COMM_ID_BYTES */);
int  ffvd_comm_init(ffvd_handle *h, int world, int rank, const void *id /* FFVD_COMM_ID_BYTES */);
int  ffvd_comm_destroy(ffvd_handle *h);
void *ffvd_comm_get(ffvd_handle *h);   /* the handle's ncclComm_t or NULL */
int  ffvd_elbo_allreduce(ffvd_handle *h, void *rccl_comm, double out_terms[8], double *out_nll);
int  ffvd_elbo_allreduce_async(ffvd_handle *h, void *rccl_comm, double *out_terms_dev);
int  ffvd_allreduce_sum_async(ffvd_handle *h, void *rccl_comm, double *buf_dev, int64_t count);
/* the same for a HOST array (staged through a device buffer the handle keeps); synchronises */
int  ffvd_allreduce_sum(ffvd_handle *h, void *rccl_comm, double *buf_host, int64_t count);

/* ---- T-shard fallback (SURVEY 8e last bullet, section 5 "long-context"): when chains x latent dims < ranks ---------
 * T is a pure reduction axis of the collapsed bound: the Gram matrices K_uf K_fu (M x M per unit) and the rows
 * delta^T K_fu, the likelihood and transition sums are all sums over t.  A T-shard handle (cfg.T = rows of the shard,
 * cfg.T_total, cfg.t_begin; FFVD_BRANCH_B, FFVD_ROUTE_GRAM, FFVD_F64; X = rows t_begin .. t_begin + T of
 * every chain, Y / control_inputs = the shard's rows) evaluates its rows' share; ONE all-reduce(sum) of the raw Gram
 * tiles + the per-chain partial sums (S_local * D * (M_p + 1) * M_p + 8 S_local doubles, M_p = M rounded up to 64)
 * follows, and EVERY rank finishes the same factorisations on the reduced sums, so every rank holds the whole-job
 * terms (out_terms[7] = S_local).  ffvd_elbo_tshard = local part + ncclAllReduce on the handle's stream + finish.
 * The three-step form lets a host carry the buffer itself (tests: two ranks on one GPU over gloo). */
int  ffvd_elbo_tshard(ffvd_handle *h, void *rccl_comm, double out_terms[8], double *out_nll);
int  ffvd_tshard_local(ffvd_handle *h);                       /* enqueue this shard's partial sums               */
int64_t ffvd_tshard_count(const ffvd_handle *h);              /* doubles in the exchange buffer                   */
int  ffvd_tshard_get(ffvd_handle *h, double *host_out);       /* copy the exchange buffer to the host (synchronises) */
int  ffvd_tshard_set(ffvd_handle *h, const double *host_in);  /* ... and the reduced sums back                    */
int  ffvd_tshard_finish(ffvd_handle *h, double out_terms[8], double *out_nll);
/* Gradient of a T-sharded job (cfg.grad = 1, every latent dim on the handle).  After the exchange of the raw tiles every shard
 * holds the job's A = K_uu + K_uf K_fu / Q, its factor, u and Gamma, so the backward pass needs no further operand from the other
 * shards: its K_fu side (the T x M product and its reductions, dgp_model.py:261-288 differentiated) runs over the shard's own rows
 * and is ADDITIVE over shards like the likelihood and transition sums; the M x M side (K_uu chain rule, tr(A^-1 G), u^T G u, the
 * priors) is identical on every shard and counted on the first (t_begin == 0) only.  ffvd_tshard_finish_grad = finish + backward
 * pass (divisor S_total, every 1 / T the job's T_total); it leaves the block [8 term sums (first shard; zero elsewhere) | dZ |
 * dlogvariance | dloglengthscales | dlog_Q | dCC | dDD | dlog_Rchols] at ffvd_train_exchange_ptr, ffvd_train_exchange_count
 * doubles: ONE all-reduce(sum) of it (ffvd_allreduce_sum_async, or ffvd_train_exchange_get / _set through the host) completes
 * the gradient and the terms on every rank; ffvd_tshard_grad_fetch then copies it out.  dX covers the shard's OWN T + 1 rows and
 * does not take part in the exchange: the row two neighbouring shards share (the last of one, the first of the next) receives a
 * part from each, which the caller adds.  ffvd_elbo_tshard_grad = all of it with the two native RCCL all-reduces. */
int  ffvd_tshard_finish_grad(ffvd_handle *h, int S_total, double out_terms[8], double *out_nll);
int  ffvd_tshard_grad_fetch(ffvd_handle *h, double out_terms[8], const ffvd_grads *gout);
int  ffvd_elbo_tshard_grad(ffvd_handle *h, void *rccl_comm, int S_total, double out_terms[8], double *out_nll,
                           const ffvd_grads *gout);
/* Optimiser step of a T-sharded job (the reference trains every variable with one AdamOptimizer, dgp_model.py:303-305).  Call after
 * ffvd_tshard_grad_fetch / ffvd_elbo_tshard_grad: the exchanged block already holds the job's shared-parameter gradients, identical
 * on every shard.  `dX_rows` = the shard's own S_local x (T + 1) x D rows of dX AFTER the caller has added the neighbouring shards'
 * parts of the first and the last row (host memory; ffvd_amd/distributed.py exchanges them with one small all-reduce).  The fused
 * Adam update then runs over every array of `train_mask`; shared parameters and both copies of a boundary row receive identical
 * gradients and carry identical state, so the shards stay in step without a broadcast.  out_terms = the job's 8 sums. */
int  ffvd_tshard_adam_apply(ffvd_handle *h, const double *dX_rows, double lr, double beta1, double beta2, double eps,
                            uint32_t train_mask, double out_terms[8], double *out_nll);
/* ... and the SG-HMC update (burn_in_op / sample_op, base_model.py:143-179) of a T-sharded job from the same exchanged block: X is
 * never an SG-HMC variable, so no rows travel; `noise` must be identical on every shard; X_N of the step size is the JOB's row
 * count (T_total + 1). */
int  ffvd_tshard_sghmc_apply(ffvd_handle *h, double epsilon, double mdecay, uint32_t sample_mask, int burn_in,
                             const ffvd_params *noise, double out_terms[8], double *out_nll);

/* the handle's HIP stream (a hipStream_t), so that a collective library or another framework can order its work after
 * ffvd_elbo_async without a host synchronisation (e.g. torch.cuda.ExternalStream around the RCCL all-reduce). */
void *ffvd_get_stream(ffvd_handle *h);
/* after a synchronous ffvd_elbo: per-chain nll values (S_local doubles, host) */
int  ffvd_chain_nll(ffvd_handle *h, double *out_nll_per_chain);
/* timing helper for benchmarks: run `iters` back-to-back ffvd_elbo_async on the resident inputs,
 * bracketed by HIP events on the handle's stream; returns total milliseconds. */
int  ffvd_time_elbo(ffvd_handle *h, int iters, float *out_ms);
/* HIP-event timing of each stage of one iteration (ms): [0] K_uu build+Cholesky+inverse,
 * [1] K_fu projection (F = K_fu L^-T), [2] Gram H = F^T F, [3] Cholesky(H)+solve, [4] reductions. */
int  ffvd_profile_stages(ffvd_handle *h, float out_ms[8]);
/* live stage timing: while enabled, every ffvd_elbo* records HIP events on the handle's stream around the
 * stages above; ffvd_stage_times() synchronises, returns the accumulated milliseconds and the number of timed
 * intervals (= kernel-launch groups) per stage since the last read, and resets the accumulators. */
int  ffvd_stage_timing(ffvd_handle *h, int enable);
int  ffvd_stage_times(ffvd_handle *h, double out_ms[8], int32_t out_launches[8]);

/* ---- operator-level entry points (host pointers in/out; they allocate temporaries) ----------
 * These mirror the reference's pure functions so that parity tests read like calls of the reference. */

/* kernel.K(X, X2) / kernel.K(X) (X2 == NULL) for ONE kernel: kernels_multi_output.py:202-214, kernels.py:270-276.
 * jitter is added to the diagonal when X2 == NULL. out: N x N2. */
int  ffvd_op_kernel_matrix(int kind, const double *X, int N, const double *X2, int N2, int P,
                           double logvariance, const double *loglengthscales, double jitter, double *out);
/* kernel.Kdiag(X): kernels_multi_output.py:199-200, kernels.py:278-281. out: N. */
int  ffvd_op_kernel_diag(int kind, const double *X, int N, int P, double logvariance, double *out);
/* batched lower Cholesky of `batch` n x n SPD matrices (tf.linalg.cholesky, conditionals_multi_output.py:28,162).
 * A and L may alias. info[b] = 0 or 1 + index of the first non-positive pivot. Returns FFVD_ENOTPD if any info != 0. */
int  ffvd_op_cholesky(const double *A, int n, int batch, double *L, int32_t *info);
/* The whitened triangular solve of base_conditional: A = Lm^-1 Kmn = tf.linalg.triangular_solve(Lm, Kmn, lower=True)
 * (conditionals_multi_output.py:34, conditionals.py:30).  L: n x n lower triangular (entries above the diagonal are ignored),
 * B: n x m, X: n x m = L^-1 B.  Runs the wavefront-level blocked substitution of the Cholesky panel step on its own. */
int  ffvd_op_trsm(const double *L, int n, const double *B, int m, double *X);
/* kernel_pre_cal (conditionals_multi_output.py:124-169): for D kernels returns the stack of L_d^{-T} (D x M x M, upper). */
int  ffvd_op_kernel_pre_cal(int kind, const double *Z, int M, int P, int D, const double *logvariance,
                            const double *loglengthscales, double jitter, double *Lm_inverse_seq);
/* collapse_after_kernel_precalculation (conditionals_multi_output.py:230-257).
 * X_combine: T x P, X: (T+1) x D, Q: D.  out3 = (-term1/Y_N, -term2/Y_N, -trace/Y_N). */
int  ffvd_op_collapse(int kind, const double *Lm_inverse_seq, const double *X_combine, const double *X,
                      const double *Z, int T, int M, int P, int D, const double *logvariance,
                      const double *loglengthscales, const double *Q, double batch_size, double Y_N, double out3[3]);
/* conditional(Xnew, Z, kern, f, white=True, full_cov=False) (conditionals_multi_output.py:73-120 -> base_conditional :6-70).
 * Xnew: N x P, f: M x D.  mean, var: N x D. */
int  ffvd_op_conditional(int kind, const double *Xnew, int N, const double *Z, int M, int P, int D,
                         const double *logvariance, const double *loglengthscales, const double *f,
                         double jitter, double *mean, double *var);

/* collapse_u_mean_after_kernel_precalculation (conditionals_multi_output.py:206-227): posterior mean of the whitened
 * inducing outputs U_mean (M x D, = H_d^-1 b_d per column) and the stack L_{H_d}^{-T} (D x M x M, upper). */
int  ffvd_op_collapse_u_mean(int kind, const double *Lm_inverse_seq, const double *X_combine, const double *X,
                             const double *Z, int T, int M, int P, int D, const double *logvariance,
                             const double *loglengthscales, const double *Q, double *U_mean, double *H_inv_sqrt);
/* conditional_after_kernel_precalculation(..., white=True, full_cov=False) (conditionals_multi_output.py:306-387).
 * q_sqrt: NULL, or the M x M slice d = 0 of the D x M x M stack -- the reference hands the whole stack to every
 * dim and `[:, :, 0]` (:322) keeps slice 0 for all of them; the caller passes that slice. */
int  ffvd_op_conditional_precalc(int kind, const double *Lm_inverse_seq, const double *Xnew, int N, const double *Z,
                                 int M, int P, int D, const double *logvariance, const double *loglengthscales,
                                 const double *f, const double *q_sqrt, double *mean, double *var);
/* Gaussian.predict_mean(X_end) = X_end @ CC + DD (likelihoods.py:76-79). X_end: N x D, out: N x Ydim. */
int  ffvd_op_predict_mean(const double *X_end, int N, int D, const double *CC, const double *DD, int Ydim, double *out);
/* logdensity_norm_diag (nonvec = 0, out: N; likelihoods.py:96-111) / logdensity_norm_diag_nonvec (nonvec = 1,
 * out: N x J; likelihoods.py:89-93).  y, ymean: N x J; Rchols: J.  No log(2 pi) constants, as in the reference. */
int  ffvd_op_logdensity_norm_diag(int nonvec, const double *y, const double *ymean, const double *Rchols, int N, int J,
                                  double *out);
/* get_rand (utils.py:11, diagonal case) with the standard-normal draw injected: out = mean + eps * sqrt(var). */
int  ffvd_op_get_rand(const double *mean, const double *var, const double *eps, int64_t n, double *out);

/* One Adam update of a flat host array (in place: theta, m, v), t = 1-based step count; same rule as ffvd_adam_step. */
int  ffvd_op_adam_step(double *theta, const double *grad, double *m, double *v, int64_t n, double lr, double beta1,
                       double beta2, double eps, int64_t t);
/* One SG-HMC update, BaseModel.generate_update_step base_model.py:143-179, of a flat host array (in place):
 * state xi, g, g2 (initialised to ones by the reference) and momentum p (zeros); noise = the standard-normal draw of
 * :169 (injected); epsilon, mdecay as DGPSSM(epsilon=0.01, mdecay=0.05) dgp_model.py:161; X_N = number of training
 * rows (:164).  burn_in != 0 = burn_in_op (xi, g, g2, theta, p move), 0 = sample_op (theta, p only).  Every
 * right-hand side reads the state from before the call. */
int  ffvd_op_sghmc_step(double *theta, const double *grad, double *xi, double *g, double *g2, double *p,
                        const double *noise, int64_t n, double epsilon, double mdecay, double X_N, int burn_in);

/* The ffvd_op_* entry points keep their device temporaries and their stream in a per-thread cache between calls (a rollout call
 * made ~27 allocations around its step loop: 2.7 ms of host work per call).  ffvd_op_release_cache frees what the calling thread's
 * cache holds; the cache also trims itself beyond 256 blocks / 8 GiB.  A host that runs ffvd_op_* calls on short-lived threads
 * calls it before a thread exits (nothing is freed at thread exit: the HIP runtime may already be shutting down then). */
int  ffvd_op_release_cache(void);

/* The prediction loop of collect_samples_formal (base_model.py:288-314) for R posterior rollouts advanced side by
 * side on the device: per step, conditional_after_kernel_precalculation at the R current states (:300, q_sqrt = the
 * d = 0 slice or NULL as in ffvd_op_conditional_precalc), then x_next = x + f_mu + eps * sqrt(f_var + Q) (:304-306).
 * x_last: D (= layers[-1].X[-1], :226); ctrl: steps x C, row t = control_inputs[Y_train.shape[0] + t] (:293), NULL when
 * C = 0; f = U_val: M x D; eps: steps x R x D standard-normal draws (injected); outputs R x steps x D:
 * predict_x (:313) and predict_var = f_var + Q (:314). */
int  ffvd_op_rollout(int kind, const double *Lm_inverse_seq, const double *Z, int M, int P, int D,
                     const double *logvariance, const double *loglengthscales, const double *f, const double *q_sqrt,
                     const double *x_last, int R, const double *ctrl, int C, int steps, const double *log_Q,
                     const double *eps, double *predict_x, double *predict_var);
/* How many ffvd_op_rollout calls of this process completed on the per-step launches because the resident-operand loop (the default
 * up to 64 rollouts, M <= 512) gave up on a bounded wait -- its workgroups must all be resident, a co-tenant can prevent that.  The
 * two forms agree to 1e-9 but not bit for bit: a seeded rollout is bit-reproducible only while this counter stands still (or with
 * FFVD_STEP_LOOP=0, which always takes the launches).  The call that fell back also leaves a warning in ffvd_last_error(NULL). */
int  ffvd_op_rollout_fallbacks(void);

/* One particle-Gibbs sweep over the latent trajectory: the INTENT of BaseModel.PG_for_X_speedup (base_model.py:78-138;
 * as written that op never updates X -- discarded TensorArray.write results (:115), an assign that is never run (:137) --
 * so there is no reference behaviour to match, see oracle/ffvd_pg_oracle.py).  n_free = PG_particles - 1 free particles
 * start at x0 (n_free x D, the N(0, I) draw of :79, injected) and advance side by side; per step tt < X_N - 1:
 * conditional_after_kernel_precalculation at [x_t, ctrl[tt]] with the explicit, whitened U (:93-97), x_{t+1} = x_t + f_mu +
 * eps[tt] * sqrt(f_var + Q) (:99-101), weights logdensity_norm(Y[tt], predict_mean(.), Rchols) of the new particles and of
 * the reference state X_ref[tt+1] (:105-109), n_free categorical draws (:113; inverse CDF of softmax(w) at the injected
 * uniforms unif[tt], first index whose cumulative probability exceeds u) and the gather (:111-115).
 * X_ref: X_N x D; Y: (X_N-1) x Ydim; ctrl: (X_N-1) x C or NULL; Rchols: Ydim x Ydim lower-triangular (= exp(log_Rchols),
 * likelihood.Rchols), Ydim <= 8; eps: (X_N-1) x n_free x D; unif: (X_N-1) x n_free in [0, 1).
 * Outputs: particles X_N x n_free x D (resampled_X of :133) and idx (X_N-1) x n_free (int32, value n_free = reference).
 * The final choice (:135-137) is the caller's: X <- particles[:, final_index] unless final_index == n_free. */
int  ffvd_op_pg_sweep(int kind, const double *Lm_inverse_seq, const double *Z, int M, int P, int D,
                      const double *logvariance, const double *loglengthscales, const double *U, const double *X_ref,
                      int X_N, const double *Y, int Ydim, const double *ctrl, int C, const double *CC, const double *DD,
                      const double *Rchols, const double *log_Q, int n_free, const double *x0, const double *eps,
                      const double *unif, double *particles, int32_t *idx);

#ifdef __cplusplus
}
#endif
#endif /* FFVD_ABI_H */
