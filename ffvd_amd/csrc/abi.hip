// C ABI of libffvd_hip.so (see include/ffvd_abi.h): handle lifetime, resident buffers, the per-iteration
// launch sequence of the ELBO, and operator-level entry points used by the Python mirror of the reference API.
#include "../../include/ffvd_abi.h"
#include "kernels.h"
#include "kernels_f32.h"
#include "grad.h"
#include "optim.h"
#include "tiny.h"
#include "step_bodies.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace ffvd;

static thread_local std::string g_last_error;
static int g_rollout_fallbacks = 0;     // ffvd_op_rollout calls of this process that fell back from the resident loop to the per-step launches

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            char buf_[512];                                                                   \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                     \
            return set_error(h, (e_ == hipErrorOutOfMemory) ? FFVD_ENOMEM : FFVD_EDEVICE, buf_); \
        }                                                                                     \
    } while (0)

struct ffvd_handle {
    ffvd_config cfg;
    int P = 0, Mp = 0, Tp = 0, Dl = 0, nbatch = 0, ng = 0, cpp = 0;
    hipStream_t stream = nullptr;
    double *dinvK = nullptr, *dinvH = nullptr;   // Cholesky scratch (kernels.h DINV_STRIDE per matrix)
    double *gpart = nullptr;                     // split-K partial tiles of the Gram kernel (few units per pass)
    int gsplit = 1;
    bool side_late = false;     // few chains, forward: the K_uu side chain as ONE dataflow launch BEHIND the tile pass (plan_schedule); decided with gsplit at create
    double *graw = nullptr;                      // unsplit first pass: raw Gram tiles for the deferred trace pass
    double *gtail = nullptr;                     // unsplit passes: blocks + counters of the tail split (kernels.h GramArgs)
    int gtail_wg = 0;
    double *lrpart = nullptr;                    // LinearK explicit-U forward: partial sums of G = C C^T and v = C u per column block (kernels.h)
    double *growpart = nullptr;                  // Gram route: per-64-row-block partial sums of delta^T K_fu from the K_fu build
    hipStream_t aux = nullptr;          // side stream: the K_uu chain runs beside the K_fu build (Gram route)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_kuu = nullptr, ev_tiles = nullptr, ev_go = nullptr;
    hipEvent_t ev_hwords = nullptr;     // recorded right behind a side-stream clear of Cholesky(A)'s progress words: the launch that trusts the clear waits for IT
    hipEvent_t ev_prior = nullptr;      // recorded behind an early prior-sums launch on the side stream (fwd_prior_sums)
    double *prior_sums = nullptr;       // [16] the ten parameter-only sums of the nll assembly, formed early in the iteration
    std::string err;
    std::vector<void *> allocs;
    int64_t ws_bytes = 0;
    // diagnostic switches, read from the environment ONCE when the handle is created
    struct Switches {
        bool fused_project = false, grad_explicit = false, no_defer_trace = false, no_late_join = false;
        bool no_main_first = false, no_kfu_first = false, atb128 = false, grad_serial = false, kuu_flow = true, kinv_gram = false, chain_rl = false;
        int small_side_rows = 32;         // FFVD_SMALL_SIDE_ROWS: block rows of the K_uu chain (Dl * 2 * Mp / 64) up to which the side chain is ONE dataflow launch
        int small_side_wgs = 512;         // FFVD_SMALL_SIDE_WGS: tile-pass workgroups (one round of the chip) up to which an iteration counts as tiny (see small_side)
        bool no_small_side = false;       // FFVD_NO_SMALL_SIDE=1: tiny iterations keep the launch-per-step K_uu chain (round 2)
        bool no_side_late = false;        // FFVD_NO_SIDE_LATE=1: few-chain forward iterations keep the K_uu chain beside the tile pass (round 4, first half)
        bool ref_row_in_gram = false;     // FFVD_REF_ROW_IN_GRAM=1: reference route, delta^T F formed by the Gram kernel's diagonal tiles (rounds 1-2)
        bool no_ref_side = false;         // FFVD_NO_REF_SIDE=1: reference route / explicit-U branch with the K_uu chain on the main stream in front of the K_fu build
        bool no_linear_lowrank = false;   // FFVD_NO_LINEAR_LOWRANK=1: LinearK explicit-U forward through the M-wide projection (rounds 1-2)
        bool lt_armed = false;            // FFVD_GRAD_LT_ARMED=1: write the L^T rows to memory (launch_set_lt_rows) even where the dataflow kernel could read L itself
        bool whiten_products = false;     // FFVD_GRAD_WHITEN_PRODUCTS=1: training forward forms H = W^T A W with two products (round 1/2) instead of arming L^T rows
        bool debug_sync = false;    // FFVD_DEBUG_SYNC: name every launch group on stderr and wait for it (locates a faulting kernel)
        int side_delay_us = 0;      // FFVD_DEBUG_SIDE_DELAY_US=n: a spin kernel of n us at the head of every side-stream fork (schedule tests: results
        int main_delay_us = 0;      //   must not depend on which stream is late); FFVD_DEBUG_MAIN_DELAY_US=n: the same on the main stream behind a fork
        bool tiny_xcd = true;       // FFVD_TINY_NO_XCD=1: the one-launch iteration with role-major workgroup ids instead of a unit's workgroups on ONE XCD
        bool no_tiny = false;       // FFVD_NO_TINY=1: the multi-kernel schedule also at the reference's own experiment size (rounds 1-3)
        bool no_tiny_a = false;     // FFVD_NO_TINY_A=1: ... for the explicit-U branch only (rounds 1-4)
        bool no_early_priors = false;   // FFVD_NO_EARLY_PRIORS=1: the parameter-only sums inside the finalize launch at the iteration's tail (rounds 1-4)
    } sw;
    // resident parameters / data (handle-owned copies)
    double *X = nullptr, *Z = nullptr, *U = nullptr, *logvar = nullptr, *loglen = nullptr, *logQ = nullptr;
    double *CC = nullptr, *DD = nullptr, *logR = nullptr, *Y = nullptr, *ctrl = nullptr;
    ffvd_params cur{};          // pointers the kernels read (resident copies or caller's device pointers)
    bool have_params = false, have_data = false;
    void *comm = nullptr;       // RCCL communicator created by ffvd_comm_init (owned by the handle), else null
    int comm_world = 1, comm_rank = 0;
    int tiny_cus = 0;           // compute units of the device (one-launch plan)
    double *tsbuf = nullptr;    // T-shard exchange buffer: [nbatch][(Mp+1) x Mp] raw Gram tiles + delta^T K_fu rows, then [S][8] chain sums
    int64_t ts_count = 0;
    double *stage = nullptr;    // staging buffer of ffvd_allreduce_sum
    int64_t stage_count = 0;
    bool kuu_flow_sched = false;   // schedule of the big unsplit Gram pass, decided in ffvd_create (see there)
    // pass-pipelined forward iteration (enqueue_elbo_pipe; FFVD_PIPE=<passes>, FFVD_PIPE_MODE=<bits>): streams for the K_fu builds
    // of later passes and for every other pass's Gram launch, one event per pass and stage, one Cholesky scratch region per pass
    int pipe_passes = 0, pipe_mode = 0;
    hipStream_t pipe_build = nullptr, pipe_gram = nullptr, pipe_chol = nullptr;
    std::vector<hipEvent_t> pipe_evB, pipe_evG;
    hipEvent_t pipe_evC = nullptr;
    double *pipe_dinv = nullptr;
    size_t pipe_dinv_stride = 0;
    // one-launch iteration of the reference's own experiment size (tiny.hip): decided once in ffvd_create
    TinyPlan tiny{};
    double *tiny_scratch = nullptr;
    int *tiny_flags = nullptr;
    TinyArgs *tiny_dargs = nullptr;     // [2] device copies of the argument block (forward / forward + backward)
    TinyArgRing tiny_ring[2];           // per copy: pinned upload slots guarded by events + what the device copy holds (tiny.h)
    bool tiny_ring_made = false;
    size_t tiny_private_bytes = 0;      // scratch per lane of the one-launch kernel as the loaded code object reports it
    bool tiny_dirty = false;       // a launch was abandoned on a bounded wait: its hand-off words are re-zeroed before the next one
    bool info_pending = false;  // an ffvd_elbo_async was enqueued whose Cholesky info flags nobody has looked at yet
    // workspace
    double *variance = nullptr, *len = nullptr, *Zs = nullptr, *zz = nullptr;
    double *Kuu = nullptr, *F = nullptr, *H = nullptr, *rowsq = nullptr, *fmean = nullptr;
    double *ucolA = nullptr;        // explicit-U branch: U columns of the local dims, zero padded to Mp
    double *Kf2 = nullptr;          // reference route, branch B: K_fu (input of the projection GEMM); F keeps K_fu L^-T
    int ngr = 0;                    // row-sum partials per unit in that path (128-column tiles)
    // fp32-contraction path (cfg.dtype == FFVD_F32C): K_fu, F = K_fu L^-T, L^-1 as fp32 GEMM operands; per-tile and
    // per-unit sums of F^2 (fp64)
    float *Kf32 = nullptr, *F32 = nullptr, *Linv32 = nullptr;
    double *sqpart = nullptr, *sqsum = nullptr;
    int nsq = 0, gram_flush = 0;
    double *Kcopy = nullptr, *Linv = nullptr, *Kinv = nullptr, *trpart = nullptr, *kterms = nullptr;   // GRAM route
    int ntiles = 0;
    double *chain_partial = nullptr;
    // backward-pass workspace (cfg.grad)
    struct GradWs {
        double *Acopy = nullptr, *u = nullptr, *LAinv = nullptr, *Gamma = nullptr, *gam_part = nullptr, *uku = nullptr;
        // whitened backward (collapsed branch): T1 = A W, later B = L_H^-1 L^-1; w = H^-1 b; b = W^T c staging; identity
        // matrix (w^T w through the u^T K u kernel); two more per-dim products of the K_uu side
        double *T1 = nullptr, *wv = nullptr, *bw = nullptr, *Ident = nullptr, *P2 = nullptr, *P3 = nullptr;
        bool whitened = false;
        double *E = nullptr, *rp = nullptr, *rsum = nullptr, *ez = nullptr, *kfu = nullptr;
        float *Gam32 = nullptr;         // fp32-contraction backward: Gamma rounded to fp32, the right operand of R = K_fu Gamma
        double *fsq = nullptr;          // reference route, fp64: sum_t |F_t|^2 per unit (the fp32 path has sqsum)
        double *cs_part = nullptr, *etx_part = nullptr, *rx2_part = nullptr, *dz_unit = nullptr, *dll_unit = nullptr, *dls_unit = nullptr;
        double *Asum = nullptr, *GamSum = nullptr, *Gs = nullptr, *gsum = nullptr, *P1 = nullptr, *KGK = nullptr, *Epsi = nullptr;
        double *rsum2 = nullptr, *ez2 = nullptr, *cs2 = nullptr, *etx2 = nullptr, *rx22 = nullptr, *dz_kuu = nullptr, *dll_kuu = nullptr, *dls_kuu = nullptr;
        double *shared_part = nullptr, *dX = nullptr, *dZ = nullptr, *dlogvar = nullptr, *dloglen = nullptr, *dlogQ = nullptr;
        size_t small_count = 0;         // doubles in the block dlogvar | dloglen | dlogQ | dCC | dDD | dlogR (one allocation)
        double *dCC = nullptr, *dDD = nullptr, *dlogR = nullptr;
        // explicit-U branch
        double *Gu = nullptr, *Gsum = nullptr, *r = nullptr, *dalpha = nullptr, *ucol = nullptr, *beta = nullptr, *du = nullptr;
        double *GammaA = nullptr, *Lclean = nullptr, *dU = nullptr, *xsq = nullptr;
        int ngam = 0, sp_stride = 0;
        // Exchange block of a sharded training step (ffvd_adam_step_allreduce): [8 term sums | dZ | dlogvar..dlogR | dU | dX],
        // every segment starting on a 256-byte boundary.  The gradient arrays above ARE these segments, so the block is
        // all-reduced in place with no packing pass; dX comes last because chain shards keep it out of the exchange.
        double *pack = nullptr;
        size_t pack_shared = 0, pack_total = 0;     // doubles up to (excluding) dX / including dX
    } gw;
    double *hterms = nullptr, *chain_terms = nullptr, *chain_nll = nullptr, *out_terms = nullptr;
    int32_t *info = nullptr;
    // Adam state for ffvd_adam_step: first/second moments per parameter array (order of FFVD_TRAIN_* bits), step count
    double *adam_m[9] = {nullptr}, *adam_v[9] = {nullptr};
    double *hmc[9][5] = {{nullptr}};     // SG-HMC state per array: xi, g, g2, p and the uploaded noise
    bool hmc_ready = false;
    int64_t adam_t = 0;
    bool adam_ready = false;
    // pinned host staging
    // result block: [8 term sums][S_local chain nll][Dl + nbatch info flags], contiguous on the device (resblk) and in
    // pinned host memory (h_res) so that one copy brings everything back; out_terms / chain_nll / info and h_out /
    // h_chain / h_info point into the two blocks
    double *resblk = nullptr, *h_res = nullptr;
    size_t res_bytes = 0;
    double *h_out = nullptr, *h_chain = nullptr, *h_sums = nullptr;
    int32_t *h_info = nullptr;
    int train_S_total = 0;
    bool stalled = false;       // check_info saw info = -1: the dataflow Cholesky gave up on a bounded wait
    int stall_recoveries = 0;   // iterations re-run with the launch-per-column Cholesky after such a stall
    int stall_hold = 0;         // > 0: this many further calls stay on the schedule without inter-workgroup waits (fetch_with_stall_recovery)
    int stall_hold_next = 16;   // length of the next hold: doubles with every stalled probe (cap 1024), back to 16 after a clean one
    long long enq_ns = 0, enq_calls = 0;      // host time spent enqueueing iterations (ffvd_debug_enqueue_us: tools)
    std::string warning;        // one-time note about the first recovery (ffvd_last_error returns it while no error is pending)      // > 0: ffvd_train_local has left a backward pass (scaled 1 / S_total) in gw.pack
    // optional live stage timing (HIP events on the handle's stream)
    bool timing_on = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_stage;
    size_t ev_used = 0;
};

static int set_error(ffvd_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    g_last_error = msg;
    return code;
}
static int set_error(std::nullptr_t, int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

template <class T>
static hipError_t dev_alloc(ffvd_handle *h, T **p, size_t count) {
    size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e == hipSuccess) {
        h->allocs.push_back((void *)*p);
        h->ws_bytes += (int64_t)bytes;
    }
    return e;
}

extern "C" const char *ffvd_last_error(const ffvd_handle *h) {
    return h ? h->err.c_str() : g_last_error.c_str();
}

extern "C" int64_t ffvd_workspace_bytes(const ffvd_handle *h) { return h ? h->ws_bytes : 0; }

extern "C" int ffvd_destroy(ffvd_handle *h) {
    if (!h) return FFVD_OK;
    hipSetDevice(h->cfg.device_id);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->comm) ffvd_comm_destroy(h);
    for (void *p : h->allocs) hipFree(p);
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->h_res) hipHostFree(h->h_res);
    if (h->tiny_ring_made) { tiny_ring_destroy(h->tiny_ring[0]); tiny_ring_destroy(h->tiny_ring[1]); }
    if (h->aux) { hipStreamSynchronize(h->aux); hipStreamDestroy(h->aux); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->ev_join2) hipEventDestroy(h->ev_join2);
    if (h->ev_kuu) hipEventDestroy(h->ev_kuu);
    if (h->ev_tiles) hipEventDestroy(h->ev_tiles);
    if (h->ev_go) hipEventDestroy(h->ev_go);
    if (h->ev_hwords) hipEventDestroy(h->ev_hwords);
    if (h->ev_prior) hipEventDestroy(h->ev_prior);
    for (hipEvent_t e : h->pipe_evB) hipEventDestroy(e);
    for (hipEvent_t e : h->pipe_evG) hipEventDestroy(e);
    if (h->pipe_evC) hipEventDestroy(h->pipe_evC);
    for (hipStream_t ps : {h->pipe_build, h->pipe_gram, h->pipe_chol})
        if (ps) { hipStreamSynchronize(ps); hipStreamDestroy(ps); }
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return FFVD_OK;
}

static int create_impl(const ffvd_config *cfg, ffvd_handle *h) {
    const ffvd_config &c = h->cfg;
    {
        auto on = [](const char *name) { const char *e = getenv(name); return e && *e && strcmp(e, "0") != 0; };
        ffvd_handle::Switches &w = h->sw;
        w.fused_project = on("FFVD_FUSED_PROJECT");   w.grad_explicit = on("FFVD_GRAD_EXPLICIT");
        w.no_defer_trace = on("FFVD_NO_DEFER_TRACE"); w.no_late_join = on("FFVD_NO_LATE_JOIN");
        w.no_main_first = on("FFVD_NO_MAIN_FIRST");   w.no_kfu_first = on("FFVD_NO_KFU_FIRST");
        w.whiten_products = on("FFVD_GRAD_WHITEN_PRODUCTS");   w.lt_armed = on("FFVD_GRAD_LT_ARMED");   w.no_small_side = on("FFVD_NO_SMALL_SIDE");   w.no_side_late = on("FFVD_NO_SIDE_LATE");   if (const char *e = getenv("FFVD_SMALL_SIDE_WGS")) w.small_side_wgs = atoi(e);   if (const char *e = getenv("FFVD_SMALL_SIDE_ROWS")) w.small_side_rows = atoi(e);   w.ref_row_in_gram = on("FFVD_REF_ROW_IN_GRAM");   w.no_ref_side = on("FFVD_NO_REF_SIDE");   w.no_linear_lowrank = on("FFVD_NO_LINEAR_LOWRANK");
        w.kuu_flow = !on("FFVD_NO_KUU_FLOW");   w.kinv_gram = on("FFVD_KINV_GRAM");   w.chain_rl = on("FFVD_CHAIN_RL");
        w.atb128 = on("FFVD_ATB128");                 w.grad_serial = on("FFVD_GRAD_SERIAL");
        w.debug_sync = on("FFVD_DEBUG_SYNC");
        w.no_tiny = on("FFVD_NO_TINY");               w.no_tiny_a = on("FFVD_NO_TINY_A");
        w.no_early_priors = on("FFVD_NO_EARLY_PRIORS");               w.tiny_xcd = !on("FFVD_TINY_NO_XCD");
        if (const char *e = getenv("FFVD_DEBUG_SIDE_DELAY_US")) w.side_delay_us = atoi(e);
        if (const char *e = getenv("FFVD_DEBUG_MAIN_DELAY_US")) w.main_delay_us = atoi(e);
    }
    h->P = c.D + c.C;
    h->Dl = c.d_count > 0 ? c.d_count : c.D;
    h->Mp = round_up(c.M, NB);
    h->Tp = round_up(c.T, STRIP);
    h->ng = (h->Mp + 511) / 512;
    h->nbatch = c.S_local * h->Dl;
    const size_t Mp = h->Mp, Tp = h->Tp, Dl = h->Dl, P = h->P;
    // chains per pass: every launch should cover as many (chain, dim) units as possible (the batched Cholesky and
    // the Gram grid need >= 1 workgroup per CU); F costs Dl*Tp*Mp*8 bytes per chain, budget 48 GiB of the 288 GB.
    if (c.chains_per_pass > 0) h->cpp = c.chains_per_pass;
    else {
        const size_t per_chain = Dl * Tp * Mp * (c.dtype == FFVD_F32C ? sizeof(float) : sizeof(double));
        size_t n = ((size_t)48 << 30) / (per_chain ? per_chain : 1);
        if (n < 1) n = 1;
        if (n > (size_t)c.S_local) n = c.S_local;
        h->cpp = (int)n;
    }
    if (h->cpp > c.S_local || c.grad || c.T_total > 0) h->cpp = c.S_local;      // backward pass, T-shards: one pass over all chains
    HIP_TRY(hipSetDevice(c.device_id));
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    {   // the side stream carries a short latency-bound chain: give it the highest priority
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, hi));
    }
    HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_kuu, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_tiles, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_go, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_hwords, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_prior, hipEventDisableTiming));
    HIP_TRY(dev_alloc(h, &h->prior_sums, 16));
    HIP_TRY(dev_alloc(h, &h->X, (size_t)c.S_local * (c.T + 1) * c.D));
    HIP_TRY(dev_alloc(h, &h->Z, (size_t)c.M * P));
    HIP_TRY(dev_alloc(h, &h->U, (size_t)c.M * c.D));
    HIP_TRY(dev_alloc(h, &h->logvar, (size_t)c.D));
    HIP_TRY(dev_alloc(h, &h->loglen, (size_t)c.D * P));
    HIP_TRY(dev_alloc(h, &h->logQ, (size_t)c.D));
    HIP_TRY(dev_alloc(h, &h->CC, (size_t)c.D * c.Ydim));
    HIP_TRY(dev_alloc(h, &h->DD, (size_t)c.Ydim));
    HIP_TRY(dev_alloc(h, &h->logR, (size_t)c.Ydim * c.Ydim));
    HIP_TRY(dev_alloc(h, &h->Y, (size_t)c.T * c.Ydim));
    HIP_TRY(dev_alloc(h, &h->ctrl, (size_t)c.T * c.C));
    HIP_TRY(dev_alloc(h, &h->variance, Dl));
    HIP_TRY(dev_alloc(h, &h->len, Dl * P));
    HIP_TRY(dev_alloc(h, &h->Zs, Dl * Mp * P));
    HIP_TRY(dev_alloc(h, &h->zz, Dl * Mp));
    HIP_TRY(dev_alloc(h, &h->Kuu, Dl * 2 * Mp * Mp));
    // reference route of the collapsed branch: K_fu is built once (kfu_build) and projected by a triangular GEMM,
    // unless FFVD_FUSED_PROJECT asks for the older kernel that generates K_fu inside the projection
    // (explicit-U branch: only from 64 units of the headline size on -- below that the single fused kernel is quicker:
    //  1.10 vs 1.21 ms on config 5's 16 units -- or when the backward pass needs K_fu anyway)
    const bool big_a = (size_t)h->nbatch * Tp * Mp >= (size_t)64 * 4096 * 512 || c.grad;
    const bool grad_ref = c.grad && c.branch == FFVD_BRANCH_B && c.route == FFVD_ROUTE_REFERENCE;    // backward pass in the reference's op order
    const bool proj_gemm = ((c.branch == FFVD_BRANCH_A && big_a) || (c.branch == FFVD_BRANCH_B && c.route == FFVD_ROUTE_REFERENCE)) &&
                           (!h->sw.fused_project || grad_ref);
    h->ngr = (proj_gemm && c.dtype != FFVD_F32C) ? (int)((Mp + 127) / 128) : 0;
    HIP_TRY(dev_alloc(h, &h->rowsq, (size_t)h->nbatch * (h->ngr > h->ng ? h->ngr : h->ng) * Tp));
    HIP_TRY(dev_alloc(h, &h->fmean, (size_t)h->nbatch * (h->ngr > h->ng ? h->ngr : h->ng) * Tp));
    if (c.branch == FFVD_BRANCH_B) {
        const size_t pass_b = (size_t)h->cpp * Dl;
        if (c.dtype == FFVD_F32C) {
            HIP_TRY(dev_alloc(h, &h->Kf32, pass_b * Tp * Mp));
            HIP_TRY(dev_alloc(h, &h->F32, pass_b * Tp * Mp));
            HIP_TRY(dev_alloc(h, &h->Linv32, Dl * Mp * Mp));
            h->nsq = proj_f32_ntiles((int)Tp, (int)Mp);
            HIP_TRY(dev_alloc(h, &h->sqpart, (size_t)h->nbatch * h->nsq));
            HIP_TRY(dev_alloc(h, &h->sqsum, (size_t)h->nbatch));
            // fp32 summation chains of the Gram product are cut every 4096 rows (128 t-tiles); FFVD_F32C_FLUSH overrides
            h->gram_flush = 128;
            if (const char *e = getenv("FFVD_F32C_FLUSH")) h->gram_flush = atoi(e);
        } else {
            HIP_TRY(dev_alloc(h, &h->F, pass_b * Tp * Mp));
            if (h->ngr) HIP_TRY(dev_alloc(h, &h->Kf2, pass_b * Tp * Mp));
        }
        const size_t hrows = c.grad ? 2 * Mp + NB : Mp + NB;     // grad: Mp identity rows (-> L_A^-T) before the b row
        HIP_TRY(dev_alloc(h, &h->H, pass_b * hrows * Mp));
        HIP_TRY(hipMemsetAsync(h->H, 0, pass_b * hrows * Mp * sizeof(double), h->stream));
    }
    h->ntiles = gram_ntiles(h->Mp);
    const bool grad_a = c.grad && c.branch == FFVD_BRANCH_A;
    if (c.branch == FFVD_BRANCH_A && (grad_a || h->ngr)) {                       // K_fu (projection GEMM input, backward pass)
        HIP_TRY(dev_alloc(h, &h->F, (size_t)(grad_a ? h->nbatch : h->cpp * Dl) * Tp * Mp));
        HIP_TRY(dev_alloc(h, &h->ucolA, Dl * Mp));
    }
    if ((c.branch == FFVD_BRANCH_B && c.route == FFVD_ROUTE_GRAM) || grad_a || grad_ref) {
        HIP_TRY(dev_alloc(h, &h->Kcopy, Dl * Mp * Mp));
        HIP_TRY(dev_alloc(h, &h->Linv, Dl * Mp * Mp));
        HIP_TRY(hipMemsetAsync(h->Linv, 0, Dl * Mp * Mp * sizeof(double), h->stream));     // blocks above the diagonal stay zero
        HIP_TRY(dev_alloc(h, &h->Kinv, Dl * Mp * Mp));
        HIP_TRY(dev_alloc(h, &h->trpart, (size_t)h->nbatch * h->ntiles));
        HIP_TRY(dev_alloc(h, &h->kterms, Dl * 2));
    }
    if (c.grad) {
        ffvd_handle::GradWs &g = h->gw;
        const size_t nbt = h->nbatch, msq = Mp * Mp, nblk = Tp / 64, nblk2 = Mp / 64, S = c.S_local, J = c.Ydim;
        g.ngam = atb_ntiles_sym64(h->Mp);        // the Gamma launch uses the 64 x 64-tile kernel
        g.sp_stride = c.D * c.Ydim + 2 * c.Ydim + (int)Dl;
        HIP_TRY(dev_alloc(h, &g.Acopy, nbt * msq));      HIP_TRY(dev_alloc(h, &g.u, nbt * Mp));
        HIP_TRY(dev_alloc(h, &g.LAinv, nbt * msq));      HIP_TRY(dev_alloc(h, &g.Gamma, nbt * msq));
        HIP_TRY(dev_alloc(h, &g.gam_part, nbt * g.ngam)); HIP_TRY(dev_alloc(h, &g.uku, nbt));
        if (c.dtype == FFVD_F32C) HIP_TRY(dev_alloc(h, &g.Gam32, nbt * msq));                          // E formed on the fly from fp32 operands
        else if (P <= 6) HIP_TRY(dev_alloc(h, &g.rp, bwd_fused_rp_doubles((int)Mp, (int)Tp, (int)nbt)));   // fused E reductions
        else HIP_TRY(dev_alloc(h, &g.E, nbt * Tp * Mp));
        if (grad_ref && c.dtype != FFVD_F32C) HIP_TRY(dev_alloc(h, &g.fsq, nbt));
        HIP_TRY(dev_alloc(h, &g.rsum, nbt * Tp));        HIP_TRY(dev_alloc(h, &g.ez, nbt * Tp * P));
        HIP_TRY(dev_alloc(h, &g.kfu, nbt * Tp));
        HIP_TRY(dev_alloc(h, &g.cs_part, nbt * nblk * Mp)); HIP_TRY(dev_alloc(h, &g.etx_part, nbt * nblk * Mp * P));
        HIP_TRY(dev_alloc(h, &g.rx2_part, nbt * nblk * P));
        HIP_TRY(dev_alloc(h, &g.dz_unit, nbt * c.M * P)); HIP_TRY(dev_alloc(h, &g.dll_unit, nbt * P));
        HIP_TRY(dev_alloc(h, &g.dls_unit, nbt));
        HIP_TRY(dev_alloc(h, &g.Asum, Dl * msq));  HIP_TRY(dev_alloc(h, &g.GamSum, Dl * msq)); HIP_TRY(dev_alloc(h, &g.Gs, Dl * msq));
        HIP_TRY(dev_alloc(h, &g.gsum, Dl * msq));  HIP_TRY(dev_alloc(h, &g.P1, Dl * msq));     HIP_TRY(dev_alloc(h, &g.KGK, Dl * msq));
        HIP_TRY(dev_alloc(h, &g.Epsi, Dl * msq));
        // (the reference route factorises H = I + F^T F / Q itself: its backward pass is the whitened one by construction)
        g.whitened = c.branch == FFVD_BRANCH_B && (!h->sw.grad_explicit || grad_ref);
        if (g.whitened) {
            HIP_TRY(dev_alloc(h, &g.T1, nbt * msq));
            HIP_TRY(dev_alloc(h, &g.wv, nbt * Mp));    HIP_TRY(dev_alloc(h, &g.bw, nbt * Mp));
            HIP_TRY(dev_alloc(h, &g.Ident, msq));      HIP_TRY(dev_alloc(h, &g.P2, Dl * msq)); HIP_TRY(dev_alloc(h, &g.P3, Dl * msq));
            launch_set_identity(h->stream, g.Ident, 0, 0, (int)Mp, 1);
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        HIP_TRY(dev_alloc(h, &g.rsum2, Dl * Mp));  HIP_TRY(dev_alloc(h, &g.ez2, Dl * Mp * P));
        HIP_TRY(dev_alloc(h, &g.cs2, Dl * nblk2 * Mp)); HIP_TRY(dev_alloc(h, &g.etx2, Dl * nblk2 * Mp * P));
        HIP_TRY(dev_alloc(h, &g.rx22, Dl * nblk2 * P));
        HIP_TRY(dev_alloc(h, &g.dz_kuu, Dl * c.M * P)); HIP_TRY(dev_alloc(h, &g.dll_kuu, Dl * P)); HIP_TRY(dev_alloc(h, &g.dls_kuu, Dl));
        HIP_TRY(dev_alloc(h, &g.shared_part, S * g.sp_stride));
        {   // every parameter gradient lives in ONE block (see GradWs::pack); the six small shared-parameter gradients are
            // contiguous inside it: one fill per backward pass instead of six memsets
            auto seg = [](size_t n) { return (n + 31) / 32 * 32; };
            g.small_count = (size_t)c.D + (size_t)c.D * P + (size_t)c.D + (size_t)c.D * J + J + J * J;
            const size_t nZ = (size_t)c.M * P, nU = grad_a ? (size_t)c.M * c.D : 0, nX = S * (c.T + 1) * c.D;
            const size_t oZ = seg(8), oS = oZ + seg(nZ), oU = oS + seg(g.small_count), oX = oU + seg(nU);
            g.pack_shared = oX; g.pack_total = oX + seg(nX);
            HIP_TRY(dev_alloc(h, &g.pack, g.pack_total));
            HIP_TRY(hipMemsetAsync(g.pack, 0, g.pack_total * sizeof(double), h->stream));     // the padding stays zero
            g.dZ = g.pack + oZ; g.dlogvar = g.pack + oS; g.dX = g.pack + oX;
            if (grad_a) g.dU = g.pack + oU;
            g.dloglen = g.dlogvar + c.D; g.dlogQ = g.dloglen + (size_t)c.D * P; g.dCC = g.dlogQ + c.D;
            g.dDD = g.dCC + (size_t)c.D * J; g.dlogR = g.dDD + J;
        }
        if (!grad_a && c.kernel_kind != FFVD_KERNEL_SE) HIP_TRY(dev_alloc(h, &g.xsq, nbt));      // LinearK, collapsed branch: sum_t |x_t|^2 per unit
        if (grad_a) {
            HIP_TRY(dev_alloc(h, &g.Gu, nbt * (Mp + NB) * Mp));   HIP_TRY(dev_alloc(h, &g.Gsum, Dl * (Mp + NB) * Mp));
            HIP_TRY(dev_alloc(h, &g.r, nbt * Tp));                HIP_TRY(dev_alloc(h, &g.dalpha, nbt));
            HIP_TRY(dev_alloc(h, &g.xsq, nbt));
            HIP_TRY(dev_alloc(h, &g.ucol, Dl * Mp));              HIP_TRY(dev_alloc(h, &g.beta, Dl * Mp));
            HIP_TRY(dev_alloc(h, &g.du, Dl * Mp));                HIP_TRY(dev_alloc(h, &g.GammaA, Dl * msq));
            HIP_TRY(dev_alloc(h, &g.Lclean, Dl * msq));           // (g.dU: a segment of g.pack)
        }
    }
    HIP_TRY(dev_alloc(h, &h->hterms, (size_t)h->nbatch * 2));
    if (c.T_total > 0) {        // the chain sums live at the tail of the exchange buffer: one all-reduce covers both
        const size_t raw = (size_t)h->nbatch * (Mp + 1) * Mp;
        h->ts_count = (int64_t)(raw + (size_t)c.S_local * 8);
        HIP_TRY(dev_alloc(h, &h->tsbuf, (size_t)h->ts_count));
        HIP_TRY(hipMemsetAsync(h->tsbuf, 0, (size_t)h->ts_count * sizeof(double), h->stream));   // tiles above the diagonal stay 0
        h->chain_terms = h->tsbuf + raw;
    } else
        HIP_TRY(dev_alloc(h, &h->chain_terms, (size_t)c.S_local * 8));
    HIP_TRY(dev_alloc(h, &h->chain_partial, (size_t)c.S_local * 128));        // [S][<= 32 row ranges][4] (launch_chain_reduce)
    {
        const size_t nS = (size_t)(c.S_local ? c.S_local : 1), ninfo = (size_t)(Dl + h->nbatch);
        const size_t ndbl = 8 + nS + (ninfo + 1) / 2;
        HIP_TRY(dev_alloc(h, &h->resblk, ndbl));
        h->res_bytes = ndbl * sizeof(double);
        h->out_terms = h->resblk;
        h->chain_nll = h->resblk + 8;
        h->info = reinterpret_cast<int32_t *>(h->resblk + 8 + nS);
        HIP_TRY(hipHostMalloc((void **)&h->h_res, h->res_bytes + 8 * sizeof(double)));
        h->h_sums = h->h_res + ndbl;                 // whole-job sums of a sharded training step (behind the result block)
        h->h_out = h->h_res;
        h->h_chain = h->h_res + 8;
        h->h_info = reinterpret_cast<int32_t *>(h->h_res + 8 + nS);
    }
    if (c.branch == FFVD_BRANCH_B && c.dtype != FFVD_F32C) {
        const int upass = h->cpp * (int)Dl;
        // the K_fu build forms delta^T K_fu (Gram route) / the projection GEMM forms delta^T F (reference route): the Gram kernel has no row
        const bool ext_row = (c.route == FFVD_ROUTE_GRAM && c.T_total == 0) || (c.route == FFVD_ROUTE_REFERENCE && h->ngr > 0 && !h->sw.ref_row_in_gram);
        // Few chains (one pass, no backward pass): the K_uu side chain goes BEHIND the tile pass as one dataflow launch (beside Cholesky(A)),
        // and the tile pass takes the row ranges that fill the chip's 512 slots exactly -- beside the pass, the chain's two dozen launches
        // only ran where the pass left slots free, which is what held the pass at three ranges (plan_schedule, side_late)
        const bool late_ok = c.route == FFVD_ROUTE_GRAM && !c.grad && c.T_total == 0 && c.S_local <= h->cpp && !h->sw.no_side_late &&
                             !h->sw.no_defer_trace && !h->sw.no_late_join && !h->sw.no_main_first && !h->sw.chain_rl &&
                             (size_t)Dl * 2 * (Mp / NB) <= 64 && (size_t)Dl * 2 * (Mp / NB) > (size_t)h->sw.small_side_rows &&
                             potrf_flow_forms_inverse((int)Mp, (int)Dl, CHOL_FLOW);
        h->side_late = late_ok && upass <= 96;
        if (h->side_late) {
            h->gsplit = gram_ksplit((int)Mp, upass, (int)Tp, ext_row ? 0 : 1, true);
            if (h->gsplit <= 1) h->side_late = false;
        }
        if (!h->side_late) h->gsplit = gram_ksplit((int)Mp, upass, (int)Tp, ext_row ? 0 : 1);
        if (ext_row) HIP_TRY(dev_alloc(h, &h->growpart, (size_t)upass * (Tp / 64) * Mp));
        h->gtail_wg = gram_tail_wg((int)Mp, upass, h->gsplit, ext_row ? 0 : 1);
        if (h->gtail_wg > 0) {
            HIP_TRY(dev_alloc(h, &h->gtail, gram_tail_doubles(h->gtail_wg)));
            HIP_TRY(hipMemsetAsync(h->gtail, 0, gram_tail_doubles(h->gtail_wg) * sizeof(double), h->stream));   // counters start at 0
        }
        if (h->gsplit > 1) HIP_TRY(dev_alloc(h, &h->gpart, gram_part_doubles((int)Mp, upass, h->gsplit)));
        else if (c.route == FFVD_ROUTE_GRAM && h->sw.kuu_flow && (size_t)upass * Tp * Mp >= (size_t)128 * 4096 * 512)
            // a K_fu build of 0.4 ms or more: the K_uu chain as ONE dataflow launch finishes beside it, the main stream
            // joins before the Gram kernel, which then forms the trace partials in its own epilogue
            h->kuu_flow_sched = true;
        else if (c.route == FFVD_ROUTE_GRAM && (size_t)upass * Tp * Mp >= (size_t)64 * 4096 * 512 && !h->sw.no_defer_trace) {
            HIP_TRY(dev_alloc(h, &h->graw, (size_t)upass * (Mp + 1) * Mp));   // raw tiles + trace pass beside Cholesky(A)
            // (unsplit pass of up to 16 chains: the chain's launches were starved by the pass and ended 0.1 ms behind Cholesky(A) -- behind
            //  the pass as one dataflow launch as well)
            h->side_late = late_ok && upass <= 64;
        }
    }
    // Pass-pipelined forward iteration (enqueue_elbo_pipe): only where the full-batch schedule runs today
    // (one buffer pass over all chains, unsplit Gram launches, K_uu chain as one dataflow launch with L^-1 and K^-1 out of it)
    if (h->kuu_flow_sched && !c.grad && c.T_total == 0 && h->cpp >= c.S_local && !h->gpart && !h->graw && h->growpart) {
        int passes = 0;
        if (const char *e = getenv("FFVD_PIPE")) passes = atoi(e);
        if (const char *e = getenv("FFVD_PIPE_MODE")) h->pipe_mode = atoi(e);
        if (passes > c.S_local) passes = c.S_local;
        if (passes >= 2 && potrf_flow_forms_inverse((int)Mp, (int)Dl, CHOL_FLOW) && !h->sw.kinv_gram) {
            h->pipe_passes = passes;
            int lo = 0, hi = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIP_TRY(hipStreamCreateWithFlags(&h->pipe_build, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&h->pipe_gram, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithPriority(&h->pipe_chol, hipStreamNonBlocking, (h->pipe_mode & 4) ? lo : hi));
            h->pipe_evB.resize(passes); h->pipe_evG.resize(passes);
            for (int i = 0; i < passes; ++i) {
                HIP_TRY(hipEventCreateWithFlags(&h->pipe_evB[i], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&h->pipe_evG[i], hipEventDisableTiming));
            }
            HIP_TRY(hipEventCreateWithFlags(&h->pipe_evC, hipEventDisableTiming));
            const int max_units = ((c.S_local + passes - 1) / passes) * (int)Dl;
            h->pipe_dinv_stride = (potrf_scratch_doubles((int)Mp, max_units) + 31) / 32 * 32;
            HIP_TRY(dev_alloc(h, &h->pipe_dinv, h->pipe_dinv_stride * passes));
        }
    }
    if (c.branch == FFVD_BRANCH_A && !c.grad && !h->sw.no_linear_lowrank && linear_lowrank_supported(c.kernel_kind, P))
        HIP_TRY(dev_alloc(h, &h->lrpart, linear_lowrank_doubles((int)Mp, (int)Dl, P)));
    if (c.dtype == FFVD_F64 && c.T_total == 0 && !h->sw.no_tiny && !(c.branch == FFVD_BRANCH_A && h->sw.no_tiny_a)) {      // (both branches since round 5)
        // The whole iteration as ONE launch when every role's workgroup fits on the chip at once (tiny.hip).  The multi-kernel
        // workspaces above stay: a launch abandoned on a bounded wait is re-run on them (fetch_with_stall_recovery).
        int cus = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c.device_id) == hipSuccess) cus = prop.multiProcessorCount;
        h->tiny = tiny_plan(c.kernel_kind, c.T, c.D, c.C, c.M, c.S_local, (int)Dl, c.grad, cus);
        h->tiny_cus = cus;
        if (h->tiny.ok) {
            // The plan needs every workgroup resident at once, and a wave is resident only with its scratch: the figure the loaded
            // code object reports must be the one this file was validated with, and the launch's total must stay within a budget
            // the runtime grants without throttling wave slots (checked where the handle is made, not assumed).
            size_t pb = 0;
            const int wpu_max = 1 + h->tiny.nstrips + (h->tiny.side ? h->tiny.NT : 0);
            if (tiny_kernel_private_bytes(h->tiny.nw, c.branch == FFVD_BRANCH_B ? 1 : 0, &pb) != hipSuccess || pb > TINY_PRIVATE_BYTES_MAX ||
                pb * 64 * h->tiny.nw * (size_t)h->tiny.nunits * wpu_max > TINY_PRIVATE_LAUNCH_BUDGET) {
                char wmsg[256];
                snprintf(wmsg, sizeof wmsg, "warning: one-launch iteration not used: its kernel reports %zu bytes of private memory per lane "
                         "(validated up to %zu; launch budget %zu MB)", pb, TINY_PRIVATE_BYTES_MAX, TINY_PRIVATE_LAUNCH_BUDGET >> 20);
                h->warning = wmsg;
                h->err = h->warning;
                h->tiny.ok = false;
            }
            h->tiny_private_bytes = pb;
        }
        if (h->tiny.ok) {
            HIP_TRY(dev_alloc(h, &h->tiny_scratch, tiny_scratch_doubles(h->tiny, c.T, (int)P, c.M, c.S_local, (int)Dl, c.D, c.Ydim, c.grad)));
            HIP_TRY(dev_alloc(h, &h->tiny_flags, tiny_flag_ints(h->tiny, c.S_local)));
            HIP_TRY(hipMemsetAsync(h->tiny_flags, 0, tiny_flag_ints(h->tiny, c.S_local) * sizeof(int), h->stream));
            HIP_TRY(dev_alloc(h, &h->tiny_dargs, 2));
            HIP_TRY(tiny_ring_create(h->tiny_ring[0]));
            HIP_TRY(tiny_ring_create(h->tiny_ring[1]));
            h->tiny_ring_made = true;
        }
    }
    HIP_TRY(dev_alloc(h, &h->dinvK, potrf_scratch_doubles((int)Mp, (int)Dl)));
    HIP_TRY(dev_alloc(h, &h->dinvH, potrf_scratch_doubles((int)Mp, h->nbatch ? h->nbatch : 1)));
    HIP_TRY(hipMemsetAsync(h->U, 0, (size_t)(c.M * c.D ? c.M * c.D : 1) * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->loglen, 0, (size_t)c.D * P * sizeof(double), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)cfg;
    return FFVD_OK;
}

extern "C" int ffvd_create(const ffvd_config *cfg, ffvd_handle **out) {
    if (!cfg || !out) return set_error(nullptr, FFVD_EINVAL, "ffvd_create: null argument");
    *out = nullptr;
    char msg[256];
    if (cfg->T < 1 || cfg->D < 1 || cfg->C < 0 || cfg->M < 1 || cfg->S_local < 1 || cfg->Ydim < 1) {
        snprintf(msg, sizeof msg, "ffvd_create: bad shape T=%d D=%d C=%d M=%d S_local=%d Ydim=%d", cfg->T, cfg->D,
                 cfg->C, cfg->M, cfg->S_local, cfg->Ydim);
        return set_error(nullptr, FFVD_EINVAL, msg);
    }
    if (cfg->D + cfg->C > MAXP) {
        snprintf(msg, sizeof msg, "ffvd_create: GP input dim P = D + C = %d exceeds %d", cfg->D + cfg->C, MAXP);
        return set_error(nullptr, FFVD_EINVAL, msg);
    }
    if (cfg->dtype != FFVD_F64 && cfg->dtype != FFVD_F32C) return set_error(nullptr, FFVD_EINVAL, "ffvd_create: unknown dtype");
    if (cfg->dtype == FFVD_F32C && (cfg->branch != FFVD_BRANCH_B || cfg->route != FFVD_ROUTE_REFERENCE))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: FFVD_F32C is the collapsed-U branch on FFVD_ROUTE_REFERENCE");
    if (cfg->kernel_kind != FFVD_KERNEL_SE && cfg->kernel_kind != FFVD_KERNEL_LINEAR)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: unknown kernel_kind");
    if (cfg->branch != FFVD_BRANCH_A && cfg->branch != FFVD_BRANCH_B)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: unknown branch");
    if (cfg->prior_type != FFVD_PRIOR_UNIFORM && cfg->prior_type != FFVD_PRIOR_NORMAL)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: unsupported prior_type (uniform / normal only)");
    const int dcount = cfg->d_count > 0 ? cfg->d_count : cfg->D;
    if (cfg->d_begin < 0 || cfg->d_begin + dcount > cfg->D)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: latent-dim shard [d_begin, d_begin + d_count) out of range");
    if (!(cfg->jitter >= 0.0)) return set_error(nullptr, FFVD_EINVAL, "ffvd_create: jitter must be >= 0");
    if (cfg->route != FFVD_ROUTE_REFERENCE && cfg->route != FFVD_ROUTE_GRAM)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: unknown route");
    if (cfg->grad && cfg->branch == FFVD_BRANCH_B && cfg->kernel_kind != FFVD_KERNEL_SE && cfg->dtype == FFVD_F32C)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: grad = 1 with fp32 contractions needs the SE kernel (the LinearK backward pass is fp64)");
    if (cfg->route == FFVD_ROUTE_GRAM && cfg->branch != FFVD_BRANCH_B)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: FFVD_ROUTE_GRAM only applies to the collapsed-U branch");
    if (cfg->T_total < 0 || cfg->t_begin < 0 || (cfg->T_total > 0 && cfg->t_begin + cfg->T > cfg->T_total))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: T-shard [t_begin, t_begin + T) outside [0, T_total)");
    if (cfg->T_total > 0 && (cfg->branch != FFVD_BRANCH_B || cfg->route != FFVD_ROUTE_GRAM || cfg->dtype != FFVD_F64))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: a T-shard (T_total > 0) is the collapsed-U branch on FFVD_ROUTE_GRAM in fp64");
    if (cfg->T_total > 0 && cfg->grad && (cfg->d_begin != 0 || (cfg->d_count > 0 && cfg->d_count != cfg->D)))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_create: grad = 1 on a T-shard needs every latent dim on the handle (T-shards and dim shards do not combine in the backward pass)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device_id < 0 || cfg->device_id >= ndev) {
        snprintf(msg, sizeof msg, "ffvd_create: device %d not available (%d HIP devices visible)", cfg->device_id, ndev);
        return set_error(nullptr, FFVD_EDEVICE, msg);
    }
    ffvd_handle *h = new (std::nothrow) ffvd_handle();
    if (!h) return set_error(nullptr, FFVD_ENOMEM, "ffvd_create: host allocation failed");
    h->cfg = *cfg;
    int rc = create_impl(cfg, h);
    if (rc != FFVD_OK) {
        std::string m = h->err;
        ffvd_destroy(h);
        return set_error(nullptr, rc, m);
    }
    *out = h;
    return FFVD_OK;
}

static int check_info(ffvd_handle *h);

extern "C" int ffvd_sync(ffvd_handle *h) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_sync: null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->info_pending) {      // report what the _async forms could not: a failed factorisation (include/ffvd_abi.h)
        h->info_pending = false;
        HIP_TRY(hipMemcpyAsync(h->h_info, h->info, (size_t)(h->Dl + h->nbatch) * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return check_info(h);
    }
    return FFVD_OK;
}

static int copy_in(ffvd_handle *h, double *dst, const double *src, size_t count, int on_device) {
    if (count == 0) return FFVD_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, count * sizeof(double), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           h->stream));
    return FFVD_OK;
}

extern "C" int ffvd_set_data(ffvd_handle *h, const double *Y, const double *control_inputs, int on_device) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_set_data: null handle");
    const ffvd_config &c = h->cfg;
    if (!Y) return set_error(h, FFVD_EINVAL, "ffvd_set_data: Y is null");
    if (c.C > 0 && !control_inputs) return set_error(h, FFVD_EINVAL, "ffvd_set_data: control_inputs is null but C > 0");
    HIP_TRY(hipSetDevice(c.device_id));
    int rc;
    if ((rc = copy_in(h, h->Y, Y, (size_t)c.T * c.Ydim, on_device))) return rc;
    if ((rc = copy_in(h, h->ctrl, control_inputs, (size_t)c.T * c.C, on_device))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_data = true;
    return FFVD_OK;
}

static int check_params(ffvd_handle *h, const ffvd_params *p, const char *who) {
    const ffvd_config &c = h->cfg;
    std::string w(who);
    if (!p) return set_error(h, FFVD_EINVAL, w + ": params is null");
    if (!p->X || !p->Z || !p->logvariance || !p->log_Q || !p->CC || !p->DD || !p->log_Rchols)
        return set_error(h, FFVD_EINVAL, w + ": a required parameter pointer is null (X, Z, logvariance, log_Q, CC, DD, log_Rchols)");
    if (c.kernel_kind == FFVD_KERNEL_SE && !p->loglengthscales)
        return set_error(h, FFVD_EINVAL, w + ": loglengthscales is null for the SquaredExponential kernel");
    if (c.branch == FFVD_BRANCH_A && !p->U)
        return set_error(h, FFVD_EINVAL, w + ": U is null in the explicit-U branch");
    return FFVD_OK;
}

extern "C" int ffvd_set_params(ffvd_handle *h, const ffvd_params *p, int on_device) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_set_params: null handle");
    int rc = check_params(h, p, "ffvd_set_params");
    if (rc) return rc;
    const ffvd_config &c = h->cfg;
    const size_t P = h->P;
    HIP_TRY(hipSetDevice(c.device_id));
    if ((rc = copy_in(h, h->X, p->X, (size_t)c.S_local * (c.T + 1) * c.D, on_device))) return rc;
    if ((rc = copy_in(h, h->Z, p->Z, (size_t)c.M * P, on_device))) return rc;
    if (p->U && (rc = copy_in(h, h->U, p->U, (size_t)c.M * c.D, on_device))) return rc;
    if ((rc = copy_in(h, h->logvar, p->logvariance, (size_t)c.D, on_device))) return rc;
    if (p->loglengthscales && (rc = copy_in(h, h->loglen, p->loglengthscales, (size_t)c.D * P, on_device))) return rc;
    if ((rc = copy_in(h, h->logQ, p->log_Q, (size_t)c.D, on_device))) return rc;
    if ((rc = copy_in(h, h->CC, p->CC, (size_t)c.D * c.Ydim, on_device))) return rc;
    if ((rc = copy_in(h, h->DD, p->DD, (size_t)c.Ydim, on_device))) return rc;
    if ((rc = copy_in(h, h->logR, p->log_Rchols, (size_t)c.Ydim * c.Ydim, on_device))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->cur.X = h->X; h->cur.Z = h->Z; h->cur.U = h->U; h->cur.logvariance = h->logvar;
    h->cur.loglengthscales = h->loglen; h->cur.log_Q = h->logQ; h->cur.CC = h->CC; h->cur.DD = h->DD;
    h->cur.log_Rchols = h->logR;
    h->have_params = true;
    return FFVD_OK;
}

// ---- the per-iteration launch sequence -------------------------------------------------------
// Stage timing: an event is recorded after each stage's launches; the interval ending at an event is
// attributed to that stage (stage -1 = origin of an iteration).  Events come from a pool owned by the handle.
// FFVD_DEBUG_SYNC=1: print the name of the launch group just enqueued and wait for both streams, so that a faulting kernel is the
// one named last (diagnostic only; read once per handle)
#define DBG_SYNC(h, name)                                                                              \
    do {                                                                                               \
        if ((h)->sw.debug_sync) {                                                                      \
            fprintf(stderr, "[ffvd debug] %s ...", name); fflush(stderr);                             \
            hipError_t e1_ = hipStreamSynchronize((h)->stream), e2_ = hipStreamSynchronize((h)->aux);  \
            fprintf(stderr, " %s\n", (e1_ == hipSuccess && e2_ == hipSuccess) ? "ok" : hipGetErrorString(e1_ != hipSuccess ? e1_ : e2_)); \
        }                                                                                              \
    } while (0)

struct StageTimer {
    ffvd_handle *h;
    void mark(int stage_id) {
        if (h->ev_used == h->ev_pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            h->ev_pool.push_back(e);
            h->ev_stage.push_back(0);
        }
        hipEventRecord(h->ev_pool[h->ev_used], h->stream);
        h->ev_stage[h->ev_used] = stage_id;
        ++h->ev_used;
    }
};

// ---- which schedule an iteration runs: EVERY flag in one place -------------------------------------------------------
// enqueue_elbo consumes these and computes none of its own; what it still tracks by itself is launch PROGRESS (which buffers a
// launch already produced: linv_done, kinv_done, hwords_zeroed, reduce_done ...).  The only run-time input besides the handle's
// constants is the Cholesky variant forced for the calling thread (stall recovery / FFVD_CHOL), through potrf_flow_selected.
struct ElboSchedule {
    bool gram_route, grad_a, grad_ref;
    bool lt_rows, lt_virtual;       // Gram-route training: L^T in the extension rows of A (read in place by the dataflow kernel)
    int first_units, ns_first;
    bool late_join;                 // split-K first pass: the K_uu chain only has to be back for the combine pass
    bool defer_trace;               //   ... one pass: the combine pass waits for the K_uu COPY only, trace partials follow on the side stream
    bool main_first;                //   ... and the main stream's K_fu build + tile pass are enqueued before the chain's launches
    bool defer_full;                // unsplit first pass beside the chain: raw tiles kept, trace pass on the side stream
    bool side_chain;                // Gram route: the K_uu chain runs on the side stream
    bool kuu_flow;                  //   ... as ONE dataflow launch resident before the K_fu build, joined before the Gram kernel
    bool kfu_first, ident_on_side;
    bool zt_rows, ref_side;         // explicit-U / reference route: the chain's dataflow launch beside the K_fu build
    bool side_late;                 // few chains: the side chain as one dataflow launch BEHIND the tile pass, trace partials and reductions behind it on the side stream
    bool small_side;                // tiny iteration on the multi-kernel path: the side chain as one dataflow launch, reductions on the main stream
    bool chain_flow_here;           // the chain is the dataflow launch on the stream that builds K_uu (the build zeroes its words)
    bool reduce_early;
    const char *name;
};
static ElboSchedule plan_schedule(const ffvd_handle *h) {
    const ffvd_config &c = h->cfg;
    const int Mp = h->Mp, Tp = h->Tp, Dl = h->Dl;
    ElboSchedule sc{};
    sc.gram_route = (c.branch == FFVD_BRANCH_B && c.route == FFVD_ROUTE_GRAM);
    sc.grad_a = c.grad && c.branch == FFVD_BRANCH_A;
    sc.grad_ref = c.grad && c.branch == FFVD_BRANCH_B && !sc.gram_route;
    sc.first_units = ((c.S_local < h->cpp) ? c.S_local : h->cpp) * Dl;
    sc.ns_first = (h->cpp <= c.S_local) ? h->cpp : c.S_local;
    sc.late_join = sc.gram_route && h->gpart && sc.first_units == h->cpp * Dl && !h->sw.no_late_join;
    sc.lt_rows = c.grad && sc.gram_route && h->gw.whitened && !h->sw.whiten_products;
    sc.lt_virtual = sc.lt_rows && !h->sw.lt_armed && potrf_flow_selected(Mp, h->nbatch, CHOL_FLOW);
    sc.defer_trace = sc.late_join && c.S_local <= h->cpp && !h->sw.no_defer_trace;
    sc.main_first = sc.defer_trace && !h->sw.no_main_first;
    sc.defer_full = sc.gram_route && !sc.late_join && h->graw;
    sc.zt_rows = h->lrpart != nullptr;
    sc.ref_side = !sc.gram_route && h->ngr > 0 && c.dtype != FFVD_F32C && h->aux && !h->sw.no_ref_side && !h->sw.chain_rl &&
                  potrf_flow_selected(Mp, Dl, CHOL_FLOW);
    sc.side_chain = sc.gram_route && (sc.late_join || (size_t)sc.first_units * Tp * Mp >= (size_t)64 * 4096 * 512);
    sc.kuu_flow = sc.side_chain && h->kuu_flow_sched && !sc.late_join;
    sc.kfu_first = sc.side_chain && (sc.main_first || (sc.defer_full && !h->sw.no_kfu_first));
    sc.ident_on_side = sc.side_chain && c.grad && (sc.defer_full || sc.defer_trace) && !sc.lt_rows;
    const bool on_side = sc.ref_side || sc.side_chain;          // sk != s
    const bool kuu_on_main = sc.ref_side || sc.kuu_flow;        // K_uu is built (and its chain launched) before the common chain code
    sc.reduce_early = sc.gram_route && on_side;
    sc.small_side = sc.defer_trace && sc.main_first && !c.grad && !kuu_on_main && on_side && !h->sw.chain_rl && !h->sw.no_small_side &&
                    (size_t)sc.first_units * h->ntiles * (h->gsplit > 1 ? h->gsplit : 1) <= (size_t)h->sw.small_side_wgs &&
                    (size_t)Dl * 2 * (Mp / NB) <= (size_t)h->sw.small_side_rows;
    // (Round 4, measured and not kept: at 8-16 chains the side chain -- a dozen dependent right-looking launches -- is starved of slots by
    //  the tile pass and its tail ends 0.14 ms behind Cholesky(A).  As ONE dataflow launch beside the K_fu build it holds its slots
    //  against the tile pass instead: that pass 1.43 instead of 1.18 ms at 16 chains, the iteration 1.91 either way; a main stream whose
    //  CU mask leaves 8-32 compute units to the side stream runs 1.2 x slower; `small_side` widened to these sizes: 1.19 vs 1.14 ms at 8.)
    sc.side_late = h->side_late && ((sc.defer_trace && sc.main_first) || (sc.defer_full && sc.kfu_first)) && !kuu_on_main && on_side && !sc.small_side &&
                   potrf_flow_forms_inverse(Mp, Dl, CHOL_FLOW);            // (a forced launch-per-column Cholesky: the first-half schedule)
    sc.chain_flow_here = !kuu_on_main && (!on_side || sc.small_side || sc.side_late) && potrf_flow_selected(Mp, Dl, CHOL_FLOW) && !h->sw.chain_rl;
    // invariants the launch code relies on (a violated one would be a silent wrong answer, not a crash)
    if ((sc.defer_full && sc.late_join) || (sc.defer_trace && !sc.late_join) || (sc.kuu_flow && sc.defer_full) || (sc.small_side && sc.kuu_flow) ||
        (sc.side_late && (sc.small_side || !(sc.main_first || sc.defer_full) || !sc.chain_flow_here)) ||
        (sc.ref_side && sc.side_chain) || (sc.main_first && !sc.defer_trace) || (sc.kuu_flow && h->graw))
        sc.name = nullptr;
    else if (!sc.gram_route) sc.name = sc.ref_side ? "projection route, K_uu chain as one dataflow launch on the side stream beside the K_fu build"
                                                   : "projection route, serial (K_uu chain on the main stream)";
    else if (sc.kuu_flow) sc.name = "full unsplit: K_uu chain as one dataflow launch beside the K_fu build, joined before the Gram kernel";
    else if (sc.side_late && sc.defer_full) sc.name = "side late: unsplit pass with raw tiles, the K_uu chain as one dataflow launch behind it beside Cholesky(A), trace pass behind the chain";
    else if (sc.defer_full && sc.side_chain) sc.name = "unsplit with raw tiles: K_uu chain beside K_fu build and Gram kernel, trace pass on the side stream, joined at finalize";
    else if (sc.side_late) sc.name = "side late: split-K one pass filling the chip, the K_uu chain as one dataflow launch behind it beside Cholesky(A), trace partials behind the chain";
    else if (sc.small_side) sc.name = "small side: split-K one pass, side chain as one dataflow launch, reductions and trace partials on the main stream";
    else if (sc.defer_trace) sc.name = "split-K one pass, late join: combine waits for the K_uu copy, trace partials on the side stream";
    else if (sc.late_join) sc.name = "split-K several passes: the first combine pass waits for the whole chain";
    else sc.name = "serial: K_uu chain on the main stream in front of the K_fu build";
    return sc;
}
// (the pipelined passes need the dataflow Cholesky: a forced launch-per-column variant -- stall recovery, FFVD_CHOL -- takes the plain schedule)
static bool pipe_selected(const ffvd_handle *h) { return h->pipe_passes >= 2 && potrf_override_current() == CHOL_FORCE_NONE; }
extern "C" const char *ffvd_schedule_name(const ffvd_handle *h) {
    if (!h) return "";
    if (h->stall_hold > 0) return "stall back-off: multi-kernel schedule with the launch-per-column Cholesky (no inter-workgroup waits) until the next probe";
    if (h->tiny.ok && potrf_override_current() == CHOL_FORCE_NONE) return "one launch (tiny.hip)";
    if (pipe_selected(h)) return "pipelined passes: K_fu build of pass p+1 | Gram kernel of pass p | Cholesky(A) of pass p-1 on separate streams";
    const ElboSchedule sc = plan_schedule(h);
    return sc.name ? sc.name : "INVALID";
}

// Fork the side stream off the main stream.  Diagnostic switches (schedule tests): a spin kernel at the head of the side stream
// (FFVD_DEBUG_SIDE_DELAY_US) and / or of the main stream behind the fork (FFVD_DEBUG_MAIN_DELAY_US) -- results must not change.
static int fork_side(ffvd_handle *h, hipEvent_t ev, hipStream_t from, hipStream_t to) {
    HIP_TRY(hipEventRecord(ev, from));
    HIP_TRY(hipStreamWaitEvent(to, ev, 0));
    if (h->sw.side_delay_us > 0) launch_spin(to, h->sw.side_delay_us);
    if (h->sw.main_delay_us > 0) launch_spin(from, h->sw.main_delay_us);
    return FFVD_OK;
}

// The iteration (with_grad: and its backward pass, gradients scaled by 1 / S_total into the arrays of gw) as ONE launch: tiny.hip.
static bool tiny_selected(const ffvd_handle *h) { return h->tiny.ok && potrf_override_current() == CHOL_FORCE_NONE; }
static int enqueue_tiny(ffvd_handle *h, double *out_dev, StageTimer *st, bool with_grad, int S_total) {
    StageTimer live{h};
    if (!st && h->timing_on) st = &live;
    const ffvd_config &c = h->cfg;
    const ffvd_params &p = h->cur;
    hipStream_t s = h->stream;
    if (st) st->mark(-1);
    if (h->tiny_dirty) {
        HIP_TRY(hipMemsetAsync(h->tiny_flags, 0, tiny_flag_ints(h->tiny, c.S_local) * sizeof(int), s));
        h->tiny_dirty = false;
    }
    TinyArgs a;
    memset(&a, 0, sizeof a);        // padding included: the block is compared bytewise with what the device copy holds (launch_tiny)
    a.kind = c.kernel_kind; a.T = c.T; a.D = c.D; a.C = c.C; a.P = h->P; a.M = c.M; a.Dl = h->Dl; a.d_begin = c.d_begin;
    a.S = c.S_local; a.Ydim = c.Ydim; a.prior_type = c.prior_type; a.shared_terms = c.shared_terms;
    a.grad = with_grad ? 1 : 0; a.S_total = S_total; a.jitter = c.jitter;
    a.branch = (c.branch == FFVD_BRANCH_B) ? 1 : 0; a.U = p.U;
    a.X = p.X; a.Z = p.Z; a.logvar = p.logvariance; a.loglen = p.loglengthscales; a.log_Q = p.log_Q; a.CC = p.CC; a.DD = p.DD;
    a.logR = p.log_Rchols; a.Y = h->Y; a.ctrl = h->ctrl;
    tiny_bind_scratch(a, h->tiny, h->tiny_scratch, h->tiny_flags);
    {   // all workgroups of a unit on one XCD when the padded grid (8 x ceil(units / 8) unit slots) still fits the chip
        const int wpu = 1 + h->tiny.nstrips + (a.side ? h->tiny.NT : 0);
        a.xcd_map = (h->sw.tiny_xcd && 8 * wpu * ((h->tiny.nunits + 7) / 8) <= h->tiny_cus) ? 1 : 0;
    }
    a.info = h->info; a.chain_nll = h->chain_nll; a.out_terms = out_dev ? out_dev : h->out_terms;
    if (with_grad) {
        const ffvd_handle::GradWs &g = h->gw;
        a.dX = g.dX; a.dZ = g.dZ; a.dlogvar = g.dlogvar; a.dloglen = g.dloglen; a.dlogQ = g.dlogQ; a.dCC = g.dCC; a.dDD = g.dDD;
        a.dlogR = g.dlogR; a.dU = g.dU;
    }
    HIP_TRY(launch_tiny(s, a, h->tiny, h->tiny_dargs + (with_grad ? 1 : 0), h->tiny_ring[with_grad ? 1 : 0]));
    if (st) { st->mark(2); st->mark(4); }
    DBG_SYNC(h, with_grad ? "one-launch iteration + backward pass" : "one-launch iteration");
    return FFVD_OK;
}


// ---- argument blocks of the forward iteration's launches: one builder each, shared by every schedule --------------------------------
static ProjectArgs elbo_project_args(const ffvd_handle *h, const HyperView &hv, int s0, int ns, bool gram_route) {
    const ffvd_config &c = h->cfg;
    const ffvd_params &p = h->cur;
    const size_t msq = (size_t)h->Mp * h->Mp, kstride = 2 * msq;
    ProjectArgs pa{};
    pa.kind = c.kernel_kind;
    pa.x = p.X; pa.x_chain_stride = (size_t)(c.T + 1) * c.D; pa.x_ld = c.D; pa.x_cols = c.D;
    pa.ctrl = h->ctrl; pa.T = c.T; pa.Tp = h->Tp; pa.C = c.C; pa.P = h->P; pa.M = c.M; pa.Mp = h->Mp; pa.Dl = h->Dl;
    pa.d_begin = c.d_begin; pa.hv = hv; pa.W = h->Kuu + msq; pa.w_stride = kstride;
    pa.U = p.U; pa.u_ld = c.D; pa.b0 = s0 * h->Dl; pa.nb = ns * h->Dl;
    pa.F = (c.branch == FFVD_BRANCH_B) ? h->F : nullptr;
    pa.rowsq = h->rowsq;
    pa.fmean = (c.branch == FFVD_BRANCH_A) ? h->fmean : nullptr;
    pa.ng = h->ng;
    pa.gpart = gram_route ? h->growpart : nullptr;      // Gram route: delta^T K_fu is summed where K_fu is made
    return pa;
}
static size_t elbo_h_stride(const ffvd_handle *h) { return (size_t)(h->cfg.grad ? 2 * h->Mp + NB : h->Mp + NB) * h->Mp; }
static GramArgs elbo_gram_args(const ffvd_handle *h, int s0, int ns, bool gram_route) {
    const ffvd_config &c = h->cfg;
    const ffvd_params &p = h->cur;
    const int Mp = h->Mp;
    const size_t msq = (size_t)Mp * Mp;
    GramArgs ga{};
    ga.mode = gram_route ? GRAM_KFU : GRAM_F;
    ga.A = h->F; ga.a_stride = (size_t)h->Tp * Mp; ga.rows = h->Tp; ga.with_row = h->growpart ? 0 : 1;
    ga.X = p.X; ga.log_Q = p.log_Q; ga.T = c.T; ga.D = c.D; ga.Mp = Mp; ga.Dl = h->Dl;
    ga.d_begin = c.d_begin; ga.b0 = s0 * h->Dl; ga.nb = ns * h->Dl; ga.yn_over_batch = 1.0;
    ga.H = h->H; ga.h_stride = elbo_h_stride(h);
    if (c.grad) ga.brow = 2 * Mp;   // rows [Mp, 2Mp) hold I (they become L_A^-T), the b row moves to 2 Mp
    ga.Kadd = h->Kcopy; ga.kadd_stride = msq; ga.Kinv = h->Kinv; ga.kinv_stride = msq; ga.trpart = h->trpart;
    if (h->gpart) { ga.ksplit = h->gsplit; ga.part = h->gpart; }     // same row ranges in every pass of this handle
    // tail split of the last partial round (full passes only: launch_gram drops it when the unit count differs)
    if (h->gtail) { ga.tail_wg = h->gtail_wg; ga.tail_part = h->gtail; }
    return ga;
}
static ReduceArgs elbo_reduce_args(const ffvd_handle *h, bool gram_route) {
    const ffvd_config &c = h->cfg;
    const ffvd_params &p = h->cur;
    ReduceArgs ra{};
    ra.kind = c.kernel_kind; ra.branch = c.branch; ra.X = p.X; ra.ctrl = h->ctrl; ra.Y = h->Y;
    ra.log_Q = p.log_Q; ra.CC = p.CC; ra.DD = p.DD; ra.log_Rchols = p.log_Rchols; ra.variance = h->variance;
    ra.T = c.T; ra.Tp = h->Tp; ra.D = c.D; ra.C = c.C; ra.Ydim = c.Ydim; ra.Dl = h->Dl; ra.d_begin = c.d_begin;
    ra.S = c.S_local; ra.ng = h->ng; ra.shared_terms = c.shared_terms;
    ra.xk = p.X; ra.xk_chain_stride = (size_t)(c.T + 1) * c.D; ra.xk_ld = c.D; ra.xk_cols = c.D;
    ra.rowsq = (gram_route || c.dtype == FFVD_F32C) ? nullptr : h->rowsq; ra.fmean = h->fmean; ra.chain_terms = h->chain_terms;
    if (h->ngr) ra.ng = h->ngr;
    if (h->lrpart) ra.ng = 1;            // LinearK through its rank: one value per (unit, row)
    return ra;
}
// (trpart / ntiles / fsq_from_trpart of the fp32-contraction path and `whitened` of the training forward are set by the caller)
static FinalizeArgs elbo_finalize_args(const ffvd_handle *h, double *out_dev, bool gram_route) {
    const ffvd_config &c = h->cfg;
    const ffvd_params &p = h->cur;
    FinalizeArgs fa{};
    fa.kind = c.kernel_kind; fa.branch = c.branch; fa.prior_type = c.prior_type; fa.shared_terms = c.shared_terms;
    fa.T = c.T; fa.D = c.D; fa.P = h->P; fa.M = c.M; fa.Ydim = c.Ydim; fa.Dl = h->Dl; fa.d_begin = c.d_begin;
    fa.S = c.S_local; fa.Z = p.Z; fa.U = p.U; fa.logvar = p.logvariance; fa.loglen = p.loglengthscales;
    fa.log_Q = p.log_Q; fa.CC = p.CC; fa.DD = p.DD; fa.log_Rchols = p.log_Rchols;
    fa.chain_terms = h->chain_terms; fa.hterms = h->hterms; fa.chain_nll = h->chain_nll;
    fa.route = gram_route ? 1 : 0; fa.kterms = h->kterms; fa.trpart = h->trpart; fa.ntiles = h->ntiles;
    fa.out_terms = out_dev ? out_dev : h->out_terms;
    fa.info = h->info; fa.ninfo = h->Dl + h->nbatch;       // any failed / abandoned factorisation of this rank -> NaN sums on every rank
    return fa;
}


// ---- pass-pipelined forward iteration ----------------------------------------------------------------------------
// The per-(s, d) units of conditionals_multi_output.py:238-255 are independent, and the full-batch iteration is three phases bound by
// three different things: the K_fu build by HBM writes, the Gram kernel by the matrix pipe, Cholesky(A) by latency.  This schedule cuts
// the chains into `pipe_passes` passes (each with its own slice of F, H and growpart -- the buffers already hold all chains) and lets
// the phases of neighbouring passes run side by side:
//
//   main   : prep, K_uu build | reductions | build(0) | Gram(0) ........... Gram(2) ...                      | wait C | finalize
//   aux    :   K_uu chain (one dataflow launch, L^-1, K^-1, log|K|) --ev_join--> (every Gram launch waits for it once per stream)
//   build  :                     wait B0 | build(1) build(2) ...            (mode bit 0: build(p) waits for Gram(p-2) -- "lazy")
//   gram   :                                  wait B1 | Gram(1) ........... Gram(3) ...     (mode bit 1: every Gram launch on main)
//   chol   : clears of all passes' progress words | wait G0 | Chol(0) finish(0) | wait G1 | Chol(1) ...     --evC-->
//
// Odd passes' Gram launches go to a second stream so that Gram(p+1) can take the slots Gram(p)'s early finishers free (stream order
// would hold it until the last workgroup of Gram(p) has left: the ends of a round are spread over 0.3 ms).
// FFVD_PIPE=<passes> turns it on for handles that would run the "full unsplit" schedule; results are bit-identical to that schedule
// (same kernels on the same units; only the launch partition differs).
static int enqueue_elbo_pipe(ffvd_handle *h, double *out_dev, StageTimer *st) {
    const ffvd_config &c = h->cfg;
    const int Mp = h->Mp, Tp = h->Tp, Dl = h->Dl, P = h->P, NP = h->pipe_passes;
    const ffvd_params &p = h->cur;
    hipStream_t s = h->stream, sk = h->aux, sb = h->pipe_build, sg = (h->pipe_mode & 2) ? h->stream : h->pipe_gram, sc = h->pipe_chol;
    const size_t msq = (size_t)Mp * Mp, kstride = 2 * msq, fstride = (size_t)Tp * Mp, hstride = elbo_h_stride(h);
    const bool lazy = (h->pipe_mode & 1) != 0;
    const int nt = (h->pipe_mode & 8) ? 0 : 1;          // streaming stores for every pass's K_fu unless bit 3 says cacheable
    if (st) st->mark(-1);
    launch_prep_hypers(s, c.kernel_kind, p.Z, c.M, Mp, P, Dl, c.d_begin, p.logvariance, p.loglengthscales,
                       h->variance, h->len, h->Zs, h->zz, h->info, Dl + h->nbatch);
    HyperView hv{h->variance, h->len, h->Zs, h->zz};
    launch_kuu_build(s, c.kernel_kind, hv, c.M, Mp, P, Dl, c.jitter, h->Kuu, h->Kcopy);
    { int rcf = fork_side(h, h->ev_fork, s, sk); if (rcf) return rcf; }
    // the chain's row workgroups must be on the chip before the first K_fu build floods it (as in the full-batch schedule)
    potrf_flow_clear(sk, h->dinvK, Dl);
    HIP_TRY(hipEventRecord(h->ev_go, sk));
    launch_potrf_ext(sk, h->Kuu, Mp, Mp, Mp, Dl, kstride, h->info, h->dinvK, CHOL_FLOW, h->Linv, msq, true, false, h->Kinv, msq);
    launch_h_finish(sk, h->Kuu, Mp, kstride, Dl, h->kterms);
    HIP_TRY(hipEventRecord(h->ev_join, sk));
    launch_chain_reduce(s, elbo_reduce_args(h, true), h->chain_partial);      // inputs only; fills the wait below
    HIP_TRY(hipStreamWaitEvent(s, h->ev_go, 0));
    // progress words of every pass's factorisation: cleared up front on the stream that runs them
    HIP_TRY(hipStreamWaitEvent(sc, h->ev_fork, 0));
    for (int q = 0; q < NP; ++q) {
        const int c0 = (int)((long long)c.S_local * q / NP), c1 = (int)((long long)c.S_local * (q + 1) / NP);
        potrf_flow_clear(sc, h->pipe_dinv + h->pipe_dinv_stride * q, (c1 - c0) * Dl);
    }
    if (st) st->mark(0);
    for (int q = 0; q < NP; ++q) {
        const int c0 = (int)((long long)c.S_local * q / NP), c1 = (int)((long long)c.S_local * (q + 1) / NP);
        const int ns = c1 - c0, u0 = c0 * Dl, nu = ns * Dl;
        // K_fu of the pass (+ the row b from the partial sums it leaves)
        hipStream_t bs = q == 0 ? s : sb;
        if (q == 1) HIP_TRY(hipStreamWaitEvent(sb, h->pipe_evB[0], 0));            // builds one after the other
        if (lazy && q >= 2) HIP_TRY(hipStreamWaitEvent(sb, h->pipe_evG[q - 2], 0));  // ... and not before Gram(q-2) has left
        ProjectArgs pa = elbo_project_args(h, hv, c0, ns, true);
        pa.F = h->F + (size_t)u0 * fstride;
        pa.gpart = h->growpart + (size_t)u0 * (Tp / 64) * Mp;
        launch_kfu_build(bs, pa, nt);
        launch_brow_finish(bs, pa.gpart, Tp / 64, Mp, Dl, c.d_begin, u0, nu, p.log_Q, 1.0, h->H + (size_t)u0 * hstride, hstride, Mp);
        HIP_TRY(hipEventRecord(h->pipe_evB[q], bs));
        if (st && q == 0) st->mark(1);
        // Gram kernel of the pass (trace partials in its epilogue: needs K_uu's copy and K^-1 from the chain)
        hipStream_t gs = (q % 2 == 0) ? s : sg;
        if (gs != bs) HIP_TRY(hipStreamWaitEvent(gs, h->pipe_evB[q], 0));
        if (q < 2) HIP_TRY(hipStreamWaitEvent(gs, h->ev_join, 0));
        GramArgs ga = elbo_gram_args(h, c0, ns, true);
        ga.A = h->F + (size_t)u0 * fstride;
        ga.H = h->H + (size_t)u0 * hstride;
        ga.tail_wg = 0; ga.tail_part = nullptr;              // (the tail split's blocks are sized for the full batch's launch)
        launch_gram(gs, ga);
        HIP_TRY(hipEventRecord(h->pipe_evG[q], gs));
        // Cholesky(A) of the pass, log|A| and b^T A^-1 b
        HIP_TRY(hipStreamWaitEvent(sc, h->pipe_evG[q], 0));
        launch_potrf_ext(sc, ga.H, Mp, NB, 0, nu, hstride, h->info + Dl + u0, h->pipe_dinv + h->pipe_dinv_stride * q, CHOL_FLOW, nullptr, 0,
                         true, true);
        launch_h_finish(sc, ga.H, Mp, hstride, nu, h->hterms + (size_t)2 * u0);
    }
    if (st) { st->mark(2); }
    HIP_TRY(hipEventRecord(h->pipe_evC, sc));
    HIP_TRY(hipStreamWaitEvent(s, h->pipe_evC, 0));
    if (NP >= 2 && sg != s) HIP_TRY(hipStreamWaitEvent(s, h->pipe_evG[NP % 2 == 0 ? NP - 1 : NP - 2], 0));   // (implied by evC; keeps the graph explicit)
    if (st) st->mark(3);
    launch_finalize(s, elbo_finalize_args(h, out_dev, true));
    if (st) st->mark(4);
    DBG_SYNC(h, "forward (pipelined passes)");
    HIP_TRY(hipGetLastError());
    return FFVD_OK;
}

// ---- the multi-kernel forward iteration, phase by phase -------------------------------------------------------
// enqueue_elbo used to be one 425-line function; it is now the sequence below, each phase a function of its own over one context.
// The SCHEDULE is sc (plan_schedule: every flag decided in one place, named, checked); the context only adds launch PROGRESS -- which
// buffers a launch has produced so far -- and the two streams.  Launch order is exactly what it was: results are bit-identical.
//
// Event graphs of the named schedules (M = main stream, S = side stream; "->e" records, "e->" waits):
//   full unsplit (kuu_flow)   M: prep, K_uu build ->fork | reductions | go-> build(K_fu), brow | join-> Gram | hwords-> Chol(A), finish | finalize
//                             S: fork-> clear ->go | chain (one dataflow launch: L, L^-1, K^-1) | clear H words ->hwords | log|K| ->join
//   unsplit + raw tiles       M: prep ->fork | build, brow | kuu-> Gram(raw) ->tiles | Chol(A) | join2-> finalize
//   (defer_full)              S: fork-> K_uu build ->kuu | chain (launches) | K^-1, log|K| ->join | reductions | tiles-> trace pass ->join2
//   split-K one pass          M: prep ->fork | build, brow, tile pass ->tiles | kuu-> combine (no trace) | Chol(A) | join2-> finalize
//   (defer_trace, main_first) S: fork-> K_uu build ->kuu | chain | K^-1, log|K| ->join | reductions | tiles-> trace partials ->join2
//   side late                 as the two above, but the chain is ONE dataflow launch enqueued behind the tile pass (tiles-> chain), the
//                             reductions in front of it, the trace pass behind it
//   small side                split-K one pass with the chain as one dataflow launch; reductions and trace partials on M (join-> trace)
//   split-K several passes    M: ... tile pass | join-> combine with trace | Chol(A) ... per pass
//   projection route, side    M: prep, K_uu build ->fork | go-> build | join-> projection GEMM, brow | Gram | Chol(H) | reductions, finalize
//   (ref_side)                S: fork-> clear ->go | chain (dataflow) | [K^-1, log|K| for a backward pass] ->join
//   serial                    everything on M in dependence order.
struct FwdCtx {
    ffvd_handle *h;
    double *out_dev;
    StageTimer *st;
    ElboSchedule sc;
    HyperView hv;
    hipStream_t s, sk;              // main stream; the stream that carries the K_uu chain (= s while nothing runs beside the main stream)
    size_t msq, kstride;
    ReduceArgs ra;
    // launch progress, not schedule
    bool kuu_on_main = false;       // K_uu is built (and its chain launched) before the common chain code
    bool linv_done = false;         // L^-1 comes / came out of the factorisation itself
    bool kinv_done = false;         // K^-1 came out of the chain's dataflow launch (no product launch)
    bool hwords_zeroed = false;     // Cholesky(A)'s progress words were cleared on the side stream (ev_hwords)
    bool ident_early = false;       // training: the identity rows of the first pass were re-armed early on the main stream
    bool reduce_done = false, reduce_launched = false, trace_on_main = false;
    bool chain_deferred = false;    // the chain is enqueued behind the first pass's Gram launch
    int prior_early = 0;            // the parameter-only sums of the assembly were formed early: 1 on the main stream, 2 on the side stream (ev_prior)

    ProjectArgs project_args(int s0, int ns) const { return elbo_project_args(h, hv, s0, ns, sc.gram_route); }
    GramArgs gram_args(int s0, int ns) const { return elbo_gram_args(h, s0, ns, sc.gram_route); }
    // Gram route: the row b = delta^T K_fu / Q of a pass, from the partial sums its K_fu build left behind (kernels.h ProjectArgs::gpart)
    void brow_finish(int s0, int ns) const {
        if (!h->growpart) return;
        const ffvd_config &c = h->cfg;
        launch_brow_finish(s, h->growpart, h->Tp / 64, h->Mp, h->Dl, c.d_begin, s0 * h->Dl, ns * h->Dl, h->cur.log_Q, 1.0, h->H,
                           elbo_h_stride(h), c.grad ? 2 * h->Mp : h->Mp);
    }
};

// The ten parameter-only sums of the nll assembly (priors, log R, log sqrt Q: finalize_priors) are a function of the parameters alone:
// formed EARLY they leave the finalize launch at the iteration's tail -- one workgroup, a chain of dependent loads -- with the
// per-chain assembly only (19 -> 9 us of every forward iteration's tail; the same code, the same bits).  Where: on the main stream
// inside its wait for the chain's kernel (full-batch and projection-route schedules), or as the side stream's first launch where
// that stream idles at the start (few-chain schedules); schedules without a slack window keep them in the finalize launch.
static int fwd_prior_sums(FwdCtx &x, hipStream_t stream) {
    ffvd_handle *h = x.h;
    if (h->sw.no_early_priors || x.prior_early) return FFVD_OK;
    launch_prior_sums(stream, elbo_finalize_args(h, x.out_dev, x.sc.gram_route), h->prior_sums);
    if (stream != x.s) {
        HIP_TRY(hipEventRecord(h->ev_prior, stream));
        x.prior_early = 2;
    } else x.prior_early = 1;
    return FFVD_OK;
}

// Projection route / explicit-U branch on the projection GEMM (fp64): only the GEMM needs the chain's W = L^-T, the K_fu build does
// not -- the chain's dataflow launch goes to the side stream (resident before the build floods the chip) and the main stream waits
// for it in front of the first projection GEMM: config 2 in the reference's op order 6.31 -> 6.05 ms.  LinearK through its rank
// (zt_rows): the chain carries Z^T instead of the identity rows.
static int fwd_chain_ref_side(FwdCtx &x) {
    ffvd_handle *h = x.h;
    const ffvd_config &c = h->cfg;
    const int Mp = h->Mp, Dl = h->Dl;
    const bool zt_rows = x.sc.zt_rows, wants_linv = x.sc.grad_a || x.sc.grad_ref;
    x.sk = h->aux;
    launch_kuu_build(x.s, c.kernel_kind, x.hv, c.M, Mp, h->P, Dl, c.jitter, h->Kuu, wants_linv ? h->Kcopy : nullptr, zt_rows);
    { int rcf = fork_side(h, h->ev_fork, x.s, x.sk); if (rcf) return rcf; }
    potrf_flow_clear(x.sk, h->dinvK, Dl);
    HIP_TRY(hipEventRecord(h->ev_go, x.sk));
    x.linv_done = wants_linv;
    launch_potrf_ext(x.sk, h->Kuu, Mp, zt_rows ? NB : Mp, zt_rows ? 0 : Mp, Dl, x.kstride, h->info, h->dinvK, CHOL_FLOW,
                     x.linv_done ? h->Linv : nullptr, x.msq, true);
    { int rcp = fwd_prior_sums(x, x.s); if (rcp) return rcp; }      // (fills the main stream's wait below)
    HIP_TRY(hipStreamWaitEvent(x.s, h->ev_go, 0));
    x.kuu_on_main = true;          // (built and factorised: fwd_chain_rest adds K^-1 / log|K| where a backward pass wants them)
    return FFVD_OK;
}

// Gram route: nothing of the K_fu build depends on K_uu, so the latency-bound K_uu chain runs on the side stream.  This is the fork
// and whatever the schedule wants on either stream BEFORE the chain's own launches (fwd_chain_rest): the full-batch schedule's
// dataflow chain and the work that fills the main stream's wait for it; the first pass's K_fu build / tile pass when the main
// stream is the critical one (kfu_first, main_first); the identity rows of a training pass on the side stream.
static int fwd_chain_fork(FwdCtx &x) {
    ffvd_handle *h = x.h;
    const ffvd_config &c = h->cfg;
    const ElboSchedule &sc = x.sc;
    const int Mp = h->Mp, Dl = h->Dl;
    hipStream_t s = x.s;
    x.sk = h->aux;
    hipStream_t sk = x.sk;
    x.kuu_on_main = sc.kuu_flow;
    if (x.kuu_on_main)
        launch_kuu_build(s, c.kernel_kind, x.hv, c.M, Mp, h->P, Dl, c.jitter, h->Kuu, h->Kcopy);
    { int rcf = fork_side(h, h->ev_fork, s, sk); if (rcf) return rcf; }
    if (!x.kuu_on_main && !sc.small_side) { int rcp = fwd_prior_sums(x, sk); if (rcp) return rcp; }      // (the side stream idles here; small side: its chain is the critical path)
    if (x.kuu_on_main) {
        HIP_TRY(hipEventRecord(h->ev_kuu, s));
        x.linv_done = potrf_flow_selected(Mp, Dl, CHOL_FLOW);      // L^-1 comes out of the factorisation itself
        if (x.linv_done) {
            // The chain's 64-odd row workgroups (76.8 KB of LDS, 256 VGPRs) must be on the chip BEFORE the K_fu build
            // floods it with small ones, or they wait for that kernel to drain: the main stream resumes one
            // cross-queue hop after the clear that sits directly in front of the chain's kernel
            potrf_flow_clear(sk, h->dinvK, Dl);
            HIP_TRY(hipEventRecord(h->ev_go, sk));
        }
        x.kinv_done = x.linv_done && !h->sw.kinv_gram && potrf_flow_forms_inverse(Mp, Dl, CHOL_FLOW);
        launch_potrf_ext(sk, h->Kuu, Mp, Mp, Mp, Dl, x.kstride, h->info, h->dinvK, CHOL_FLOW, h->Linv, x.msq, x.linv_done, false,
                         x.kinv_done ? h->Kinv : nullptr, x.msq);
        if (x.linv_done) {
            launch_chain_reduce(s, x.ra, h->chain_partial);      // inputs only; fills the wait below
            x.reduce_done = true;
            { int rcp = fwd_prior_sums(x, s); if (rcp) return rcp; }      // (likewise)
            if (c.grad && c.branch == FFVD_BRANCH_B && !sc.lt_rows) {
                // training: the identity rows that become L_A^-T are re-armed here too (rows the K_fu build and the Gram
                // kernel do not touch) instead of between the K_fu build and the Gram kernel
                const GramArgs gi = x.gram_args(0, sc.ns_first);
                launch_set_identity(s, h->H, gi.h_stride, Mp, Mp, sc.ns_first * Dl);
                x.ident_early = true;
            }
            HIP_TRY(hipStreamWaitEvent(s, h->ev_go, 0));
            if (c.S_local <= h->cpp) {          // single pass: the words of Cholesky(A) are cleared here, off the main stream
                potrf_flow_clear(sk, h->dinvH, h->nbatch);
                HIP_TRY(hipEventRecord(h->ev_hwords, sk));      // the factorisation that trusts this clear waits for THIS event
                x.hwords_zeroed = true;
            }
        }
    }
    if (sc.kfu_first) {         // the first pass's K_fu build goes to the main stream before the chain is enqueued
        if (x.st) x.st->mark(0);
        launch_kfu_build(s, x.project_args(0, sc.ns_first));
        x.brow_finish(0, sc.ns_first);
        if (x.st) x.st->mark(1);
    }
    if (sc.main_first) {        // ... and so does the split-K tile pass (it needs neither K_uu nor K^-1)
        const GramArgs ga = x.gram_args(0, c.S_local);
        launch_gram(s, ga, 1);
        HIP_TRY(hipEventRecord(h->ev_tiles, s));
    }
    // training: the identity rows that become L_A^-T are re-armed on the side stream, ahead of the K_uu build whose
    // event the main stream waits for anyway (0.05 ms off the critical path at the full batch)
    if (sc.ident_on_side) {
        const GramArgs ga = x.gram_args(0, sc.ns_first);
        launch_set_identity(sk, h->H, ga.h_stride, Mp, Mp, sc.ns_first * Dl);
    }
    return FFVD_OK;
}

// The K_uu chain's own launches on x.sk (unless the prelude already enqueued its factorisation): Cholesky with the L^-T rows, then --
// Gram route and backward passes -- K^-1 = L^-T L^-1 and log|K|, the join event, and the per-chain reductions where they ride behind
// the chain.
static int fwd_chain_rest(FwdCtx &x) {
    ffvd_handle *h = x.h;
    const ffvd_config &c = h->cfg;
    const ElboSchedule &sc = x.sc;
    const int Mp = h->Mp, Dl = h->Dl;
    hipStream_t s = x.s, sk = x.sk;
    const size_t msq = x.msq, kstride = x.kstride;
    const bool zt_rows = sc.zt_rows, needs_inverse = sc.gram_route || sc.grad_a || sc.grad_ref;
    if (!x.kuu_on_main) {
        // chain on the main stream = on the critical path with nothing beside it: the dataflow launch; on the side stream
        // (beside the K_fu build / tile pass of a small batch) the right-looking launches, whose workgroups come and go
        const bool chain_flow = sc.chain_flow_here;
        if (chain_flow && needs_inverse) x.linv_done = true;
        if (chain_flow && (sc.small_side || sc.side_late) && sc.gram_route && potrf_flow_forms_inverse(Mp, Dl, CHOL_FLOW)) x.kinv_done = true;
        if (sc.side_late) {
            // the reductions of the inputs first (they need nothing), then the chain once the tile pass has left the chip
            if (sc.reduce_early && !x.reduce_done) { launch_chain_reduce(sk, x.ra, h->chain_partial); x.reduce_launched = true; }
            HIP_TRY(hipStreamWaitEvent(sk, h->ev_tiles, 0));
        }
        launch_potrf_ext(sk, h->Kuu, Mp, zt_rows ? NB : Mp, zt_rows ? 0 : Mp, Dl, kstride, h->info, h->dinvK,
                         chain_flow ? CHOL_FLOW : CHOL_AUTO, x.linv_done ? h->Linv : nullptr, msq, chain_flow /* words zeroed by the build */,
                         false, x.kinv_done ? h->Kinv : nullptr, msq, nullptr, 0, 1, sc.small_side || sc.side_late);
    }
    if (needs_inverse) {
        // K^-1 = L^-T L^-1 (shared by all chains) and log|K|
        if (!x.linv_done) launch_transpose(sk, h->Kuu + msq, kstride, h->Linv, msq, Mp, Dl);
        GramArgs gk{};
        gk.mode = GRAM_PLAIN; gk.A = h->Linv; gk.a_stride = msq; gk.rows = Mp; gk.with_row = 0; gk.Mp = Mp; gk.Dl = Dl;
        gk.d_begin = c.d_begin; gk.b0 = 0; gk.nb = Dl; gk.yn_over_batch = 1.0; gk.H = h->Kinv; gk.h_stride = msq;
        if (x.kinv_done) {
            // (the identity-row workgroups of the chain's launch have formed it: kernels.hip, df_inverse_tiles)
        } else if (c.grad || (x.kuu_on_main && !h->sw.kinv_gram)) {
            // the backward pass reads K^-1 everywhere, the Gram kernel only writes lower tiles; and beside the K_fu build
            // (kuu_on_main: the main stream waits for this product) 256-thread workgroups find a slot where the Gram
            // kernel's 1024-thread ones wait for the build to drain (0.19 against 0.05 ms)
            AtbArgs ak{};
            ak.mode = ATB_PLAIN; ak.A = h->Linv; ak.a_stride = msq; ak.lda = Mp; ak.nA = Mp; ak.B = h->Linv; ak.b_stride = msq;
            ak.ldb = Mp; ak.nB = Mp; ak.rows = Mp; ak.C = h->Kinv; ak.c_stride = msq; ak.ldc = Mp; ak.nb = Dl; ak.Dl = Dl;
            ak.k_lower = 1;                                  // L^-1 is lower triangular
            ak.small_tiles = x.kuu_on_main ? 1 : 0;
            launch_atb(sk, ak);
        } else launch_gram(sk, gk);     // (split-K here was measured slower: 0.42 vs 0.37 ms for the K_uu stage)
        launch_h_finish(sk, h->Kuu, Mp, kstride, Dl, h->kterms);
        if (sk != s) HIP_TRY(hipEventRecord(h->ev_join, sk));
    } else if (sc.ref_side) HIP_TRY(hipEventRecord(h->ev_join, sk));
    // the per-chain likelihood / transition reductions depend on the inputs only (Gram route: no row sums of F), so
    // they ride on the side stream behind the K_uu chain and are back long before finalize needs them (kuu_on_main: they
    // already ran on the main stream while it waited for the chain's kernel to be dispatched)
    if (sc.reduce_early && !x.reduce_done && !sc.small_side /* reductions on the main stream */ && !x.reduce_launched) {
        launch_chain_reduce(sk, x.ra, h->chain_partial);
        HIP_TRY(hipEventRecord(h->ev_join2, sk));
    }
    return FFVD_OK;
}

// K(X_combine, Z) of one pass and what turns it into the pass's F / fmean / row sums: five forms, by route, branch and arithmetic.
static int fwd_pass_project(FwdCtx &x, int s0, int ns) {
    ffvd_handle *h = x.h;
    const ffvd_config &c = h->cfg;
    const ElboSchedule &sc = x.sc;
    const ffvd_params &p = h->cur;
    const int Mp = h->Mp, Tp = h->Tp, Dl = h->Dl;
    hipStream_t s = x.s, sk = x.sk;
    const size_t msq = x.msq, kstride = x.kstride;
    ProjectArgs pa = x.project_args(s0, ns);
    if (sc.gram_route) {
        if (!(sc.kfu_first && s0 == 0)) { launch_kfu_build(s, pa); x.brow_finish(s0, ns); }
        if (s0 == 0 && sk != s && !sc.late_join) HIP_TRY(hipStreamWaitEvent(s, sc.defer_full ? h->ev_kuu : h->ev_join, 0));
    } else if (c.dtype == FFVD_F32C) {
        if (s0 == 0) launch_linv_f32(s, h->Kuu + msq, kstride, h->Linv32, Mp, Dl);     // L^-1 as the fp32 B operand
        launch_kfu_build_f32(s, pa, h->Kf32);                     // K(X_combine, Z)           (:240)
        ProjF32Args pg{};
        pg.Kf = h->Kf32; pg.kf_stride = (size_t)Tp * Mp; pg.LinvT = h->Linv32; pg.F = h->F32; pg.f_stride = (size_t)Tp * Mp;
        pg.sqpart = h->sqpart; pg.Tp = Tp; pg.Mp = Mp; pg.Dl = Dl; pg.b0 = s0 * Dl; pg.nb = ns * Dl;
        launch_proj_gemm_f32(s, pg);                              // tilde_F = Knm Lm^-T, sum F^2  (:242,:255)
    } else if (h->ngr && c.branch == FFVD_BRANCH_B) {
        pa.F = h->Kf2;
        launch_kfu_build(s, pa);                                  // K(X_combine, Z)           (:240)
        ProjGemmArgs pg{};
        pg.Kf = h->Kf2; pg.kf_stride = (size_t)Tp * Mp; pg.W = h->Kuu + msq; pg.w_stride = kstride;
        pg.F = h->F; pg.f_stride = (size_t)Tp * Mp; pg.rowsq = h->rowsq; pg.Tp = Tp; pg.Mp = Mp; pg.Dl = Dl;
        pg.b0 = s0 * Dl; pg.nb = ns * Dl;
        pg.gpart = h->growpart; pg.X = p.X; pg.T = c.T; pg.D = c.D; pg.d_begin = c.d_begin;      // delta^T F by 128-row tiles (:247-248)
        if (sc.ref_side && s0 == 0) HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));               // W = L^-T from the side stream's chain
        launch_proj_gemm(s, pg);                                  // tilde_F = Knm Lm^-T, sum F^2  (:242,:255)
        if (h->growpart)
            launch_brow_finish(s, h->growpart, (int)((Tp + 127) / 128), Mp, Dl, c.d_begin, s0 * Dl, ns * Dl, p.log_Q, 1.0, h->H,
                               elbo_h_stride(h), c.grad ? 2 * Mp : Mp);
    } else if (h->lrpart) {
        // explicit-U branch, LinearK, forward only: fmean and sum_j F^2 through the kernel's rank P (no K_fu, no F)
        if (sk != s && s0 == 0) HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
        launch_linear_lowrank(s, pa, h->lrpart);
    } else if (c.branch == FFVD_BRANCH_A && h->ngr) {
        // explicit-U branch: K_fu once, then the triangular GEMM with fvar / fmean folded into its epilogue (F unstored)
        pa.F = h->F + (sc.grad_a ? (size_t)s0 * Dl * Tp * Mp : 0);
        launch_kfu_build(s, pa);
        if (sc.ref_side && s0 == 0) HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));               // W = L^-T from the side stream's chain
        if (s0 == 0) launch_ucols(s, p.U, c.M, Mp, c.D, c.d_begin, Dl, h->ucolA);
        ProjGemmArgs pg{};
        pg.Kf = pa.F; pg.kf_stride = (size_t)Tp * Mp; pg.W = h->Kuu + msq; pg.w_stride = kstride;
        pg.F = nullptr; pg.f_stride = 0; pg.rowsq = h->rowsq; pg.fmean = h->fmean; pg.u = h->ucolA; pg.u_stride = Mp;
        pg.Tp = Tp; pg.Mp = Mp; pg.Dl = Dl; pg.b0 = s0 * Dl; pg.nb = ns * Dl;
        launch_proj_gemm(s, pg);
    } else {
        if (sc.grad_a) {                    // (FFVD_FUSED_PROJECT) the backward pass still needs K_fu itself
            pa.F = h->F;
            launch_kfu_build(s, pa);
            pa.F = nullptr;
        }
        launch_project(s, pa);
    }
    if (x.st && !(sc.kfu_first && s0 == 0)) x.st->mark(1);
    DBG_SYNC(h, "forward: K_fu / projection");
    return FFVD_OK;
}

// H = F^T F / Q + I (projection route) or A = K_uu + K_uf K_fu / Q (Gram route) of one pass: the tile pass + combine of a split-K
// launch, the unsplit launch with raw tiles beside the chain, the fp32 product, or the plain unsplit launch.
static int fwd_pass_gram(FwdCtx &x, int s0, int ns, GramArgs &ga, bool &trace_pending, bool &acopy_done) {
    ffvd_handle *h = x.h;
    const ffvd_config &c = h->cfg;
    const ElboSchedule &sc = x.sc;
    const ffvd_params &p = h->cur;
    const int Mp = h->Mp, Tp = h->Tp, Dl = h->Dl;
    hipStream_t s = x.s, sk = x.sk;
    if (sc.lt_rows) {
        // (armed right in front of the factorisation, when at all: L comes from the K_uu chain)
    } else if (c.grad && !((sc.ident_on_side || x.ident_early) && s0 == 0)) launch_set_identity(s, h->H, ga.h_stride, Mp, Mp, ns * Dl);
    if (s0 == 0 && sc.late_join) {
        if (!sc.main_first) launch_gram(s, ga, 1);
        if (sc.defer_trace) {
            if (!sc.main_first) HIP_TRY(hipEventRecord(h->ev_tiles, s));
            HIP_TRY(hipStreamWaitEvent(s, h->ev_kuu, 0));
            ga.trace_mode = 1;
            // side late: the trace pass runs long after this one (behind the K_uu chain) -- it reads the summed raw tiles this
            // pass leaves in partial 0 instead of all the row ranges again (134 MB at 4 chains: 20-27 us of the iteration's tail)
            ga.raw_summed = (sc.side_late || sc.small_side) ? 1 : 0;      // (small side: the trace pass follows on this stream)
            launch_gram(s, ga, 2);
            if (sc.side_late) HIP_TRY(hipEventRecord(h->ev_tiles, s));      // (the trace pass waits for THIS record)
            trace_pending = true;      // enqueued behind the factorisation: the main stream is the critical one
        } else {
            HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
            launch_gram(s, ga, 2);
        }
    } else if (s0 == 0 && sc.defer_full && sk != s) {
        ga.mode = GRAM_KFU_RAW; ga.part = h->graw; ga.ksplit = 1;
        if (c.grad) { ga.Hcopy = h->gw.Acopy; ga.hcopy_stride = x.msq; acopy_done = true; }
        launch_gram(s, ga);
        HIP_TRY(hipEventRecord(h->ev_tiles, s));
        trace_pending = true;
    } else if (c.dtype == FFVD_F32C) {
        GramF32Args gf{};
        gf.F = h->F32; gf.f_stride = (size_t)Tp * Mp; gf.rows = Tp; gf.with_row = 1; gf.brow = c.grad ? 2 * Mp : Mp;
        gf.X = p.X; gf.log_Q = p.log_Q; gf.T = c.T; gf.D = c.D; gf.Mp = Mp; gf.Dl = Dl; gf.d_begin = c.d_begin;
        gf.b0 = s0 * Dl; gf.nb = ns * Dl; gf.yn_over_batch = 1.0; gf.H = h->H; gf.h_stride = ga.h_stride;
        gf.flush = h->gram_flush;
        launch_gram_f32(s, gf);                               // H = F^T F / Q + I, b = delta^T F / Q  (:246-248)
    } else {
        // training: the unsplit Gram kernel stores the symmetric copy of A itself (no copy + symmetrize launches: 0.25 ms)
        if (c.grad && sc.gram_route && !ga.part) { ga.Hcopy = h->gw.Acopy; ga.hcopy_stride = x.msq; acopy_done = true; }
        launch_gram(s, ga);
    }
    if (x.st) x.st->mark(2);
    DBG_SYNC(h, "forward: Gram");
    return FFVD_OK;
}

// Gram-route training without L^T rows (FFVD_GRAD_WHITEN_PRODUCTS): H = W^T A W (W = L^-T of K_uu) replaces A in the slab, b = W^T c
// replaces c: the factorisation, the explicit inverse and everything the backward pass derives from them then live in the whitened
// variables, where cond(H) is about 1e4 instead of the 1e7 of A
static int fwd_whiten_products(FwdCtx &x, int ns, const GramArgs &ga) {
    ffvd_handle *h = x.h;
    ffvd_handle::GradWs &g = h->gw;
    const int Mp = h->Mp, Dl = h->Dl, nbp = ns * Dl;
    const size_t msq = x.msq, kstride = x.kstride;
    hipStream_t s = x.s;
    const int small = h->sw.atb128 ? 0 : 1;
    if (x.sk != s) HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));       // W and L^-1 come from the K_uu chain
    launch_symmetrize(s, g.Acopy, Mp, nbp);                            // (the saved copy holds the lower triangle)
    AtbArgs t1{};
    t1.mode = ATB_PLAIN; t1.A = g.Acopy; t1.a_stride = msq; t1.lda = Mp; t1.nA = Mp;
    t1.B = h->Kuu + msq; t1.b_stride = kstride; t1.ldb = Mp; t1.nB = Mp; t1.b_per_dim = 1; t1.rows = Mp;
    t1.C = g.T1; t1.c_stride = msq; t1.ldc = Mp; t1.nb = nbp; t1.Dl = Dl; t1.krange = 8; t1.small_tiles = small;      // W upper triangular
    launch_atb(s, t1);                                                  // T1 = A W
    AtbArgs t2{};
    t2.mode = ATB_PLAIN; t2.A = h->Kuu + msq; t2.a_stride = kstride; t2.lda = Mp; t2.nA = Mp; t2.a_per_dim = 1;
    t2.B = g.T1; t2.b_stride = msq; t2.ldb = Mp; t2.nB = Mp; t2.rows = Mp;
    t2.C = h->H; t2.c_stride = ga.h_stride; t2.ldc = Mp; t2.nb = nbp; t2.Dl = Dl; t2.krange = 4; t2.sym = 1; t2.small_tiles = small;
    launch_atb(s, t2);                                                  // H = W^T T1 into rows [0, Mp)
    launch_matvec(s, h->Linv, msq, h->H + 2 * msq, ga.h_stride, Mp, g.bw, 1, Mp, Mp, nbp, Dl);   // b = W^T c
    HIP_TRY(hipMemcpy2DAsync(h->H + 2 * msq, ga.h_stride * sizeof(double), g.bw, (size_t)Mp * sizeof(double),
                             (size_t)Mp * sizeof(double), (size_t)nbp, hipMemcpyDeviceToDevice, s));
    return FFVD_OK;
}

// Cholesky of the pass's H / A slabs (one dataflow launch, the row b as a vector; training: with the L^T / identity rows), log|.| and
// the quadratic term; in front of it the trace pass on the side stream where the schedule defers it, behind it the trace pass /
// reductions of the small-side schedule.
static int fwd_pass_factor(FwdCtx &x, int s0, int ns, GramArgs &ga, bool &trace_pending, const bool acopy_done) {
    ffvd_handle *h = x.h;
    const ffvd_config &c = h->cfg;
    const ElboSchedule &sc = x.sc;
    const int Mp = h->Mp, Dl = h->Dl;
    hipStream_t s = x.s, sk = x.sk;
    const size_t msq = x.msq, kstride = x.kstride;
    int rc;
    if (trace_pending && sc.small_side) {
        // tiny iteration: the trace partials wait for the chain on the MAIN stream, behind Cholesky(A) -- one cross-stream
        // hop (chain -> here) instead of two (tile pass -> side stream, side stream -> finalize)
        x.trace_on_main = true;
    } else if (trace_pending) {
        if (x.chain_deferred && s0 == 0 && (rc = fwd_chain_rest(x)) != FFVD_OK) return rc;      // (waits for ev_tiles itself)
        // trace partials from the raw tiles, on the side stream (K^-1 precedes in its order).  Enqueued AHEAD of the
        // factorisation: that is one launch whose row workgroups hold every slot of the chip for most of its length,
        // and a kernel that arrives behind it only starts when they leave (finalize then waited 0.26 ms for this pass)
        HIP_TRY(hipStreamWaitEvent(sk, h->ev_tiles, 0));
        launch_gram(sk, ga, 3);
        HIP_TRY(hipEventRecord(h->ev_join2, sk));                 // supersedes the record after the reductions
        trace_pending = false;
    }
    const bool words = x.hwords_zeroed && s0 == 0;
    if (c.grad) {       // keep A = K_uu + K_uf K_fu / Q: the factorisation overwrites it in place
        if (!acopy_done)
            HIP_TRY(hipMemcpy2DAsync(h->gw.Acopy, msq * sizeof(double), h->H, ga.h_stride * sizeof(double),
                                     msq * sizeof(double), (size_t)ns * Dl, hipMemcpyDeviceToDevice, s));
        if (sc.lt_rows) {
            if (sk != s && s0 == 0) HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));       // L (and W, L^-1 for the backward pass) come from the K_uu chain
            if (!sc.lt_virtual) launch_set_lt_rows(s, h->Kuu, kstride, Dl, h->H, ga.h_stride, Mp, Mp, ns * Dl);
        } else if (h->gw.whitened && sc.gram_route) {
            if ((rc = fwd_whiten_products(x, ns, ga)) != FFVD_OK) return rc;
        }
        if (words) HIP_TRY(hipStreamWaitEvent(s, h->ev_hwords, 0));      // the side-stream clear of these words
        launch_potrf_ext(s, h->H, Mp, Mp + NB, Mp, ns * Dl, ga.h_stride, h->info + Dl + s0 * Dl, h->dinvH, CHOL_FLOW, nullptr, 0,
                         words, true, nullptr, 0, sc.lt_virtual ? h->Kuu : nullptr, kstride, Dl);
        launch_h_finish(s, h->H, Mp, ga.h_stride, ns * Dl, h->hterms + (size_t)2 * s0 * Dl, 2 * Mp);
    } else {
        if (words) HIP_TRY(hipStreamWaitEvent(s, h->ev_hwords, 0));      // the side-stream clear of these words
        launch_potrf_ext(s, h->H, Mp, NB, 0, ns * Dl, ga.h_stride, h->info + Dl + s0 * Dl, h->dinvH, CHOL_FLOW, nullptr, 0, words, true);
        launch_h_finish(s, h->H, Mp, ga.h_stride, ns * Dl, h->hterms + (size_t)2 * s0 * Dl);
    }
    if (sc.small_side && !x.reduce_launched) { launch_chain_reduce(s, x.ra, h->chain_partial); x.reduce_launched = true; }
    if (trace_pending && x.trace_on_main) {
        HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
        launch_gram(s, ga, 3);
        trace_pending = false;
    }
    if (x.st) x.st->mark(3);
    DBG_SYNC(h, "forward: Cholesky(H) + solves");
    return FFVD_OK;
}

// the per-chain reductions where nothing ran them early, the join with the side stream, the nll assembly
static int fwd_finalize(FwdCtx &x) {
    ffvd_handle *h = x.h;
    const ffvd_config &c = h->cfg;
    const ElboSchedule &sc = x.sc;
    hipStream_t s = x.s;
    if (!sc.reduce_early) launch_chain_reduce(s, x.ra, h->chain_partial);
    else if (!x.reduce_done && !x.trace_on_main) HIP_TRY(hipStreamWaitEvent(s, h->ev_join2, 0));
    FinalizeArgs fa = elbo_finalize_args(h, x.out_dev, sc.gram_route);
    if (x.prior_early) {
        if (x.prior_early == 2) HIP_TRY(hipStreamWaitEvent(s, h->ev_prior, 0));
        fa.prior_sums = h->prior_sums;
    }
    fa.whitened = (c.grad && h->gw.whitened && sc.gram_route && !sc.lt_rows) ? 1 : 0;      // L^T rows: the slab holds the factor of A itself
    if (c.dtype == FFVD_F32C) {           // sum_t |F_t|^2 per unit from the projection's fp64 tile sums
        launch_sum_partials(s, h->sqpart, h->nsq, h->nbatch, h->sqsum);
        fa.trpart = h->sqsum; fa.ntiles = 1; fa.fsq_from_trpart = 1;
    } else if (sc.grad_ref)               // the backward pass wants the same per-unit sum (dl/dalpha): all row sums of F^2 of a unit
        launch_sum_partials(s, h->rowsq, h->ngr * h->Tp, h->nbatch, h->gw.fsq);
    launch_finalize(s, fa);
    if (x.st) x.st->mark(4);
    DBG_SYNC(h, "forward: reductions + finalize");
    HIP_TRY(hipGetLastError());
    return FFVD_OK;
}

static int enqueue_elbo(ffvd_handle *h, double *out_dev, StageTimer *st) {
    if (tiny_selected(h)) return enqueue_tiny(h, out_dev, st, false, h->cfg.S_local);
    StageTimer live{h};
    if (!st && h->timing_on) st = &live;
    if (pipe_selected(h)) return enqueue_elbo_pipe(h, out_dev, st);
    const ffvd_config &c = h->cfg;
    const ffvd_params &p = h->cur;
    const int Mp = h->Mp, Dl = h->Dl;
    if (st) st->mark(-1);
    launch_prep_hypers(h->stream, c.kernel_kind, p.Z, c.M, Mp, h->P, Dl, c.d_begin, p.logvariance, p.loglengthscales,
                       h->variance, h->len, h->Zs, h->zz, h->info, Dl + h->nbatch);
    FwdCtx x{};
    x.h = h; x.out_dev = out_dev; x.st = st;
    x.sc = plan_schedule(h);
    if (!x.sc.name) return set_error(h, FFVD_EINVAL, "internal: inconsistent schedule flags (plan_schedule)");
    x.hv = HyperView{h->variance, h->len, h->Zs, h->zz};
    x.s = h->stream; x.sk = h->stream;
    x.msq = (size_t)Mp * Mp; x.kstride = 2 * x.msq;
    x.ra = elbo_reduce_args(h, x.sc.gram_route);
    const ElboSchedule &sc = x.sc;
    int rc;
    // ---- the K_uu chain: where it runs, what is enqueued around it ----
    if (sc.ref_side && (rc = fwd_chain_ref_side(x)) != FFVD_OK) return rc;
    if (sc.side_chain && (rc = fwd_chain_fork(x)) != FFVD_OK) return rc;
    if (!x.kuu_on_main) {
        launch_kuu_build(x.sk, c.kernel_kind, x.hv, c.M, Mp, h->P, Dl, c.jitter, h->Kuu,
                         (sc.gram_route || sc.grad_a || sc.grad_ref) ? h->Kcopy : nullptr, sc.zt_rows, sc.chain_flow_here ? h->dinvK : nullptr);
        if (sc.defer_trace || (sc.defer_full && x.sk != x.s)) HIP_TRY(hipEventRecord(h->ev_kuu, x.sk));
    }
    x.chain_deferred = sc.side_late && sc.defer_full;       // the chain is enqueued behind the first pass's Gram launch (fwd_pass_factor)
    if (!x.chain_deferred && (rc = fwd_chain_rest(x)) != FFVD_OK) return rc;
    DBG_SYNC(h, "forward: K_uu chain");
    if (st && !sc.kfu_first) st->mark(0);
    // ---- the passes over the chains: projection / K_fu, Gram, factorisation ----
    for (int s0 = 0; s0 < c.S_local; s0 += h->cpp) {
        const int ns = (s0 + h->cpp <= c.S_local) ? h->cpp : c.S_local - s0;
        if ((rc = fwd_pass_project(x, s0, ns)) != FFVD_OK) return rc;
        if (c.branch != FFVD_BRANCH_B) continue;
        GramArgs ga = x.gram_args(s0, ns);
        bool trace_pending = false, acopy_done = false;      // acopy_done: the Gram kernel stored the copy of A itself
        if ((rc = fwd_pass_gram(x, s0, ns, ga, trace_pending, acopy_done)) != FFVD_OK) return rc;
        if ((rc = fwd_pass_factor(x, s0, ns, ga, trace_pending, acopy_done)) != FFVD_OK) return rc;
    }
    return fwd_finalize(x);
}

static int ready(ffvd_handle *h, const char *who) {
    if (!h->have_data) return set_error(h, FFVD_EINVAL, std::string(who) + ": call ffvd_set_data first");
    if (!h->have_params) return set_error(h, FFVD_EINVAL, std::string(who) + ": no parameters bound (ffvd_set_params or pass params)");
    return FFVD_OK;
}

static int check_info(ffvd_handle *h) {
    const int Dl = h->Dl;
    h->stalled = false;
    for (int i = 0; i < Dl + h->nbatch; ++i) {
        if (h->h_info[i] < 0) {        // the dataflow Cholesky bounds every wait (kernels.hip, potrf_df_kernel); so does tiny.hip
            h->stalled = true;
            h->tiny_dirty = true;
            return set_error(h, FFVD_EDEVICE, "blocked Cholesky abandoned: a block row waited more than 1 s for the row above it");
        }
        if (h->h_info[i] != 0) {
            char msg[256];
            if (i < Dl)
                snprintf(msg, sizeof msg, "Cholesky of K_uu + jitter*I failed: latent dim %d, pivot %d is not positive",
                         h->cfg.d_begin + i, h->h_info[i] - 1);
            else
                snprintf(msg, sizeof msg, "Cholesky of H failed: chain %d, latent dim %d, pivot %d is not positive",
                         (i - Dl) / Dl, h->cfg.d_begin + (i - Dl) % Dl, h->h_info[i] - 1);
            return set_error(h, FFVD_ENOTPD, msg);
        }
    }
    return FFVD_OK;
}

// Stall recovery (VERDICT r2 W10 / ADVICE r2): the dataflow Cholesky's forward progress rests on in-order workgroup dispatch; its
// waits are bounded, and when one fires (info = -1) the whole launch is abandoned.  The iteration is a pure function of resident
// inputs, so it is simply enqueued AGAIN, once, in this process, with the left-looking launch-per-column Cholesky, which has no
// inter-workgroup waits (every launch re-zeroes what it needs; the dataflow launches of later iterations clear their own words).
// A second failure is reported as FFVD_EDEVICE.  The first recovery leaves a note for ffvd_last_error.  Collective entry points
// (ffvd_elbo_allreduce, ffvd_*_step_allreduce, T-shards) do NOT retry: the other ranks have already moved on.
// The override is thread-local state: it is reset on EVERY exit path of the scope that set it (an error return between
// set and reset left the launch-per-column variant selected for every later factorisation of the thread).
struct CholOverrideGuard {
    bool armed = false;
    void force_left() { potrf_override_variant(CHOL_FORCE_LEFT); armed = true; }
    ~CholOverrideGuard() { if (armed) potrf_override_variant(CHOL_FORCE_NONE); }
};

// A stall is remembered.  The one-launch iteration and the dataflow Cholesky need every workgroup resident at once; a
// co-tenant that keeps compute units busy makes EVERY call wait out its bounded spin (1 s) before the recovery runs -- FFVD_OK each
// time, an hour of waiting over the 4000 outer iterations of FFVD_Main.py.  So after a recovery the handle stays on the schedule without
// inter-workgroup waits (launch-per-column Cholesky, multi-kernel iteration) for `stall_hold` calls, then probes the fast path again;
// a probe that stalls doubles the hold (16, 32, ... 1024 calls), one that succeeds resets it.  ffvd_schedule_name reports the state.
static constexpr int STALL_HOLD_FIRST = 16, STALL_HOLD_MAX = 1024;

template <class Enqueue>
static int fetch_with_stall_recovery(ffvd_handle *h, Enqueue enqueue) {
    int rc;
    const bool held = h->stall_hold > 0;        // back-off: this call does not try the paths that stalled
    if (held) --h->stall_hold;
    for (int attempt = held ? 1 : 0;; ++attempt) {
        {
            CholOverrideGuard guard;
            if (attempt == 1) guard.force_left();
            const auto t0 = std::chrono::steady_clock::now();
            rc = enqueue();
            h->enq_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            ++h->enq_calls;
        }
        if (rc) return rc;
        hipError_t e1 = hipMemcpyAsync(h->h_res, h->resblk, h->res_bytes, hipMemcpyDeviceToHost, h->stream);      // terms, chain nll, info flags
        hipError_t e2 = (e1 == hipSuccess) ? hipStreamSynchronize(h->stream) : e1;
        if (e2 != hipSuccess) return set_error(h, FFVD_EDEVICE, std::string("result copy failed: ") + hipGetErrorString(e2));
        rc = check_info(h);
        if (rc == FFVD_OK && attempt == 1 && !held) {
            if (h->stall_recoveries++ == 0)
                h->warning = "warning: the one-launch (dataflow) Cholesky gave up on a bounded wait; the iteration was re-run with "
                             "the launch-per-column Cholesky and completed";
            h->err = h->warning;
            h->stall_hold = h->stall_hold_next;
            h->stall_hold_next = std::min(2 * h->stall_hold_next, STALL_HOLD_MAX);
        } else if (rc == FFVD_OK && attempt == 0 && h->stall_hold_next != STALL_HOLD_FIRST) {
            h->stall_hold_next = STALL_HOLD_FIRST;      // the fast path ran through again: forget the history
        }
        if (!(rc == FFVD_EDEVICE && h->stalled && attempt == 0)) return rc;
    }
}

extern "C" int ffvd_stall_recoveries(const ffvd_handle *h) { return h ? h->stall_recoveries : 0; }
extern "C" int ffvd_stall_hold(const ffvd_handle *h) { return h ? h->stall_hold : 0; }

// Debug (tools/sync_step.py): average host time of one iteration's enqueue in microseconds, -1 before the first call.
extern "C" double ffvd_debug_enqueue_us(const ffvd_handle *h) {
    return (h && h->enq_calls > 0) ? (double)h->enq_ns / (double)h->enq_calls * 1e-3 : -1.0;
}

extern "C" int ffvd_single_launch(const ffvd_handle *h) { return (h && h->tiny.ok) ? h->tiny.nw : 0; }
// Debug / tests: scratch bytes per lane of the one-launch kernel (0 when the shape has no one-launch plan); how often the argument
// block of the forward (which = 0) / forward + backward (1) launch travelled to the device.
extern "C" int64_t ffvd_debug_tiny_private_bytes(const ffvd_handle *h) { return h ? (int64_t)h->tiny_private_bytes : 0; }
extern "C" int ffvd_debug_tiny_uploads(const ffvd_handle *h, int which) {
    return (h && h->tiny_ring_made && (which == 0 || which == 1)) ? h->tiny_ring[which].uploads : 0;
}

// Debug: copy `count` doubles of the one-launch path's scratch block, starting at `offset`, to the host (tools only).
extern "C" int64_t ffvd_debug_tiny_scratch(ffvd_handle *h, int64_t offset, int64_t count, double *out) {
    if (!h || !h->tiny.ok) return -1;
    const ffvd_config &c = h->cfg;
    const int64_t total = (int64_t)tiny_scratch_doubles(h->tiny, c.T, h->P, c.M, c.S_local, h->Dl, c.D, c.Ydim, c.grad);
    if (!out) return total;
    if (offset < 0 || offset + count > total) return -1;
    hipStreamSynchronize(h->stream);
    return hipMemcpy(out, h->tiny_scratch + offset, (size_t)count * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess ? total : -1;
}

extern "C" int ffvd_elbo(ffvd_handle *h, const ffvd_params *p, uint32_t flags, double out_terms[8], double *out_nll) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_elbo: null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if (p) {
        if (flags & FFVD_PARAMS_ON_DEVICE) {
            if ((rc = check_params(h, p, "ffvd_elbo"))) return rc;
            h->cur = *p;
            if (!h->cur.U) h->cur.U = h->U;
            if (!h->cur.loglengthscales) h->cur.loglengthscales = h->loglen;
            h->have_params = true;
        } else if ((rc = ffvd_set_params(h, p, 0))) return rc;
    }
    if ((rc = ready(h, "ffvd_elbo"))) return rc;
    if ((rc = fetch_with_stall_recovery(h, [&] { return enqueue_elbo(h, nullptr, nullptr); }))) return rc;
    if (out_terms) memcpy(out_terms, h->h_out, 8 * sizeof(double));
    if (out_nll) *out_nll = h->h_out[FFVD_TERM_NLL] / (double)h->cfg.S_local;
    return FFVD_OK;
}

extern "C" int ffvd_elbo_async(ffvd_handle *h, double *out_terms_dev) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_elbo_async: null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if ((rc = ready(h, "ffvd_elbo_async"))) return rc;
    h->info_pending = true;
    return enqueue_elbo(h, out_terms_dev, nullptr);
}

extern "C" void *ffvd_get_stream(ffvd_handle *h) { return h ? (void *)h->stream : nullptr; }

extern "C" int ffvd_chain_nll(ffvd_handle *h, double *out) {
    if (!h || !out) return set_error(h, FFVD_EINVAL, "ffvd_chain_nll: null argument");
    memcpy(out, h->h_chain, (size_t)h->cfg.S_local * sizeof(double));
    return FFVD_OK;
}

extern "C" int ffvd_time_elbo(ffvd_handle *h, int iters, float *out_ms) {
    if (!h || !out_ms || iters < 1) return set_error(h, FFVD_EINVAL, "ffvd_time_elbo: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if ((rc = ready(h, "ffvd_time_elbo"))) return rc;
    struct EventPair {          // destroyed on every exit path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~EventPair() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); }
    } ev;
    HIP_TRY(hipEventCreate(&ev.e0));
    HIP_TRY(hipEventCreate(&ev.e1));
    HIP_TRY(hipEventRecord(ev.e0, h->stream));
    for (int i = 0; i < iters; ++i)
        if ((rc = enqueue_elbo(h, nullptr, nullptr))) return rc;
    HIP_TRY(hipEventRecord(ev.e1, h->stream));
    HIP_TRY(hipMemcpyAsync(h->h_info, h->info, (size_t)(h->Dl + h->nbatch) * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipEventElapsedTime(out_ms, ev.e0, ev.e1));
    return check_info(h);
}

static void drain_stage_times(ffvd_handle *h, double ms[8], int32_t n[8]) {
    for (int i = 0; i < 8; ++i) { ms[i] = 0.0; n[i] = 0; }
    for (size_t i = 1; i < h->ev_used; ++i) {
        const int st = h->ev_stage[i];
        if (st < 0 || st >= 8) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, h->ev_pool[i - 1], h->ev_pool[i]) == hipSuccess) { ms[st] += t; ++n[st]; }
    }
    h->ev_used = 0;
}

extern "C" int ffvd_profile_stages(ffvd_handle *h, float out_ms[8]) {
    if (!h || !out_ms) return set_error(h, FFVD_EINVAL, "ffvd_profile_stages: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if ((rc = ready(h, "ffvd_profile_stages"))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->ev_used = 0;
    StageTimer st{h};
    if ((rc = enqueue_elbo(h, nullptr, &st))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    double ms[8];
    int32_t n[8];
    drain_stage_times(h, ms, n);
    for (int i = 0; i < 8; ++i) out_ms[i] = (float)ms[i];
    return FFVD_OK;
}

extern "C" int ffvd_stage_timing(ffvd_handle *h, int enable) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_stage_timing: null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->timing_on = enable != 0;
    h->ev_used = 0;
    return FFVD_OK;
}

extern "C" int ffvd_stage_times(ffvd_handle *h, double out_ms[8], int32_t out_launches[8]) {
    if (!h || !out_ms || !out_launches) return set_error(h, FFVD_EINVAL, "ffvd_stage_times: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipStreamSynchronize(h->stream));
    drain_stage_times(h, out_ms, out_launches);
    return FFVD_OK;
}

// ---- backward pass (see grad.hip); runs on the handle's stream right after the forward kernels ----------
// Backward pass of the explicit-U branch (closed form: oracle/ffvd_grad_oracle.py nll_grad_explicit_u).  The T x M work
// reuses the collapsed branch's kernels: one Gram pass (G = K_uf K_fu and g_r = K_uf r per unit) and the fused E
// product with Gamma := alpha K^-1 / 2, delta := r, u := beta = L^-T u; everything else is M x M per latent dim.
static int enqueue_grad_a(ffvd_handle *h, int S_total) {
    const ffvd_config &c = h->cfg;
    ffvd_handle::GradWs &g = h->gw;
    const int Mp = h->Mp, Tp = h->Tp, Dl = h->Dl, P = h->P, nb = h->nbatch, S = c.S_local;
    const size_t msq = (size_t)Mp * Mp, kstride = 2 * msq, fstride = (size_t)Tp * Mp, gstride = (size_t)(Mp + NB) * Mp;
    hipStream_t s = h->stream;
    const ffvd_params &p = h->cur;
    const double *W = h->Kuu + msq;                                                     // L^-T rows, per dim stride kstride
    // beta = W u,  r = delta - mean,  dl/dalpha per unit
    launch_ucols(s, p.U, c.M, Mp, c.D, c.d_begin, Dl, g.ucol);
    launch_matvec(s, W, kstride, g.ucol, Mp, Mp, g.beta, 1, Mp, Mp, Dl);
    const int kind = c.kernel_kind;
    launch_resid_a(s, kind, p.X, h->ctrl, c.C, h->fmean, h->rowsq, h->variance, p.log_Q, c.T, Tp, c.D, Dl, c.d_begin,
                   h->ngr ? h->ngr : h->ng, nb, g.r, g.dalpha, g.xsq);
    // G = K_uf K_fu (lower tiles) and g_r = K_uf r (row Mp) per unit, then summed over the chains
    GramArgs gg{};
    gg.mode = GRAM_PLAIN; gg.A = h->F; gg.a_stride = fstride; gg.rows = Tp; gg.with_row = 1; gg.brow = Mp; gg.rvec = g.r;
    gg.X = p.X; gg.log_Q = p.log_Q; gg.T = c.T; gg.D = c.D; gg.Mp = Mp; gg.Dl = Dl; gg.d_begin = c.d_begin; gg.b0 = 0; gg.nb = nb;
    gg.yn_over_batch = 1.0; gg.H = g.Gu; gg.h_stride = gstride;
    launch_gram(s, gg);
    launch_chain_sum(s, g.Gu, gstride, S, Dl, (size_t)(Mp + 1) * Mp, g.Gsum, gstride);
    // M x M chain per dim:  dW = alpha (g_r u^T + G W);  P = W dW^T W;  dL = -tril(P);  Phi = sym(tril(L^T dL), diag/2);
    // dK = W Phi W^T;  E_psi = dK o K_uu.  Temporaries: Asum (G sym), Gs (T1), gsum (dW), P1, KGK, GamSum
    HIP_TRY(hipMemcpy2DAsync(g.Asum, msq * sizeof(double), g.Gsum, gstride * sizeof(double), msq * sizeof(double), Dl,
                             hipMemcpyDeviceToDevice, s));
    launch_symmetrize(s, g.Asum, Mp, Dl);
    AtbArgs ap{};
    ap.mode = ATB_PLAIN; ap.lda = Mp; ap.nA = Mp; ap.ldb = Mp; ap.nB = Mp; ap.rows = Mp; ap.ldc = Mp; ap.nb = Dl; ap.Dl = Dl;
    ap.c_stride = msq;
    ap.A = g.Asum; ap.a_stride = msq; ap.B = W; ap.b_stride = kstride; ap.C = g.Gs;
    launch_atb(s, ap);                                                                  // T1 = G W
    launch_dw_a(s, g.Gs, g.Gsum + msq, gstride, g.ucol, p.log_Q, Mp, Dl, c.d_begin, g.gsum);   // dW
    ap.A = g.gsum; ap.a_stride = msq; ap.B = W; ap.b_stride = kstride; ap.C = g.P1;
    launch_atb(s, ap);                                                                  // Q1 = dW^T W
    ap.A = h->Linv; ap.a_stride = msq; ap.B = g.P1; ap.b_stride = msq; ap.C = g.KGK;
    launch_atb(s, ap);                                                                  // P = W Q1
    launch_tril_neg(s, g.KGK, Mp, Dl, g.GamSum);                                        // dL
    launch_tril_copy(s, h->Kuu, kstride, Mp, Dl, g.Lclean);
    ap.A = g.Lclean; ap.a_stride = msq; ap.B = g.GamSum; ap.b_stride = msq; ap.C = g.P1;
    launch_atb(s, ap);                                                                  // S = L^T dL
    launch_phi(s, g.P1, Mp, Dl, g.KGK);                                                 // Phi
    ap.A = g.KGK; ap.a_stride = msq; ap.B = h->Linv; ap.b_stride = msq; ap.C = g.P1;
    launch_atb(s, ap);                                                                  // Q2 = Phi W^T
    ap.A = h->Linv; ap.a_stride = msq; ap.B = g.P1; ap.b_stride = msq; ap.C = g.KGK;
    launch_atb(s, ap);                                                                  // dK = W Q2
    launch_epsi_a(s, kind, g.KGK, h->Kcopy, c.M, Mp, Dl, c.jitter, g.Epsi);
    EReduceArgs ek{};
    ek.kind = kind; ek.variance = h->variance;
    ek.E = g.Epsi; ek.e_stride = msq; ek.Kf = nullptr; ek.u = nullptr; ek.x_is_z = 1; ek.Z = p.Z; ek.len = h->len;
    ek.T = c.M; ek.Tp = Mp; ek.M = c.M; ek.Mp = Mp; ek.P = P; ek.Dl = Dl; ek.b0 = 0; ek.nb = Dl; ek.nblk = Mp / 64;
    ek.rsum = g.rsum2; ek.ez = g.ez2; ek.kfu = nullptr; ek.cs_part = g.cs2; ek.etx_part = g.etx2; ek.rx2_part = g.rx22;
    launch_e_reduce(s, ek);
    launch_e_finish(s, ek, g.dz_kuu, g.dll_kuu, g.dls_kuu);
    // du = W^T g_r (per dim) for dU
    launch_matvec(s, h->Linv, msq, g.Gsum + msq, gstride, Mp, g.du, 1, Mp, Mp, Dl);
    // K_fu side: E = (alpha K_fu K^-1 + alpha r beta^T) o K_fu, reduced in the fused kernel
    launch_scale_kinv(s, h->Kinv, p.log_Q, Mp, Dl, c.d_begin, g.GammaA);
    EReduceArgs er{};
    er.kind = kind; er.variance = h->variance; er.u_per_dim = 1;
    er.E = g.E; er.e_stride = fstride; er.Kf = h->F; er.u = g.beta; er.u_stride = Mp; er.x_is_z = 0;
    er.x = p.X; er.x_chain_stride = (size_t)(c.T + 1) * c.D; er.x_ld = c.D; er.x_cols = c.D; er.ctrl = h->ctrl; er.C = c.C;
    er.Z = p.Z; er.len = h->len; er.T = c.T; er.Tp = Tp; er.M = c.M; er.Mp = Mp; er.P = P; er.Dl = Dl; er.b0 = 0; er.nb = nb;
    er.nblk = Tp / 64; er.rsum = g.rsum; er.ez = g.ez; er.kfu = g.kfu; er.cs_part = g.cs_part; er.etx_part = g.etx_part;
    er.rx2_part = g.rx2_part;
    BwdFusedArgs bf{};
    bf.Kf = h->F; bf.kf_stride = fstride; bf.Gamma = g.GammaA; bf.g_stride = msq; bf.u = g.beta; bf.u_stride = Mp;
    bf.per_dim = 1; bf.rvec = g.r;
    bf.X = p.X; bf.ctrl = h->ctrl; bf.Z = p.Z; bf.log_Q = p.log_Q; bf.T = c.T; bf.Tp = Tp; bf.D = c.D; bf.C = c.C;
    bf.M = c.M; bf.Mp = Mp; bf.P = P; bf.Dl = Dl; bf.d_begin = c.d_begin; bf.b0 = 0; bf.nb = nb; bf.rp = g.rp;
    bf.cs_part = g.cs_part; bf.etx_part = g.etx_part; bf.rsum = g.rsum; bf.ez = g.ez; bf.kfu = g.kfu; bf.rx2_part = g.rx2_part;
    bf.linear = kind != FFVD_KERNEL_SE;
    if (g.rp) launch_bwd_fused(s, bf);
    else {
        // P > 6 (BASELINE config 5: P = 17): materialise E = alpha (K_fu K^-1 + r beta^T) [o K_fu for the SE kernel] and
        // reduce it in a second kernel
        AtbArgs ae{};
        ae.mode = ATB_BWD_E; ae.A = h->F; ae.a_stride = fstride; ae.lda = Mp; ae.nA = Tp; ae.a_rowmajor = 1;
        ae.B = g.GammaA; ae.b_stride = msq; ae.ldb = Mp; ae.nB = Mp; ae.b_per_dim = 1; ae.rows = Mp;
        ae.C = g.E; ae.c_stride = fstride; ae.ldc = Mp; ae.nb = nb; ae.b0 = 0; ae.Dl = Dl; ae.d_begin = c.d_begin;
        ae.log_Q = p.log_Q; ae.u = g.beta; ae.u_stride = Mp; ae.u_per_dim = 1; ae.X = p.X; ae.T = c.T; ae.D = c.D;
        ae.Kf = h->F; ae.kf_stride = fstride; ae.ldkf = Mp; ae.rvec = g.r; ae.no_hadamard = kind != FFVD_KERNEL_SE;
        launch_atb(s, ae);
        launch_e_reduce(s, er);
    }
    launch_e_finish(s, er, g.dz_unit, g.dll_unit, g.dls_unit);
    DxArgs dx{};
    dx.kind = kind; dx.variance = h->variance;
    dx.X = p.X; dx.Y = h->Y; dx.CC = p.CC; dx.DD = p.DD; dx.log_Rchols = p.log_Rchols; dx.log_Q = p.log_Q; dx.len = h->len;
    dx.rsum = g.rsum; dx.ez = g.ez; dx.kfu = g.kfu; dx.S = S; dx.S_total = S_total; dx.T = c.T; dx.Tp = Tp; dx.D = c.D;
    dx.P = P; dx.Ydim = c.Ydim; dx.Dl = Dl; dx.d_begin = c.d_begin; dx.shared_terms = c.shared_terms; dx.dX = g.dX;
    dx.T_norm = c.T_total; dx.skip_x0 = (c.T_total > 0 && c.t_begin > 0) ? 1 : 0;      // T-shards: the job's 1 / T, x_0 on the first shard
    launch_dx(s, dx);
    launch_shared_partials(s, dx, g.shared_part, g.sp_stride);
    GradFinalArgs gf{};
    gf.T = c.T; gf.D = c.D; gf.P = P; gf.M = c.M; gf.Mp = Mp; gf.Ydim = c.Ydim; gf.Dl = Dl; gf.d_begin = c.d_begin; gf.S = S;
    gf.S_total = S_total; gf.shared_terms = c.shared_terms; gf.prior_type = c.prior_type;
    gf.Z = p.Z; gf.logvar = p.logvariance; gf.loglen = p.loglengthscales; gf.log_Q = p.log_Q; gf.CC = p.CC; gf.DD = p.DD;
    gf.log_Rchols = p.log_Rchols; gf.dz_unit = g.dz_unit; gf.dll_unit = g.dll_unit; gf.dls_unit = g.dls_unit;
    gf.dz_kuu = g.dz_kuu; gf.dll_kuu = g.dll_kuu; gf.dls_kuu = g.dls_kuu; gf.shared_part = g.shared_part;
    gf.sp_stride = g.sp_stride; gf.dZ = g.dZ; gf.dlogvar = g.dlogvar; gf.dloglen = g.dloglen; gf.dlogQ = g.dlogQ;
    gf.dCC = g.dCC; gf.dDD = g.dDD; gf.dlogR = g.dlogR;
    gf.branch_a = 1; gf.dalpha_unit = g.dalpha; gf.du_dim = g.du; gf.U = p.U; gf.dU = g.dU;
    gf.kind = kind; gf.xsq_unit = g.xsq;
    launch_fill(s, g.dlogvar, g.small_count, 0.0);        // dlogvar | dloglen | dlogQ | dCC | dDD | dlogR
    launch_grad_finalize(s, gf);
    HIP_TRY(hipGetLastError());
    return FFVD_OK;
}

static int enqueue_grad_b(ffvd_handle *h, int S_total) {
    const ffvd_config &c = h->cfg;
    ffvd_handle::GradWs &g = h->gw;
    const int Mp = h->Mp, Tp = h->Tp, Dl = h->Dl, P = h->P, nb = h->nbatch, S = c.S_local;
    const size_t msq = (size_t)Mp * Mp, hstride = (size_t)(2 * Mp + NB) * Mp, fstride = (size_t)Tp * Mp;
    hipStream_t s = h->stream;
    const ffvd_params &p = h->cur;
    const size_t kstride = (size_t)2 * Mp * Mp;
    const bool wh = g.whitened;
    // Reference route (F = K_fu L^-T, H = F^T F / Q + I factorised by the forward pass; fp64 or fp32 contractions): the slab already
    // holds the factor of the whitened H, so the M x M side below is the whitened one as it stands; what differs is where K_fu
    // lives (Kf2, or the fp32 copy), that sum_s H_s is read instead of W^T (sum_s A_s) W, and where sum_t |F_t|^2 comes from.
    const bool ref = c.route == FFVD_ROUTE_REFERENCE;
    const bool f32c = c.dtype == FFVD_F32C;
    const double *Kf64 = ref ? h->Kf2 : h->F;
    // explicit form: u = A^-1 c = L_A^-T (L_A^-1 c), Gamma = alpha/2 (K^-1 - A^-1 - u u^T) with A^-1 from the factor of A.
    // whitened form (default): the slab holds the factor of H = W^T A W and y = L_H^-1 b.  Then w = H^-1 b, u = W w, and
    // A^-1 = B^T B with B = L_H^-1 L^-1 (a product of two accurate triangular factors; K^-1 = (L^-1)^T L^-1 is formed
    // the same way in the forward pass), so the same Gamma launch runs on B instead of on the inverse factor of the
    // ill-conditioned A -- dZ at M = 512 then agrees with central differences to 7 digits instead of 3.
    AtbArgs ag{};
    ag.mode = ATB_GAMMA; ag.a_stride = msq; ag.lda = Mp; ag.nA = Mp;
    ag.b_stride = msq; ag.ldb = Mp; ag.nB = Mp; ag.b_per_dim = 0; ag.rows = Mp;
    ag.C = g.Gamma; ag.c_stride = msq; ag.ldc = Mp; ag.nb = nb; ag.b0 = 0; ag.Dl = Dl; ag.d_begin = c.d_begin;
    ag.log_Q = p.log_Q; ag.u = g.u; ag.u_stride = Mp; ag.Kinv = h->Kinv; ag.Kcopy = h->Kcopy; ag.k_stride = msq; ag.ldk = Mp;
    ag.part = g.gam_part; ag.k_lower = 1; ag.sym = 1; ag.small_tiles = 1;   // inverse factor lower triangular, its Gram symmetric
    // K_uu side, first half: K^-1 (sum_s A_s - S K) K^-1 needs the saved A-matrices and the K_uu chain only, not Gamma.  When the E
    // product is short (tiny problems: the side stream's launches are the critical path of the backward pass) it starts here,
    // ahead of w / u / B / Gamma; Gamma's sum joins it through ev_go.
    hipStream_t sk = h->sw.grad_serial ? s : h->aux;
    const bool tiny = (size_t)nb * Tp * Mp <= (size_t)64 * 1024 * 128 && sk != s;
    auto kgk_chain = [&]() -> int {
        launch_chain_sum(sk, g.Acopy, msq, S, Dl, msq, g.Asum, msq);      // (reference route: the saved matrices are the H_s)
        launch_symmetrize(sk, g.Asum, Mp, Dl);
        if (!ref) launch_axpby(sk, g.Asum, h->Kcopy, 1.0, -(double)S, p.log_Q, c.d_begin, 0, msq, Dl, g.Gs);
        AtbArgs ap{};
        ap.mode = ATB_PLAIN; ap.a_stride = msq; ap.lda = Mp; ap.nA = Mp; ap.b_stride = msq;
        ap.ldb = Mp; ap.nB = Mp; ap.rows = Mp; ap.c_stride = msq; ap.ldc = Mp; ap.nb = Dl; ap.Dl = Dl;
        if (wh) {       // K^-1 Gs K^-1 = W (W^T Gs W) W^T, conjugated step by step (Gs and W^T Gs W are symmetric)
            if (ref) launch_sub_identity(sk, g.Asum, (double)S, Mp, Dl, g.P2);     // W^T Gs W = sum_s (H_s - I): no products needed
            else {
                ap.A = g.Gs; ap.B = h->Kuu + msq; ap.b_stride = kstride; ap.C = g.P1; ap.krange = 8;
                launch_atb(sk, ap);                             // P1 = Gs W
                ap.A = h->Kuu + msq; ap.a_stride = kstride; ap.B = g.P1; ap.b_stride = msq; ap.C = g.P2; ap.krange = 4;
                launch_atb(sk, ap);                             // P2 = W^T Gs W
            }
            ap.a_stride = msq; ap.b_stride = msq;
            ap.A = g.P2; ap.a_stride = msq; ap.B = h->Linv; ap.C = g.P3; ap.krange = 2;
            launch_atb(sk, ap);                             // P3 = P2 W^T
            ap.A = h->Linv; ap.B = g.P3; ap.C = g.KGK; ap.krange = 1;
            launch_atb(sk, ap);                             // KGK = W P3
        } else {
            ap.A = g.Gs; ap.B = h->Kinv; ap.C = g.P1;
            launch_atb(sk, ap);                             // P1 = Gs^T K^-1 = Gs K^-1
            ap.A = g.P1; ap.C = g.KGK;
            launch_atb(sk, ap);                             // P1^T K^-1 = K^-1 Gs K^-1
        }
        return FFVD_OK;
    };
    if (tiny) {
        { int rcf = fork_side(h, h->ev_fork, s, sk); if (rcf) return rcf; }
        kgk_chain();
    }
    if (wh) {
        launch_matvec(s, h->H + msq, hstride, h->H + 2 * msq, hstride, Mp, g.wv, 1, Mp, Mp, nb);          // w = L_H^-T y
        launch_matvec(s, h->Kuu + msq, kstride, g.wv, Mp, Mp, g.u, 1, Mp, Mp, nb, Dl);                     // u = W w
        AtbArgs tb{};       // B[i][j] = sum_k L_H^-T[k][i] L^-1[k][j]: the extension rows as they are, no transpose
        tb.mode = ATB_PLAIN; tb.A = h->H + msq; tb.a_stride = hstride; tb.lda = Mp; tb.nA = Mp;
        tb.B = h->Linv; tb.b_stride = msq; tb.ldb = Mp; tb.nB = Mp; tb.b_per_dim = 1; tb.rows = Mp;
        tb.C = g.T1; tb.c_stride = msq; tb.ldc = Mp; tb.nb = nb; tb.Dl = Dl; tb.krange = 2 | 4; tb.small_tiles = h->sw.atb128 ? 0 : 1;      // k in [tile tj, tile (ti + 1))
        launch_atb(s, tb);
        ag.A = g.T1; ag.B = g.T1;
    } else {
        launch_matvec(s, h->H + msq, hstride, h->H + 2 * msq, hstride, Mp, g.u, 1, Mp, Mp, nb);
        launch_transpose(s, h->H + msq, hstride, g.LAinv, msq, Mp, nb);
        ag.A = g.LAinv; ag.B = g.LAinv;
    }
    launch_atb(s, ag);
    DBG_SYNC(h, "backward: w, u, B, Gamma");
    // K_uu side: Psi_d = sum_s Gamma_s / alpha_d - 1/2 K^-1 (sum_s A_s - S K) K^-1.  It needs Gamma and the saved
    // A-matrices only, so its dozen small launches go to the side stream and run beside the E product
    // (enqueued after it: the main stream must not wait for their launch overhead).
    if (sk != s) { int rcf = fork_side(h, tiny ? h->ev_go : h->ev_fork, s, sk); if (rcf) return rcf; }          // Gamma is there
    EReduceArgs er{};
    er.E = g.E; er.e_stride = fstride; er.Kf = Kf64; er.u = g.u; er.u_stride = Mp; er.x_is_z = 0;
    er.x = p.X; er.x_chain_stride = (size_t)(c.T + 1) * c.D; er.x_ld = c.D; er.x_cols = c.D; er.ctrl = h->ctrl; er.C = c.C;
    er.Z = p.Z; er.len = h->len; er.T = c.T; er.Tp = Tp; er.M = c.M; er.Mp = Mp; er.P = P; er.Dl = Dl; er.b0 = 0; er.nb = nb;
    er.nblk = Tp / 64; er.rsum = g.rsum; er.ez = g.ez; er.kfu = g.kfu; er.cs_part = g.cs_part; er.etx_part = g.etx_part;
    er.rx2_part = g.rx2_part;
    // LinearK (kernels.py:270-281) in the collapsed branch: K = (x s2) z^T has no Hadamard factor in its chain rule and its Kdiag_t =
    // s2 |x_t|^2 depends on the inputs -- the switches the explicit-U branch already uses (enqueue_grad_a), plus sum_t |x_t|^2 per unit
    const int kind = c.kernel_kind;
    const bool lin = kind != FFVD_KERNEL_SE;
    er.kind = kind; er.variance = h->variance;
    if (lin) launch_xsq_unit(sk, p.X, h->ctrl, c.T, c.D, c.C, S, Dl, g.xsq);
    if (f32c) {
        // fp32 contractions (BASELINE configs[3]): Gamma rounded once, R = K_fu Gamma on v_mfma_f32_32x32x2_f32 into the buffer F
        // occupied in the forward pass, then E_tm = (2 R_tm + alpha delta_t u_m) K_tm formed on the fly inside the reduction
        // kernel with every sum in fp64 (E itself is never stored)
        launch_to_f32(s, g.Gamma, g.Gam32, (size_t)nb * msq);
        ProjF32Args pg{};
        pg.Kf = h->Kf32; pg.kf_stride = fstride; pg.Bunit = g.Gam32; pg.bunit_stride = msq; pg.F = h->F32; pg.f_stride = fstride;
        pg.sqpart = nullptr; pg.Tp = Tp; pg.Mp = Mp; pg.Dl = Dl; pg.b0 = 0; pg.nb = nb;
        launch_proj_gemm_f32(s, pg);
        er.E = nullptr; er.Kf = nullptr; er.R32 = h->F32; er.Kf32 = h->Kf32; er.Xd = p.X; er.log_Q = p.log_Q; er.D = c.D;
        er.d_begin = c.d_begin;
        launch_e_reduce(s, er);
    } else if (g.rp) {
        // E = (2 Kf Gamma + alpha delta u^T) o Kf formed and reduced tile by tile: it never reaches HBM
        BwdFusedArgs bf{};
        bf.Kf = Kf64; bf.kf_stride = fstride; bf.Gamma = g.Gamma; bf.g_stride = msq; bf.u = g.u; bf.u_stride = Mp;
        bf.X = p.X; bf.ctrl = h->ctrl; bf.Z = p.Z; bf.log_Q = p.log_Q; bf.T = c.T; bf.Tp = Tp; bf.D = c.D; bf.C = c.C;
        bf.M = c.M; bf.Mp = Mp; bf.P = P; bf.Dl = Dl; bf.d_begin = c.d_begin; bf.b0 = 0; bf.nb = nb; bf.rp = g.rp;
        bf.cs_part = g.cs_part; bf.etx_part = g.etx_part; bf.rsum = g.rsum; bf.ez = g.ez; bf.kfu = g.kfu;
        bf.rx2_part = g.rx2_part;
        bf.linear = lin ? 1 : 0;
        launch_bwd_fused(s, bf);
    } else {
        // P > 6: materialise E and reduce it in a second kernel
        AtbArgs ae{};
        ae.mode = ATB_BWD_E; ae.A = Kf64; ae.a_stride = fstride; ae.lda = Mp; ae.nA = Tp; ae.a_rowmajor = 1;   // K_fu itself
        ae.B = g.Gamma; ae.b_stride = msq; ae.ldb = Mp; ae.nB = Mp; ae.rows = Mp;
        ae.C = g.E; ae.c_stride = fstride; ae.ldc = Mp; ae.nb = nb; ae.b0 = 0; ae.Dl = Dl; ae.d_begin = c.d_begin;
        ae.log_Q = p.log_Q; ae.u = g.u; ae.u_stride = Mp; ae.X = p.X; ae.T = c.T; ae.D = c.D;
        ae.Kf = Kf64; ae.kf_stride = fstride; ae.ldkf = Mp; ae.no_hadamard = lin ? 1 : 0;
        launch_atb(s, ae);
        launch_e_reduce(s, er);
    }
    DBG_SYNC(h, "backward: E product + reductions");
    launch_e_finish(s, er, g.dz_unit, g.dll_unit, g.dls_unit);
    DBG_SYNC(h, "backward: e_finish");
    // latent trajectories (after the E product) and the per-chain partials of the shared parameters (inputs only: side)
    DxArgs dx{};
    dx.kind = kind; dx.variance = h->variance;
    dx.X = p.X; dx.Y = h->Y; dx.CC = p.CC; dx.DD = p.DD; dx.log_Rchols = p.log_Rchols; dx.log_Q = p.log_Q; dx.len = h->len;
    dx.rsum = g.rsum; dx.ez = g.ez; dx.kfu = g.kfu; dx.S = S; dx.S_total = S_total; dx.T = c.T; dx.Tp = Tp; dx.D = c.D;
    dx.P = P; dx.Ydim = c.Ydim; dx.Dl = Dl; dx.d_begin = c.d_begin; dx.shared_terms = c.shared_terms; dx.dX = g.dX;
    dx.T_norm = c.T_total; dx.skip_x0 = (c.T_total > 0 && c.t_begin > 0) ? 1 : 0;      // T-shards: the job's 1 / T, x_0 on the first shard
    // u^T K u and the per-chain partials of the shared parameters feed grad_finalize only.  Beside a long E product they ride on
    // the side stream; when that product is a few dozen microseconds (the reference's own experiment sizes) the side stream's
    // dozen launches ARE the backward pass's critical path and these two go to the main stream, which has the slack there
    hipStream_t su = ((size_t)nb * Tp * Mp <= (size_t)64 * 1024 * 128 && !h->sw.grad_serial) ? s : sk;
    if (wh) launch_utu(su, g.wv, Mp, Mp, nb, g.uku);                       // u^T K u = w^T w
    else launch_uku(su, g.u, Mp, h->Kcopy, msq, Mp, Dl, nb, g.uku);   // u^T K u per unit: only grad_finalize reads it
    launch_shared_partials(su, dx, g.shared_part, g.sp_stride);
    if (!tiny) kgk_chain();
    launch_chain_sum(sk, g.Gamma, msq, S, Dl, msq, g.GamSum, msq);
    launch_axpby(sk, g.GamSum, nullptr, 1.0, 0.0, p.log_Q, c.d_begin, 1, msq, Dl, g.gsum);
    launch_psi_e(sk, g.gsum, g.KGK, h->Kcopy, c.M, Mp, Dl, c.jitter, g.Epsi, kind);
    EReduceArgs ek{};
    ek.E = g.Epsi; ek.e_stride = msq; ek.Kf = nullptr; ek.u = nullptr; ek.x_is_z = 1; ek.Z = p.Z; ek.len = h->len;
    ek.T = c.M; ek.Tp = Mp; ek.M = c.M; ek.Mp = Mp; ek.P = P; ek.Dl = Dl; ek.b0 = 0; ek.nb = Dl; ek.nblk = Mp / 64;
    ek.rsum = g.rsum2; ek.ez = g.ez2; ek.kfu = nullptr; ek.cs_part = g.cs2; ek.etx_part = g.etx2; ek.rx2_part = g.rx22;
    ek.kind = kind; ek.variance = h->variance;
    launch_e_reduce(sk, ek);
    launch_e_finish(sk, ek, g.dz_kuu, g.dll_kuu, g.dls_kuu);
    if (sk != s) HIP_TRY(hipEventRecord(h->ev_join, sk));
    if (sk != s) HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
    DBG_SYNC(h, "backward: K_uu side");
    launch_dx(s, dx);
    GradFinalArgs gf{};
    gf.T = c.T; gf.D = c.D; gf.P = P; gf.M = c.M; gf.Mp = Mp; gf.Ydim = c.Ydim; gf.Dl = Dl; gf.d_begin = c.d_begin; gf.S = S;
    gf.S_total = S_total; gf.shared_terms = c.shared_terms; gf.prior_type = c.prior_type;
    gf.T_norm = c.T_total; gf.replicated_skip = (c.T_total > 0 && c.t_begin > 0) ? 1 : 0;
    gf.Z = p.Z; gf.logvar = p.logvariance; gf.loglen = p.loglengthscales; gf.log_Q = p.log_Q; gf.CC = p.CC; gf.DD = p.DD;
    gf.log_Rchols = p.log_Rchols; gf.dz_unit = g.dz_unit; gf.dll_unit = g.dll_unit; gf.dls_unit = g.dls_unit;
    gf.dz_kuu = g.dz_kuu; gf.dll_kuu = g.dll_kuu; gf.dls_kuu = g.dls_kuu; gf.gam_part = g.gam_part; gf.ngam = g.ngam;
    gf.trpart = h->trpart; gf.ntr = h->ntiles; gf.hterms = h->hterms;
    gf.uku = g.uku; gf.shared_part = g.shared_part;
    gf.kind = kind; gf.xsq_unit = g.xsq;
    if (ref) { gf.trpart = f32c ? h->sqsum : g.fsq; gf.ntr = 1; }       // sum_t |F_t|^2 per unit (Gram route: tr(K^-1 K_uf K_fu) by tiles)
    gf.sp_stride = g.sp_stride; gf.dZ = g.dZ; gf.dlogvar = g.dlogvar; gf.dloglen = g.dloglen; gf.dlogQ = g.dlogQ;
    gf.dCC = g.dCC; gf.dDD = g.dDD; gf.dlogR = g.dlogR;
    // entries this handle does not own (other ranks' dims; the shared terms off rank 0) stay zero for the all-reduce
    launch_fill(s, g.dlogvar, g.small_count, 0.0);        // dlogvar | dloglen | dlogQ | dCC | dDD | dlogR
    launch_grad_finalize(s, gf);
    DBG_SYNC(h, "backward: dx + finalize");
    HIP_TRY(hipGetLastError());
    return FFVD_OK;
}

static int enqueue_grad(ffvd_handle *h, int S_total) {
    return h->cfg.branch == FFVD_BRANCH_A ? enqueue_grad_a(h, S_total) : enqueue_grad_b(h, S_total);
}
// forward + backward pass of one training iteration: ONE launch where the handle has the small-problem plan (tiny.hip)
static int enqueue_forward_backward(ffvd_handle *h, int S_total) {
    if (h->cfg.grad && tiny_selected(h)) return enqueue_tiny(h, nullptr, nullptr, true, S_total);
    const int r = enqueue_elbo(h, nullptr, nullptr);
    return r ? r : enqueue_grad(h, S_total);
}

// the gradient arrays of the handle to the caller's host arrays (enqueued on the main stream; the caller synchronises)
static int copy_grads_out(ffvd_handle *h, const ffvd_grads *gout) {
    const ffvd_config &c = h->cfg;
    ffvd_handle::GradWs &g = h->gw;
    hipStream_t s = h->stream;
    const size_t P = h->P, J = c.Ydim;
    if (gout->X) HIP_TRY(hipMemcpyAsync(gout->X, g.dX, (size_t)c.S_local * (c.T + 1) * c.D * sizeof(double), hipMemcpyDeviceToHost, s));
    if (gout->Z) HIP_TRY(hipMemcpyAsync(gout->Z, g.dZ, (size_t)c.M * P * sizeof(double), hipMemcpyDeviceToHost, s));
    if (gout->logvariance) HIP_TRY(hipMemcpyAsync(gout->logvariance, g.dlogvar, (size_t)c.D * sizeof(double), hipMemcpyDeviceToHost, s));
    if (gout->loglengthscales) HIP_TRY(hipMemcpyAsync(gout->loglengthscales, g.dloglen, (size_t)c.D * P * sizeof(double), hipMemcpyDeviceToHost, s));
    if (gout->log_Q) HIP_TRY(hipMemcpyAsync(gout->log_Q, g.dlogQ, (size_t)c.D * sizeof(double), hipMemcpyDeviceToHost, s));
    if (gout->CC) HIP_TRY(hipMemcpyAsync(gout->CC, g.dCC, (size_t)c.D * J * sizeof(double), hipMemcpyDeviceToHost, s));
    if (gout->DD) HIP_TRY(hipMemcpyAsync(gout->DD, g.dDD, J * sizeof(double), hipMemcpyDeviceToHost, s));
    if (gout->log_Rchols) HIP_TRY(hipMemcpyAsync(gout->log_Rchols, g.dlogR, J * J * sizeof(double), hipMemcpyDeviceToHost, s));
    if (gout->U) {
        if (g.dU) HIP_TRY(hipMemcpyAsync(gout->U, g.dU, (size_t)c.M * c.D * sizeof(double), hipMemcpyDeviceToHost, s));
        else memset(gout->U, 0, (size_t)c.M * c.D * sizeof(double));        // collapsed branch: U is integrated out
    }
    return FFVD_OK;
}

extern "C" int ffvd_elbo_grad(ffvd_handle *h, const ffvd_params *p, uint32_t flags, int S_total, double out_terms[8],
                              double *out_nll, const ffvd_grads *gout) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_elbo_grad: null handle");
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_elbo_grad: the handle was created without grad = 1");
    if (!gout) return set_error(h, FFVD_EINVAL, "ffvd_elbo_grad: null gradient struct");
    if (S_total < h->cfg.S_local) return set_error(h, FFVD_EINVAL, "ffvd_elbo_grad: S_total < S_local");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if (p) {
        if (flags & FFVD_PARAMS_ON_DEVICE) {
            if ((rc = check_params(h, p, "ffvd_elbo_grad"))) return rc;
            h->cur = *p;
            if (!h->cur.U) h->cur.U = h->U;
            if (!h->cur.loglengthscales) h->cur.loglengthscales = h->loglen;
            h->have_params = true;
        } else if ((rc = ffvd_set_params(h, p, 0))) return rc;
    }
    if ((rc = ready(h, "ffvd_elbo_grad"))) return rc;
    if ((rc = fetch_with_stall_recovery(h, [&] { return enqueue_forward_backward(h, S_total); })))
        return rc;
    if ((rc = copy_grads_out(h, gout))) return rc;
    const ffvd_config &c = h->cfg;
    hipStream_t s = h->stream;
    HIP_TRY(hipStreamSynchronize(s));
    if (out_terms) memcpy(out_terms, h->h_out, 8 * sizeof(double));
    if (out_nll) *out_nll = h->h_out[FFVD_TERM_NLL] / (double)c.S_local;
    return FFVD_OK;
}

// ---- optimiser steps on the resident parameters --------------------------------------
static constexpr int NPARAM = 9;       // order of the FFVD_TRAIN_* bits: X, Z, logvariance, loglengthscales, log_Q, CC, DD, log_Rchols, U
static void param_table(ffvd_handle *h, double *theta[NPARAM], const double *grad[NPARAM], size_t n[NPARAM]) {
    const ffvd_config &c = h->cfg;
    const size_t P = h->P, J = c.Ydim;
    const ffvd_handle::GradWs &g = h->gw;
    const ffvd_params &p = h->cur;
    double *th[NPARAM] = {const_cast<double *>(p.X), const_cast<double *>(p.Z), const_cast<double *>(p.logvariance),
                          const_cast<double *>(p.loglengthscales), const_cast<double *>(p.log_Q), const_cast<double *>(p.CC),
                          const_cast<double *>(p.DD), const_cast<double *>(p.log_Rchols), const_cast<double *>(p.U)};
    const double *gr[NPARAM] = {g.dX, g.dZ, g.dlogvar, g.dloglen, g.dlogQ, g.dCC, g.dDD, g.dlogR, g.dU};
    const size_t nn[NPARAM] = {(size_t)c.S_local * (c.T + 1) * c.D, (size_t)c.M * P, (size_t)c.D, (size_t)c.D * P, (size_t)c.D,
                               (size_t)c.D * J, J, J * J, g.dU ? (size_t)c.M * c.D : 0};      // U only where it has a gradient
    for (int i = 0; i < NPARAM; ++i) { theta[i] = th[i]; grad[i] = gr[i]; n[i] = nn[i]; }
}

extern "C" int ffvd_optimizer_reset(ffvd_handle *h) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_optimizer_reset: null handle");
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_optimizer_reset: the handle was created without grad = 1");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    double *theta[NPARAM]; const double *grad[NPARAM]; size_t n[NPARAM];
    param_table(h, theta, grad, n);
    for (int i = 0; i < NPARAM; ++i) {
        if (!n[i]) continue;
        if (!h->adam_m[i]) { HIP_TRY(dev_alloc(h, &h->adam_m[i], n[i])); HIP_TRY(dev_alloc(h, &h->adam_v[i], n[i])); }
        HIP_TRY(hipMemsetAsync(h->adam_m[i], 0, n[i] * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->adam_v[i], 0, n[i] * sizeof(double), h->stream));
    }
    h->adam_t = 0;
    h->adam_ready = true;
    return FFVD_OK;
}

static int adam_update(ffvd_handle *h, double lr, double beta1, double beta2, double eps, uint32_t train_mask);
static int sghmc_prepare(ffvd_handle *h, uint32_t sample_mask, const ffvd_params *noise, const char *who);
static int sghmc_update(ffvd_handle *h, double epsilon, double mdecay, uint32_t sample_mask, int burn_in);

extern "C" int ffvd_adam_step(ffvd_handle *h, double lr, double beta1, double beta2, double eps, uint32_t train_mask,
                              double out_terms[8], double *out_nll) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_adam_step: null handle");
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_adam_step: the handle was created without grad = 1");
    if (h->Dl != h->cfg.D)
        return set_error(h, FFVD_EINVAL, "ffvd_adam_step: a latent-dim shard holds partial gradients; use ffvd_adam_step_allreduce");
    if (h->comm_world > 1)
        return set_error(h, FFVD_EINVAL, "ffvd_adam_step: this handle is one rank of a multi-rank communicator, its gradient is a share of "
                                         "the job's; use ffvd_adam_step_allreduce");
    if (!(lr > 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return set_error(h, FFVD_EINVAL, "ffvd_adam_step: bad hyper-parameter");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if ((rc = ready(h, "ffvd_adam_step"))) return rc;
    if (!h->adam_ready && (rc = ffvd_optimizer_reset(h))) return rc;
    // (a failed factorisation leaves the parameters untouched)
    if ((rc = fetch_with_stall_recovery(h, [&] { return enqueue_forward_backward(h, h->cfg.S_local); })))
        return rc;
    if ((rc = adam_update(h, lr, beta1, beta2, eps, train_mask))) return rc;
    if (out_terms) memcpy(out_terms, h->h_out, 8 * sizeof(double));
    if (out_nll) *out_nll = h->h_out[FFVD_TERM_NLL] / (double)h->cfg.S_local;
    return FFVD_OK;
}

extern "C" int ffvd_get_params(ffvd_handle *h, const ffvd_params *out) {
    if (!h || !out) return set_error(h, FFVD_EINVAL, "ffvd_get_params: null argument");
    if (!h->have_params) return set_error(h, FFVD_EINVAL, "ffvd_get_params: no parameters bound");
    const ffvd_config &c = h->cfg;
    const size_t P = h->P, J = c.Ydim;
    HIP_TRY(hipSetDevice(c.device_id));
    hipStream_t s = h->stream;
    const ffvd_params &p = h->cur;
    const void *src[9] = {p.X, p.Z, p.U, p.logvariance, p.loglengthscales, p.log_Q, p.CC, p.DD, p.log_Rchols};
    const void *dst[9] = {out->X, out->Z, out->U, out->logvariance, out->loglengthscales, out->log_Q, out->CC, out->DD,
                          out->log_Rchols};
    const size_t n[9] = {(size_t)c.S_local * (c.T + 1) * c.D, (size_t)c.M * P, (size_t)c.M * c.D, (size_t)c.D, (size_t)c.D * P,
                         (size_t)c.D, (size_t)c.D * J, J, J * J};
    for (int i = 0; i < 9; ++i)
        if (dst[i] && src[i] && n[i])
            HIP_TRY(hipMemcpyAsync(const_cast<void *>(dst[i]), src[i], n[i] * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return FFVD_OK;
}

// ffvd_update_params: overwrite some of the bound parameter arrays from host memory (NULL = keep).
extern "C" int ffvd_update_params(ffvd_handle *h, const ffvd_params *p) {
    if (!h || !p) return set_error(h, FFVD_EINVAL, "ffvd_update_params: null argument");
    if (!h->have_params) return set_error(h, FFVD_EINVAL, "ffvd_update_params: no parameters bound yet (ffvd_set_params)");
    const ffvd_config &c = h->cfg;
    const size_t P = h->P, J = c.Ydim;
    HIP_TRY(hipSetDevice(c.device_id));
    const ffvd_params &cur = h->cur;
    const void *dst[9] = {cur.X, cur.Z, cur.U, cur.logvariance, cur.loglengthscales, cur.log_Q, cur.CC, cur.DD, cur.log_Rchols};
    const void *src[9] = {p->X, p->Z, p->U, p->logvariance, p->loglengthscales, p->log_Q, p->CC, p->DD, p->log_Rchols};
    const size_t n[9] = {(size_t)c.S_local * (c.T + 1) * c.D, (size_t)c.M * P, (size_t)c.M * c.D, (size_t)c.D, (size_t)c.D * P,
                         (size_t)c.D, (size_t)c.D * J, J, J * J};
    for (int i = 0; i < 9; ++i)
        if (src[i] && dst[i] && n[i])
            HIP_TRY(hipMemcpyAsync(const_cast<void *>(dst[i]), src[i], n[i] * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return FFVD_OK;
}

extern "C" int ffvd_sghmc_step(ffvd_handle *h, double epsilon, double mdecay, uint32_t sample_mask, int burn_in,
                               const ffvd_params *noise, double out_terms[8], double *out_nll) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_sghmc_step: null handle");
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step: the handle was created without grad = 1");
    if (h->Dl != h->cfg.D)
        return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step: a latent-dim shard holds partial gradients; use ffvd_sghmc_step_allreduce");
    if (h->comm_world > 1)
        return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step: this handle is one rank of a multi-rank communicator, its gradient is a share of "
                                         "the job's; use ffvd_sghmc_step_allreduce");
    if (!noise || !(epsilon > 0.0) || !(mdecay >= 0.0))
        return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step: bad argument");
    if (sample_mask & FFVD_TRAIN_X)
        return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step: X is never an SG-HMC variable (dgp_model.py:213-244)");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if ((rc = ready(h, "ffvd_sghmc_step"))) return rc;
    if ((rc = sghmc_prepare(h, sample_mask, noise, "ffvd_sghmc_step"))) return rc;
    if ((rc = fetch_with_stall_recovery(h, [&] { return enqueue_forward_backward(h, h->cfg.S_local); })))
        return rc;
    if ((rc = sghmc_update(h, epsilon, mdecay, sample_mask, burn_in))) return rc;
    if (out_terms) memcpy(out_terms, h->h_out, 8 * sizeof(double));
    if (out_nll) *out_nll = h->h_out[FFVD_TERM_NLL] / (double)h->cfg.S_local;
    return FFVD_OK;
}

// ---- sharded, device-resident training step (multi-GPU counterpart of adam.minimize(nll), dgp_model.py:303-305 /
// train_hypers, base_model.py:944-950, and of one burn_in_op / sample_op, base_model.py:143-179) ---------------------------
// Every rank: forward + backward with the whole job's chain count as the divisor, so that its gradient block is its ADDITIVE
// share; ONE all-reduce(sum) of [8 term sums | shared-parameter gradients (| dX for latent-dim shards)] in place in HBM;
// then the fused update from the reduced block.  No gradient crosses PCIe; the only host traffic is the result block.
static int train_local(ffvd_handle *h, int S_total, const char *who) {
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, std::string(who) + ": the handle was created without grad = 1");
    if (S_total < h->cfg.S_local) return set_error(h, FFVD_EINVAL, std::string(who) + ": S_total < S_local");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if ((rc = ready(h, who))) return rc;
    h->train_S_total = 0;
    if ((rc = enqueue_forward_backward(h, S_total))) return rc;
    HIP_TRY(hipMemcpyAsync(h->gw.pack, h->out_terms, 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    h->train_S_total = S_total;
    return FFVD_OK;
}

// doubles of the block that take part in the exchange: chain shards keep dX (their own chains' rows) out of it
static size_t train_exchange_count(const ffvd_handle *h) {
    return (h->Dl != h->cfg.D) ? h->gw.pack_total : h->gw.pack_shared;
}

// result block + the (reduced) sums to the host, one synchronisation; a failed factorisation anywhere leaves the parameters as they are
static int train_fetch(ffvd_handle *h, const char *who) {
    if (h->train_S_total <= 0) return set_error(h, FFVD_EINVAL, std::string(who) + ": no pending backward pass (ffvd_train_local first)");
    hipStream_t s = h->stream;
    HIP_TRY(hipMemcpyAsync(h->h_res, h->resblk, h->res_bytes, hipMemcpyDeviceToHost, s));      // this rank's chain nll + info flags
    HIP_TRY(hipMemcpyAsync(h->h_sums, h->gw.pack, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->train_S_total = 0;
    int rc;
    if ((rc = check_info(h))) return rc;
    for (int i = 0; i < 8; ++i)
        if (!std::isfinite(h->h_sums[i]))
            return set_error(h, FFVD_ENOTPD, std::string(who) + ": non-finite sums after the exchange (a factorisation failed or was abandoned on another rank); parameters untouched");
    return FFVD_OK;
}

static void train_report(ffvd_handle *h, double out_terms[8], double *out_nll) {
    if (out_terms) memcpy(out_terms, h->h_sums, 8 * sizeof(double));
    if (out_nll) *out_nll = h->h_sums[FFVD_TERM_NLL] / h->h_sums[FFVD_TERM_COUNT];
}

static int adam_update(ffvd_handle *h, double lr, double beta1, double beta2, double eps, uint32_t train_mask) {
    double *theta[NPARAM]; const double *grad[NPARAM]; size_t n[NPARAM];
    param_table(h, theta, grad, n);
    OptTable tab{};
    for (int i = 0; i < NPARAM; ++i) {
        if (!(train_mask & (1u << i)) || n[i] == 0) continue;
        OptTensor &t = tab.t[tab.count++];
        t.theta = theta[i]; t.grad = grad[i]; t.s0 = h->adam_m[i]; t.s1 = h->adam_v[i]; t.n = (int64_t)n[i];
    }
    h->adam_t += 1;
    const double lr_t = lr * sqrt(1.0 - pow(beta2, (double)h->adam_t)) / (1.0 - pow(beta1, (double)h->adam_t));
    launch_adam(h->stream, tab, lr_t, beta1, beta2, eps);
    HIP_TRY(hipGetLastError());
    return FFVD_OK;
}

extern "C" int ffvd_train_local(ffvd_handle *h, int S_total) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_train_local: null handle");
    return train_local(h, S_total, "ffvd_train_local");
}

extern "C" int64_t ffvd_train_exchange_count(const ffvd_handle *h) { return (h && h->cfg.grad) ? (int64_t)train_exchange_count(h) : 0; }

extern "C" void *ffvd_train_exchange_ptr(ffvd_handle *h) { return (h && h->cfg.grad) ? (void *)h->gw.pack : nullptr; }

extern "C" int ffvd_train_exchange_get(ffvd_handle *h, double *host_out) {
    if (!h || !host_out || !h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_train_exchange_get: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipMemcpyAsync(host_out, h->gw.pack, train_exchange_count(h) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return FFVD_OK;
}

extern "C" int ffvd_train_exchange_set(ffvd_handle *h, const double *host_in) {
    if (!h || !host_in || !h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_train_exchange_set: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipMemcpyAsync(h->gw.pack, host_in, train_exchange_count(h) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));          // the host array is the caller's
    return FFVD_OK;
}

extern "C" int ffvd_adam_apply(ffvd_handle *h, double lr, double beta1, double beta2, double eps, uint32_t train_mask,
                               double out_terms[8], double *out_nll) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_adam_apply: null handle");
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_adam_apply: the handle was created without grad = 1");
    if (!(lr > 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return set_error(h, FFVD_EINVAL, "ffvd_adam_apply: bad hyper-parameter");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if (!h->adam_ready && (rc = ffvd_optimizer_reset(h))) return rc;
    if ((rc = train_fetch(h, "ffvd_adam_apply"))) return rc;
    if ((rc = adam_update(h, lr, beta1, beta2, eps, train_mask))) return rc;
    train_report(h, out_terms, out_nll);
    return FFVD_OK;
}

extern "C" int ffvd_adam_step_allreduce(ffvd_handle *h, void *rccl_comm, int S_total, double lr, double beta1, double beta2,
                                        double eps, uint32_t train_mask, double out_terms[8], double *out_nll) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_adam_step_allreduce: null handle");
    if (!(lr > 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return set_error(h, FFVD_EINVAL, "ffvd_adam_step_allreduce: bad hyper-parameter");
    if (!rccl_comm && !h->comm)
        return set_error(h, FFVD_EINVAL, "ffvd_adam_step_allreduce: no communicator (pass one or call ffvd_comm_init)");
    int rc;
    if (h->cfg.grad && !h->adam_ready && (rc = ffvd_optimizer_reset(h))) return rc;
    if ((rc = train_local(h, S_total, "ffvd_adam_step_allreduce"))) return rc;
    if ((rc = ffvd_allreduce_sum_async(h, rccl_comm, h->gw.pack, (int64_t)train_exchange_count(h)))) { h->train_S_total = 0; return rc; }
    if ((rc = train_fetch(h, "ffvd_adam_step_allreduce"))) return rc;
    if ((rc = adam_update(h, lr, beta1, beta2, eps, train_mask))) return rc;
    train_report(h, out_terms, out_nll);
    return FFVD_OK;
}

static int sghmc_prepare(ffvd_handle *h, uint32_t sample_mask, const ffvd_params *noise, const char *who) {
    double *theta[NPARAM]; const double *grad[NPARAM]; size_t n[NPARAM];
    param_table(h, theta, grad, n);
    const double *nz[NPARAM] = {noise->X, noise->Z, noise->logvariance, noise->loglengthscales, noise->log_Q, noise->CC,
                                noise->DD, noise->log_Rchols, noise->U};
    for (int i = 1; i < NPARAM; ++i) {
        if (!(sample_mask & (1u << i)) || n[i] == 0) continue;
        if (!nz[i]) return set_error(h, FFVD_EINVAL, std::string(who) + ": a sampled array has no noise array");
        if (!h->hmc[i][0]) {            // xi, g, g2 <- 1, p <- 0 (base_model.py:151-154)
            std::vector<double> ones(n[i], 1.0);
            for (int k = 0; k < 5; ++k) HIP_TRY(dev_alloc(h, &h->hmc[i][k], n[i]));
            for (int k = 0; k < 3; ++k)
                HIP_TRY(hipMemcpy(h->hmc[i][k], ones.data(), n[i] * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(hipMemset(h->hmc[i][3], 0, n[i] * sizeof(double)));
        }
        HIP_TRY(hipMemcpyAsync(h->hmc[i][4], nz[i], n[i] * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    return FFVD_OK;
}

static int sghmc_update(ffvd_handle *h, double epsilon, double mdecay, uint32_t sample_mask, int burn_in) {
    double *theta[NPARAM]; const double *grad[NPARAM]; size_t n[NPARAM];
    param_table(h, theta, grad, n);
    OptTable tab{};
    for (int i = 1; i < NPARAM; ++i) {
        if (!(sample_mask & (1u << i)) || n[i] == 0) continue;
        OptTensor &t = tab.t[tab.count++];
        t.theta = theta[i]; t.grad = grad[i]; t.s0 = h->hmc[i][0]; t.s1 = h->hmc[i][1]; t.s2 = h->hmc[i][2];
        t.s3 = h->hmc[i][3]; t.noise = h->hmc[i][4]; t.n = (int64_t)n[i];
    }
    // X_N = rows of X (dgp_model.py:203) -- of the JOB's trajectory: a T-shard handle holds T_r + 1 of its T_total + 1 rows
    const double x_n = (double)((h->cfg.T_total > 0 ? h->cfg.T_total : h->cfg.T) + 1);
    launch_sghmc(h->stream, tab, epsilon, mdecay, x_n, burn_in);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));           // the noise arrays are the caller's: done with them on return
    return FFVD_OK;
}

extern "C" int ffvd_sghmc_step_allreduce(ffvd_handle *h, void *rccl_comm, int S_total, double epsilon, double mdecay,
                                         uint32_t sample_mask, int burn_in, const ffvd_params *noise, double out_terms[8],
                                         double *out_nll) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_sghmc_step_allreduce: null handle");
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step_allreduce: the handle was created without grad = 1");
    if (!noise || !(epsilon > 0.0) || !(mdecay >= 0.0)) return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step_allreduce: bad argument");
    if (sample_mask & FFVD_TRAIN_X)
        return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step_allreduce: X is never an SG-HMC variable (dgp_model.py:213-244)");
    if (!rccl_comm && !h->comm)
        return set_error(h, FFVD_EINVAL, "ffvd_sghmc_step_allreduce: no communicator (pass one or call ffvd_comm_init)");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if ((rc = ready(h, "ffvd_sghmc_step_allreduce"))) return rc;
    if ((rc = sghmc_prepare(h, sample_mask, noise, "ffvd_sghmc_step_allreduce"))) return rc;
    if ((rc = train_local(h, S_total, "ffvd_sghmc_step_allreduce"))) return rc;
    if ((rc = ffvd_allreduce_sum_async(h, rccl_comm, h->gw.pack, (int64_t)train_exchange_count(h)))) { h->train_S_total = 0; return rc; }
    if ((rc = train_fetch(h, "ffvd_sghmc_step_allreduce"))) return rc;
    if ((rc = sghmc_update(h, epsilon, mdecay, sample_mask, burn_in))) return rc;
    train_report(h, out_terms, out_nll);
    return FFVD_OK;
}

extern "C" int ffvd_sghmc_apply(ffvd_handle *h, double epsilon, double mdecay, uint32_t sample_mask, int burn_in,
                                const ffvd_params *noise, double out_terms[8], double *out_nll) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_sghmc_apply: null handle");
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, "ffvd_sghmc_apply: the handle was created without grad = 1");
    if (!noise || !(epsilon > 0.0) || !(mdecay >= 0.0)) return set_error(h, FFVD_EINVAL, "ffvd_sghmc_apply: bad argument");
    if (sample_mask & FFVD_TRAIN_X) return set_error(h, FFVD_EINVAL, "ffvd_sghmc_apply: X is never an SG-HMC variable (dgp_model.py:213-244)");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int rc;
    if ((rc = sghmc_prepare(h, sample_mask, noise, "ffvd_sghmc_apply"))) return rc;
    if ((rc = train_fetch(h, "ffvd_sghmc_apply"))) return rc;
    if ((rc = sghmc_update(h, epsilon, mdecay, sample_mask, burn_in))) return rc;
    train_report(h, out_terms, out_nll);
    return FFVD_OK;
}

// ---- operator-level entry points (temporaries per call; not the hot path) -------------------------------------------------------------
// Round 5: the temporaries come from a per-thread cache instead of hipMalloc / hipFree / hipStreamCreate per call.  A rollout call made
// ~27 allocations and as many frees around 200 steps of 15 us: 2.7 ms of host work per call (profiles/r04_next_rows.json: 27.6 us per
// step of a 200-step call against 15.3 marginal).  A block is handed out again when its size fits (smallest block that is large enough
// and at most 4 x the request); the cache is bounded (256 blocks / 8 GiB: beyond that everything idle is freed) and can be released
// with ffvd_op_release_cache().  Nothing relies on fresh memory being zero (hipMalloc never promised that).
namespace {
struct OpCache {
    struct Block { void *p; size_t bytes; bool busy; };
    std::vector<Block> blocks;
    size_t total = 0;
    hipStream_t stream = nullptr;
    int device = -1;
    void *pinned = nullptr;             // page-locked staging block for large results (a device-to-host copy into fresh pageable memory
    size_t pinned_bytes = 0;            //  pins its pages on the fly: 13 MB took up to 30 ms; through this block + memcpy: ~3)
    void release_idle() {
        std::vector<Block> keep;
        for (Block &b : blocks) {
            if (b.busy) keep.push_back(b);
            else { hipFree(b.p); total -= b.bytes; }
        }
        blocks.swap(keep);
    }
    void release_all() {
        release_idle();
        if (stream && blocks.empty()) { hipStreamDestroy(stream); stream = nullptr; }
        if (pinned) { hipHostFree(pinned); pinned = nullptr; pinned_bytes = 0; }
    }
    ~OpCache() { /* process exit: the runtime may already be gone; leave the memory to it */ }
};
thread_local OpCache g_op_cache;

struct Scratch {
    std::vector<size_t> mine;           // indices into the cache of the blocks this call holds
    hipStream_t stream = nullptr;
    bool begin() {
        OpCache &c = g_op_cache;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (c.device != dev) {           // (another device was made current on this thread: start over)
            c.release_all();
            c.device = dev;
        }
        if (!c.stream && hipStreamCreate(&c.stream) != hipSuccess) return false;
        stream = c.stream;
        return true;
    }
    ~Scratch() {
        if (stream) hipStreamSynchronize(stream);
        OpCache &c = g_op_cache;
        for (size_t i : mine) c.blocks[i].busy = false;
        if (c.blocks.size() > 256 || c.total > ((size_t)8 << 30)) c.release_idle();
    }
    template <class T>
    T *alloc(size_t n) {
        OpCache &c = g_op_cache;
        const size_t bytes = ((n ? n : 1) * sizeof(T) + 255) / 256 * 256;
        size_t best = (size_t)-1;
        for (size_t i = 0; i < c.blocks.size(); ++i) {
            const OpCache::Block &b = c.blocks[i];
            if (!b.busy && b.bytes >= bytes && b.bytes <= 4 * bytes && (best == (size_t)-1 || b.bytes < c.blocks[best].bytes)) best = i;
        }
        if (best == (size_t)-1) {
            void *p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) {
                (void)hipGetLastError();
                c.release_idle();                                   // make room and try once more
                mine.clear();
                for (size_t i = 0; i < c.blocks.size(); ++i) if (c.blocks[i].busy) mine.push_back(i);      // (indices moved; every busy block is this call's: one call per thread at a time)
                if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            }
            c.blocks.push_back({p, bytes, false});
            c.total += bytes;
            best = c.blocks.size() - 1;
        }
        c.blocks[best].busy = true;
        mine.push_back(best);
        // FFVD_OP_CACHE_POISON=1 (tests): every block handed out starts as NaN bytes -- nothing may rely on what a block held before
        static const bool poison = [] { const char *e = getenv("FFVD_OP_CACHE_POISON"); return e && *e && strcmp(e, "0") != 0; }();
        if (poison && stream) (void)hipMemsetAsync(c.blocks[best].p, 0xFF, c.blocks[best].bytes, stream);
        return (T *)c.blocks[best].p;
    }
    // device -> host for a large result: through the thread's page-locked block when it has (or gets) one of that size, else directly.
    // Synchronises the stream.
    bool download(void *dst, const void *src, size_t bytes) {
        if (void *pin = pinned_block(bytes)) {
            if (hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
            if (hipStreamSynchronize(stream) != hipSuccess) return false;
            memcpy(dst, pin, bytes);
            return true;
        }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    }
    // the thread's page-locked block, grown to `bytes` (nullptr when it cannot be had or the size is outside 1 MB .. 256 MB)
    void *pinned_block(size_t bytes) {
        OpCache &c = g_op_cache;
        if (bytes < ((size_t)1 << 20) || bytes > ((size_t)256 << 20)) return nullptr;
        if (c.pinned_bytes < bytes) {
            if (c.pinned) { hipHostFree(c.pinned); c.pinned = nullptr; c.pinned_bytes = 0; }
            if (hipHostMalloc(&c.pinned, bytes, hipHostMallocDefault) == hipSuccess) c.pinned_bytes = bytes;
            else { (void)hipGetLastError(); c.pinned = nullptr; }
        }
        return c.pinned;
    }
    // host -> device, asynchronous, straight from the caller's memory.  (Measured: large uploads through the page-locked block + a
    // synchronisation made a 200-step rollout call 0.4 ms slower and changed nothing for the sweep -- the 28 ms some calls spend here in
    // tools/bench_next.py's process are a wait of the stream behind the 0.2 ms DMA, whichever memory it reads; FFVD_PG_TIMING=1 shows it.)
    double *upload(const double *src, size_t n) {
        double *d = alloc<double>(n);
        if (d && n && hipMemcpyAsync(d, src, n * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) return nullptr;
        return d;
    }
};
}  // namespace

// Free what the calling thread's operator cache holds (device buffers of finished ffvd_op_* calls, their stream).
extern "C" int ffvd_op_release_cache(void) {
    g_op_cache.release_all();
    return FFVD_OK;
}

#define OP_BEGIN(name)                                                                       \
    ffvd_handle *h = nullptr;                                                                \
    (void)h;                                                                                 \
    Scratch sc;                                                                              \
    if (!sc.begin())                                                                         \
        return set_error(nullptr, FFVD_EDEVICE, name ": no usable HIP device / stream creation failed");
#define OP_CHECK(ptr, name) \
    if (!(ptr)) return set_error(nullptr, FFVD_ENOMEM, name ": device allocation or upload failed");

extern "C" int ffvd_op_kernel_matrix(int kind, const double *X, int N, const double *X2, int N2, int P,
                                     double logvariance, const double *loglengthscales, double jitter, double *out) {
    if (!X || !out || N < 0 || P < 1 || (X2 && N2 < 0) || (kind == FFVD_KERNEL_SE && !loglengthscales) ||
        (kind != FFVD_KERNEL_SE && kind != FFVD_KERNEL_LINEAR))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_kernel_matrix: bad argument");
    OP_BEGIN("ffvd_op_kernel_matrix");
    const int same = (X2 == nullptr);
    if (same) N2 = N;
    if ((size_t)N * N2 == 0) return FFVD_OK;
    double *dX = sc.upload(X, (size_t)N * P);
    OP_CHECK(dX, "ffvd_op_kernel_matrix");
    double *dX2 = same ? dX : sc.upload(X2, (size_t)N2 * P);
    OP_CHECK(dX2, "ffvd_op_kernel_matrix");
    double *dl = sc.alloc<double>(P);
    OP_CHECK(dl, "ffvd_op_kernel_matrix");
    if (loglengthscales) HIP_TRY(hipMemcpyAsync(dl, loglengthscales, P * sizeof(double), hipMemcpyHostToDevice, sc.stream));
    double *dO = sc.alloc<double>((size_t)N * N2);
    OP_CHECK(dO, "ffvd_op_kernel_matrix");
    launch_kernel_matrix(sc.stream, kind, dX, N, dX2, N2, P, logvariance, dl, jitter, same, dO);
    HIP_TRY(hipMemcpyAsync(out, dO, (size_t)N * N2 * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

extern "C" int ffvd_op_kernel_diag(int kind, const double *X, int N, int P, double logvariance, double *out) {
    if (!X || !out || N < 0 || P < 1) return set_error(nullptr, FFVD_EINVAL, "ffvd_op_kernel_diag: bad argument");
    OP_BEGIN("ffvd_op_kernel_diag");
    if (N == 0) return FFVD_OK;
    double *dX = sc.upload(X, (size_t)N * P);
    OP_CHECK(dX, "ffvd_op_kernel_diag");
    double *dO = sc.alloc<double>(N);
    OP_CHECK(dO, "ffvd_op_kernel_diag");
    launch_kernel_diag(sc.stream, kind, dX, N, P, logvariance, dO);
    HIP_TRY(hipMemcpyAsync(out, dO, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

extern "C" int ffvd_op_cholesky(const double *A, int n, int batch, double *L, int32_t *info) {
    if (!A || !L || n < 1 || batch < 0) return set_error(nullptr, FFVD_EINVAL, "ffvd_op_cholesky: bad argument");
    OP_BEGIN("ffvd_op_cholesky");
    if (batch == 0) return FFVD_OK;
    const int np = round_up(n, NB);
    const size_t slab = (size_t)np * np;
    std::vector<double> pad(slab * batch);
    auto fill_pad = [&] {
        std::fill(pad.begin(), pad.end(), 0.0);
        for (int b = 0; b < batch; ++b) {
            double *S = pad.data() + slab * b;
            for (int i = 0; i < np; ++i) {
                if (i < n) memcpy(S + (size_t)i * np, A + ((size_t)b * n + i) * n, (size_t)n * sizeof(double));
                else S[(size_t)i * np + i] = 1.0;
            }
        }
    };
    fill_pad();
    double *dA = sc.upload(pad.data(), pad.size());
    OP_CHECK(dA, "ffvd_op_cholesky");
    int32_t *dinfo = sc.alloc<int32_t>(batch);
    OP_CHECK(dinfo, "ffvd_op_cholesky");
    double *dinv = sc.alloc<double>(potrf_scratch_doubles(np, batch));
    OP_CHECK(dinv, "ffvd_op_cholesky");
    std::vector<int32_t> hinfo(batch, 0);
    bool recovered = false;
    for (int attempt = 0;; ++attempt) {
        // attempt 1 (only after the dataflow launch gave up on a bounded wait): the same batch again with the launch-per-column
        // Cholesky, which has no inter-workgroup waits (stall recovery, see fetch_with_stall_recovery)
        {
            CholOverrideGuard guard;            // reset on the error returns below as well
            if (attempt == 1) {
                fill_pad();
                HIP_TRY(hipMemcpyAsync(dA, pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice, sc.stream));
                guard.force_left();
            }
            HIP_TRY(hipMemsetAsync(dinfo, 0, batch * sizeof(int32_t), sc.stream));
            launch_potrf_ext(sc.stream, dA, np, 0, 0, batch, slab, dinfo, dinv);
        }
        HIP_TRY(hipMemcpyAsync(pad.data(), dA, pad.size() * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
        HIP_TRY(hipMemcpyAsync(hinfo.data(), dinfo, batch * sizeof(int32_t), hipMemcpyDeviceToHost, sc.stream));
        HIP_TRY(hipStreamSynchronize(sc.stream));
        bool stalled = false;
        for (int b = 0; b < batch; ++b) stalled = stalled || hinfo[b] < 0;
        if (!stalled) { recovered = attempt == 1; break; }
        if (attempt == 1)
            return set_error(nullptr, FFVD_EDEVICE, "ffvd_op_cholesky: abandoned, a block row waited more than 1 s for the row above it");
    }
    int bad = -1;
    for (int b = 0; b < batch; ++b) {
        const double *S = pad.data() + slab * b;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) L[((size_t)b * n + i) * n + j] = (j <= i) ? S[(size_t)i * np + j] : 0.0;
        if (info) info[b] = hinfo[b];
        if (hinfo[b] != 0 && bad < 0) bad = b;
    }
    if (bad >= 0) {
        char msg[160];
        snprintf(msg, sizeof msg, "ffvd_op_cholesky: matrix %d is not positive definite (pivot %d)", bad, hinfo[bad] - 1);
        return set_error(nullptr, FFVD_ENOTPD, msg);
    }
    if (recovered)
        set_error(nullptr, FFVD_OK, "warning: ffvd_op_cholesky: the one-launch (dataflow) Cholesky gave up on a bounded wait; the batch was "
                                    "re-run with the launch-per-column Cholesky and completed");
    return FFVD_OK;
}

// shared by kernel_pre_cal / conditional: device-side hypers + K_uu + extended Cholesky
struct KuuWork {
    double *variance, *len, *Zs, *zz, *Kuu, *logvar, *loglen, *dinv;
    int32_t *info;
    int Mp;
};
static int build_kuu(Scratch &sc, int kind, const double *Z, int M, int P, int D, const double *logvariance,
                     const double *loglengthscales, double jitter, KuuWork &w, double **dZ_out) {
    const int Mp = round_up(M, NB);
    w.Mp = Mp;
    double *dZ = sc.upload(Z, (size_t)M * P);
    w.logvar = sc.upload(logvariance, D);
    w.loglen = sc.alloc<double>((size_t)D * P);
    w.variance = sc.alloc<double>(D);
    w.len = sc.alloc<double>((size_t)D * P);
    w.Zs = sc.alloc<double>((size_t)D * Mp * P);
    w.zz = sc.alloc<double>((size_t)D * Mp);
    w.Kuu = sc.alloc<double>((size_t)D * 2 * Mp * Mp);
    w.info = sc.alloc<int32_t>(D);
    w.dinv = sc.alloc<double>(potrf_scratch_doubles(Mp, D));
    if (!dZ || !w.logvar || !w.loglen || !w.variance || !w.len || !w.Zs || !w.zz || !w.Kuu || !w.info || !w.dinv) return FFVD_ENOMEM;
    if (loglengthscales &&
        hipMemcpyAsync(w.loglen, loglengthscales, (size_t)D * P * sizeof(double), hipMemcpyHostToDevice, sc.stream) != hipSuccess)
        return FFVD_EDEVICE;
    if (hipMemsetAsync(w.info, 0, D * sizeof(int32_t), sc.stream) != hipSuccess) return FFVD_EDEVICE;
    launch_prep_hypers(sc.stream, kind, dZ, M, Mp, P, D, 0, w.logvar, w.loglen, w.variance, w.len, w.Zs, w.zz);
    HyperView hv{w.variance, w.len, w.Zs, w.zz};
    launch_kuu_build(sc.stream, kind, hv, M, Mp, P, D, jitter, w.Kuu, nullptr);
    launch_potrf_ext(sc.stream, w.Kuu, Mp, Mp, Mp, D, (size_t)2 * Mp * Mp, w.info, w.dinv);
    if (dZ_out) *dZ_out = dZ;
    return FFVD_OK;
}
static int check_kuu_info(Scratch &sc, const KuuWork &w, int D, const char *who) {
    std::vector<int32_t> hinfo(D, 0);
    if (hipMemcpyAsync(hinfo.data(), w.info, D * sizeof(int32_t), hipMemcpyDeviceToHost, sc.stream) != hipSuccess ||
        hipStreamSynchronize(sc.stream) != hipSuccess)
        return set_error(nullptr, FFVD_EDEVICE, std::string(who) + ": device error while reading Cholesky status");
    for (int d = 0; d < D; ++d)
        if (hinfo[d]) {
            char msg[200];
            snprintf(msg, sizeof msg, "%s: Cholesky of K_uu + jitter*I failed: latent dim %d, pivot %d is not positive", who, d,
                     hinfo[d] - 1);
            return set_error(nullptr, FFVD_ENOTPD, msg);
        }
    return FFVD_OK;
}

extern "C" int ffvd_op_trsm(const double *L, int n, const double *B, int m, double *X) {
    if (!L || !B || !X || n < 1 || m < 0) return set_error(nullptr, FFVD_EINVAL, "ffvd_op_trsm: bad argument");
    for (int i = 0; i < n; ++i)
        if (!(L[(size_t)i * n + i] != 0.0)) return set_error(nullptr, FFVD_EINVAL, "ffvd_op_trsm: zero on the diagonal of L");
    OP_BEGIN("ffvd_op_trsm");
    if (m == 0) return FFVD_OK;
    // slab = [ L (np x np, identity padding) ; R = B^T (mp x np, zero padding) ];  X^T = R L^-T
    const int np = round_up(n, NB), mp = round_up(m, NB);
    std::vector<double> slab((size_t)(np + mp) * np, 0.0);
    for (int i = 0; i < np; ++i) {
        if (i < n) for (int j = 0; j <= i; ++j) slab[(size_t)i * np + j] = L[(size_t)i * n + j];
        else slab[(size_t)i * np + i] = 1.0;
    }
    for (int c = 0; c < m; ++c)
        for (int i = 0; i < n; ++i) slab[(size_t)(np + c) * np + i] = B[(size_t)i * m + c];
    double *dA = sc.upload(slab.data(), slab.size());
    OP_CHECK(dA, "ffvd_op_trsm");
    double *dinv = sc.alloc<double>(DINV_STRIDE);
    OP_CHECK(dinv, "ffvd_op_trsm");
    launch_trsm_ext(sc.stream, dA, np, mp, 1, slab.size(), dinv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(slab.data(), dA, slab.size() * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    for (int c = 0; c < m; ++c)
        for (int i = 0; i < n; ++i) X[(size_t)i * m + c] = slab[(size_t)(np + c) * np + i];
    return FFVD_OK;
}

extern "C" int ffvd_op_kernel_pre_cal(int kind, const double *Z, int M, int P, int D, const double *logvariance,
                                      const double *loglengthscales, double jitter, double *Lm_inverse_seq) {
    if (!Z || !logvariance || !Lm_inverse_seq || M < 1 || P < 1 || P > MAXP || D < 1 ||
        (kind == FFVD_KERNEL_SE && !loglengthscales))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_kernel_pre_cal: bad argument");
    OP_BEGIN("ffvd_op_kernel_pre_cal");
    KuuWork w{};
    int rc = build_kuu(sc, kind, Z, M, P, D, logvariance, loglengthscales, jitter, w, nullptr);
    if (rc) return set_error(nullptr, rc, "ffvd_op_kernel_pre_cal: device allocation or upload failed");
    const size_t Mp = w.Mp;
    std::vector<double> host((size_t)D * 2 * Mp * Mp);
    HIP_TRY(hipMemcpyAsync(host.data(), w.Kuu, host.size() * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    if ((rc = check_kuu_info(sc, w, D, "ffvd_op_kernel_pre_cal"))) return rc;
    for (int d = 0; d < D; ++d) {
        const double *Wd = host.data() + (size_t)d * 2 * Mp * Mp + Mp * Mp;
        for (int i = 0; i < M; ++i)
            memcpy(Lm_inverse_seq + ((size_t)d * M + i) * M, Wd + (size_t)i * Mp, (size_t)M * sizeof(double));
    }
    return FFVD_OK;
}

extern "C" int ffvd_op_collapse(int kind, const double *Lm_inverse_seq, const double *X_combine, const double *X,
                                const double *Z, int T, int M, int P, int D, const double *logvariance,
                                const double *loglengthscales, const double *Q, double batch_size, double Y_N,
                                double out3[3]) {
    if (!Lm_inverse_seq || !X_combine || !X || !Z || !logvariance || !Q || !out3 || T < 1 || M < 1 || P < 1 ||
        P > MAXP || D < 1 || (kind == FFVD_KERNEL_SE && !loglengthscales) || !(batch_size > 0.0))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_collapse: bad argument");
    OP_BEGIN("ffvd_op_collapse");
    const int Mp = round_up(M, NB), Tp = round_up(T, STRIP), ng = (Mp + 511) / 512;
    // pad the caller's L^{-T} stack to Mp with an identity block
    std::vector<double> Wp((size_t)D * Mp * Mp, 0.0);
    for (int d = 0; d < D; ++d)
        for (int i = 0; i < Mp; ++i) {
            double *row = Wp.data() + ((size_t)d * Mp + i) * Mp;
            if (i < M) memcpy(row, Lm_inverse_seq + ((size_t)d * M + i) * M, (size_t)M * sizeof(double));
            else row[i] = 1.0;
        }
    std::vector<double> logQ(D);
    for (int d = 0; d < D; ++d) logQ[d] = log(Q[d]);
    double *dW = sc.upload(Wp.data(), Wp.size());
    double *dXc = sc.upload(X_combine, (size_t)T * P);
    double *dX = sc.upload(X, (size_t)(T + 1) * D);
    double *dZ = sc.upload(Z, (size_t)M * P);
    double *dlv = sc.upload(logvariance, D);
    double *dll = sc.alloc<double>((size_t)D * P);
    double *dlq = sc.upload(logQ.data(), D);
    double *variance = sc.alloc<double>(D), *len = sc.alloc<double>((size_t)D * P);
    double *Zs = sc.alloc<double>((size_t)D * Mp * P), *zz = sc.alloc<double>((size_t)D * Mp);
    double *F = sc.alloc<double>((size_t)D * Tp * Mp);
    const size_t hstride = (size_t)(Mp + NB) * Mp;
    double *H = sc.alloc<double>((size_t)D * hstride);
    double *rowsq = sc.alloc<double>((size_t)D * ng * Tp);
    double *hterms = sc.alloc<double>((size_t)D * 2), *cterms = sc.alloc<double>(8);
    int32_t *info = sc.alloc<int32_t>(D);
    double *dinv = sc.alloc<double>(potrf_scratch_doubles(Mp, D));
    if (!dW || !dXc || !dX || !dZ || !dlv || !dll || !dlq || !variance || !len || !Zs || !zz || !F || !H || !rowsq ||
        !hterms || !cterms || !info || !dinv)
        return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_collapse: device allocation or upload failed");
    if (loglengthscales)
        HIP_TRY(hipMemcpyAsync(dll, loglengthscales, (size_t)D * P * sizeof(double), hipMemcpyHostToDevice, sc.stream));
    HIP_TRY(hipMemsetAsync(H, 0, (size_t)D * hstride * sizeof(double), sc.stream));
    HIP_TRY(hipMemsetAsync(info, 0, D * sizeof(int32_t), sc.stream));
    launch_prep_hypers(sc.stream, kind, dZ, M, Mp, P, D, 0, dlv, dll, variance, len, Zs, zz);
    HyperView hv{variance, len, Zs, zz};
    ProjectArgs pa{};
    pa.kind = kind; pa.x = dXc; pa.x_chain_stride = 0; pa.x_ld = P; pa.x_cols = P; pa.ctrl = nullptr;
    pa.T = T; pa.Tp = Tp; pa.C = 0; pa.P = P; pa.M = M; pa.Mp = Mp; pa.Dl = D; pa.d_begin = 0; pa.hv = hv;
    pa.W = dW; pa.w_stride = (size_t)Mp * Mp; pa.U = nullptr; pa.u_ld = 0; pa.b0 = 0; pa.nb = D; pa.F = F;
    pa.rowsq = rowsq; pa.fmean = nullptr; pa.ng = ng;
    launch_project(sc.stream, pa);
    GramArgs ga{};
    ga.mode = GRAM_F; ga.A = F; ga.a_stride = (size_t)Tp * Mp; ga.rows = Tp; ga.with_row = 1;
    ga.X = dX; ga.log_Q = dlq; ga.T = T; ga.D = D; ga.Mp = Mp; ga.Dl = D; ga.d_begin = 0;
    ga.b0 = 0; ga.nb = D; ga.yn_over_batch = Y_N / batch_size; ga.H = H; ga.h_stride = hstride;
    launch_gram(sc.stream, ga);
    launch_potrf_ext(sc.stream, H, Mp, NB, 0, D, hstride, info, dinv);
    launch_h_finish(sc.stream, H, Mp, hstride, D, hterms);
    ReduceArgs ra{};
    ra.kind = kind; ra.branch = FFVD_BRANCH_B; ra.X = dX; ra.ctrl = nullptr; ra.Y = nullptr; ra.log_Q = dlq;
    ra.CC = nullptr; ra.DD = nullptr; ra.log_Rchols = nullptr; ra.variance = variance; ra.T = T; ra.Tp = Tp; ra.D = D;
    ra.C = 0; ra.Ydim = 0; ra.Dl = D; ra.d_begin = 0; ra.S = 1; ra.ng = ng; ra.shared_terms = 0;
    ra.xk = dXc; ra.xk_chain_stride = 0; ra.xk_ld = P; ra.xk_cols = P; ra.rowsq = rowsq; ra.fmean = nullptr;
    ra.chain_terms = cterms;
    double *cpart = sc.alloc<double>(128);
    if (!cpart) return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_collapse: device allocation failed");
    launch_chain_reduce(sc.stream, ra, cpart);
    std::vector<double> ht((size_t)D * 2), ct(8);
    std::vector<int32_t> hinfo(D);
    HIP_TRY(hipMemcpyAsync(ht.data(), hterms, ht.size() * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(ct.data(), cterms, 8 * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(hinfo.data(), info, D * sizeof(int32_t), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    for (int d = 0; d < D; ++d)
        if (hinfo[d]) {
            char msg[160];
            snprintf(msg, sizeof msg, "ffvd_op_collapse: Cholesky of H failed: latent dim %d, pivot %d is not positive", d, hinfo[d] - 1);
            return set_error(nullptr, FFVD_ENOTPD, msg);
        }
    double term1 = 0.0, term2 = 0.0;
    for (int d = 0; d < D; ++d) { term1 += -0.5 * ht[2 * d]; term2 += 0.5 * ht[2 * d + 1]; }
    out3[0] = -term1 / Y_N;
    out3[1] = -term2 / Y_N;
    out3[2] = -ct[2] / Y_N;      // chain_reduce's trace sum: sum_d sum_t -0.5 (Kdiag - |F_t|^2) / Q_d
    return FFVD_OK;
}

extern "C" int ffvd_op_conditional(int kind, const double *Xnew, int N, const double *Z, int M, int P, int D,
                                   const double *logvariance, const double *loglengthscales, const double *f,
                                   double jitter, double *mean, double *var) {
    if (!Xnew || !Z || !logvariance || !f || !mean || !var || N < 0 || M < 1 || P < 1 || P > MAXP || D < 1 ||
        (kind == FFVD_KERNEL_SE && !loglengthscales))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_conditional: bad argument");
    OP_BEGIN("ffvd_op_conditional");
    if (N == 0) return FFVD_OK;
    KuuWork w{};
    int rc = build_kuu(sc, kind, Z, M, P, D, logvariance, loglengthscales, jitter, w, nullptr);
    if (rc) return set_error(nullptr, rc, "ffvd_op_conditional: device allocation or upload failed");
    const int Mp = w.Mp, Tp = round_up(N, STRIP), ng = (Mp + 511) / 512;
    double *dX = sc.upload(Xnew, (size_t)N * P);
    double *dU = sc.upload(f, (size_t)M * D);
    double *rowsq = sc.alloc<double>((size_t)D * ng * Tp), *fmean = sc.alloc<double>((size_t)D * ng * Tp);
    double *dmean = sc.alloc<double>((size_t)N * D), *dvar = sc.alloc<double>((size_t)N * D);
    if (!dX || !dU || !rowsq || !fmean || !dmean || !dvar)
        return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_conditional: device allocation or upload failed");
    HyperView hv{w.variance, w.len, w.Zs, w.zz};
    ProjectArgs pa{};
    pa.kind = kind; pa.x = dX; pa.x_chain_stride = 0; pa.x_ld = P; pa.x_cols = P; pa.ctrl = nullptr;
    pa.T = N; pa.Tp = Tp; pa.C = 0; pa.P = P; pa.M = M; pa.Mp = Mp; pa.Dl = D; pa.d_begin = 0; pa.hv = hv;
    pa.W = w.Kuu + (size_t)Mp * Mp; pa.w_stride = (size_t)2 * Mp * Mp; pa.U = dU; pa.u_ld = D; pa.b0 = 0; pa.nb = D;
    pa.F = nullptr; pa.rowsq = rowsq; pa.fmean = fmean; pa.ng = ng;
    launch_project(sc.stream, pa);
    launch_conditional_finish(sc.stream, kind, dX, N, P, w.variance, rowsq, fmean, ng, Tp, D, dmean, dvar, nullptr);
    HIP_TRY(hipMemcpyAsync(mean, dmean, (size_t)N * D * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(var, dvar, (size_t)N * D * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    return check_kuu_info(sc, w, D, "ffvd_op_conditional");
}

extern "C" int ffvd_op_predict_mean(const double *X_end, int N, int D, const double *CC, const double *DD, int Ydim,
                                    double *out) {
    if (!X_end || !CC || !DD || !out || N < 0 || D < 1 || Ydim < 1)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_predict_mean: bad argument");
    OP_BEGIN("ffvd_op_predict_mean");
    if (N == 0) return FFVD_OK;
    double *dX = sc.upload(X_end, (size_t)N * D), *dC = sc.upload(CC, (size_t)D * Ydim), *dD = sc.upload(DD, Ydim);
    double *dO = sc.alloc<double>((size_t)N * Ydim);
    if (!dX || !dC || !dD || !dO) return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_predict_mean: device allocation or upload failed");
    launch_predict_mean(sc.stream, dX, N, D, dC, dD, Ydim, dO);
    HIP_TRY(hipMemcpyAsync(out, dO, (size_t)N * Ydim * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

extern "C" int ffvd_op_logdensity_norm_diag(int nonvec, const double *y, const double *ymean, const double *Rchols, int N,
                                            int J, double *out) {
    if (!y || !ymean || !Rchols || !out || N < 0 || J < 1)
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_logdensity_norm_diag: bad argument");
    OP_BEGIN("ffvd_op_logdensity_norm_diag");
    if (N == 0) return FFVD_OK;
    const size_t nout = nonvec ? (size_t)N * J : (size_t)N;
    double *dy = sc.upload(y, (size_t)N * J), *dm = sc.upload(ymean, (size_t)N * J), *dR = sc.upload(Rchols, J);
    double *dO = sc.alloc<double>(nout);
    if (!dy || !dm || !dR || !dO) return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_logdensity_norm_diag: device allocation or upload failed");
    launch_logdensity(sc.stream, nonvec ? 1 : 0, dy, dm, dR, N, J, dO);
    HIP_TRY(hipMemcpyAsync(out, dO, nout * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

extern "C" int ffvd_op_get_rand(const double *mean, const double *var, const double *eps, int64_t n, double *out) {
    if (!mean || !var || !eps || !out || n < 0) return set_error(nullptr, FFVD_EINVAL, "ffvd_op_get_rand: bad argument");
    OP_BEGIN("ffvd_op_get_rand");
    if (n == 0) return FFVD_OK;
    double *dm = sc.upload(mean, (size_t)n), *dv = sc.upload(var, (size_t)n), *de = sc.upload(eps, (size_t)n);
    double *dO = sc.alloc<double>((size_t)n);
    if (!dm || !dv || !de || !dO) return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_get_rand: device allocation or upload failed");
    launch_get_rand(sc.stream, dm, dv, de, (size_t)n, dO);
    HIP_TRY(hipMemcpyAsync(out, dO, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

// shared by ffvd_op_collapse_u_mean / ffvd_op_conditional_precalc: pad a caller-supplied stack of D M x M matrices
// (M a multiple of the block size: nothing to pad -- the callers upload the caller's array itself, pad_needed)
static bool pad_needed(int M, int Mp) { return M != Mp; }
static std::vector<double> pad_stack(const double *src, int D, int M, int Mp) {
    std::vector<double> out((size_t)D * Mp * Mp, 0.0);
    for (int d = 0; d < D; ++d)
        for (int i = 0; i < Mp; ++i) {
            double *row = out.data() + ((size_t)d * Mp + i) * Mp;
            if (i < M) memcpy(row, src + ((size_t)d * M + i) * M, (size_t)M * sizeof(double));
            else row[i] = 1.0;
        }
    return out;
}

// upload a stack of D M x M matrices padded to Mp (`keep` owns the padded host copy until the call returns; none when M == Mp)
static double *upload_stack(Scratch &sc, const double *src, int D, int M, int Mp, std::vector<double> &keep) {
    if (!pad_needed(M, Mp)) return sc.upload(src, (size_t)D * M * M);
    keep = pad_stack(src, D, M, Mp);
    return sc.upload(keep.data(), keep.size());
}

extern "C" int ffvd_op_collapse_u_mean(int kind, const double *Lm_inverse_seq, const double *X_combine, const double *X,
                                       const double *Z, int T, int M, int P, int D, const double *logvariance,
                                       const double *loglengthscales, const double *Q, double *U_mean,
                                       double *H_inv_sqrt) {
    if (!Lm_inverse_seq || !X_combine || !X || !Z || !logvariance || !Q || !U_mean || !H_inv_sqrt || T < 1 || M < 1 ||
        P < 1 || P > MAXP || D < 1 || (kind == FFVD_KERNEL_SE && !loglengthscales))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_collapse_u_mean: bad argument");
    OP_BEGIN("ffvd_op_collapse_u_mean");
    const int Mp = round_up(M, NB), Tp = round_up(T, STRIP), ng = (Mp + 511) / 512;
    std::vector<double> Wp;
    std::vector<double> logQ(D);
    for (int d = 0; d < D; ++d) logQ[d] = log(Q[d]);
    // slab per dim: rows [0,Mp) H, rows [Mp,2Mp) identity -> L_H^-T, row 2Mp carries b -> L_H^-1 b
    const size_t hstride = (size_t)(2 * Mp + NB) * Mp;
    std::vector<double> Hinit((size_t)D * hstride, 0.0);
    for (int d = 0; d < D; ++d)
        for (int i = 0; i < Mp; ++i) Hinit[(size_t)d * hstride + (size_t)(Mp + i) * Mp + i] = 1.0;
    double *dW = upload_stack(sc, Lm_inverse_seq, D, M, Mp, Wp);
    double *dXc = sc.upload(X_combine, (size_t)T * P), *dX = sc.upload(X, (size_t)(T + 1) * D);
    double *dZ = sc.upload(Z, (size_t)M * P), *dlv = sc.upload(logvariance, D), *dlq = sc.upload(logQ.data(), D);
    double *dll = sc.alloc<double>((size_t)D * P);
    double *variance = sc.alloc<double>(D), *len = sc.alloc<double>((size_t)D * P);
    double *Zs = sc.alloc<double>((size_t)D * Mp * P), *zz = sc.alloc<double>((size_t)D * Mp);
    double *F = sc.alloc<double>((size_t)D * Tp * Mp), *rowsq = sc.alloc<double>((size_t)D * ng * Tp);
    double *H = sc.upload(Hinit.data(), Hinit.size());
    double *dU = sc.alloc<double>((size_t)M * D);
    int32_t *info = sc.alloc<int32_t>(D);
    double *dinv = sc.alloc<double>(potrf_scratch_doubles(Mp, D));
    if (!dW || !dXc || !dX || !dZ || !dlv || !dlq || !dll || !variance || !len || !Zs || !zz || !F || !rowsq || !H || !dU || !info || !dinv)
        return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_collapse_u_mean: device allocation or upload failed");
    if (loglengthscales)
        HIP_TRY(hipMemcpyAsync(dll, loglengthscales, (size_t)D * P * sizeof(double), hipMemcpyHostToDevice, sc.stream));
    HIP_TRY(hipMemsetAsync(info, 0, D * sizeof(int32_t), sc.stream));
    launch_prep_hypers(sc.stream, kind, dZ, M, Mp, P, D, 0, dlv, dll, variance, len, Zs, zz);
    HyperView hv{variance, len, Zs, zz};
    ProjectArgs pa{};
    pa.kind = kind; pa.x = dXc; pa.x_chain_stride = 0; pa.x_ld = P; pa.x_cols = P; pa.ctrl = nullptr;
    pa.T = T; pa.Tp = Tp; pa.C = 0; pa.P = P; pa.M = M; pa.Mp = Mp; pa.Dl = D; pa.d_begin = 0; pa.hv = hv;
    pa.W = dW; pa.w_stride = (size_t)Mp * Mp; pa.U = nullptr; pa.u_ld = 0; pa.b0 = 0; pa.nb = D; pa.F = F;
    pa.rowsq = rowsq; pa.fmean = nullptr; pa.ng = ng;
    launch_project(sc.stream, pa);
    GramArgs ga{};
    ga.mode = GRAM_F; ga.A = F; ga.a_stride = (size_t)Tp * Mp; ga.rows = Tp; ga.with_row = 1; ga.brow = 2 * Mp;
    ga.X = dX; ga.log_Q = dlq; ga.T = T; ga.D = D; ga.Mp = Mp; ga.Dl = D; ga.d_begin = 0;
    ga.b0 = 0; ga.nb = D; ga.yn_over_batch = 1.0; ga.H = H; ga.h_stride = hstride;     // :215,:217 (no batch rescaling)
    launch_gram(sc.stream, ga);
    launch_potrf_ext(sc.stream, H, Mp, Mp + NB, Mp, D, hstride, info, dinv);
    // U_mean[:, d] = H^-1 b = L_H^-T (L_H^-1 b)   (tf.linalg.solve, :219)
    launch_matvec(sc.stream, H + (size_t)Mp * Mp, hstride, H + (size_t)2 * Mp * Mp, hstride, Mp, dU, D, 1, M, D);
    std::vector<double> hH((size_t)D * hstride);
    std::vector<int32_t> hinfo(D);
    HIP_TRY(hipMemcpyAsync(hH.data(), H, hH.size() * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(U_mean, dU, (size_t)M * D * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(hinfo.data(), info, D * sizeof(int32_t), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    for (int d = 0; d < D; ++d)
        if (hinfo[d]) {
            char msg[160];
            snprintf(msg, sizeof msg, "ffvd_op_collapse_u_mean: Cholesky of H failed: latent dim %d, pivot %d is not positive", d, hinfo[d] - 1);
            return set_error(nullptr, FFVD_ENOTPD, msg);
        }
    for (int d = 0; d < D; ++d)          // Lm_inverse_dd_seq = L_H^-T (:222)
        for (int i = 0; i < M; ++i)
            memcpy(H_inv_sqrt + ((size_t)d * M + i) * M, hH.data() + (size_t)d * hstride + (size_t)(Mp + i) * Mp,
                   (size_t)M * sizeof(double));
    return FFVD_OK;
}

extern "C" int ffvd_op_conditional_precalc(int kind, const double *Lm_inverse_seq, const double *Xnew, int N,
                                           const double *Z, int M, int P, int D, const double *logvariance,
                                           const double *loglengthscales, const double *f, const double *q_sqrt,
                                           double *mean, double *var) {
    if (!Lm_inverse_seq || !Xnew || !Z || !logvariance || !f || !mean || !var || N < 0 || M < 1 || M > 2048 || P < 1 ||
        P > MAXP || D < 1 || (kind == FFVD_KERNEL_SE && !loglengthscales))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_conditional_precalc: bad argument");
    OP_BEGIN("ffvd_op_conditional_precalc");
    if (N == 0) return FFVD_OK;
    const int Mp = round_up(M, NB), Tp = round_up(N, STRIP), ng = (Mp + 511) / 512;
    std::vector<double> Wp;
    double *dW = upload_stack(sc, Lm_inverse_seq, D, M, Mp, Wp);
    double *dX = sc.upload(Xnew, (size_t)N * P), *dZ = sc.upload(Z, (size_t)M * P), *dU = sc.upload(f, (size_t)M * D);
    double *dlv = sc.upload(logvariance, D), *dll = sc.alloc<double>((size_t)D * P);
    double *variance = sc.alloc<double>(D), *len = sc.alloc<double>((size_t)D * P);
    double *Zs = sc.alloc<double>((size_t)D * Mp * P), *zz = sc.alloc<double>((size_t)D * Mp);
    double *F = sc.alloc<double>((size_t)D * Tp * Mp);
    double *rowsq = sc.alloc<double>((size_t)D * ng * Tp), *fmean = sc.alloc<double>((size_t)D * ng * Tp);
    double *dmean = sc.alloc<double>((size_t)N * D), *dvar = sc.alloc<double>((size_t)N * D);
    double *dQs = q_sqrt ? sc.upload(q_sqrt, (size_t)M * M) : nullptr;     // slice d = 0 only (the reference quirk)
    double *extra = q_sqrt ? sc.alloc<double>((size_t)D * Tp) : nullptr;
    if (!dW || !dX || !dZ || !dU || !dlv || !dll || !variance || !len || !Zs || !zz || !F || !rowsq || !fmean || !dmean ||
        !dvar || (q_sqrt && (!dQs || !extra)))
        return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_conditional_precalc: device allocation or upload failed");
    if (loglengthscales)
        HIP_TRY(hipMemcpyAsync(dll, loglengthscales, (size_t)D * P * sizeof(double), hipMemcpyHostToDevice, sc.stream));
    launch_prep_hypers(sc.stream, kind, dZ, M, Mp, P, D, 0, dlv, dll, variance, len, Zs, zz);
    HyperView hv{variance, len, Zs, zz};
    ProjectArgs pa{};
    pa.kind = kind; pa.x = dX; pa.x_chain_stride = 0; pa.x_ld = P; pa.x_cols = P; pa.ctrl = nullptr;
    pa.T = N; pa.Tp = Tp; pa.C = 0; pa.P = P; pa.M = M; pa.Mp = Mp; pa.Dl = D; pa.d_begin = 0; pa.hv = hv;
    pa.W = dW; pa.w_stride = (size_t)Mp * Mp; pa.U = dU; pa.u_ld = D; pa.b0 = 0; pa.nb = D;
    pa.F = q_sqrt ? F : nullptr; pa.rowsq = rowsq; pa.fmean = fmean; pa.ng = ng;
    launch_project(sc.stream, pa);                                           // A^T = K_fu L^-T (:349), mean (:365), sum A^2 (:356)
    if (q_sqrt) launch_qsqrt_inflation(sc.stream, F, (size_t)Tp * Mp, Tp, Mp, M, dQs, extra, N, D);
    launch_conditional_finish(sc.stream, kind, dX, N, P, variance, rowsq, fmean, ng, Tp, D, dmean, dvar, extra);
    HIP_TRY(hipMemcpyAsync(mean, dmean, (size_t)N * D * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(var, dvar, (size_t)N * D * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

extern "C" int ffvd_op_adam_step(double *theta, const double *grad, double *m, double *v, int64_t n, double lr,
                                 double beta1, double beta2, double eps, int64_t t) {
    if (!theta || !grad || !m || !v || n < 0 || t < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_adam_step: bad argument");
    OP_BEGIN("ffvd_op_adam_step");
    if (n == 0) return FFVD_OK;
    double *dth = sc.upload(theta, n), *dm = sc.upload(m, n), *dv = sc.upload(v, n);
    const double *dg = sc.upload(grad, n);
    if (!dth || !dm || !dv || !dg) return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_adam_step: device allocation or upload failed");
    OptTable tab{};
    tab.count = 1;
    tab.t[0].theta = dth; tab.t[0].grad = dg; tab.t[0].s0 = dm; tab.t[0].s1 = dv; tab.t[0].n = n;
    const double lr_t = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
    launch_adam(sc.stream, tab, lr_t, beta1, beta2, eps);
    HIP_TRY(hipMemcpyAsync(theta, dth, n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(m, dm, n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(v, dv, n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

extern "C" int ffvd_op_sghmc_step(double *theta, const double *grad, double *xi, double *g, double *g2, double *p,
                                  const double *noise, int64_t n, double epsilon, double mdecay, double X_N, int burn_in) {
    if (!theta || !grad || !xi || !g || !g2 || !p || !noise || n < 0 || !(X_N > 0.0))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_sghmc_step: bad argument");
    OP_BEGIN("ffvd_op_sghmc_step");
    if (n == 0) return FFVD_OK;
    double *dth = sc.upload(theta, n), *dxi = sc.upload(xi, n), *dg_ = sc.upload(g, n), *dg2 = sc.upload(g2, n);
    double *dp = sc.upload(p, n);
    const double *dgrad = sc.upload(grad, n), *dnoise = sc.upload(noise, n);
    if (!dth || !dxi || !dg_ || !dg2 || !dp || !dgrad || !dnoise)
        return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_sghmc_step: device allocation or upload failed");
    OptTable tab{};
    tab.count = 1;
    OptTensor &t = tab.t[0];
    t.theta = dth; t.grad = dgrad; t.s0 = dxi; t.s1 = dg_; t.s2 = dg2; t.s3 = dp; t.noise = dnoise; t.n = n;
    launch_sghmc(sc.stream, tab, epsilon, mdecay, X_N, burn_in);
    HIP_TRY(hipMemcpyAsync(theta, dth, n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(p, dp, n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    if (burn_in) {
        HIP_TRY(hipMemcpyAsync(xi, dxi, n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
        HIP_TRY(hipMemcpyAsync(g, dg_, n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
        HIP_TRY(hipMemcpyAsync(g2, dg2, n * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    }
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

extern "C" int ffvd_op_rollout_fallbacks(void) { return g_rollout_fallbacks; }

extern "C" int ffvd_op_rollout(int kind, const double *Lm_inverse_seq, const double *Z, int M, int P, int D,
                               const double *logvariance, const double *loglengthscales, const double *f,
                               const double *q_sqrt, const double *x_last, int R, const double *ctrl, int C, int steps,
                               const double *log_Q, const double *eps, double *predict_x, double *predict_var) {
    if (!Lm_inverse_seq || !Z || !logvariance || !f || !x_last || !log_Q || !eps || !predict_x || !predict_var || R < 1 ||
        steps < 0 || M < 1 || M > 2048 || D < 1 || C < 0 || P != D + C || P > MAXP || (C > 0 && !ctrl) ||
        (kind == FFVD_KERNEL_SE && !loglengthscales))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_rollout: bad argument");
    OP_BEGIN("ffvd_op_rollout");
    if (steps == 0) return FFVD_OK;
    // K_fu rows once per step, then the triangular projection GEMM with fvar / fmean in its epilogue (one workgroup
    // per 128-column tile and dim): 16 workgroups instead of the 4 of the fused projection kernel, whose single
    // workgroup per dim made a step compute-bound on one CU
    const int Mp = round_up(M, NB), Tp = round_up(R, STRIP), ng = (Mp + 127) / 128;
    std::vector<double> Wp;
    std::vector<double> xc0((size_t)R * P);
    for (int r = 0; r < R; ++r) {
        for (int d = 0; d < D; ++d) xc0[(size_t)r * P + d] = x_last[d];              // x_t = X[-1] (:226), every rollout
        for (int c = 0; c < C; ++c) xc0[(size_t)r * P + D + c] = ctrl[c];            // control row of step 0 (:293)
    }
    double *dW = upload_stack(sc, Lm_inverse_seq, D, M, Mp, Wp);
    double *dxc = sc.upload(xc0.data(), xc0.size()), *dZ = sc.upload(Z, (size_t)M * P), *dU = sc.upload(f, (size_t)M * D);
    double *dxc2 = sc.alloc<double>(xc0.size());          // the rows of step t + 1 (the step kernel reads one buffer and writes the other)
    double *dlv = sc.upload(logvariance, D), *dll = sc.alloc<double>((size_t)D * P), *dlq = sc.upload(log_Q, D);
    double *deps = sc.upload(eps, (size_t)steps * R * D);
    double *dctrl = C ? sc.upload(ctrl, (size_t)steps * C) : nullptr;
    double *variance = sc.alloc<double>(D), *len = sc.alloc<double>((size_t)D * P);
    double *Zs = sc.alloc<double>((size_t)D * Mp * P), *zz = sc.alloc<double>((size_t)D * Mp);
    double *F = q_sqrt ? sc.alloc<double>((size_t)D * Tp * Mp) : nullptr;
    double *Kf = sc.alloc<double>((size_t)D * Tp * Mp), *ucol = sc.alloc<double>((size_t)D * Mp);
    // few rows per step (R rollouts): the skinny product (kernels.hip) with one partial sum per 16-column slab replaces the
    // 128 x 128-tile projection GEMM (50 -> 7 us per step) and the per-row q_sqrt kernel (37 us)
    const bool skinny = R <= 512;
    const int ngs = skinny ? Mp / 16 : ng;
    double *rowsq = sc.alloc<double>((size_t)D * ngs * Tp), *fmean = sc.alloc<double>((size_t)D * ngs * Tp);
    double *dmean = sc.alloc<double>((size_t)R * D), *dvar = sc.alloc<double>((size_t)R * D);
    double *dpx = sc.alloc<double>((size_t)R * steps * D), *dpv = sc.alloc<double>((size_t)R * steps * D);
    std::vector<double> Qp;
    // W = L^-T is upper triangular; when q_sqrt (slice 0) is too -- the reference hands over L_H^-T -- so is W q_sqrt, and the second
    // right-hand side of a step's product stops at a slab's last row like the first (exact zeros are skipped: half of its k range)
    int q_upper = 0;
    if (q_sqrt && skinny) {
        q_upper = 1;
        for (int i = 1; i < M && q_upper; ++i)
            for (int j = 0; j < i; ++j)
                if (q_sqrt[(size_t)i * M + j] != 0.0) { q_upper = 0; break; }
    }
    static const bool no_upper2 = getenv("FFVD_NO_QSQRT_UPPER") != nullptr;       // A/B switch (read once)
    if (no_upper2) q_upper = 0;
    double *dQs = q_sqrt ? (skinny ? upload_stack(sc, q_sqrt, 1, M, Mp, Qp) /* (F is zero in the padded columns) */ : sc.upload(q_sqrt, (size_t)M * M)) : nullptr;   // slice d = 0 only
    double *extra = q_sqrt ? sc.alloc<double>((size_t)D * (skinny ? ngs : 1) * Tp) : nullptr;
    // skinny path: W q_sqrt once per call (D products of M^3), so that the inflation term is a second right-hand side of the
    // step's one product instead of a dependent product on F
    double *dWQ = (q_sqrt && skinny) ? sc.alloc<double>((size_t)D * Mp * Mp) : nullptr;
    if (!dW || !dxc || !dxc2 || !dZ || !dU || !dlv || !dll || !dlq || !deps || (C && !dctrl) || !variance || !len || !Zs || !zz ||
        !rowsq || !fmean || !dmean || !dvar || !dpx || !dpv || !Kf || !ucol || (q_sqrt && (!dQs || !extra || !F)) || (q_sqrt && skinny && !dWQ))
        return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_rollout: device allocation or upload failed");
    if (loglengthscales)
        HIP_TRY(hipMemcpyAsync(dll, loglengthscales, (size_t)D * P * sizeof(double), hipMemcpyHostToDevice, sc.stream));
    launch_prep_hypers(sc.stream, kind, dZ, M, Mp, P, D, 0, dlv, dll, variance, len, Zs, zz);
    HyperView hv{variance, len, Zs, zz};
    ProjectArgs pa{};
    pa.kind = kind; pa.x = dxc; pa.x_chain_stride = 0; pa.x_ld = P; pa.x_cols = P; pa.ctrl = nullptr;
    pa.T = R; pa.Tp = Tp; pa.C = 0; pa.P = P; pa.M = M; pa.Mp = Mp; pa.Dl = D; pa.d_begin = 0; pa.hv = hv;
    pa.W = dW; pa.w_stride = (size_t)Mp * Mp; pa.U = dU; pa.u_ld = D; pa.b0 = 0; pa.nb = D;
    pa.F = Kf; pa.rowsq = rowsq; pa.fmean = fmean; pa.ng = ng;
    launch_ucols(sc.stream, dU, M, Mp, D, 0, D, ucol);
    ProjGemmArgs pg{};
    pg.Kf = Kf; pg.kf_stride = (size_t)Tp * Mp; pg.W = dW; pg.w_stride = (size_t)Mp * Mp; pg.F = F; pg.f_stride = (size_t)Tp * Mp;
    pg.rowsq = rowsq; pg.fmean = fmean; pg.u = ucol; pg.u_stride = Mp; pg.Tp = Tp; pg.Mp = Mp; pg.Dl = D; pg.b0 = 0; pg.nb = D;
    // the whole loop is enqueued at once: steps x (K_fu rows, projection, [q_sqrt inflation], conditional, update)
    if (dWQ)
        launch_skinny_gemm(sc.stream, dW, (size_t)Mp * Mp, Mp, dQs, 0, Mp, 0, Mp, Mp, Mp, D, Mp, dWQ, (size_t)Mp * Mp, Mp, nullptr, 0,
                           nullptr, nullptr);
    double *xbuf[2] = {dxc, dxc2};                                           // input rows of step t in xbuf[t & 1]
    auto step_launches = [&]() {                                              // three dependent launches per step
        for (int t = 0; t < steps; ++t) {
            pa.x = xbuf[t & 1];
            if (skinny) {
                launch_kfu_build_t(sc.stream, pa, Tp);                               // K(x_t, Z) per dim, m-major (coalesced operand loads)
                // conditional_after_kernel_precalculation (:300) and, in the same launch, sum_j (F q_sqrt)_j^2 = |K (W q_sqrt)|^2 (:371-380)
                launch_skinny_gemm(sc.stream, Kf, (size_t)Tp * Mp, Tp, dW, (size_t)Mp * Mp, Mp, 1, R, Mp, Mp, D, Tp,
                                   nullptr, 0, 0, ucol, Mp, rowsq, fmean, q_sqrt ? dWQ : nullptr, (size_t)Mp * Mp, Mp, Mp, extra, 1, q_upper);
            } else {
                launch_kfu_build(sc.stream, pa);                                     // K(x_t, Z) per dim
                launch_proj_gemm(sc.stream, pg);
                if (q_sqrt) launch_qsqrt_inflation(sc.stream, F, (size_t)Tp * Mp, Tp, Mp, M, dQs, extra, R, D);
            }
            // conditional epilogue + x <- x + f_mu + eps sqrt(f_var + Q) in one launch (three dependent launches per step instead of five)
            launch_rollout_finish_update(sc.stream, kind, variance, rowsq, fmean, skinny ? ngs : ng, Tp, extra, skinny ? ngs : 1, dlq,
                                         deps + (size_t)t * R * D, (C && t + 1 < steps) ? dctrl + (size_t)(t + 1) * C : nullptr, R, D, C, t,
                                         steps, xbuf[t & 1], xbuf[(t + 1) & 1], dpx, dpv);
        }
    };
    // Round 4: the whole loop as ONE persistent launch (loops.hip; the same kernel bodies: bit-identical) -- first
    // with a grid-wide barrier between the phases (64.7 against 32.7 us per step at 32 rollouts: 160-512 workgroups on one counter),
    // then with one role per workgroup and per-unit counters (40.3 against 36.0; 93 against 57 at 100 rollouts).  MEASURED SLOWER both
    // times: ~170 agent-scope releases / acquires per step cost more than the three kernel boundaries they replace -- so it is opt-in:
    // FFVD_STEP_LOOP=1.
    const char *nsl = getenv("FFVD_STEP_LOOP");
    const int loop_mode = (nsl && *nsl) ? atoi(nsl) : -1;
    // Third form (loops.hip, rollout_resident_kernel; FFVD_STEP_LOOP=2): L^-T and W q_sqrt stay in LDS for the whole loop, three hand-offs
    // per step.  Measured as the slope between 200 and 800 steps at M = 512, D = 4 (tools/rollout_modes.py): 16.6-19.2 us per step at 16
    // rollouts, 19.3-23.9 at 32, 27-50 at 64 against 19.6-20.8 / 20.1-20.6 / 20.5-25.7 for the launches -- a hand-off among 32 workgroups
    // costs 4-5 us here (release 2, wait + acquire 2.5-3: FFVD_RR_STAMPS=1), three of them are what three kernel boundaries cost.
    // (Second measurement, with the hand-offs rebuilt without cache maintenance -- write-through stores and sc1 loads instead of fences, as in
    //  the Gram kernel's tail exchange: 10.1-11.8 us per step at 16 rollouts, 13.6-17.9 at 32, 20.5-27 at 64.  The default up to 32 rollouts.)
    // (Third measurement, the products' MFMA section branch-free per chunk: 10.4-12.5 / 12.5-15.4 / 17.9-22.5 us per step.  The default
    //  wherever it applies: up to 64 rollouts, M <= 512, 8 latent dims.)
    // (Round 5: with K^T operands, the XCD-mapped grid and long / short slabs paired per CU the per-step launches take 15.4-15.6 us
    //  (18.4-21.8 with q_sqrt) from 40 to 64 rollouts, the resident loop 16.6-18.6 (20.3-23.4); at 32: 15.0 / 16.6 against 12.7 / 14.9,
    //  profiles/r05_rollout_modes.txt -- the resident loop is the default up to 32 rollouts, FFVD_STEP_LOOP=2 asks for it up to 64.)
    const bool resident = skinny && rollout_resident_ok(R, D, P, Mp) && (loop_mode == 2 || (loop_mode < 0 && R <= 32));
    const bool use_loop = skinny && loop_mode == 1;
    bool resident_done = false;
    if (resident) {
        const int RT = (R + 15) / 16, RP = 16 * RT, NS = Mp / 16;
        int32_t *words = sc.alloc<int32_t>(rollout_resident_words());
        double *Kt = sc.alloc<double>((size_t)D * Mp * RP), *part = sc.alloc<double>((size_t)D * NS * RP * 4);
        double *xb = sc.alloc<double>((size_t)2 * RP * D), *dxl = sc.upload(x_last, D);
        OP_CHECK(words && Kt && part && xb && dxl, "ffvd_op_rollout");
        HIP_TRY(hipMemsetAsync(words, 0, (size_t)rollout_resident_words() * sizeof(int32_t), sc.stream));
        RolloutResidentArgs ra{};
        ra.kind = kind; ra.R = R; ra.RT = RT; ra.D = D; ra.C = C; ra.P = P; ra.M = M; ra.Mp = Mp; ra.steps = steps; ra.NS = NS;
        ra.hv = hv; ra.W = dW; ra.w_stride = (size_t)Mp * Mp; ra.WQ = q_sqrt ? dWQ : nullptr; ra.wq_upper = q_sqrt ? q_upper : 0; ra.ucol = ucol;
        ra.log_Q = dlq; ra.eps = deps; ra.ctrl = dctrl; ra.x_last = dxl; ra.Kt = Kt; ra.part = part; ra.xbuf = xb;
        ra.predict_x = dpx; ra.predict_var = dpv; ra.words = words; ra.abort_w = words + 1;
        long long *dst = nullptr;
        if (getenv("FFVD_RR_STAMPS")) { dst = sc.alloc<long long>(32); if (dst) HIP_TRY(hipMemsetAsync(dst, 0, 32 * sizeof(long long), sc.stream)); }
        ra.stamps = dst;
        ra.test_stall = getenv("FFVD_RR_TEST_STALL") ? 1 : 0;
        const int lrc = launch_rollout_resident(sc.stream, ra);
        if (lrc == 0) {
            int32_t hw[4] = {0, 0, 0, 0};
            HIP_TRY(hipMemcpyAsync(hw, words, sizeof hw, hipMemcpyDeviceToHost, sc.stream));
            HIP_TRY(hipStreamSynchronize(sc.stream));
            resident_done = hw[1] == 0;          // (a wait gave up: the per-step launches below, from the initial rows)
            if (!resident_done) {
                // the two forms agree to 1e-9, not bit for bit, so a fallback decided by run-time contention must not be
                // silent -- counted (ffvd_op_rollout_fallbacks) and left as a warning for ffvd_last_error(NULL)
                ++g_rollout_fallbacks;
                g_last_error = "warning: ffvd_op_rollout: the resident-operand loop gave up on a bounded wait (its workgroups were not all "
                               "resident: another tenant on the GPU?); the call was completed by the per-step launches, whose results "
                               "agree to 1e-9 but are not bit-identical (FFVD_STEP_LOOP=0 selects them always)";
            }
            if (dst && steps > 10) {             // tools: where step 10 spent its time (us since its start), slab 0 (the updater) and the last slab of dim 0
                long long hs[32];
                HIP_TRY(hipMemcpy(hs, dst, sizeof hs, hipMemcpyDeviceToHost));
                static const char *nm[12] = {"start", "x_t seen", "inputs staged", "K stored", "K counted", "all K seen", "products", "partials stored",
                                             "partials counted", "all partials seen", "updated", "x counted"};
                for (int w = 0; w < 2; ++w) {
                    fprintf(stderr, "rollout_resident R=%d q=%d slab %s:", R, q_sqrt ? 1 : 0, w ? "last" : "0");
                    for (int i = 1; i < 12; ++i) if (hs[16 * w + i]) fprintf(stderr, "  %s %.2f", nm[i], (double)(hs[16 * w + i] - hs[16 * w]) / 100.0);
                    fprintf(stderr, "\n");
                }
            }
        } else (void)hipGetLastError();
    }
    if (resident_done) {
    } else if (use_loop) {
        const int nwords = loop_words(D);
        int32_t *words = sc.alloc<int32_t>(nwords);
        OP_CHECK(words, "ffvd_op_rollout");
        HIP_TRY(hipMemsetAsync(words, 0, (size_t)nwords * sizeof(int32_t), sc.stream));
        RolloutLoopArgs la{};
        la.pa = pa;
        la.sk = SkinnyArgs{Kf, (size_t)Tp * Mp, Mp, dW, (size_t)Mp * Mp, Mp, 1, R, Mp, Mp, D, Tp, nullptr, 0, 0, ucol, (size_t)Mp, rowsq, fmean,
                           q_sqrt ? dWQ : nullptr, (size_t)Mp * Mp, Mp, q_sqrt ? Mp : 0, extra, 0, q_sqrt ? q_upper : 0};
        la.f = FinishIn{kind, D + C, ngs, Tp, D, ngs, variance, rowsq, fmean, extra};
        la.log_Q = dlq; la.eps = deps; la.ctrl = dctrl; la.R = R; la.C = C; la.steps = steps;
        la.xbuf0 = xbuf[0]; la.xbuf1 = xbuf[1]; la.predict_x = dpx; la.predict_var = dpv;
        la.bar = reinterpret_cast<unsigned *>(words); la.abort_w = words + 1;
        launch_rollout_loop(sc.stream, la);
        int32_t hw[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(hw, words, sizeof hw, hipMemcpyDeviceToHost, sc.stream));
        HIP_TRY(hipStreamSynchronize(sc.stream));
        if (hw[1] != 0) {           // a wait gave up (the grid was not resident at once): the per-step launches, from the initial rows
            HIP_TRY(hipMemcpyAsync(dxc, xc0.data(), xc0.size() * sizeof(double), hipMemcpyHostToDevice, sc.stream));
            step_launches();
        }
    } else step_launches();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(predict_x, dpx, (size_t)R * steps * D * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipMemcpyAsync(predict_var, dpv, (size_t)R * steps * D * sizeof(double), hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
    return FFVD_OK;
}

extern "C" int ffvd_op_pg_sweep(int kind, const double *Lm_inverse_seq, const double *Z, int M, int P, int D,
                                const double *logvariance, const double *loglengthscales, const double *U,
                                const double *X_ref, int X_N, const double *Y, int Ydim, const double *ctrl, int C,
                                const double *CC, const double *DD, const double *Rchols, const double *log_Q, int n_free,
                                const double *x0, const double *eps, const double *unif, double *particles, int32_t *idx) {
    if (!Lm_inverse_seq || !Z || !logvariance || !U || !X_ref || !Y || !CC || !DD || !Rchols || !log_Q || !x0 || !eps ||
        !unif || !particles || !idx || n_free < 1 || n_free + 1 > 1024 || X_N < 1 || M < 1 || M > 2048 || D < 1 || C < 0 ||
        P != D + C || P > MAXP || Ydim < 1 || Ydim > 8 || (C > 0 && !ctrl) || (kind == FFVD_KERNEL_SE && !loglengthscales))
        return set_error(nullptr, FFVD_EINVAL, "ffvd_op_pg_sweep: bad argument");
    for (int j = 0; j < Ydim; ++j)
        if (!(Rchols[(size_t)j * Ydim + j] > 0.0)) return set_error(nullptr, FFVD_EINVAL, "ffvd_op_pg_sweep: Rchols diagonal must be positive");
    OP_BEGIN("ffvd_op_pg_sweep");
    const auto t_begin = std::chrono::steady_clock::now();
    const int R = n_free, steps = X_N - 1;
    memcpy(particles, x0, (size_t)R * D * sizeof(double));                                   // particles[0] (:87)
    if (steps == 0) return FFVD_OK;
    const int Mp = round_up(M, NB), Tp = round_up(R, STRIP), ng = (Mp + 127) / 128;
    std::vector<double> Wp;
    std::vector<double> xc0((size_t)R * P);
    for (int r = 0; r < R; ++r) {
        for (int d = 0; d < D; ++d) xc0[(size_t)r * P + d] = x0[(size_t)r * D + d];
        for (int c = 0; c < C; ++c) xc0[(size_t)r * P + D + c] = ctrl[c];                     // control row of step 0 (:93)
    }
    const bool pg_timing = getenv("FFVD_PG_TIMING") != nullptr;
    auto lap = [&](const char *what) {
        if (pg_timing) fprintf(stderr, "ffvd_op_pg_sweep:   %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    lap("begin");
    double *dW = upload_stack(sc, Lm_inverse_seq, D, M, Mp, Wp);
    lap("L^-T uploaded");
    double *dxc = sc.upload(xc0.data(), xc0.size()), *dZ = sc.upload(Z, (size_t)M * P), *dU = sc.upload(U, (size_t)M * D);
    double *dlv = sc.upload(logvariance, D), *dll = sc.alloc<double>((size_t)D * P), *dlq = sc.upload(log_Q, D);
    double *deps = sc.upload(eps, (size_t)steps * R * D), *dun = sc.upload(unif, (size_t)steps * R);
    lap("draws uploaded");
    double *dctrl = C ? sc.upload(ctrl, (size_t)steps * C) : nullptr;
    double *dXr = sc.upload(X_ref, (size_t)X_N * D), *dY = sc.upload(Y, (size_t)steps * Ydim);
    double *dCC = sc.upload(CC, (size_t)D * Ydim), *dDD = sc.upload(DD, Ydim), *dR = sc.upload(Rchols, (size_t)Ydim * Ydim);
    double *variance = sc.alloc<double>(D), *len = sc.alloc<double>((size_t)D * P);
    double *Zs = sc.alloc<double>((size_t)D * Mp * P), *zz = sc.alloc<double>((size_t)D * Mp);
    double *Kf = sc.alloc<double>((size_t)D * Tp * Mp), *ucol = sc.alloc<double>((size_t)D * Mp);
    const bool skinny = R <= 512;                       // see ffvd_op_rollout
    const int ngs = skinny ? Mp / 16 : ng;
    double *rowsq = sc.alloc<double>((size_t)D * ngs * Tp), *fmean = sc.alloc<double>((size_t)D * ngs * Tp);
    double *dmean = sc.alloc<double>((size_t)R * D), *dvar = sc.alloc<double>((size_t)R * D);
    double *cand = sc.alloc<double>((size_t)(R + 1) * D);
    double *dparts = sc.alloc<double>((size_t)steps * R * D);
    int32_t *didx = sc.alloc<int32_t>((size_t)steps * R);
    lap("temporaries allocated");
    if (!dW || !dxc || !dZ || !dU || !dlv || !dll || !dlq || !deps || !dun || (C && !dctrl) || !dXr || !dY || !dCC || !dDD ||
        !dR || !variance || !len || !Zs || !zz || !Kf || !ucol || !rowsq || !fmean || !dmean || !dvar || !cand || !dparts || !didx)
        return set_error(nullptr, FFVD_ENOMEM, "ffvd_op_pg_sweep: device allocation or upload failed");
    if (loglengthscales)
        HIP_TRY(hipMemcpyAsync(dll, loglengthscales, (size_t)D * P * sizeof(double), hipMemcpyHostToDevice, sc.stream));
    launch_prep_hypers(sc.stream, kind, dZ, M, Mp, P, D, 0, dlv, dll, variance, len, Zs, zz);
    HyperView hv{variance, len, Zs, zz};
    ProjectArgs pa{};
    pa.kind = kind; pa.x = dxc; pa.x_chain_stride = 0; pa.x_ld = P; pa.x_cols = P; pa.ctrl = nullptr;
    pa.T = R; pa.Tp = Tp; pa.C = 0; pa.P = P; pa.M = M; pa.Mp = Mp; pa.Dl = D; pa.d_begin = 0; pa.hv = hv;
    pa.W = dW; pa.w_stride = (size_t)Mp * Mp; pa.U = dU; pa.u_ld = D; pa.b0 = 0; pa.nb = D;
    pa.F = Kf; pa.rowsq = rowsq; pa.fmean = fmean; pa.ng = ng;
    launch_ucols(sc.stream, dU, M, Mp, D, 0, D, ucol);
    ProjGemmArgs pg{};
    pg.Kf = Kf; pg.kf_stride = (size_t)Tp * Mp; pg.W = dW; pg.w_stride = (size_t)Mp * Mp; pg.F = nullptr; pg.f_stride = 0;
    pg.rowsq = rowsq; pg.fmean = fmean; pg.u = ucol; pg.u_stride = Mp; pg.Tp = Tp; pg.Mp = Mp; pg.Dl = D; pg.b0 = 0; pg.nb = D;
    // the whole sweep is enqueued at once: steps x (K_fu rows, projection, conditional, propagate + weight + resample)
    auto step_launches = [&]() {
        for (int t = 0; t < steps; ++t) {
            if (skinny) {                               // K(x_t, Z) m-major: the skinny product's operand loads coalesce
                launch_kfu_build_t(sc.stream, pa, Tp);
                launch_skinny_gemm(sc.stream, Kf, (size_t)Tp * Mp, Tp, dW, (size_t)Mp * Mp, Mp, 1, R, Mp, Mp, D, Tp,
                                   nullptr, 0, 0, ucol, Mp, rowsq, fmean, nullptr, 0, 0, 0, nullptr, 1);
            } else { launch_kfu_build(sc.stream, pa); launch_proj_gemm(sc.stream, pg); }                                // conditional_after_kernel_precalculation (:95-97)
            const double *ctrl_next = (C && t + 1 < steps) ? dctrl + (size_t)(t + 1) * C : nullptr;
            launch_conditional_finish(sc.stream, kind, dxc, R, P, variance, rowsq, fmean, ngs, Tp, D, dmean, dvar, nullptr);
            launch_pg_step(sc.stream, dmean, dvar, dlq, deps + (size_t)t * R * D, dun + (size_t)t * R, dY + (size_t)t * Ydim,
                           dXr + (size_t)(t + 1) * D, dCC, dDD, dR, ctrl_next,
                           R, D, C, Ydim, dxc, cand, dparts + (size_t)t * R * D, didx + (size_t)t * R);
        }
    };
    // Round 4: ONE persistent launch for the sweep (loops.hip), three grid-wide barriers per step instead of four dependent launches:
    // measured slower (86 against 43 us per step with roles and per-unit counters, 151 with a grid-wide barrier; see ffvd_op_rollout), opt-in with FFVD_STEP_LOOP=1
    const char *nsl = getenv("FFVD_STEP_LOOP");
    const bool use_loop = skinny && nsl && *nsl && strcmp(nsl, "0") != 0;
    if (use_loop) {
        const int nwords = loop_words(D);
        int32_t *words = sc.alloc<int32_t>(nwords);
        OP_CHECK(words, "ffvd_op_pg_sweep");
        HIP_TRY(hipMemsetAsync(words, 0, (size_t)nwords * sizeof(int32_t), sc.stream));
        PgLoopArgs la{};
        la.pa = pa;
        la.sk = SkinnyArgs{Kf, (size_t)Tp * Mp, Mp, dW, (size_t)Mp * Mp, Mp, 1, R, Mp, Mp, D, Tp, nullptr, 0, 0, ucol, (size_t)Mp, rowsq, fmean,
                           nullptr, 0, 0, 0, nullptr};
        la.kind = kind; la.R = R; la.D = D; la.C = C; la.Ydim = Ydim; la.steps = steps; la.ngs = ngs;
        la.variance = variance; la.rowsq = rowsq; la.fmean = fmean; la.mean = dmean; la.var = dvar; la.cand = cand; la.parts = dparts;
        la.idx = didx; la.log_Q = dlq; la.eps = deps; la.unif = dun; la.Y = dY; la.X_ref = dXr; la.CC = dCC; la.DD = dDD; la.Rch = dR;
        la.ctrl = dctrl; la.bar = reinterpret_cast<unsigned *>(words); la.abort_w = words + 1;
        launch_pg_loop(sc.stream, la);
        int32_t hw[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(hw, words, sizeof hw, hipMemcpyDeviceToHost, sc.stream));
        HIP_TRY(hipStreamSynchronize(sc.stream));
        if (hw[1] != 0) {
            HIP_TRY(hipMemcpyAsync(dxc, xc0.data(), xc0.size() * sizeof(double), hipMemcpyHostToDevice, sc.stream));
            step_launches();
        }
    } else {
        const bool timing = getenv("FFVD_PG_TIMING") != nullptr;          // debug: how long the host takes to enqueue the sweep
        if (timing) {
            hipStreamSynchronize(sc.stream);
            fprintf(stderr, "ffvd_op_pg_sweep: uploads and set-up %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
        }
        const auto t0 = std::chrono::steady_clock::now();
        step_launches();
        if (timing) {
            const auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(sc.stream);
            const auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "ffvd_op_pg_sweep: %d steps enqueued in %.2f ms, drained %.2f ms later\n", steps,
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
    }
    HIP_TRY(hipGetLastError());
    const auto t_dl = std::chrono::steady_clock::now();
    if (!sc.download(particles + (size_t)R * D, dparts, (size_t)steps * R * D * sizeof(double)) ||
        !sc.download(idx, didx, (size_t)steps * R * sizeof(int32_t)))
        return set_error(nullptr, FFVD_EDEVICE, "ffvd_op_pg_sweep: copying the results to the host failed");
    if (getenv("FFVD_PG_TIMING"))
        fprintf(stderr, "ffvd_op_pg_sweep: results to the host %.2f ms (after the sweep had drained)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dl).count());
    return FFVD_OK;
}

// ---- T-shard fallback (include/ffvd_abi.h "T-shard") ------------------------------------------------------
static int tshard_ready(ffvd_handle *h, const char *who) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, std::string(who) + ": null handle");
    if (h->cfg.T_total <= 0) return set_error(h, FFVD_EINVAL, std::string(who) + ": the handle is not a T-shard (cfg.T_total = 0)");
    return ready(h, who);
}

// this shard's rows: raw Gram tiles K_uf K_fu + delta^T K_fu rows into the exchange buffer, likelihood / transition /
// trace sums into its tail.  The K_uu chain (identical on every rank) runs first on the same stream.
static int enqueue_tshard_local(ffvd_handle *h) {
    const ffvd_config &c = h->cfg;
    const int Mp = h->Mp, Tp = h->Tp, Dl = h->Dl, P = h->P;
    hipStream_t s = h->stream;
    const ffvd_params &p = h->cur;
    const size_t msq = (size_t)Mp * Mp, kstride = 2 * msq;
    launch_prep_hypers(s, c.kernel_kind, p.Z, c.M, Mp, P, Dl, c.d_begin, p.logvariance, p.loglengthscales, h->variance, h->len,
                       h->Zs, h->zz, h->info, Dl + h->nbatch);
    HyperView hv{h->variance, h->len, h->Zs, h->zz};
    launch_kuu_build(s, c.kernel_kind, hv, c.M, Mp, P, Dl, c.jitter, h->Kuu, h->Kcopy);
    launch_potrf_ext(s, h->Kuu, Mp, Mp, Mp, Dl, kstride, h->info, h->dinvK);
    launch_transpose(s, h->Kuu + msq, kstride, h->Linv, msq, Mp, Dl);
    GramArgs gk{};
    gk.mode = GRAM_PLAIN; gk.A = h->Linv; gk.a_stride = msq; gk.rows = Mp; gk.with_row = 0; gk.Mp = Mp; gk.Dl = Dl;
    gk.d_begin = c.d_begin; gk.b0 = 0; gk.nb = Dl; gk.yn_over_batch = 1.0; gk.H = h->Kinv; gk.h_stride = msq;
    launch_gram(s, gk);                                                     // K^-1 = L^-T L^-1
    launch_h_finish(s, h->Kuu, Mp, kstride, Dl, h->kterms);                 // log|K|
    ProjectArgs pa{};
    pa.kind = c.kernel_kind;
    pa.x = p.X; pa.x_chain_stride = (size_t)(c.T + 1) * c.D; pa.x_ld = c.D; pa.x_cols = c.D;
    pa.ctrl = h->ctrl; pa.T = c.T; pa.Tp = Tp; pa.C = c.C; pa.P = P; pa.M = c.M; pa.Mp = Mp; pa.Dl = Dl;
    pa.d_begin = c.d_begin; pa.hv = hv; pa.b0 = 0; pa.nb = h->nbatch; pa.F = h->F; pa.ng = h->ng;
    launch_kfu_build(s, pa);
    GramArgs gr{};
    gr.mode = GRAM_PLAIN; gr.A = h->F; gr.a_stride = (size_t)Tp * Mp; gr.rows = Tp; gr.with_row = 1; gr.brow = Mp;
    gr.X = p.X; gr.log_Q = p.log_Q; gr.T = c.T; gr.D = c.D; gr.Mp = Mp; gr.Dl = Dl; gr.d_begin = c.d_begin; gr.b0 = 0;
    gr.nb = h->nbatch; gr.yn_over_batch = 1.0; gr.H = h->tsbuf; gr.h_stride = (size_t)(Mp + 1) * Mp;
    if (h->gpart) { gr.ksplit = h->gsplit; gr.part = h->gpart; }
    launch_gram(s, gr);                                                     // raw sum over this shard's rows
    ReduceArgs ra{};
    ra.kind = c.kernel_kind; ra.branch = c.branch; ra.X = p.X; ra.ctrl = h->ctrl; ra.Y = h->Y;
    ra.log_Q = p.log_Q; ra.CC = p.CC; ra.DD = p.DD; ra.log_Rchols = p.log_Rchols; ra.variance = h->variance;
    ra.T = c.T; ra.Tp = Tp; ra.D = c.D; ra.C = c.C; ra.Ydim = c.Ydim; ra.Dl = Dl; ra.d_begin = c.d_begin;
    ra.S = c.S_local; ra.ng = h->ng; ra.shared_terms = c.shared_terms;
    ra.xk = p.X; ra.xk_chain_stride = (size_t)(c.T + 1) * c.D; ra.xk_ld = c.D; ra.xk_cols = c.D;
    ra.rowsq = nullptr; ra.fmean = h->fmean; ra.chain_terms = h->chain_terms;
    ra.skip_x0 = c.t_begin > 0;
    ra.info = h->info; ra.ninfo = Dl;       // a failed / abandoned K_uu chain of THIS shard turns its chain sums into NaN: they are part of the exchange
    launch_chain_reduce(s, ra, h->chain_partial);
    HIP_TRY(hipGetLastError());
    return FFVD_OK;
}

// on the all-reduced sums: A = K_uu + K_uf K_fu / Q, the trace partials, Cholesky(A), the solve, the assembly
static int enqueue_tshard_finish(ffvd_handle *h) {
    const ffvd_config &c = h->cfg;
    const int Mp = h->Mp, Dl = h->Dl, P = h->P;
    hipStream_t s = h->stream;
    const ffvd_params &p = h->cur;
    const size_t msq = (size_t)Mp * Mp;
    GramArgs ga{};
    ga.mode = GRAM_KFU; ga.rows = h->Tp; ga.with_row = 1; ga.brow = Mp;
    ga.X = p.X; ga.log_Q = p.log_Q; ga.T = c.T; ga.D = c.D; ga.Mp = Mp; ga.Dl = Dl; ga.d_begin = c.d_begin; ga.b0 = 0;
    ga.nb = h->nbatch; ga.yn_over_batch = 1.0; ga.H = h->H; ga.h_stride = (size_t)(Mp + NB) * Mp;
    ga.Kadd = h->Kcopy; ga.kadd_stride = msq; ga.Kinv = h->Kinv; ga.kinv_stride = msq; ga.trpart = h->trpart;
    ga.part = h->tsbuf; ga.ksplit = 1;
    if (c.grad) {
        // training forward (round 4: T-shards have a backward pass): the slab keeps Mp extension rows in front of the b row -- L^T there,
        // so that the factorisation of A leaves L_H^-T and y = L_A^-1 c, exactly what enqueue_grad_b reads -- and
        // A itself is saved before it is overwritten
        ga.brow = 2 * Mp; ga.h_stride = (size_t)(2 * Mp + NB) * Mp;
        launch_gram(s, ga, 4);
        HIP_TRY(hipMemcpy2DAsync(h->gw.Acopy, msq * sizeof(double), h->H, ga.h_stride * sizeof(double), msq * sizeof(double),
                                 (size_t)h->nbatch, hipMemcpyDeviceToDevice, s));
        // (FFVD_GRAD_EXPLICIT: identity rows instead, the factorisation then leaves L_A^-T -- the unwhitened form of the backward pass)
        const bool lt_virtual = h->gw.whitened && !h->sw.lt_armed && potrf_flow_selected(Mp, h->nbatch, CHOL_FLOW);
        if (!h->gw.whitened) launch_set_identity(s, h->H, ga.h_stride, Mp, Mp, h->nbatch);
        else if (!lt_virtual) launch_set_lt_rows(s, h->Kuu, (size_t)2 * msq, Dl, h->H, ga.h_stride, Mp, Mp, h->nbatch);
        launch_potrf_ext(s, h->H, Mp, Mp + NB, Mp, h->nbatch, ga.h_stride, h->info + Dl, h->dinvH, CHOL_FLOW, nullptr, 0, false, true,
                         nullptr, 0, lt_virtual ? h->Kuu : nullptr, (size_t)2 * msq, Dl);
        launch_h_finish(s, h->H, Mp, ga.h_stride, h->nbatch, h->hterms, 2 * Mp);
    } else {
        launch_gram(s, ga, 4);
        launch_potrf_ext(s, h->H, Mp, NB, 0, h->nbatch, ga.h_stride, h->info + Dl, h->dinvH, CHOL_FLOW, nullptr, 0, false, true);
        launch_h_finish(s, h->H, Mp, ga.h_stride, h->nbatch, h->hterms);
    }
    FinalizeArgs fa{};
    fa.kind = c.kernel_kind; fa.branch = c.branch; fa.prior_type = c.prior_type; fa.shared_terms = c.shared_terms;
    fa.T = c.T_total;                                   // every /T of dgp_model.py:261-297 is the whole job's
    fa.D = c.D; fa.P = P; fa.M = c.M; fa.Ydim = c.Ydim; fa.Dl = Dl; fa.d_begin = c.d_begin;
    fa.S = c.S_local; fa.Z = p.Z; fa.U = p.U; fa.logvar = p.logvariance; fa.loglen = p.loglengthscales;
    fa.log_Q = p.log_Q; fa.CC = p.CC; fa.DD = p.DD; fa.log_Rchols = p.log_Rchols;
    fa.chain_terms = h->chain_terms; fa.hterms = h->hterms; fa.chain_nll = h->chain_nll;
    fa.route = 1; fa.kterms = h->kterms; fa.trpart = h->trpart; fa.ntiles = h->ntiles;
    fa.out_terms = h->out_terms;
    fa.info = h->info; fa.ninfo = Dl + h->nbatch;
    launch_finalize(s, fa);
    HIP_TRY(hipGetLastError());
    return FFVD_OK;
}

extern "C" int64_t ffvd_tshard_count(const ffvd_handle *h) { return h ? h->ts_count : 0; }

extern "C" int ffvd_tshard_local(ffvd_handle *h) {
    int rc;
    if ((rc = tshard_ready(h, "ffvd_tshard_local"))) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    return enqueue_tshard_local(h);
}

extern "C" int ffvd_tshard_get(ffvd_handle *h, double *host_out) {
    int rc;
    if ((rc = tshard_ready(h, "ffvd_tshard_get"))) return rc;
    if (!host_out) return set_error(h, FFVD_EINVAL, "ffvd_tshard_get: null output");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipMemcpyAsync(host_out, h->tsbuf, (size_t)h->ts_count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return FFVD_OK;
}

extern "C" int ffvd_tshard_set(ffvd_handle *h, const double *host_in) {
    int rc;
    if ((rc = tshard_ready(h, "ffvd_tshard_set"))) return rc;
    if (!host_in) return set_error(h, FFVD_EINVAL, "ffvd_tshard_set: null input");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipMemcpyAsync(h->tsbuf, host_in, (size_t)h->ts_count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return FFVD_OK;
}

// The finish (S_total > 0: and the backward pass behind it) with the recovery of an abandoned dataflow Cholesky(A).
static int tshard_finish_run(ffvd_handle *h, int S_total, const char *who, double out_terms[8], double *out_nll) {
    int rc;
    // The finish is a pure function of the exchanged buffer (read-only here) and of this rank's K_uu chain, and no collective
    // follows inside the call: a dataflow Cholesky(A) that gave up on a bounded wait is re-run ONCE with the launch-per-column
    // variant, like the single-rank entry points (fetch_with_stall_recovery).  A failure of the K_uu chain itself (local phase,
    // BEFORE the exchange) has already turned the exchanged chain sums into NaN on every rank.
    for (int attempt = 0;; ++attempt) {
        {
            CholOverrideGuard guard;
            if (attempt == 1) {
                guard.force_left();
                HIP_TRY(hipMemsetAsync(h->info + h->Dl, 0, (size_t)h->nbatch * sizeof(int32_t), h->stream));
            }
            if ((rc = enqueue_tshard_finish(h))) return rc;
            if (S_total > 0) {
                // backward pass on the job's factorisation: this shard's ADDITIVE share of every gradient (grad_finalize: the
                // M x M-side terms and the priors count on the first shard only), dX for the shard's own rows.  The term sums
                // head the block like in a sharded training step -- the job's on the first shard, zero elsewhere -- so that
                // ONE all-reduce(sum) of ffvd_train_exchange_count doubles completes both.
                if ((rc = enqueue_grad_b(h, S_total))) return rc;
                if (h->cfg.t_begin == 0)
                    HIP_TRY(hipMemcpyAsync(h->gw.pack, h->out_terms, 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
                else HIP_TRY(hipMemsetAsync(h->gw.pack, 0, 8 * sizeof(double), h->stream));
            }
        }
        HIP_TRY(hipMemcpyAsync(h->h_res, h->resblk, h->res_bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        rc = check_info(h);
        bool kuu_ok = true;
        for (int i = 0; i < h->Dl; ++i) kuu_ok = kuu_ok && h->h_info[i] == 0;
        if (rc == FFVD_EDEVICE && h->stalled && kuu_ok && attempt == 0) continue;
        if (rc == FFVD_OK && attempt == 1) {
            if (h->stall_recoveries++ == 0)
                h->warning = "warning: the one-launch (dataflow) Cholesky gave up on a bounded wait; the T-shard finish was re-run with "
                             "the launch-per-column Cholesky and completed";
            h->err = h->warning;
        }
        break;
    }
    if (rc) return rc;
    for (int i = 0; i < 7; ++i)
        if (!std::isfinite(h->h_out[i]))
            return set_error(h, FFVD_ENOTPD, std::string(who) + ": non-finite sums after the exchange (a factorisation failed or was abandoned on another rank)");
    if (out_terms) memcpy(out_terms, h->h_out, 8 * sizeof(double));
    if (out_nll) *out_nll = h->h_out[FFVD_TERM_NLL] / (double)h->cfg.S_local;
    return FFVD_OK;
}

extern "C" int ffvd_tshard_finish(ffvd_handle *h, double out_terms[8], double *out_nll) {
    int rc;
    if ((rc = tshard_ready(h, "ffvd_tshard_finish"))) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    return tshard_finish_run(h, 0, "ffvd_tshard_finish", out_terms, out_nll);
}

// ---- gradient of a T-sharded job -------------------------------------------------------------------------
// After the exchange of the raw tiles every shard holds the job's A, its factor, u and Gamma; the K_fu side of the backward pass
// (E = (2 K_fu Gamma + alpha delta u^T) o K_fu and its reductions) runs over the shard's own rows and is additive over shards,
// like the likelihood / transition sums; the M x M side (K_uu chain rule, tr(A^-1 G), u^T G u, priors) is the same on every shard
// and counted on the first.  dX covers the shard's own T + 1 rows: the row two neighbouring shards share (the last of one, the
// first of the next) gets a part from each -- the caller adds them.
static int tshard_grad_ready(ffvd_handle *h, const char *who, int S_total) {
    int rc;
    if ((rc = tshard_ready(h, who))) return rc;
    if (!h->cfg.grad) return set_error(h, FFVD_EINVAL, std::string(who) + ": the handle was created without grad = 1");
    if (S_total < h->cfg.S_local) return set_error(h, FFVD_EINVAL, std::string(who) + ": S_total < S_local");
    return FFVD_OK;
}

extern "C" int ffvd_tshard_finish_grad(ffvd_handle *h, int S_total, double out_terms[8], double *out_nll) {
    int rc;
    if ((rc = tshard_grad_ready(h, "ffvd_tshard_finish_grad", S_total))) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    return tshard_finish_run(h, S_total, "ffvd_tshard_finish_grad", out_terms, out_nll);
}

extern "C" int ffvd_tshard_grad_fetch(ffvd_handle *h, double out_terms[8], const ffvd_grads *gout) {
    int rc;
    if ((rc = tshard_grad_ready(h, "ffvd_tshard_grad_fetch", h ? h->cfg.S_local : 0))) return rc;
    if (!gout) return set_error(h, FFVD_EINVAL, "ffvd_tshard_grad_fetch: null gradient struct");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if ((rc = copy_grads_out(h, gout))) return rc;
    HIP_TRY(hipMemcpyAsync(h->h_sums, h->gw.pack, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 8; ++i)
        if (!std::isfinite(h->h_sums[i]))
            return set_error(h, FFVD_ENOTPD, "ffvd_tshard_grad_fetch: non-finite sums in the exchanged block (a factorisation failed on another rank)");
    if (out_terms) memcpy(out_terms, h->h_sums, 8 * sizeof(double));
    return FFVD_OK;
}

// Optimiser step of a T-sharded job (dgp_model.py:303-305 trains every variable).  After ffvd_tshard_grad_fetch /
// ffvd_elbo_tshard_grad the exchanged block in gw.pack holds the WHOLE job's shared-parameter gradients, identical on every shard;
// dX holds this shard's own T + 1 rows, whose first and last row each lack the part of the neighbouring shard.  The caller adds
// those parts (one small exchange of boundary rows, ffvd_amd/distributed.py) and hands the rows back here: they replace gw.dX and the
// fused Adam update runs over every parameter array.  Shared parameters receive the same gradient and carry the same optimiser
// state on every shard, the two copies of a boundary row likewise: the shards' parameters stay identical without a broadcast.
extern "C" int ffvd_tshard_adam_apply(ffvd_handle *h, const double *dX_rows, double lr, double beta1, double beta2, double eps,
                                      uint32_t train_mask, double out_terms[8], double *out_nll) {
    int rc;
    if ((rc = tshard_grad_ready(h, "ffvd_tshard_adam_apply", h ? h->cfg.S_local : 0))) return rc;
    if (!dX_rows) return set_error(h, FFVD_EINVAL, "ffvd_tshard_adam_apply: null dX rows");
    if (!(lr > 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return set_error(h, FFVD_EINVAL, "ffvd_tshard_adam_apply: bad hyper-parameter");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (!h->adam_ready && (rc = ffvd_optimizer_reset(h))) return rc;
    const ffvd_config &c = h->cfg;
    HIP_TRY(hipMemcpyAsync(h->h_sums, h->gw.pack, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(h->gw.dX, dX_rows, (size_t)c.S_local * (c.T + 1) * c.D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));          // the host rows are the caller's
    for (int i = 0; i < 8; ++i)
        if (!std::isfinite(h->h_sums[i]))
            return set_error(h, FFVD_ENOTPD, "ffvd_tshard_adam_apply: non-finite sums in the exchanged block (a factorisation failed on a shard); parameters untouched");
    if ((rc = adam_update(h, lr, beta1, beta2, eps, train_mask))) return rc;
    train_report(h, out_terms, out_nll);
    return FFVD_OK;
}

// ... and the SG-HMC update (burn_in_op / sample_op, base_model.py:143-179) of a T-sharded job: X is never an SG-HMC variable
// (dgp_model.py:213-244), so the exchanged block is all it needs -- every shard applies the same update with the same noise.
extern "C" int ffvd_tshard_sghmc_apply(ffvd_handle *h, double epsilon, double mdecay, uint32_t sample_mask, int burn_in,
                                       const ffvd_params *noise, double out_terms[8], double *out_nll) {
    int rc;
    if ((rc = tshard_grad_ready(h, "ffvd_tshard_sghmc_apply", h ? h->cfg.S_local : 0))) return rc;
    if (!noise || !(epsilon > 0.0) || !(mdecay >= 0.0)) return set_error(h, FFVD_EINVAL, "ffvd_tshard_sghmc_apply: bad argument");
    if (sample_mask & FFVD_TRAIN_X) return set_error(h, FFVD_EINVAL, "ffvd_tshard_sghmc_apply: X is never an SG-HMC variable (dgp_model.py:213-244)");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if ((rc = sghmc_prepare(h, sample_mask, noise, "ffvd_tshard_sghmc_apply"))) return rc;
    HIP_TRY(hipMemcpyAsync(h->h_sums, h->gw.pack, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 8; ++i)
        if (!std::isfinite(h->h_sums[i]))
            return set_error(h, FFVD_ENOTPD, "ffvd_tshard_sghmc_apply: non-finite sums in the exchanged block (a factorisation failed on a shard); parameters untouched");
    if ((rc = sghmc_update(h, epsilon, mdecay, sample_mask, burn_in))) return rc;
    train_report(h, out_terms, out_nll);
    return FFVD_OK;
}

extern "C" int ffvd_allreduce_sum_async(ffvd_handle *h, void *rccl_comm, double *buf_dev, int64_t count);
extern "C" int ffvd_elbo_tshard(ffvd_handle *h, void *rccl_comm, double out_terms[8], double *out_nll) {
    int rc;
    if ((rc = tshard_ready(h, "ffvd_elbo_tshard"))) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if ((rc = enqueue_tshard_local(h))) return rc;
    if ((rc = ffvd_allreduce_sum_async(h, rccl_comm, h->tsbuf, h->ts_count))) return rc;      // the ONE exchange step
    return ffvd_tshard_finish(h, out_terms, out_nll);
}
// nll + gradient of a T-sharded job: two exchange steps (raw tiles + chain sums; then the gradient block), both native RCCL
extern "C" int ffvd_elbo_tshard_grad(ffvd_handle *h, void *rccl_comm, int S_total, double out_terms[8], double *out_nll,
                                     const ffvd_grads *gout) {
    int rc;
    if ((rc = tshard_grad_ready(h, "ffvd_elbo_tshard_grad", S_total))) return rc;
    if (!gout) return set_error(h, FFVD_EINVAL, "ffvd_elbo_tshard_grad: null gradient struct");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if ((rc = enqueue_tshard_local(h))) return rc;
    if ((rc = ffvd_allreduce_sum_async(h, rccl_comm, h->tsbuf, h->ts_count))) return rc;
    // a failed finish still takes part in the second exchange (NaN sums at the head of its block): the ranks stay in step
    const int rc_fin = tshard_finish_run(h, S_total, "ffvd_elbo_tshard_grad", nullptr, nullptr);
    if (rc_fin) {
        std::vector<double> nan8(8, std::nan(""));
        HIP_TRY(hipMemcpy(h->gw.pack, nan8.data(), 8 * sizeof(double), hipMemcpyHostToDevice));
    }
    const std::string first_err = rc_fin ? h->err : std::string();
    if ((rc = ffvd_allreduce_sum_async(h, rccl_comm, h->gw.pack, (int64_t)h->gw.pack_shared))) return rc;
    if (rc_fin) { HIP_TRY(hipStreamSynchronize(h->stream)); return set_error(h, rc_fin, first_err); }
    double sums[8];
    if ((rc = ffvd_tshard_grad_fetch(h, sums, gout))) return rc;
    if (out_terms) memcpy(out_terms, sums, 8 * sizeof(double));
    if (out_nll) *out_nll = sums[FFVD_TERM_NLL] / sums[FFVD_TERM_COUNT];
    return FFVD_OK;
}

// ---- native RCCL collectives (SURVEY 8b `ffvd_elbo_allreduce(h, rccl_comm)`, 8e) ---------------------------------
// The only exchange step of the path is an all-reduce(sum) of the 8 partial sums over xGMI.  librccl is bound at run time
// (dlopen): the library that is already mapped in the process wins (a host that also runs PyTorch has torch's bundled
// RCCL mapped, and two RCCL copies in one process must be avoided), then $FFVD_RCCL_LIB, then the system library.  A
// build box or a host without RCCL therefore still loads libffvd_hip.so; the collective entry points then fail loudly.
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
void rccl_bind(RcclApi &api);
RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;             // handles of different threads may ask at the same time
    std::call_once(once, [] { rccl_bind(api); });
    return &api;
}
void rccl_bind(RcclApi &api) {
    const char *env = getenv("FFVD_RCCL_LIB");
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)
        if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);           // whatever the process already runs on
    if (!api.lib && env && *env) api.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (const char *n : names)
        if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) api.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) {
        const char *why = dlerror();
        api.why = std::string("librccl not found: ") + (why ? why : "?");
        return;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
        api.why = "librccl lacks an expected symbol";
        api.lib = nullptr;
    }
}
}  // namespace

#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            char buf_[512];                                                                        \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, api->GetErrorString(r_), __FILE__, __LINE__); \
            return set_error(h, FFVD_EDEVICE, buf_);                                               \
        }                                                                                          \
    } while (0)

extern "C" int ffvd_comm_unique_id(void *id_out) {
    ffvd_handle *h = nullptr;
    if (!id_out) return set_error(nullptr, FFVD_EINVAL, "ffvd_comm_unique_id: null argument");
    RcclApi *api = rccl_api();
    if (!api->lib) return set_error(nullptr, FFVD_EDEVICE, "ffvd_comm_unique_id: " + api->why);
    ncclUniqueId id;
    RCCL_TRY(api->GetUniqueId(&id));
    memcpy(id_out, &id, FFVD_COMM_ID_BYTES);
    return FFVD_OK;
}

extern "C" int ffvd_comm_init(ffvd_handle *h, int world, int rank, const void *id) {
    if (!h || !id || world < 1 || rank < 0 || rank >= world)
        return set_error(h, FFVD_EINVAL, "ffvd_comm_init: bad argument");
    if (h->comm) return set_error(h, FFVD_EINVAL, "ffvd_comm_init: the handle already owns a communicator");
    RcclApi *api = rccl_api();
    if (!api->lib) return set_error(h, FFVD_EDEVICE, "ffvd_comm_init: " + api->why);
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    ncclUniqueId uid;
    static_assert(sizeof(uid) == FFVD_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    RCCL_TRY(api->CommInitRank(&comm, world, uid, rank));
    h->comm = (void *)comm;
    h->comm_world = world;
    h->comm_rank = rank;
    return FFVD_OK;
}

extern "C" int ffvd_comm_destroy(ffvd_handle *h) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_comm_destroy: null handle");
    if (!h->comm) return FFVD_OK;
    RcclApi *api = rccl_api();
    hipSetDevice(h->cfg.device_id);
    hipStreamSynchronize(h->stream);
    ncclComm_t comm = (ncclComm_t)h->comm;
    h->comm = nullptr;
    if (api->lib) RCCL_TRY(api->CommDestroy(comm));
    return FFVD_OK;
}

extern "C" void *ffvd_comm_get(ffvd_handle *h) { return h ? h->comm : nullptr; }

extern "C" int ffvd_allreduce_sum_async(ffvd_handle *h, void *rccl_comm, double *buf_dev, int64_t count) {
    if (!h || !buf_dev || count < 0) return set_error(h, FFVD_EINVAL, "ffvd_allreduce_sum_async: bad argument");
    void *comm = rccl_comm ? rccl_comm : h->comm;
    if (!comm) return set_error(h, FFVD_EINVAL, "ffvd_allreduce_sum_async: no communicator (pass one or call ffvd_comm_init)");
    RcclApi *api = rccl_api();
    if (!api->lib) return set_error(h, FFVD_EDEVICE, "ffvd_allreduce_sum_async: " + api->why);
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (count == 0) return FFVD_OK;
    RCCL_TRY(api->AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)comm, h->stream));
    return FFVD_OK;
}

extern "C" int ffvd_allreduce_sum(ffvd_handle *h, void *rccl_comm, double *buf_host, int64_t count) {
    if (!h || !buf_host || count < 0) return set_error(h, FFVD_EINVAL, "ffvd_allreduce_sum: bad argument");
    if (count == 0) return FFVD_OK;
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (h->stage_count < count) {          // device staging buffer, grown on demand and kept by the handle
        double *d = nullptr;
        HIP_TRY(dev_alloc(h, &d, (size_t)count));
        h->stage = d;
        h->stage_count = count;
    }
    HIP_TRY(hipMemcpyAsync(h->stage, buf_host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = ffvd_allreduce_sum_async(h, rccl_comm, h->stage, count);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(buf_host, h->stage, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return FFVD_OK;
}

extern "C" int ffvd_elbo_allreduce_async(ffvd_handle *h, void *rccl_comm, double *out_terms_dev) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_elbo_allreduce_async: null handle");
    int rc;
    double *dst = out_terms_dev ? out_terms_dev : h->out_terms;
    if ((rc = ffvd_elbo_async(h, dst))) return rc;
    return ffvd_allreduce_sum_async(h, rccl_comm, dst, 8);
}

extern "C" int ffvd_elbo_allreduce(ffvd_handle *h, void *rccl_comm, double out_terms[8], double *out_nll) {
    if (!h) return set_error(nullptr, FFVD_EINVAL, "ffvd_elbo_allreduce: null handle");
    int rc;
    // kernels -> finalize (8 partial sums in HBM) -> ncclAllReduce on the same stream -> one copy back: the only host
    // synchronisation of the step is the final one
    if ((rc = ffvd_elbo_allreduce_async(h, rccl_comm, nullptr))) return rc;
    h->info_pending = false;
    HIP_TRY(hipMemcpyAsync(h->h_res, h->resblk, h->res_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if ((rc = check_info(h))) return rc;              // this rank's factorisations
    bool finite = true;
    for (int i = 0; i < 8; ++i) finite = finite && std::isfinite(h->h_out[i]);
    if (!finite)                                      // a failed factorisation on ANOTHER rank poisons the sums with NaN
        return set_error(h, FFVD_ENOTPD, "ffvd_elbo_allreduce: non-finite partial sums after the all-reduce (a factorisation failed or was abandoned on another rank)");
    if (out_terms) memcpy(out_terms, h->h_out, 8 * sizeof(double));
    if (out_nll) *out_nll = h->h_out[FFVD_TERM_NLL] / h->h_out[FFVD_TERM_COUNT];
    return FFVD_OK;
}
