// The whole ELBO iteration of the reference's own experiment size -- and its backward pass -- as ONE launch
// (FFVD_Main.py:356-369: T <= 512, M = 100, D = 4; 4000 outer iterations of models.py:142-182, each 1 or 22 evaluations of
// nll and its gradient, base_model.py:915-950).  See tiny.hip for the design; abi.hip decides per handle (tiny_plan).
#pragma once
#include "kernels.h"

namespace ffvd {

constexpr int TINY_MPMAX = 128;     // inducing points, padded to 16 (not 64: this path has no 64-blocked kernel)
constexpr int TINY_PMAX = 8;        // GP input dimension D + C (all six reference datasets: 5)
constexpr int TINY_TMAX = 2048;     // transitions

struct TinyArgs {
    int kind, T, D, C, P, M, Mp, NT, Dl, d_begin, S, Ydim;
    int SR, nstrips, nunits;        // rows per strip workgroup (16 per wavefront), strips per unit, units = S * Dl
    int side;                       // backward: NT more workgroups per unit take the row blocks of the K_uu side (else the strips do)
    int xcd_map;                    // workgroup id -> (unit, role): all workgroups of a unit on ONE XCD (blockIdx % 8 = unit % 8)
    int prior_type, shared_terms, grad, S_total;
    int branch;                     // 1: collapsed U (dgp_model.py:267-288);  0: explicit U (:289-297, regularizer :337-359; cases 1/2/3/6)
    double jitter;
    const double *X, *Z, *logvar, *loglen, *log_Q, *CC, *DD, *logR, *Y, *ctrl;
    const double *U;                // [M][D] whitened inducing outputs (explicit-U branch)
    // scratch (tiny_scratch_doubles; all of it is rewritten by every launch)
    double *Wg, *Wt;                // [nunits][Mp*Mp]  W = L^-T (upper block triangle) and its transpose L^-1
    double *Pp;                     // [nunits][nstrips][pstride]  per-strip F^T F tiles, F^T delta, chain-term partials
    double *Hs, *Nw, *Nm2;          // [nunits][Mp*Mp]  backward: H - I, N = I - H^-1 - w w^T, (scratch);  explicit U: alpha F^T F, 2 Phi, Lbar (tiny.hip, head)
    double *wv;                     // [nunits][Mp]     backward: w = H^-1 b
    double *hterms;                 // [nunits][2]      log|H|, b^T H^-1 b   (FinalizeArgs::hterms)
    double *uterms;                 // [nunits][8]      backward scalars of a unit: 0 dl/dalpha
    double *Qp;                     // [nunits][nstrips][qstride]  backward: per-strip column sums / E^T x / r x^2 partials
    double *Kst;                    // [nunits][nstrips][threads][32]  backward: each lane's K_fu elements (accumulator layout) parked in L2
    double *dxc;                    // [nunits][Tp][P + 1]  backward: rows of dl/dx_comb (p < P) and dl/ddelta (slot P)
    double *dz2;                    // [nunits][Mp][TINY_PMAX] K_uu side: rows of dl/dZ;  kuu_part [nunits][NT][TINY_PMAX + 1]
    double *kuu_part;
    double *unit_out;               // [nunits][M*P + P + 2]  per-unit totals: dl/dZ, dl/dloglen, dl/dlogvar-part
    double *chain_terms;            // [S][8]
    double *prior_sums;             // [16] the ten parameter-only sums of the nll assembly (finalize_priors), formed early by head 0
    double *chain_part;             // [S][sp_stride]  backward: dCC, dDD, dlog_Rchols row 0, transition part of dlog_Q per local dim
    int sp_stride;
    int *flags;                     // [nunits*4 + S + 8] hand-off words, all zero between launches (the last workgroup re-arms them)
    int32_t *info;                  // [Dl + nunits]
    double *chain_nll, *out_terms;
    double *du_unit;                // [nunits][Mp]  explicit-U backward: alpha F^T r per unit (= dl/du in whitened variables)
    double *dX, *dZ, *dlogvar, *dloglen, *dlogQ, *dCC, *dDD, *dlogR;   // backward outputs (layout of ffvd_grads)
    double *dU;                     // [M][D] explicit-U branch
};

struct TinyPlan {
    bool ok;
    int nw;             // wavefronts per workgroup: 4 (64-row strips) or 8 (128-row strips)
    int Mp, NT, SR, nstrips, nunits;
    int side;           // grad: workgroups of their own for the K_uu side of the backward pass (they fit beside heads and strips)
    size_t lds_bytes;
};
// Can this shape run as one launch on a chip with `cus` compute units?  (Every workgroup must be resident at once: the roles wait
// for each other.)
TinyPlan tiny_plan(int kind, int T, int D, int C, int M, int S, int Dl, int grad, int cus);
size_t tiny_scratch_doubles(const TinyPlan &pl, int T, int P, int M, int S, int Dl, int D, int Ydim, int grad);
size_t tiny_flag_ints(const TinyPlan &pl, int S);
// carve `scratch` / `flags` into the pointers of `a` (shapes already filled in)
void tiny_bind_scratch(TinyArgs &a, const TinyPlan &pl, double *scratch, int *flags);
// The argument block travels to the device only when it differs from what the device copy holds.  Uploads are asynchronous copies
// out of pinned memory, so a slot must not be rewritten before the copy that reads it has run (two back-to-back enqueues with
// different blocks -- alternating output buffers of ffvd_elbo_async -- would otherwise hand launch 1 the block of launch 2).  A ring of
// pinned slots, each guarded by an event recorded right behind its upload; what the device holds is remembered in PLAIN memory and
// compared bytewise (callers zero the whole block, padding included, before filling it).
struct TinyArgRing {
    static constexpr int N = 4;
    TinyArgs *pinned = nullptr;     // [N] pinned staging slots (hipHostMalloc, owned by the handle)
    hipEvent_t ev[N] = {};          // recorded behind the upload that reads slot i
    bool used[N] = {};
    int next = 0;
    int uploads = 0;                // how many times the block travelled (tests)
    TinyArgs held;                  // what the device copy holds
    bool held_valid = false;
};
constexpr size_t TINY_PRIVATE_BYTES_MAX = 1024;                 // scratch per lane the kernels were validated with (576 at the time of writing)
constexpr size_t TINY_PRIVATE_LAUNCH_BUDGET = (size_t)128 << 20;    // scratch of all resident waves of one launch
hipError_t tiny_kernel_private_bytes(int nw, int branch /* 1 collapsed, 0 explicit U */, size_t *bytes);
hipError_t tiny_ring_create(TinyArgRing &r);
void tiny_ring_destroy(TinyArgRing &r);
// dev_args: a TinyArgs in device memory the kernel reads its arguments from (one per launch flavour, with its ring).
hipError_t launch_tiny(hipStream_t stream, const TinyArgs &a, const TinyPlan &pl, TinyArgs *dev_args, TinyArgRing &ring);

}  // namespace ffvd
