// The step loops of the posterior rollouts (collect_samples_formal, base_model.py:288-314) and of the particle-Gibbs sweep
// (PG_for_X_speedup, :99-115) as ONE persistent launch per call.
//
// A step is three (rollouts) or four (particle Gibbs) dependent kernels of 5-10 us each -- K_fu rows of the current states, the
// skinny product against L^-T (and W q_sqrt), the conditional epilogue + update (+ weights / resampling) -- and the per-step launches
// were back to back on the GPU at 33 / 37 us per step (tools/prof_rollout.sh): what a step costs is the NUMBER of dependent launches.
// Here the grid stays resident for the whole loop: its workgroups walk the virtual blocks of each phase (the SAME bodies as the
// per-step kernels, step_bodies.h: results are bit-identical) and meet at a grid-wide barrier between phases -- a monotone counter in
// device memory with the release / acquire hand-off of the dataflow Cholesky, every wait bounded by the wall clock (an abort word
// ends the launch; the caller then runs the per-step launches instead).  noise and uniforms stay injected.
#include "kernels.h"
#include "dev_common.h"
#include "step_bodies.h"

namespace ffvd {

constexpr long long LOOP_SPIN_TICKS = 100000000LL;      // 1 s of the 100 MHz wall clock

// ---- hand-offs between the roles of the loop -------------------------------------------------------------------------------------
// Round 4, first form: every workgroup walked every phase and all of them met at ONE counter between phases -- 64.7 us per rollout
// step against 32.7 for the per-step launches: 160-512 workgroups adding to and polling one word is a 15 us barrier.  This form: a
// workgroup keeps ONE role for the whole loop (K_fu rows of a unit / a slab of the skinny product / the update) and waits only for what
// its role reads -- the K_fu rows of ITS unit (8-16 arrivals on that unit's word), the slabs of every unit (one word per unit), the
// update (one word) -- every word on a cache line of its own.  Same bodies, same order of every sum: bit-identical to the launches.
#ifndef LOOP_POLL_SLEEP
#define LOOP_POLL_SLEEP 8              // x 64 cycles between two polls of a word (dozens of workgroups poll the same one)
#endif
constexpr int LOOP_WORD_STRIDE = 16;                     // ints between two counters (64 bytes)
__device__ __forceinline__ int *loop_word(int *base, int i) { return base + (size_t)(2 + i) * LOOP_WORD_STRIDE; }

// thread 0 polls until *word >= need (bounded by the wall clock; an abort word ends the launch), acquires, tells the workgroup
__device__ __forceinline__ bool loop_wait(int *word, const int need, int *abort_w, int *slot) {
    if (threadIdx.x == 0) {
        int ok = 1;
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            const long long t0 = wall_clock64();
            for (;;) {
                __builtin_amdgcn_s_sleep(LOOP_POLL_SLEEP);
                if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
                if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (wall_clock64() - t0 > LOOP_SPIN_TICKS) {
                    __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *slot = ok;
    }
    __syncthreads();
    const int ok = *slot;
    __syncthreads();
    return ok != 0;
}
// the workgroup's stores of this piece of work are in memory: count it
__device__ __forceinline__ void loop_arrive(int *word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// words: [0] unused, [1] abort, then one counter per LOOP_WORD_STRIDE ints: 0 = updates done, 1 .. nb = K_fu row blocks of unit u,
// nb + 1 .. 2 nb = product slabs of unit u (all of them counts of pieces of work, monotone over the steps)
template <int KIND, int NQ>
__global__ __launch_bounds__(256) void rollout_loop_kernel(RolloutLoopArgs a, const int GA, const int GB, const int GC) {
    __shared__ int slot;
    ProjectArgs pa = a.pa;
    const int nAx = pa.Tp / 64, nAy = pa.Mp / 64, perA = nAx * nAy, nA = perA * pa.nb;
    const int nBx = a.sk.N / 16 + (a.sk.B2 ? a.sk.N2 / 16 : 0), nBy = (a.sk.rows + 31) / 32, perB = nBx * nBy, nB = perB * a.sk.nb;
    const int R = a.R, D = a.f.D, nC = (R * D * 16 + 255) / 256, nb = pa.nb;
    int *base = reinterpret_cast<int *>(a.bar), *abort_w = a.abort_w;
    const int bid = blockIdx.x;
    const int role = bid < GA ? 0 : (bid < GA + GB ? 1 : 2);
    const int first = role == 0 ? bid : (role == 1 ? bid - GA : bid - GA - GB), stride = role == 0 ? GA : (role == 1 ? GB : GC);
    for (int t = 0; t < a.steps; ++t) {
        const double *xin = (t & 1) ? a.xbuf1 : a.xbuf0;
        double *xout = (t & 1) ? a.xbuf0 : a.xbuf1;
        if (role == 0) {                                                              // K(x_t, Z) per dim
            if (t > 0 && !loop_wait(loop_word(base, 0), nC * t, abort_w, &slot)) return;      // every row of x_t is written
            pa.x = xin;
            for (int vb = first; vb < nA; vb += stride) {
                kfu_build_body<KIND, NQ>(pa, vb % nAx, (vb / nAx) % nAy, vb / perA);
                loop_arrive(loop_word(base, 1 + vb / perA));
            }
        } else if (role == 1) {                                                       // F = K W (and |K (W q_sqrt)|^2): partial sums per slab
            for (int vb = first; vb < nB; vb += stride) {
                const int u = vb / perB;
                if (!loop_wait(loop_word(base, 1 + u), perA * (t + 1), abort_w, &slot)) return;
                skinny_body(a.sk, vb % nBx, (vb / nBx) % nBy, u);
                loop_arrive(loop_word(base, 1 + nb + u));
            }
        } else {                                                                      // conditional epilogue + x <- x + f_mu + eps sqrt(f_var + Q)
            for (int u = 0; u < nb; ++u)
                if (!loop_wait(loop_word(base, 1 + nb + u), perB * (t + 1), abort_w, &slot)) return;
            for (int vb = first; vb < nC; vb += stride) {
                rollout_finish_update_body(vb, a.f, a.log_Q, a.eps + (size_t)t * R * D,
                                           (a.C && a.ctrl && t + 1 < a.steps) ? a.ctrl + (size_t)(t + 1) * a.C : nullptr, R, a.C, t, a.steps, xin,
                                           xout, a.predict_x, a.predict_var);
                loop_arrive(loop_word(base, 0));
            }
        }
    }
}

template <int KIND, int NQ>
__global__ __launch_bounds__(256) void pg_loop_kernel(PgLoopArgs a, const int GA, const int GB) {
    __shared__ int slot;
    const ProjectArgs &pa = a.pa;
    const int nAx = pa.Tp / 64, nAy = pa.Mp / 64, perA = nAx * nAy, nA = perA * pa.nb;
    const int nBx = a.sk.N / 16, nBy = (a.sk.rows + 31) / 32, perB = nBx * nBy, nB = perB * a.sk.nb;
    const int R = a.R, D = a.D, nC = (R * D * 16 + 255) / 256, nb = pa.nb;
    int *base = reinterpret_cast<int *>(a.bar), *abort_w = a.abort_w;
    const int bid = blockIdx.x;
    const int role = bid < GA ? 0 : (bid < GA + GB ? 1 : 2);
    const int first = role == 0 ? bid : (role == 1 ? bid - GA : 0), stride = role == 0 ? GA : GB;
    for (int t = 0; t < a.steps; ++t) {
        if (role == 0) {
            if (t > 0 && !loop_wait(loop_word(base, 0), t, abort_w, &slot)) return;   // the particles of step t are in place
            for (int vb = first; vb < nA; vb += stride) {
                kfu_build_body<KIND, NQ>(pa, vb % nAx, (vb / nAx) % nAy, vb / perA);
                loop_arrive(loop_word(base, 1 + vb / perA));
            }
        } else if (role == 1) {
            for (int vb = first; vb < nB; vb += stride) {
                const int u = vb / perB;
                if (!loop_wait(loop_word(base, 1 + u), perA * (t + 1), abort_w, &slot)) return;
                skinny_body(a.sk, vb % nBx, (vb / nBx) % nBy, u);
                loop_arrive(loop_word(base, 1 + nb + u));
            }
        } else {
            // conditional_after_kernel_precalculation's epilogue (:95-97) for every particle, then propagate + weight + resample (:99-115):
            // one workgroup (the cumulative sum of the weights is sequential); mean / var go through memory as in the per-step launches
            for (int u = 0; u < nb; ++u)
                if (!loop_wait(loop_word(base, 1 + nb + u), perB * (t + 1), abort_w, &slot)) return;
            for (int vb = 0; vb < nC; ++vb)
                conditional_finish_body(vb, a.kind, pa.x, R, pa.P, a.variance, a.rowsq, a.fmean, a.ngs, pa.Tp, D, a.mean, a.var, nullptr, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            pg_step_body<1>(a.mean, a.var, a.log_Q, a.eps + (size_t)t * R * D, a.unif + (size_t)t * R, a.Y + (size_t)t * a.Ydim,
                            a.X_ref + (size_t)(t + 1) * D, a.CC, a.DD, a.Rch, (a.C && a.ctrl && t + 1 < a.steps) ? a.ctrl + (size_t)(t + 1) * a.C : nullptr,
                            R, D, a.C, a.Ydim, const_cast<double *>(pa.x), a.cand, a.parts + (size_t)t * R * D, a.idx + (size_t)t * R);
            loop_arrive(loop_word(base, 0));
        }
    }
}

// ---- third form (round 4, end): the loop with RESIDENT operands ----------------------------------------------------------------------
// The two forms above replace kernel boundaries by hand-offs and keep everything else: every step still streams L^-T (2 MB per dim,
// 4 with W q_sqrt) through 170 workgroups that each announce and await something.  This one is built around what a hand-off costs
// (tools/probes/handoff_probe.hip: 1.6 us per 128 KB inside an XCD, 2.8 us across) and has THREE of them per step:
//   workgroup (d, s) owns 16 columns of dim d's W = L^-T (and of W q_sqrt) -- loaded into LDS ONCE, before the loop -- and the 16
//   inducing points of the same index range; all workgroups of a dim sit on one XCD (id = 8 s + d).  Per step:
//   1. K(x_t, z_m) for its 16 inducing points and all rollouts -> Kt[d][m][r] (rollouts contiguous: the products' A fragments are
//      then coalesced loads straight from L2, no staging)                                                  -> count on cK[d]
//   2. (all slabs of the dim arrived)  F = K W and K (W q_sqrt) for its 16 columns on the matrix cores (W is upper triangular: slab s
//      contracts over 16 (s + 1) rows), the wavefronts split the contraction or the row tiles, partial sums meet in LDS;
//      per rollout sum_j F^2, sum_j F u_j, sum_j (F q)^2 of the slab -> part[d][s][r]                      -> count on cP[d]
//   3. workgroup (d, 0): (all slabs arrived) adds the slabs in fixed order, x_{t+1,d} = x_td + f_mu + eps sqrt(f_var + Q_d)  (:300-314),
//      predict_x / predict_var                                                                              -> count on cX
//   and everyone starts step t + 1 when all D dims have counted on cX (the only hand-off that crosses XCDs).
// Every wait is bounded; an abort makes the caller run the per-step launches.  Not bit-identical to them (the sums over a
// row of F are formed per slab in another order): tests hold it to the oracle and to the launches at 1e-9.
#if !defined(__gfx942__) && !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "rollout_resident_kernel's hand-offs rely on gfx942 / gfx950 semantics (agent-scope accesses = sc1: write-through stores counted in vmcnt, loads past the non-coherent caches)"
#endif
// Hand-offs without cache maintenance (what the stamps of the fenced version asked for: release 2 us, wait + acquire 2.5-3 us among the
// 32 workgroups of a dim).  Every word another workgroup reads is written with an agent-scope (sc1, write-through) store and read with an
// agent-scope load, i.e. past the caches that are not coherent across compute units / XCDs; the producer waits for its stores
// (s_waitcnt vmcnt(0)) in front of the workgroup barrier behind which one lane counts, the consumer's loads are issued behind the
// barrier that follows its poll.  No fence on either side -- the Gram kernel's tail exchange (kernels.hip) is the same construction.
__device__ __forceinline__ void rr_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double rr_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr long long RR_SPIN_TICKS = 10000000LL;         // 0.1 s of the 100 MHz wall clock: a step is 10-25 us; a wait this long means that the
                                                         // launch's workgroups are not all on the chip (another tenant): give up, the caller runs the launches
// -DFFVD_RR_FENCED (build variant `rrfenced`): the same hand-offs INSIDE the HIP memory model -- an agent-scope release in
// front of every count, an agent-scope acquire behind every satisfied wait (what the first measurement of this form used: 4-5 us per
// hand-off instead of 2-3).  tests/test_gpu_ops.py runs the oracle-parity rollout test on both builds; the product build's
// fence-free form is verified by that comparison and by the soak only -- it rests on gfx942 / gfx950 semantics (the #error above).
__device__ __forceinline__ bool rr_wait(int *word, const int need, int *abort_w, int *slot) {
    if (threadIdx.x == 0) {
        int ok = 1;
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            const long long t0 = wall_clock64();
            for (int spin = 0;; ++spin) {
#ifdef RR_POLL_SLEEP
                __builtin_amdgcn_s_sleep(RR_POLL_SLEEP);
#endif
                if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
                if ((spin & 63) != 63) continue;                   // (the abort word and the clock once in 64 polls)
                if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (wall_clock64() - t0 > RR_SPIN_TICKS) {
                    __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
#ifdef FFVD_RR_FENCED
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        *slot = ok;
    }
    __syncthreads();
    const int ok = *slot;
    __syncthreads();
    return ok != 0;
}
__device__ __forceinline__ void rr_arrive(int *word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
#ifdef FFVD_RR_FENCED
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
constexpr int RR_RED = 4 * 2 * 4 * 64;            // doubles: accumulators of four wavefronts, two right-hand sides
constexpr int RR_PF = 16;                         // A fragments in flight per lane
template <int KIND>
__global__ __launch_bounds__(256) void rollout_resident_kernel(RolloutResidentArgs a) {
    extern __shared__ double rr_lds[];
    __shared__ int slot;
    const int d = blockIdx.x & 7, s = blockIdx.x >> 3;
    if (d >= a.D) return;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mp = a.Mp, P = a.P, D = a.D, R = a.R, RT = a.RT, RP = 16 * RT, NS = a.NS, C = a.C;
    const bool hasq = a.WQ != nullptr;
    double *Wp = rr_lds;                                   // [mlim][16]
    double *WQp = Wp + (size_t)Mp * 16;                    // [Mp][16] (only with q_sqrt)
    double *red = hasq ? WQp + (size_t)Mp * 16 : WQp;      // [4][2][4][64]
    double *zsl = red + RR_RED;                            // [16][8]  this slab's inducing inputs (scaled)
    double *xrow = zsl + 128;                              // [64][8]  the step's inputs (scaled)
    double *usl = xrow + 512;                              // [16]
    double *sums = usl + 16;                               // [4][64][3] (the updater's quarter sums)
    const int mlim = 16 * (s + 1), j0 = 16 * s;
    {
        const double *Wd = a.W + (size_t)d * a.w_stride;
        for (int e = tid; e < mlim * 16; e += 256) Wp[e] = Wd[(size_t)(e >> 4) * Mp + j0 + (e & 15)];
        if (hasq) {
            const double *Qd = a.WQ + (size_t)d * a.w_stride;
            for (int e = tid; e < Mp * 16; e += 256) WQp[e] = Qd[(size_t)(e >> 4) * Mp + j0 + (e & 15)];
        }
        if (tid < 128) zsl[tid] = ((tid & 7) < P) ? a.hv.Zs[((size_t)d * Mp + j0 + (tid >> 3)) * P + (tid & 7)] : 0.0;
        if (tid < 16) usl[tid] = a.ucol[(size_t)d * Mp + j0 + tid];
    }
    const double var = a.hv.variance[d];
    const double zzm = a.hv.zz[(size_t)d * Mp + j0 + (tid >> 4)];
    const bool mok = j0 + (tid >> 4) < a.M;
    double lenp = 1.0;                                     // thread (r, p) of the input staging keeps its lengthscale
    if (KIND == 0 && (tid & 7) < P) lenp = a.hv.len[(size_t)d * P + (tid & 7)];
    const double Qd = exp(a.log_Q[d]);
    int *base = a.words, *abort_w = a.abort_w;
    int *cX = loop_word(base, 0), *cK = loop_word(base, 1 + d), *cP = loop_word(base, 9 + d);
    // which part of the products this wavefront forms: row tile rt, part kp of nkp of the contraction
    int nkp, rt, kp;
    if (RT == 1) { nkp = 4; rt = 0; kp = wave; }
    else if (RT == 2) { nkp = 2; rt = wave & 1; kp = wave >> 1; }
    else { nkp = 1; rt = wave; kp = 0; }
    const bool active = rt < RT;
    // the contraction in PAIRS of k-steps (8 inducing points)
    const int npW = mlim / 8, ppW = (npW + nkp - 1) / nkp, pw0 = kp * ppW, pw1 = (pw0 + ppW < npW) ? pw0 + ppW : npW;
    const int npQ = a.wq_upper ? npW : Mp / 8, ppQ = (npQ + nkp - 1) / nkp,          // (an upper-triangular W q_sqrt ends where W ends)
          pq0 = kp * ppQ, pq1 = (pq0 + ppQ < npQ) ? pq0 + ppQ : npQ;
    double *Ktd = a.Kt + (size_t)d * Mp * RP;
    __syncthreads();
    long long *stp = (a.stamps && d == 0 && (s == 0 || s == NS - 1)) ? a.stamps + (s == 0 ? 0 : 16) : nullptr;
#define RR_STAMP(i) do { if (stp && t == 10 && tid == 0) stp[i] = wall_clock64(); } while (0)
    for (int t = 0; t < a.steps; ++t) {
        if (a.test_stall && d == 0 && s == 0 && t == 3) return;
        RR_STAMP(0);
        if (t > 0 && !rr_wait(cX, D * t, abort_w, &slot)) return;                     // every dim of x_t is written
        RR_STAMP(1);
        const double *xin = a.xbuf + (size_t)(t & 1) * RP * D;
        double *xout = a.xbuf + (size_t)((t + 1) & 1) * RP * D;
        for (int e = tid; e < RP * 8; e += 256) {                                         // (256 = 32 rows x 8: a thread keeps its p)
            const int r = e >> 3, p = e & 7;
            double v = 0.0;
            if (r < R && p < P) {
                v = (p < D) ? (t == 0 ? a.x_last[p] : rr_load(xin + (size_t)r * D + p)) : a.ctrl[(size_t)t * C + (p - D)];
                v = (KIND == 0) ? v / lenp : v * var;
            }
            xrow[e] = v;
        }
        __syncthreads();
        RR_STAMP(2);
        {   // 1. K(x_t, z_m): thread = (inducing point m = tid >> 4, rollout r = tid & 15 of every row tile)
            const int m = tid >> 4;
            double zr[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) zr[p] = zsl[m * 8 + p];
            for (int pass = 0; pass < RT; ++pass) {
                const int r = 16 * pass + (tid & 15);
                double dot = 0.0, xx = 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) { const double xv = xrow[r * 8 + p]; dot += xv * zr[p]; if (KIND == 0) xx += xv * xv; }
                double v = kernel_value<KIND>(dot, xx, zzm, var);
                if (r >= R || !mok) v = 0.0;
                rr_store(Ktd + (size_t)(j0 + m) * RP + r, v);
            }
        }
        RR_STAMP(3);
        rr_arrive(cK);
        RR_STAMP(4);
        if (!rr_wait(cK, NS * (t + 1), abort_w, &slot)) return;
        RR_STAMP(5);
        // 2. the products for this slab's 16 columns
        d4 accW = (d4){0.0, 0.0, 0.0, 0.0}, accQ = accW, accW1 = accW, accQ1 = accW;
        if (active) {
            // A fragments straight from L2 (8 bytes per lane and k-step), RR_PF pairs of k-steps in flight ahead of the MFMAs that use them
            const double *Ka = Ktd + (size_t)lk * RP + 16 * rt + lr;                      // element (m, r) at m RP + r: a wavefront's load is four full lines
            const int p0 = hasq ? pq0 : pw0, p1 = hasq ? pq1 : pw1;                       // (with q_sqrt both products walk the same range)
            auto lda = [&](int pr) {                                                       // k-steps 2 pr and 2 pr + 1
                const double *q = Ka + (size_t)((pr < p1) ? pr : p1 - 1) * 8 * RP;
                return make_double2(rr_load(q), rr_load(q + (size_t)4 * RP));
            };
            // (the MFMA section of a chunk of RR_PF pairs is branch-free when the whole chunk has the same right-hand sides -- with a
            //  test per MFMA every one of them waited for its own LDS read: 8.7 us for 128 MFMAs)
            auto run = [&](auto hq) {
                constexpr bool HQ = decltype(hq)::value;
                double2 cur[RR_PF], nxt[RR_PF];
#pragma unroll
                for (int i = 0; i < RR_PF; ++i) cur[i] = lda(p0 + i);
                for (int pb = p0; pb < p1; pb += RR_PF) {
                    if (pb + RR_PF < p1) {
#pragma unroll
                        for (int i = 0; i < RR_PF; ++i) nxt[i] = lda(pb + RR_PF + i);
                    }
                    const bool full = pb + RR_PF <= p1;
                    if (full && pb + RR_PF <= npW) {                  // every pair: W (and W q_sqrt)
#pragma unroll
                        for (int i = 0; i < RR_PF; ++i) {
                            const int m0 = 8 * (pb + i) + lk;
                            if (HQ) accQ = mfma_f64(cur[i].x, WQp[m0 * 16 + lr], accQ);
                            accW = mfma_f64(cur[i].x, Wp[m0 * 16 + lr], accW);
                            if (HQ) accQ1 = mfma_f64(cur[i].y, WQp[(m0 + 4) * 16 + lr], accQ1);
                            accW1 = mfma_f64(cur[i].y, Wp[(m0 + 4) * 16 + lr], accW1);
                        }
                    } else if (HQ && full && pb >= npW) {             // every pair: W q_sqrt only
#pragma unroll
                        for (int i = 0; i < RR_PF; ++i) {
                            const int m0 = 8 * (pb + i) + lk;
                            accQ = mfma_f64(cur[i].x, WQp[m0 * 16 + lr], accQ);
                            accQ1 = mfma_f64(cur[i].y, WQp[(m0 + 4) * 16 + lr], accQ1);
                        }
                    } else {                                          // a partial chunk or one that straddles the end of W's rows
#pragma unroll
                        for (int i = 0; i < RR_PF; ++i) {
                            const int pr = pb + i;
                            if (pr < p1) {
                                const int m0 = 8 * pr + lk;
                                if (HQ) accQ = mfma_f64(cur[i].x, WQp[m0 * 16 + lr], accQ);
                                if (pr < npW) accW = mfma_f64(cur[i].x, Wp[m0 * 16 + lr], accW);
                                if (HQ) accQ1 = mfma_f64(cur[i].y, WQp[(m0 + 4) * 16 + lr], accQ1);
                                if (pr < npW) accW1 = mfma_f64(cur[i].y, Wp[(m0 + 4) * 16 + lr], accW1);
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < RR_PF; ++i) cur[i] = nxt[i];
                }
            };
            if (hasq) run(std::true_type{});
            else run(std::false_type{});
        }
        RR_STAMP(6);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            red[((wave * 2 + 0) * 4 + q) * 64 + lane] = accW[q] + accW1[q];
            red[((wave * 2 + 1) * 4 + q) * 64 + lane] = accQ[q] + accQ1[q];
        }
        __syncthreads();
        if (active && kp == 0) {
            const double uj = usl[lr];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double f = 0.0, fq = 0.0;
                for (int pp = 0; pp < nkp; ++pp) {                                        // fixed order over the parts of the contraction
                    const int w2 = (RT == 1) ? pp : ((RT == 2) ? rt + 2 * pp : rt);
                    f += red[((w2 * 2 + 0) * 4 + q) * 64 + lane];
                    fq += red[((w2 * 2 + 1) * 4 + q) * 64 + lane];
                }
                double rs = f * f, fm = f * uj, ex = fq * fq;
#pragma unroll
                for (int mm = 1; mm < 16; mm <<= 1) { rs += __shfl_xor(rs, mm); fm += __shfl_xor(fm, mm); ex += __shfl_xor(ex, mm); }
                if (lr == 0) {
                    double *pp = a.part + (((size_t)d * NS + s) * RP + 16 * rt + lk + 4 * q) * 4;
                    rr_store(pp, rs); rr_store(pp + 1, fm); rr_store(pp + 2, ex);
                }
            }
        }
        RR_STAMP(7);
        rr_arrive(cP);
        RR_STAMP(8);
        if (s == 0) {
            // 3. conditional epilogue + update of dim d (conditionals_multi_output.py:355-387, base_model.py:300-314)
            if (!rr_wait(cP, NS * (t + 1), abort_w, &slot)) return;
            RR_STAMP(9);
            {
                const int r = tid & 63, qtr = tid >> 6, nq = NS / 4;                      // nq <= 8
                double rs = 0.0, fm = 0.0, ex = 0.0;
                if (r < RP) {
                    double v0[8], v1[8], v2[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {                                          // all loads first, then the adds in slab order
                        const double *pp = a.part + (((size_t)d * NS + qtr * nq + ((i < nq) ? i : 0)) * RP + r) * 4;
                        v0[i] = rr_load(pp); v1[i] = rr_load(pp + 1); v2[i] = rr_load(pp + 2);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < nq) { rs += v0[i]; fm += v1[i]; ex += v2[i]; }
                }
                sums[(qtr * 64 + r) * 3 + 0] = rs; sums[(qtr * 64 + r) * 3 + 1] = fm; sums[(qtr * 64 + r) * 3 + 2] = ex;
            }
            __syncthreads();
            if (tid < R) {
                const int r = tid;
                double rs = 0.0, fm = 0.0, ex = 0.0;
#pragma unroll
                for (int qtr = 0; qtr < 4; ++qtr) { rs += sums[(qtr * 64 + r) * 3]; fm += sums[(qtr * 64 + r) * 3 + 1]; ex += sums[(qtr * 64 + r) * 3 + 2]; }
                const double xd = (t == 0) ? a.x_last[d] : rr_load(xin + (size_t)r * D + d);
                double kd = var;
                if (KIND == 1) {
                    kd = 0.0;
                    for (int p = 0; p < P; ++p) {
                        const double xv = (p < D) ? (t == 0 ? a.x_last[p] : rr_load(xin + (size_t)r * D + p)) : a.ctrl[(size_t)t * C + (p - D)];
                        kd += (xv * xv) * var;
                    }
                }
                double vv = kd - rs;
                if (hasq) vv = vv + ex;
                const double v = vv + Qd;
                const double xn = (fm + xd) + a.eps[((size_t)t * R + r) * D + d] * sqrt(v);
                const size_t o = ((size_t)r * a.steps + t) * D + d;
                a.predict_x[o] = xn;
                a.predict_var[o] = v;
                rr_store(xout + (size_t)r * D + d, xn);
            }
            RR_STAMP(10);
            rr_arrive(cX);
            RR_STAMP(11);
        }
    }
#undef RR_STAMP
}

bool rollout_resident_ok(int R, int D, int P, int Mp) { return R >= 1 && R <= 64 && D >= 1 && D <= 8 && P <= 8 && Mp >= 64 && Mp <= 512 && Mp % 64 == 0; }
int rollout_resident_words() { return (2 + 17) * LOOP_WORD_STRIDE; }
int launch_rollout_resident(hipStream_t stream, const RolloutResidentArgs &a) {
    const size_t lds = ((size_t)a.Mp * 16 * (a.WQ ? 2 : 1) + RR_RED + 128 + 512 + 16 + 768) * sizeof(double);
    auto go = [&](auto kind) -> int {
        auto fn = rollout_resident_kernel<decltype(kind)::value>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(fn, dim3(8 * a.NS), dim3(256), lds, stream, a);
        return (int)hipGetLastError();
    };
    return a.kind == 0 ? go(std::integral_constant<int, 0>{}) : go(std::integral_constant<int, 1>{});
}

// every workgroup must be resident at once: 256 threads and ~25 KB of LDS each, two per CU at most
static int loop_capacity() {
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
        return n;
    }();
    return 2 * cus;
}
int loop_words(int nb) { return (2 + 1 + 2 * nb) * LOOP_WORD_STRIDE; }

template <class Args, class F>
static void dispatch_kfu(int kind, int P, F &&launch) {
    if (P <= 8) {
        const int nq = (P + 1) / 2;
        if (kind == 0) {
            if (nq <= 2) launch(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            else if (nq == 3) launch(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
            else launch(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
        } else launch(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
    } else {
        if (kind == 0) launch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        else launch(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    }
}

void launch_rollout_loop(hipStream_t stream, const RolloutLoopArgs &a) {
    const ProjectArgs &pa = a.pa;
    const int nA = (pa.Tp / 64) * (pa.Mp / 64) * pa.nb;
    const int nB = (a.sk.N / 16 + (a.sk.B2 ? a.sk.N2 / 16 : 0)) * ((a.sk.rows + 31) / 32) * a.sk.nb;
    const int nC = (a.R * a.f.D * 16 + 255) / 256;
    const int cap = loop_capacity();
    const int GC = nC < 16 ? nC : 16, GA = nA < cap / 4 ? nA : cap / 4, GB = nB < cap - GA - GC ? nB : cap - GA - GC;
    dispatch_kfu<RolloutLoopArgs>(pa.kind, pa.P, [&](auto kind, auto nq) {
        hipLaunchKernelGGL((rollout_loop_kernel<decltype(kind)::value, decltype(nq)::value>), dim3(GA + GB + GC), dim3(256), 0, stream, a, GA, GB, GC);
    });
}

void launch_pg_loop(hipStream_t stream, const PgLoopArgs &a) {
    const ProjectArgs &pa = a.pa;
    const int nA = (pa.Tp / 64) * (pa.Mp / 64) * pa.nb;
    const int nB = (a.sk.N / 16) * ((a.sk.rows + 31) / 32) * a.sk.nb;
    const int cap = loop_capacity();
    const int GA = nA < cap / 4 ? nA : cap / 4, GB = nB < cap - GA - 1 ? nB : cap - GA - 1;
    dispatch_kfu<PgLoopArgs>(pa.kind, pa.P, [&](auto kind, auto nq) {
        hipLaunchKernelGGL((pg_loop_kernel<decltype(kind)::value, decltype(nq)::value>), dim3(GA + GB + 1), dim3(256), 0, stream, a, GA, GB);
    });
}

}  // namespace ffvd
