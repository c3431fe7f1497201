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
