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

// All workgroups of the grid have finished the phase; `target` = (phases so far) x (workgroups).  Returns false when a wait gave up.
__device__ __forceinline__ bool loop_grid_sync(unsigned *count, int *abort_w, const unsigned target, int *slot) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            const long long t0 = wall_clock64();
            for (;;) {
                __builtin_amdgcn_s_sleep(1);
                if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
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

template <int KIND, int NQ>
__global__ __launch_bounds__(256) void rollout_loop_kernel(RolloutLoopArgs a) {
    __shared__ int slot;
    const unsigned G = gridDim.x;
    unsigned phase = 0;
    ProjectArgs pa = a.pa;
    const int nAx = pa.Tp / 64, nAy = pa.Mp / 64, nA = nAx * nAy * pa.nb;
    const int nBx = a.sk.N / 16 + (a.sk.B2 ? a.sk.N2 / 16 : 0), nBy = (a.sk.rows + 31) / 32, nB = nBx * nBy * a.sk.nb;
    const int R = a.R, D = a.f.D, nC = (R * D * 16 + 255) / 256;
    for (int t = 0; t < a.steps; ++t) {
        const double *xin = (t & 1) ? a.xbuf1 : a.xbuf0;
        double *xout = (t & 1) ? a.xbuf0 : a.xbuf1;
        pa.x = xin;
        for (int vb = blockIdx.x; vb < nA; vb += G)                                   // K(x_t, Z) per dim
            kfu_build_body<KIND, NQ>(pa, vb % nAx, (vb / nAx) % nAy, vb / (nAx * nAy));
        if (!loop_grid_sync(a.bar, a.abort_w, ++phase * G, &slot)) return;
        for (int vb = blockIdx.x; vb < nB; vb += G)                                   // F = K W (and |K (W q_sqrt)|^2): partial sums per slab
            skinny_body(a.sk, vb % nBx, (vb / nBx) % nBy, vb / (nBx * nBy));
        if (!loop_grid_sync(a.bar, a.abort_w, ++phase * G, &slot)) return;
        for (int vb = blockIdx.x; vb < nC; vb += G)                                   // conditional epilogue + x <- x + f_mu + eps sqrt(f_var + Q)
            rollout_finish_update_body(vb, a.f, a.log_Q, a.eps + (size_t)t * R * D,
                                       (a.C && a.ctrl && t + 1 < a.steps) ? a.ctrl + (size_t)(t + 1) * a.C : nullptr, R, a.C, t, a.steps, xin,
                                       xout, a.predict_x, a.predict_var);
        if (t + 1 < a.steps && !loop_grid_sync(a.bar, a.abort_w, ++phase * G, &slot)) return;
    }
}

template <int KIND, int NQ>
__global__ __launch_bounds__(256) void pg_loop_kernel(PgLoopArgs a) {
    __shared__ int slot;
    const unsigned G = gridDim.x;
    unsigned phase = 0;
    const ProjectArgs &pa = a.pa;
    const int nAx = pa.Tp / 64, nAy = pa.Mp / 64, nA = nAx * nAy * pa.nb;
    const int nBx = a.sk.N / 16, nBy = (a.sk.rows + 31) / 32, nB = nBx * nBy * a.sk.nb;
    const int R = a.R, D = a.D, nC = (R * D * 16 + 255) / 256;
    for (int t = 0; t < a.steps; ++t) {
        for (int vb = blockIdx.x; vb < nA; vb += G)
            kfu_build_body<KIND, NQ>(pa, vb % nAx, (vb / nAx) % nAy, vb / (nAx * nAy));
        if (!loop_grid_sync(a.bar, a.abort_w, ++phase * G, &slot)) return;
        for (int vb = blockIdx.x; vb < nB; vb += G)
            skinny_body(a.sk, vb % nBx, (vb / nBx) % nBy, vb / (nBx * nBy));
        if (!loop_grid_sync(a.bar, a.abort_w, ++phase * G, &slot)) return;
        if (blockIdx.x == 0) {
            // conditional_after_kernel_precalculation's epilogue (:95-97) for every particle, then propagate + weight + resample (:99-115):
            // one workgroup (the cumulative sum of the weights is sequential); mean / var go through memory as in the per-step launches
            for (int vb = 0; vb < nC; ++vb)
                conditional_finish_body(vb, a.kind, pa.x, R, pa.P, a.variance, a.rowsq, a.fmean, a.ngs, pa.Tp, D, a.mean, a.var, nullptr, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            pg_step_body<1>(a.mean, a.var, a.log_Q, a.eps + (size_t)t * R * D, a.unif + (size_t)t * R, a.Y + (size_t)t * a.Ydim,
                            a.X_ref + (size_t)(t + 1) * D, a.CC, a.DD, a.Rch, (a.C && a.ctrl && t + 1 < a.steps) ? a.ctrl + (size_t)(t + 1) * a.C : nullptr,
                            R, D, a.C, a.Ydim, const_cast<double *>(pa.x), a.cand, a.parts + (size_t)t * R * D, a.idx + (size_t)t * R);
        }
        if (t + 1 < a.steps && !loop_grid_sync(a.bar, a.abort_w, ++phase * G, &slot)) return;
    }
}

static int loop_grid(int nmax) {
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
        return n;
    }();
    const int cap = 2 * cus;                      // every workgroup must be resident: 256 threads and ~25 KB of LDS each, two per CU at most
    return nmax < cap ? (nmax < 1 ? 1 : nmax) : cap;
}

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
    const int G = loop_grid(nA > nB ? nA : nB);
    dispatch_kfu<RolloutLoopArgs>(pa.kind, pa.P, [&](auto kind, auto nq) {
        hipLaunchKernelGGL((rollout_loop_kernel<decltype(kind)::value, decltype(nq)::value>), dim3(G), dim3(256), 0, stream, a);
    });
}

void launch_pg_loop(hipStream_t stream, const PgLoopArgs &a) {
    const ProjectArgs &pa = a.pa;
    const int nA = (pa.Tp / 64) * (pa.Mp / 64) * pa.nb;
    const int nB = (a.sk.N / 16) * ((a.sk.rows + 31) / 32) * a.sk.nb;
    const int G = loop_grid(nA > nB ? nA : nB);
    dispatch_kfu<PgLoopArgs>(pa.kind, pa.P, [&](auto kind, auto nq) {
        hipLaunchKernelGGL((pg_loop_kernel<decltype(kind)::value, decltype(nq)::value>), dim3(G), dim3(256), 0, stream, a);
    });
}

}  // namespace ffvd
