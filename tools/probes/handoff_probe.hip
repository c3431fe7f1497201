// Hand-off probe: one producer workgroup writes a block, announces it; one consumer workgroup waits, makes the block visible to
// itself and reads it.  What does the read cost, and which of the cheaper "make visible" forms are still CORRECT, when both sit on
// ONE XCD (one L2) and when they do not?  hipcc --offload-arch=gfx950 -O3 handoff_probe.hip -o handoff_probe
//   consumer modes: 0 agent-scope acquire fence + plain loads (what the kernels do)
//                   1 buffer_inv sc0 (workgroup scope: the vector L1 only) + plain loads
//                   2 no invalidate, loads with sc0 (miss the L1, may hit the L2)
//                   3 no invalidate, loads with sc1 (agent scope)
//   producer modes: 0 plain stores + agent-scope release fence        1 plain stores + s_waitcnt vmcnt(0) only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ long long wall() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ long long wclk() { long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

// eight 16-byte loads in flight with the given cache bits, then the wait (asm: the compiler does not track these loads)
#define LD8(BITS)                                                                                                                   \
    asm volatile("global_load_dwordx4 %0, %8, off " BITS "\n\tglobal_load_dwordx4 %1, %9, off " BITS "\n\t"                          \
                 "global_load_dwordx4 %2, %10, off " BITS "\n\tglobal_load_dwordx4 %3, %11, off " BITS "\n\t"                        \
                 "global_load_dwordx4 %4, %12, off " BITS "\n\tglobal_load_dwordx4 %5, %13, off " BITS "\n\t"                        \
                 "global_load_dwordx4 %6, %14, off " BITS "\n\tglobal_load_dwordx4 %7, %15, off " BITS "\n\ts_waitcnt vmcnt(0)"     \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])             \
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])                             \
                 : "memory")
template <int CM>
__device__ __forceinline__ void ld8(double2 (&v)[8], const double2 *const (&p)[8]) {
    if (CM == 2) LD8("sc0");
    else if (CM == 3) LD8("sc1");
    else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *p[q];
    }
}
constexpr int SPIN_MAX = 20000000;        // a wait that never ends would take the box down: give up instead

template <int CM, int PM>
__global__ __launch_bounds__(256) void handoff(double *buf, int *flag, int *back, long long *tout, int *stale, int n2 /* double2 per thread */,
                                                int rounds, int prod_id, int cons_id) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x == prod_id) {
        for (int r = 1; r <= rounds; ++r) {
            if (tid == 0 && r > 1) { int n = 0; while (__hip_atomic_load(back, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r - 1 && ++n < SPIN_MAX) __builtin_amdgcn_s_sleep(1); }
            __syncthreads();
            double2 *b = reinterpret_cast<double2 *>(buf);
            for (int i = 0; i < n2; ++i) b[(size_t)i * 256 + tid] = make_double2((double)r, (double)(r + tid));
            if (PM == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if ((int)blockIdx.x == cons_id) {
        long long total = 0;
        int bad = 0;
        for (int r = 1; r <= rounds; ++r) {
            if (tid == 0) { int n = 0; while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r && ++n < SPIN_MAX) __builtin_amdgcn_s_sleep(1); }
            __syncthreads();
            const long long t0 = wclk();
            if (CM == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (CM == 1) asm volatile("buffer_inv sc0" ::: "memory");
            const double2 *b = reinterpret_cast<const double2 *>(buf);
            double sx = 0.0, sy = 0.0;
            for (int i0 = 0; i0 < n2; i0 += 8) {
                double2 v[8];
                const double2 *p[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) p[q] = b + (size_t)(i0 + q) * 256 + tid;
                ld8<CM>(v, p);
#pragma unroll
                for (int q = 0; q < 8; ++q) { sx += v[q].x; sy += v[q].y; }
            }
            if (sx != (double)r * n2 || sy != (double)(r + tid) * n2) ++bad;
            __syncthreads();
            total += wclk() - t0;
            if (tid == 0) __hip_atomic_store(back, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) tout[0] = total;
        if (bad) atomicAdd(stale, 1);
    }
}

template <int CM, int PM>
static void run(const char *label, double *buf, int *words, long long *t, int n2, int rounds, int prod, int cons) {
    hipMemset(words, 0, 4 * sizeof(int));
    hipMemset(t, 0, sizeof(long long));
    hipLaunchKernelGGL((handoff<CM, PM>), dim3(16), dim3(256), 0, 0, buf, words, words + 1, t, words + 2, n2, rounds, prod, cons);
    hipDeviceSynchronize();
    long long h; int w[4];
    hipMemcpy(&h, t, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(w, words, sizeof(w), hipMemcpyDeviceToHost);
    printf("%-70s %s  read of %4d KB: %6.2f us per round   threads that saw a stale block: %d\n", label, (prod & 7) == (cons & 7) ? "same XCD " : "other XCD",
           n2 * 256 * 16 / 1024, (double)h * 10.0 / rounds / 1000.0, w[2]);
}

int main() {
    double *buf; int *words; long long *t;
    const int n2 = 32;                     // 32 x 256 x 16 B = 128 KB
    hipMalloc(&buf, (size_t)n2 * 256 * 16);
    hipMalloc(&words, 64);
    hipMalloc(&t, 64);
    const int rounds = 2000;
    for (int pass = 0; pass < 2; ++pass) {
        const int prod = 0, cons = pass == 0 ? 8 : 1;          // ids 0 and 8: the same XCD; 0 and 1: neighbours
        run<0, 0>("agent acquire fence + plain loads | release fence", buf, words, t, n2, rounds, prod, cons);
        run<1, 0>("buffer_inv sc0 + plain loads       | release fence", buf, words, t, n2, rounds, prod, cons);
        run<2, 0>("no invalidate, sc0 loads           | release fence", buf, words, t, n2, rounds, prod, cons);
        run<3, 0>("no invalidate, sc1 loads           | release fence", buf, words, t, n2, rounds, prod, cons);
        run<0, 1>("agent acquire fence + plain loads | vmcnt(0) only", buf, words, t, n2, rounds, prod, cons);
        run<1, 1>("buffer_inv sc0 + plain loads       | vmcnt(0) only", buf, words, t, n2, rounds, prod, cons);
        run<2, 1>("no invalidate, sc0 loads           | vmcnt(0) only", buf, words, t, n2, rounds, prod, cons);
        run<3, 1>("no invalidate, sc1 loads           | vmcnt(0) only", buf, words, t, n2, rounds, prod, cons);
    }
    return 0;
}
