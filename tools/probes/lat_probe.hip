// Dependent-latency probe (one wavefront): ns per dependent v_fma_f64, per v_rsq_f64 + refinement, per readlane -> fma round trip, and
// per 16 x 16 right-looking pivot chain (the inner loop of every Cholesky in this repo).  hipcc --offload-arch=gfx950 -O3 lat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../ffvd_amd/csrc/kernels.h"
#include "../../ffvd_amd/csrc/dev_common.h"
using namespace ffvd;
__global__ void probe(double *out, long long *t, int n, double seed) {
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fma(a, b, c);
    }
    long long t1 = wall_clock64();
    double r = a;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double y = __builtin_amdgcn_rsq(r);
            const double e = fma(-(r * y), y, 1.0);
            y = fma(y, e * fma(0.375, e, 0.5), y);
            r = r * y + 1.5;
        }
    }
    long long t2 = wall_clock64();
    double q = r;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) q = fma(q, readlane_f64(q, k), c);
    }
    long long t3 = wall_clock64();
    // 16-pivot chain on a diagonally dominant tile held lane = row
    double m[16];
    double acc = 0.0;
    for (int i = 0; i < n / 4 + 1; ++i) {
#pragma unroll
        for (int cidx = 0; cidx < 16; ++cidx) m[cidx] = ((threadIdx.x & 15) == cidx ? 20.0 + q * 1e-30 : 0.5 + 0.01 * cidx) + i * 1e-3;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double ajj = readlane_f64(m[j], j);
            double y = __builtin_amdgcn_rsq(ajj);
            const double e = fma(-(ajj * y), y, 1.0);
            y = fma(y, e * fma(0.375, e, 0.5), y);
            m[j] *= y;
#pragma unroll
            for (int cc = j + 1; cc < 16; ++cc) m[cc] = fma(-m[j], readlane_f64(m[j], cc), m[cc]);
        }
#pragma unroll
        for (int cidx = 0; cidx < 16; ++cidx) acc += m[cidx];
    }
    long long t4 = wall_clock64();
    out[threadIdx.x] = a + r + q + acc;
    if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; }
}

// the product's 16-pivot chain as tiny.hip / kernels.hip have it (tile from LDS, identity rows in lanes 16-31, first bad pivot tracked,
// L_ss and L_ss^-T back to LDS)
template <bool BAD, bool LOAD, bool STORE>
__device__ __forceinline__ int chain16_product(const double (*Sc)[17], double *Am, const int LD, const int s0, double (*Dv)[17], const int lane) {
    const int lr = lane & 15;
    int bad = 0;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = LOAD ? ((lane < 16) ? Sc[lr][c] : ((lane < 32 && lr == c) ? 1.0 : 0.0)) : ((lr == c) ? 20.0 + s0 : 0.5 + 0.01 * c + 0.02 * lr);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double ajj = readlane_f64(a[j], j);
        if (BAD && !(ajj > 0.0) && bad == 0) bad = j + 1;
        double piv, y;
        pivot_sqrt(ajj, piv, y);
        a[j] *= y;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = fma(-a[j], readlane_f64(a[j], c), a[c]);
    }
    if (STORE) {
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Am[(size_t)(s0 + lr) * LD + s0 + c] = (c <= lr) ? a[c] : 0.0;
    } else if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Dv[lr][c] = a[c];
    }
    } else { double v = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) v += a[c];
        if (v == 1.2345e300) Am[lane] = v; }
    return bad;
}
template <bool BAD, bool LOAD, bool STORE>
__global__ void probe2(double *out, long long *t, int n) {
    __shared__ double Sc[16][17], Am[16 * 17], Dv[16][17];
    const int lane = threadIdx.x;
    if (lane < 16) for (int c = 0; c < 16; ++c) Sc[lane][c] = (lane == c) ? 20.0 : 0.5 + 0.01 * c + 0.02 * lane;
    __syncthreads();
    int bad = 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        bad += chain16_product<BAD, LOAD, STORE>(Sc, Am, 17, i & 0, Dv, lane);
        wave_lds_order();
    }
    long long t1 = wall_clock64();
    out[lane] = Am[lane] + Dv[lane & 15][0] + bad;
    if (lane == 0) t[0] = t1 - t0;
}
// candidate: bad pivot tracked per lane off the critical path (lane j looks at its own diagonal entry at step j), one store stream for
// both halves (lanes 0-15: L_ss row, lanes 16-31: row of L_ss^-T), loads as shipped
__device__ __forceinline__ int chain16_v2(const double (*Sc)[17], double *Am, const int LD, const int s0, double (*Dv)[17], const int lane) {
    const int lr = lane & 15;
    int flag = 0;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = (lane < 16) ? Sc[lr][c] : ((lane < 32 && lr == c) ? 1.0 : 0.0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        flag = ((int)(lane == j) & (int)!(a[j] > 0.0) & (int)(flag == 0)) ? j + 1 : flag;        // (branch-free vector select: nothing scalar waits for it)
        const double ajj = readlane_f64(a[j], j);
        double piv, y;
        pivot_sqrt(ajj, piv, y);
        a[j] *= y;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = fma(-a[j], readlane_f64(a[j], c), a[c]);
    }
    if (lane < 32) {
        double *base = (lane < 16) ? Am + (size_t)(s0 + lr) * LD + s0 : &Dv[lr][0];
#pragma unroll
        for (int c = 0; c < 16; ++c) base[c] = (lane >= 16 || c <= lr) ? a[c] : 0.0;
    }
    // first bad pivot of the tile: the flags are ascending in the lane index (lane j holds j + 1 or 0)
    const unsigned long long m = __ballot(flag != 0);
    return m ? (int)__builtin_ctzll(m) + 1 : 0;
}
template <int V>
__global__ void probe3(double *out, long long *t, int n) {
    __shared__ double Sc[16][17], Am[16 * 17], Dv[16][17];
    const int lane = threadIdx.x;
    if (lane < 16) for (int c = 0; c < 16; ++c) Sc[lane][c] = (lane == c) ? 20.0 : 0.5 + 0.01 * c + 0.02 * lane;
    __syncthreads();
    int bad = 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        bad += chain16_v2(Sc, Am, 17, i & 0, Dv, lane);
        wave_lds_order();
    }
    long long t1 = wall_clock64();
    out[lane] = Am[lane] + Dv[lane & 15][0] + bad;
    if (lane == 0) t[0] = t1 - t0;
}
// candidate v4: loads through a per-lane base pointer (lanes 16-31 read an identity tile kept in LDS: no exec-masked load per element),
// no bad-pivot test inside the chain (a non-positive or NaN pivot leaves NaN on the diagonal of L from there on: looked for once,
// afterwards), one store stream
__device__ __forceinline__ int chain16_v4(const double (*Sc)[17], const double (*Id)[17], double *Am, const int LD, const int s0, double (*Dv)[17], const int lane) {
    const int lr = lane & 15;
    double a[16];
    const double *src = (lane < 16) ? &Sc[lr][0] : &Id[lr][0];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = src[c];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double ajj = readlane_f64(a[j], j);
        double piv, y;
        pivot_sqrt(ajj, piv, y);
        a[j] *= y;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = fma(-a[j], readlane_f64(a[j], c), a[c]);
    }
    double diag = a[0];
#pragma unroll
    for (int c = 1; c < 16; ++c) diag = (lr == c) ? a[c] : diag;
    if (lane < 32) {
        double *base = (lane < 16) ? Am + (size_t)(s0 + lr) * LD + s0 : &Dv[lr][0];
#pragma unroll
        for (int c = 0; c < 16; ++c) base[c] = (lane >= 16 || c <= lr) ? a[c] : 0.0;
    }
    const unsigned long long m = __ballot((lane < 16) & !(diag > 0.0));
    return m ? (int)__builtin_ctzll(m) + 1 : 0;
}
__global__ void probe4(double *out, long long *t, int n) {
    __shared__ double Sc[16][17], Id[16][17], Am[16 * 17], Dv[16][17];
    const int lane = threadIdx.x;
    if (lane < 16) for (int c = 0; c < 16; ++c) { Sc[lane][c] = (lane == c) ? 20.0 : 0.5 + 0.01 * c + 0.02 * lane; Id[lane][c] = (lane == c) ? 1.0 : 0.0; }
    __syncthreads();
    int bad = 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        bad += chain16_v4(Sc, Id, Am, 17, i & 0, Dv, lane);
        wave_lds_order();
    }
    long long t1 = wall_clock64();
    out[lane] = Am[lane] + Dv[lane & 15][0] + bad;
    if (lane == 0) t[0] = t1 - t0;
}
// the SE-kernel element of tiny.hip's K_uu build: 8-component dot against an LDS row, expanded-form distance, exp, store to LDS
template <int MODE>
__global__ void probe_kbuild(double *out, long long *t, int n) {
    __shared__ double zs[128 * 9], zz[128], Am[128 * 113];
    const int tid = threadIdx.x, r = (tid & 255) >> 4, cj = tid & 15;
    for (int e = tid; e < 128 * 9; e += blockDim.x) zs[e] = 0.01 * (e % 97);
    for (int e = tid; e < 128; e += blockDim.x) zz[e] = 0.3 + 0.001 * e;
    __syncthreads();
    const double var = 1.3;
    long long t0 = wall_clock64();
    for (int rep = 0; rep < n; ++rep)
        for (int ti = 0; ti < 7; ++ti) {
            const int i = 16 * ti + r;
            double zi[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) zi[p] = zs[i * 9 + p];
            const double zzi = zz[i];
#pragma unroll 2
            for (int tj = 0; tj <= ti; ++tj) {
                const int j = 16 * tj + cj;
                double dot = 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) dot += zi[p] * zs[j * 9 + p];
                double v;
                if (MODE == 0) v = kernel_value<0>(dot, zzi, zz[j], var);
                else if (MODE == 1) v = var * exp(-(-2.0 * dot + (zzi + zz[j])) * 0.5);
                else v = dot + zzi + zz[j];                                  // no exp: the cost of everything else
                Am[i * 113 + j] = v + rep;
            }
        }
    long long t1 = wall_clock64();
    __syncthreads();
    out[tid & 63] = Am[tid];
    if (tid == 0) t[0] = t1 - t0;
}
// cost of the hand-off primitives for ONE workgroup on an otherwise idle chip: n rounds of (256 plain 8-byte stores; release; relaxed
// atomic add) -- and the same with sc1 write-through stores and no fence -- and n rounds of (acquire; 256 loads)
template <int MODE>
__global__ void probe_fence(double *buf, int *word, long long *t, int n) {
    const int tid = threadIdx.x;
    typedef __attribute__((address_space(1))) double gdouble;
    long long t0 = wall_clock64();
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {            // plain stores + release + add
            buf[tid] = i;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (MODE == 1) {     // write-through stores, no fence
            __hip_atomic_store((gdouble *)buf + tid, (double)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {                    // acquire + loads
            if (tid == 0) {
                (void)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            acc += buf[tid + 256 * (i & 3)];
        }
        __syncthreads();
    }
    long long t1 = wall_clock64();
    if (acc == 1.2345e300) buf[0] = acc;
    if (tid == 0) t[0] = t1 - t0;
}
int main() {
    double *out; long long *t;
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 4 * 8);
    const int n = 4096;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, t, n, 1.25);
        hipDeviceSynchronize();
    }
    long long h[4];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const double tick = 10.0;       // ns per wall_clock64 tick (100 MHz)
    printf("dependent v_fma_f64:            %.2f ns per op   (= %.1f cycles at 2.4 GHz)\n", h[0] * tick / (n * 16.0), h[0] * tick / (n * 16.0) * 2.4);
    printf("rsq + Halley + mul-add:         %.2f ns per pivot-like step (= %.1f cycles)\n", h[1] * tick / (n * 4.0), h[1] * tick / (n * 4.0) * 2.4);
    printf("readlane -> fma:                %.2f ns per round trip (= %.1f cycles)\n", h[2] * tick / (n * 16.0), h[2] * tick / (n * 16.0) * 2.4);
    printf("16-pivot right-looking chain:   %.2f ns per chain = %.2f ns per pivot (= %.1f cycles)\n", h[3] * tick / (n / 4 + 1), h[3] * tick / (n / 4 + 1) / 16.0, h[3] * tick / (n / 4 + 1) / 16.0 * 2.4);
#define RUN(B, L, S, label) hipLaunchKernelGGL((probe2<B, L, S>), dim3(1), dim3(64), 0, 0, out, t, 1024); hipDeviceSynchronize(); \
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost); printf("product chain16, %s: %.2f ns per chain\n", label, h[0] * tick / 1024.0);
    RUN(true, true, true, "as shipped (LDS in/out, bad-pivot tracking)")
    RUN(false, true, true, "without the bad-pivot tracking")
    RUN(true, false, true, "without the LDS loads")
    RUN(true, true, false, "without the LDS stores")
    RUN(false, false, false, "bare")
    hipLaunchKernelGGL((probe3<0>), dim3(1), dim3(64), 0, 0, out, t, 1024); hipDeviceSynchronize();
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost); printf("candidate v2 (per-lane bad flag, one store stream): %.2f ns per chain\n", h[0] * tick / 1024.0);
    hipLaunchKernelGGL(probe4, dim3(1), dim3(64), 0, 0, out, t, 1024); hipDeviceSynchronize();
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost); printf("candidate v4 (identity tile, NaN check afterwards, one store stream): %.2f ns per chain\n", h[0] * tick / 1024.0);
#define RUNK(M, label) hipLaunchKernelGGL((probe_kbuild<M>), dim3(1), dim3(256), 0, 0, out, t, 64); hipDeviceSynchronize(); \
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost); printf("K build element loop (28 elements per thread, 256 threads), %s: %.2f us per pass = %.1f ns per element per wavefront\n", label, h[0] * tick / 64.0 / 1000.0, h[0] * tick / 64.0 / 28.0);
    RUNK(0, "exp_kernel") RUNK(1, "library exp") RUNK(2, "no exp")
    {
        double *fb; int *fw; hipMalloc(&fb, 2048 * 8); hipMalloc(&fw, 64); hipMemset(fb, 0, 2048 * 8); hipMemset(fw, 0, 64);
#define RUNF(M, label) hipLaunchKernelGGL((probe_fence<M>), dim3(1), dim3(256), 0, 0, fb, fw, t, 2000); hipDeviceSynchronize(); \
        hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost); printf("hand-off primitive, %s: %.2f us per round\n", label, h[0] * tick / 2000.0 / 1000.0);
        RUNF(0, "2 KB of plain stores + agent-scope release + atomic add") RUNF(1, "2 KB of sc1 write-through stores + atomic add, no fence") RUNF(2, "atomic load + agent-scope acquire + 2 KB of loads")
    }
    return 0;
}
