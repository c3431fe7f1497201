// Reduced form of the round-4 corruption: a kernel takes a ~416-byte block BY VALUE, hands a
// reference to it (= a generic pointer to its private-memory copy) to a non-inlined callee, runs spill-heavy inlined code, and hands it
// over again.  The callee folds every field into a checksum; the two checksums of a lane must be equal and equal to the host's.
// Build + run: hipcc -O3 --offload-arch=gfx950 tools/probes/byval_probe.hip -o tools/probes/byval_probe && tools/probes/byval_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct Block { int i[22]; double jit; const double *p[40]; };                // 22 ints + 1 double + 40 pointers = 416 bytes
__device__ __noinline__ uint64_t fold(const Block &b, int salt) {
    uint64_t s = (uint64_t)salt;
    for (int k = 0; k < 22; ++k) s = s * 1000003u + (uint64_t)b.i[k];
    for (int k = 0; k < 40; ++k) s = s * 1000003u + (uint64_t)(uintptr_t)b.p[k];
    return s;
}
template <int N> __device__ __forceinline__ double heavy(const double *src, int n, double seed) {      // N live doubles per lane
    double r[N];
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = src[k * 64 + threadIdx.x % 64] + seed;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int k = 0; k < N; ++k) r[k] = __builtin_fma(r[k], r[(k + 7) % N], r[(k + 13) % N]);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) s += r[k];
    return s;
}
__global__ __launch_bounds__(512, 1) void probe(Block b, const double *src, int n, uint64_t *out, double *sink) {
    extern __shared__ double lds[];
    const int t = blockIdx.x * 512 + threadIdx.x;
    const uint64_t before = fold(b, 1);
    double v = 0;
    if (blockIdx.x % 5 == 0) v = heavy<160>(src, n, (double)b.i[3]);         // "heads": 320 VGPRs' worth of live values -> spills
    lds[threadIdx.x] = v; __syncthreads();
    const uint64_t after = fold(b, 1);
    out[2 * t] = before; out[2 * t + 1] = after; sink[t] = v + lds[(threadIdx.x + 1) % 512];
}
int main() {
    const int grid = 200, nt = grid * 512;
    Block b{}; for (int k = 0; k < 22; ++k) b.i[k] = 100 + k; b.jit = 1e-5;
    double *src; uint64_t *out; double *sink;
    hipMalloc(&src, 160 * 64 * 8); hipMemset(src, 0, 160 * 64 * 8); hipMalloc(&out, nt * 16); hipMalloc(&sink, nt * 8);
    for (int k = 0; k < 40; ++k) b.p[k] = src + 17 * k;
    uint64_t want = 1; for (int k = 0; k < 22; ++k) want = want * 1000003u + (uint64_t)b.i[k];
    for (int k = 0; k < 40; ++k) want = want * 1000003u + (uint64_t)(uintptr_t)b.p[k];
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)probe);
    printf("localSizeBytes %zu numRegs %d\n", (size_t)fa.localSizeBytes, fa.numRegs);
    uint64_t *h = new uint64_t[2 * nt]; long bad = 0;
    for (int rep = 0; rep < 50; ++rep) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 100 * 1024, 0, b, src, 4 + rep % 3, out, sink);
        hipMemcpy(h, out, nt * 16, hipMemcpyDeviceToHost);
        for (int k = 0; k < 2 * nt; ++k) if (h[k] != want) { if (bad++ < 8) printf("rep %d thread %d %s: %llx != %llx\n", rep, k / 2, k & 1 ? "after" : "before", (unsigned long long)h[k], (unsigned long long)want); }
    }
    printf("byval_probe: %ld mismatching checksums in 50 launches of %d threads\n", bad, nt);
    return bad ? 1 : 0;
}
