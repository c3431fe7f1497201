// Pure write stream of 2 GiB in the K_fu build's pattern (a workgroup = a 64 x 64 tile of a row-major T x M matrix, M = 512; a wavefront
// writes 16 rows): 8-byte stores (one column per lane, 512 B per row and instruction) against 16-byte stores (two columns per lane, half a
// wavefront per row), plain and nontemporal.  hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W16, int NT>
__global__ __launch_bounds__(256) void wr(double *out, int Mp, double v) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t t0 = (size_t)blockIdx.x * 64, m0 = (size_t)blockIdx.y * 64;
    if (!W16) {
        double *o = out + (t0 + wave * 16) * Mp + m0 + lane;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            if (NT) __builtin_nontemporal_store(v + i, &o[(size_t)i * Mp]);
            else o[(size_t)i * Mp] = v + i;
        }
    } else {
        // two rows per instruction: lanes 0-31 row 2 i, lanes 32-63 row 2 i + 1, 16 bytes each
        double2 *o = reinterpret_cast<double2 *>(out + (t0 + wave * 16 + (lane >> 5)) * Mp + m0 + 2 * (lane & 31));
#pragma unroll 4
        for (int i = 0; i < 8; ++i) {
            double2 *q = reinterpret_cast<double2 *>(reinterpret_cast<double *>(o) + (size_t)2 * i * Mp);
            const double2 val = make_double2(v + i, v - i);
            if (NT) { __builtin_nontemporal_store(val.x, &q->x); __builtin_nontemporal_store(val.y, &q->y); }
            else *q = val;
        }
    }
}
// whole rows: a workgroup writes 8 rows of 512 doubles, a wavefront two full rows (8 KB contiguous) as 16-byte stores
template <int NT>
__global__ __launch_bounds__(256) void wr_rows(double *out, int Mp, double v) {
    const int tid = threadIdx.x;
    double *o = out + ((size_t)blockIdx.x * 8 + (tid >> 5)) * Mp + (tid & 31) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (NT) { __builtin_nontemporal_store(v + i, &o[64 * i]); __builtin_nontemporal_store(v - i, &o[64 * i + 1]); }
        else *reinterpret_cast<double2 *>(o + 64 * i) = make_double2(v + i, v - i);
    }
}
template <int NT>
static void run_rows(const char *label, double *buf, size_t T, int Mp) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wr_rows<NT>), dim3(T / 8), dim3(256), 0, 0, buf, Mp, 1.0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((wr_rows<NT>), dim3(T / 8), dim3(256), 0, 0, buf, Mp, 1.0 + i);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms per pass  %.2f TB/s\n", label, ms / 20, (double)T * Mp * 8 / (ms / 20) / 1e9);
}
template <int W16, int NT>
static void run(const char *label, double *buf, size_t T, int Mp) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wr<W16, NT>), dim3(T / 64, Mp / 64), dim3(256), 0, 0, buf, Mp, 1.0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((wr<W16, NT>), dim3(T / 64, Mp / 64), dim3(256), 0, 0, buf, Mp, 1.0 + i);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms per pass  %.2f TB/s\n", label, ms / 20, (double)T * Mp * 8 / (ms / 20) / 1e9);
}
int main() {
    const int Mp = 512;
    const size_t T = (size_t)128 * 4096;            // 128 units of config 2: 2.15 GB
    double *buf;
    if (hipMalloc(&buf, T * Mp * 8) != hipSuccess) return 1;
    run<0, 0>("8-byte stores", buf, T, Mp);
    run<0, 1>("8-byte nontemporal stores", buf, T, Mp);
    run<1, 0>("16-byte stores", buf, T, Mp);
    run<1, 1>("16-byte (2 x 8 nontemporal) stores", buf, T, Mp);
    run_rows<0>("whole rows (8 KB per wavefront), 16-byte", buf, T, Mp);
    run_rows<1>("whole rows, nontemporal", buf, T, Mp);
    return 0;
}
