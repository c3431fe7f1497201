// What limits a realistic fp64 MFMA stream on gfx950?  Variants of a 4x4-tile inner product step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a,b,c) __builtin_amdgcn_mfma_f64_16x16x4f64(a,b,c,0,0,0)

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(double *out, const double *in, int iters) {
    __shared__ double lds[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += NT) lds[i] = in[i];
    __syncthreads();
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = lds[tid & 63 + 64 * i]; b[i] = lds[256 + (tid & 63) + 64 * i]; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 3) {   // operands re-read from LDS every step
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = lds[((it & 3) * 512) + (tid & 63) + 64 * i]; b[i] = lds[((it & 3) * 512) + 256 + (tid & 63) + 64 * i]; }
        }
        if (MODE == 0 || MODE == 1) {   // A-major order: same A, four B (as the compiler emitted for gram)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = MF(a[i], b[j], acc[i * 4 + j]);
        } else if (MODE == 2 || MODE == 3) {   // 4 accumulators only (one active column tile)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = MF(a[i], b[0], acc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = MF(a[i], b[1], acc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = MF(a[i], b[2], acc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = MF(a[i], b[3], acc[i]);
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * NT + tid] = s;
}
template <class F> float time_ms(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
template <int MODE, int NT> void run(double *out, double *in, int wgs_per_cu, const char *name) {
    const int CU = 256, iters = 4000;
    float ms = time_ms([&] { hipLaunchKernelGGL((k<MODE, NT>), dim3(CU * wgs_per_cu), dim3(NT), 0, 0, out, in, iters); }, 5);
    double fl = (double)CU * wgs_per_cu * (NT / 64) * (double)iters * 16 * 2048.0;
    printf("%-52s NT=%4d x%d WG/CU: %.1f TFLOP/s\n", name, NT, wgs_per_cu, fl / ms / 1e9);
}
int main() {
    double *out, *in;
    hipMalloc(&out, 256 * 8 * 1024 * 8); hipMalloc(&in, 2048 * 8); hipMemset(in, 0, 2048 * 8);
    run<0, 256>(out, in, 1, "16 acc, operands fixed (1 wave/SIMD)");
    run<0, 256>(out, in, 2, "16 acc, operands fixed (2 waves/SIMD, 2 WGs)");
    run<0, 512>(out, in, 1, "16 acc, operands fixed (2 waves/SIMD, 1 WG)");
    run<1, 256>(out, in, 1, "16 acc, operands from LDS each step (1 wave/SIMD)");
    run<1, 256>(out, in, 2, "16 acc, operands from LDS each step (2 waves/SIMD)");
    run<1, 512>(out, in, 1, "16 acc, operands from LDS each step (2w/SIMD, 1 WG)");
    run<2, 512>(out, in, 1, "4 acc reused every 4 MFMAs, fixed operands (2w/SIMD)");
    run<3, 512>(out, in, 1, "4 acc reused every 4 MFMAs, LDS operands (2w/SIMD)");
    run<3, 1024>(out, in, 1, "4 acc reused every 4 MFMAs, LDS operands (4w/SIMD)");
    run<1, 1024>(out, in, 1, "16 acc, LDS operands (4w/SIMD)");
    return 0;
}
