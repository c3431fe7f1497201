// Standalone probe for gfx950: (1) verifies the v_mfma_f64_16x16x4_f64 operand/accumulator lane maps the
// kernels rely on, (2) measures the sustained fp64 MFMA and fp64 VALU FMA rates (the roofline peaks),
// (3) measures a plain HBM copy.  Build: hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double *A /*16x4*/, const double *B /*4x16*/, double *D /*16x16*/) {
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate_kernel(double *out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// in-kernel clock: shader cycles (s_memtime) per 100 MHz reference tick (s_memrealtime) around an MFMA loop
__global__ __launch_bounds__(256) void mfma_clock_kernel(double *out, unsigned long long *stamps, int iters) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// co-execution: per workgroup of 512 threads, waves 0-3 run fp64 MFMA, waves 4-7 run fp64 VALU FMA (one of each per SIMD)
__global__ __launch_bounds__(512) void coexec_kernel(double *out, int iters_mfma, int iters_fma, int mode) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    const bool do_mfma = (mode == 0) ? (wave < 4) : (mode == 1);
    if (do_mfma) {
        d4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double x[16];
        for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-9 + i;
        const double a = 1.0000001, b = 1e-9;
        for (int it = 0; it < iters_fma; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
        }
        for (int i = 0; i < 16; ++i) s += x[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_rate_kernel(double *out, int iters) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-9 + i;
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void exp_rate_kernel(double *out, int iters) {
    double x[4];
    for (int i = 0; i < 4; ++i) x[i] = -1e-3 * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = exp(x[i]) - 1.0001;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

__global__ void copy_kernel(const double2 *in, double2 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) out[i] = in[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class F>
float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    // ---- layout ----
    std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;     // asymmetric integers
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 + j * 5 + (j * j) % 3;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += (D[i] != ref[i]);
    printf("mfma_f64_16x16x4 layout (A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4r][l&15]): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    // ---- rates ----
    const int CU = prop.multiProcessorCount;
    double *dout; CK(hipMalloc(&dout, (size_t)CU * 8 * 1024 * 8));
    const int iters = 2000;
    for (int wpc : {1, 2}) {     // workgroups of 256 threads per CU (1 or 2 waves per SIMD)
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel<8>, dim3(CU * wpc), dim3(256), 0, 0, dout, iters); }, 5);
        double fl = (double)CU * wpc * 4 * iters * 8 * 2048.0;
        printf("fp64 MFMA 16x16x4, 8 accumulators, %d wave/SIMD: %.1f TFLOP/s\n", wpc, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(CU * wpc), dim3(256), 0, 0, dout, iters); }, 5);
        fl = (double)CU * wpc * 4 * iters * 2 * 2048.0;
        printf("fp64 MFMA 16x16x4, 2 accumulators, %d wave/SIMD: %.1f TFLOP/s\n", wpc, fl / ms / 1e9);
    }
    {   // sustained run (~1 s) + in-kernel clock
        unsigned long long *dst_; CK(hipMalloc(&dst_, (size_t)CU * 4 * 16));
        for (int wpc : {1, 2, 4}) {
            const int it2 = 40000;
            float ms = time_ms([&] { hipLaunchKernelGGL(mfma_clock_kernel, dim3(CU * wpc), dim3(256), 0, 0, dout, dst_, it2); }, 10);
            std::vector<unsigned long long> st((size_t)CU * wpc * 2);
            CK(hipMemcpy(st.data(), dst_, st.size() * 8, hipMemcpyDeviceToHost));
            double cyc = 0, ref = 0;
            for (int i = 0; i < CU * wpc; ++i) { cyc += st[2 * i]; ref += st[2 * i + 1]; }
            double fl = (double)CU * wpc * 4 * (double)it2 * 8 * 2048.0;
            printf("sustained fp64 MFMA, %d wave/SIMD, %.0f ms/launch: %.1f TFLOP/s, in-kernel clock %.2f GHz, %.1f cycles per MFMA per wave\n",
                   wpc, ms, fl / ms / 1e9, cyc / ref * 0.1, cyc / (CU * wpc) / ((double)it2 * 8));
        }
    }
    {   // MFMA and VALU fp64 side by side on every SIMD
        const int im = 20000, iff = 20000 * 8 * 140 / (16 * 8);   // roughly equal run times when alone
        float t_m = time_ms([&] { hipLaunchKernelGGL(coexec_kernel, dim3(CU), dim3(512), 0, 0, dout, im, iff, 1); }, 5);
        float t_f = time_ms([&] { hipLaunchKernelGGL(coexec_kernel, dim3(CU), dim3(512), 0, 0, dout, im, iff, 2); }, 5);
        float t_c = time_ms([&] { hipLaunchKernelGGL(coexec_kernel, dim3(CU), dim3(512), 0, 0, dout, im, iff, 0); }, 5);
        double fl_m8 = (double)CU * 8 * (double)im * 8 * 2048.0, fl_f8 = (double)CU * 512 * (double)iff * 16 * 2.0;
        printf("co-exec: 8 MFMA waves/CU alone %.1f TF (%.2f ms); 8 FMA waves/CU alone %.1f TF (%.2f ms); 4 MFMA + 4 FMA waves/CU: %.2f ms => MFMA %.1f TF + VALU %.1f TF = %.1f TF\n",
               fl_m8 / t_m / 1e9, t_m, fl_f8 / t_f / 1e9, t_f, t_c, fl_m8 / 2 / t_c / 1e9, fl_f8 / 2 / t_c / 1e9, (fl_m8 + fl_f8) / 2 / t_c / 1e9);
    }
    for (int wpc : {1, 2, 4}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_rate_kernel, dim3(CU * wpc), dim3(256), 0, 0, dout, iters); }, 5);
        double fl = (double)CU * wpc * 256 * (double)iters * 16 * 2.0;
        printf("fp64 VALU FMA, %d wave/SIMD: %.1f TFLOP/s\n", wpc, fl / ms / 1e9);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(exp_rate_kernel, dim3(CU * 4), dim3(256), 0, 0, dout, 500); }, 5);
        double n = (double)CU * 4 * 256 * 500.0 * 4;
        printf("fp64 exp(): %.2f Gexp/s chip-wide (4 waves/SIMD)\n", n / ms / 1e6);
    }
    // ---- HBM copy ----
    size_t nbytes = (size_t)2 << 30;
    double2 *src, *dst; CK(hipMalloc(&src, nbytes)); CK(hipMalloc(&dst, nbytes));
    CK(hipMemset(src, 1, nbytes));
    float ms = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(CU * 8), dim3(256), 0, 0, src, dst, nbytes / 16); }, 5);
    printf("HBM copy 2 GiB: %.2f TB/s (read+write)\n", 2.0 * nbytes / ms / 1e9);
    return bad ? 2 : 0;
}
