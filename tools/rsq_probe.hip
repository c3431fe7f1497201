// accuracy of v_rsq_f64 + k Newton steps (tools probe; hipcc --offload-arch=gfx950 tools/rsq_probe.hip -o tools/rsq_probe)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *a, double *y0, double *y1, double *y2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i];
    double y = __builtin_amdgcn_rsq(x);
    y0[i] = y;
    double hx = 0.5 * x;
    y = y * (1.5 - hx * y * y);
    y1[i] = y;
    y = y * (1.5 - hx * y * y);
    y2[i] = y;
    // Halley-type cubic step from the raw rsq
    double yh = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * yh), yh, 1.0);
    yh = fma(yh, e * fma(0.375, e, 0.5), yh);
    y0[i] = yh;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> a(n), r0(n), r1(n), r2(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = std::ldexp(1.0 + (s >> 11) * 0x1.0p-53, (int)(s % 40) - 20); }
    double *da, *d0, *d1, *d2;
    hipMalloc(&da, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, d0, d1, d2, n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double ex = 1.0L / sqrtl((long double)a[i]);
        e0 = fmax(e0, (double)fabsl((r0[i] - ex) / ex)); e1 = fmax(e1, (double)fabsl((r1[i] - ex) / ex));
        e2 = fmax(e2, (double)fabsl((r2[i] - ex) / ex));
    }
    printf("max rel err: Halley step %.3e, +1 Newton %.3e, +2 Newton %.3e (eps = 1.1e-16)\n", e0, e1, e2);
    return 0;
}
