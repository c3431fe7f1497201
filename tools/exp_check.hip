#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__device__ __forceinline__ double exp_kernel(double x) {
    const double n = __builtin_rint(x * __longlong_as_double(0x3ff71547652b82feLL));
    double r = __builtin_fma(__longlong_as_double(0xbfe62e42fefa39efLL), n, x);
    r = __builtin_fma(__longlong_as_double(0xbc7abc9e3b39803fLL), n, r);
    double p = __builtin_fma(__longlong_as_double(0x3e5ade156a5dcb37LL), r, __longlong_as_double(0x3e928af3fca7ab0cLL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3ec71dee623fde64LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3efa01997c89e6b0LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3f2a01a014761f6eLL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3f81111111122322LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3fa55555555502a1LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3fc5555555555511LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3fe000000000000bLL));
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return ldexp(p, (int)n);
}
__global__ void k(const double *x, double *a, double *b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = exp(x[i]); b[i] = exp_kernel(x[i]); }
}
int main() {
    const int n = 1 << 22;
    double *hx = new double[n], *ha = new double[n], *hb = new double[n];
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double u = (st >> 11) * (1.0 / 9007199254740992.0);
        int m = i & 7;
        hx[i] = (m < 4) ? -60.0 * u : (m < 6) ? -760.0 * u : (m == 6) ? -1e-3 * u : 1e-9 * (u - 0.5);
    }
    hx[0] = 0.0; hx[1] = -745.2; hx[2] = -800.0; hx[3] = -1200.0; hx[4] = NAN; hx[5] = -708.4; hx[6] = -1074.9; hx[7] = -1e-300;
    double *dx, *da, *db;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    hipMemcpy(ha, da, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 8, hipMemcpyDeviceToHost);
    long diff = 0;
    for (int i = 0; i < n; ++i) if (memcmp(&ha[i], &hb[i], 8) != 0 && !(std::isnan(ha[i]) && std::isnan(hb[i]))) { if (diff < 5) printf("x=%.17g exp=%.17g mine=%.17g\n", hx[i], ha[i], hb[i]); ++diff; }
    printf("EXPCHECK n=%d bitwise differences=%ld; edge: %g %g %g %g %g\n", n, diff, hb[1], hb[2], hb[3], hb[4], hb[6]);
    return diff != 0;
}
