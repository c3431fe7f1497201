// What does the per-chunk structure of the fp64 MFMA kernels cost on gfx950?  A 128 x 128 tile per 512-thread workgroup (8 wavefronts of
// 64 x 32, 8 accumulators each -- the shape of bwd_fused / the Gram kernel's standard body), operands out of a double-buffered LDS chunk
// of 16 k, two workgroups per CU.  MODE 0: no barrier, the same chunk over and over (the ceiling of this fragment / MFMA pattern);
// 1: a workgroup barrier per chunk; 2: barrier + the first k-step's fragments of the NEXT chunk loaded behind the last k-step's MFMAs of
// this one (fragments cross the barrier in registers); 3: as 1 plus the register-staged global loads and LDS stores of the next chunk.
// 4: as 3 with the across-barrier fragments of 2.
// Build + run: hipcc -O3 --offload-arch=gfx950 tools/probes/chunk_probe.hip -o tools/probes/chunk_probe && tools/probes/chunk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
constexpr int AT = 16, LD = 144, LDT = 145;        // LDT: the transposed A chunk's odd row stride (as gemm_rowmajor_a)
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(double *out, const double *A, const double *B, int nchunk, int ld) {
    __shared__ double As[2][AT][LDT], Bs[2][AT][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3, lr = lane & 15, lk = lane >> 4;
    for (int i = tid; i < 2 * AT * LD; i += 512) { (&As[0][0][0])[i] = A[i % 4096] + 1.0; (&Bs[0][0][0])[i] = B[i % 4096] - 1.0; }
    constexpr bool GL = (MODE == 3 || MODE == 4 || MODE == 5 || MODE >= 7) && MODE < 11, ST = MODE == 3 || MODE == 4 || MODE == 6 || MODE == 7;
    __syncthreads();
    d4 acc[4][2];
    for (int x = 0; x < 4; ++x) for (int y = 0; y < 2; ++y) acc[x][y] = (d4){0, 0, 0, 0};
    const double *Ag = A + (size_t)((MODE == 7 || MODE == 12) ? blockIdx.x / 4 : blockIdx.x) * 128 * ld + (size_t)(tid >> 2) * ld + 4 * (tid & 3);      // this workgroup's own 128 x ld panel
    const double *Bg = B + (size_t)(tid >> 6) * 128 + 2 * lane;                                           // B: ld x 128, shared
    double2 ra[2] = {{1.0, 2.0}, {3.0, 4.0}}, rb[2] = {{5.0, 6.0}, {7.0, 8.0}};
    double af[4], bf[2];
    auto frag = [&](int buf, int ks) {
#pragma unroll
        for (int x = 0; x < 4; ++x) af[x] = As[buf][4 * ks + lk][wr * 64 + 16 * x + lr];
#pragma unroll
        for (int y = 0; y < 2; ++y) bf[y] = Bs[buf][4 * ks + lk][wc * 32 + 16 * y + lr];
    };
    auto mm = [&]() {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = MF(af[x], bf[y], acc[x][y]);
    };
    auto gload = [&](int c) {
        const int cc = c % (ld / AT);               // (the panel is ld wide: walk it again and again)
        ra[0] = *reinterpret_cast<const double2 *>(Ag + (size_t)cc * AT); ra[1] = *reinterpret_cast<const double2 *>(Ag + (size_t)cc * AT + 2);
        for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const double2 *>(Bg + ((size_t)cc * AT + 8 * i) * 128);
    };
    auto lstore = [&](int buf) {
        const int il = tid >> 2, as = 4 * (tid & 3);
        As[buf][as][il] = ra[0].x; As[buf][as + 1][il] = ra[0].y; As[buf][as + 2][il] = ra[1].x; As[buf][as + 3][il] = ra[1].y;
        for (int i = 0; i < 2; ++i) *reinterpret_cast<double2 *>(&Bs[buf][(tid >> 6) + 8 * i][2 * lane]) = rb[i];
    };
    if (MODE == 2 || MODE == 4) frag(0, 0);
    for (int c = 0; c < nchunk; ++c) {
        const int buf = (MODE == 0) ? 0 : (c & 1);
        if (GL && MODE != 10 && c + 1 < nchunk) gload(c + 1);
        if (MODE == 10 && c == 0) gload(1);
        if (MODE == 2 || MODE == 4) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) { double a2[4], b2[2]; mm(); frag(buf, ks + 1); }      // (fragments of ks + 1 requested right behind the MFMAs of ks)
            if (MODE == 4 && c + 1 < nchunk) lstore(buf ^ 1);
            __syncthreads();
            mm();                                   // last k-step of this chunk: operands in registers
            frag(buf ^ 1, 0);                       // first k-step of the next chunk, behind those MFMAs
        } else if (MODE == 11 || MODE == 12) {
            // B chunk: LDS-DMA, no registers (rows of 128 doubles = 1 KiB = one wavefront-instruction; LDS row stride 144 doubles);
            // A chunk: registers, loaded TWO chunks ahead, stored behind k-step 0 into the buffer the last barrier freed.
            typedef __attribute__((address_space(3))) void lvoid;
            auto glds = [&](const double *base, const void *lds_row) {
                const unsigned dst = (unsigned)(uintptr_t)(lvoid *)lds_row, voff = (unsigned)(2 * lane * sizeof(double));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
            };
            if (c == 0) {
                ra[0] = *reinterpret_cast<const double2 *>(Ag + (size_t)(1 % (ld / AT)) * AT); ra[1] = *reinterpret_cast<const double2 *>(Ag + (size_t)(1 % (ld / AT)) * AT + 2);
            }
            if (c + 1 < nchunk) {
                const int cc = (c + 1) % (ld / AT);
                const int wv = __builtin_amdgcn_readfirstlane(wave);
                const double *brow = B + ((size_t)cc * AT + wv) * 128;
                glds(brow, &Bs[buf ^ 1][wv][0]);
                glds(brow + (size_t)8 * 128, &Bs[buf ^ 1][wv + 8][0]);
            }
            frag(buf, 0); mm();
            if (c + 1 < nchunk) {
                const int il = tid >> 2, as = 4 * (tid & 3);
                As[buf ^ 1][as][il] = ra[0].x; As[buf ^ 1][as + 1][il] = ra[0].y; As[buf ^ 1][as + 2][il] = ra[1].x; As[buf ^ 1][as + 3][il] = ra[1].y;
            }
            if (c + 2 < nchunk) {
                const int cc = (c + 2) % (ld / AT);
                ra[0] = *reinterpret_cast<const double2 *>(Ag + (size_t)cc * AT); ra[1] = *reinterpret_cast<const double2 *>(Ag + (size_t)cc * AT + 2);
            }
            frag(buf, 1); mm();
            frag(buf, 2); mm();
            frag(buf, 3); mm();
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");        // the two DMAs have landed (the A loads of chunk c + 2, issued behind them, may still fly)
            __syncthreads();
        } else if (MODE == 13) {
            // both operands K-major in memory (the Gram kernel's case): both chunks by LDS-DMA one chunk ahead, no staging registers at all
            typedef __attribute__((address_space(3))) void lvoid;
            auto glds = [&](const double *base, const void *lds_row) {
                const unsigned dst = (unsigned)(uintptr_t)(lvoid *)lds_row, voff = (unsigned)(2 * lane * sizeof(double));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
            };
            if (c + 1 < nchunk) {
                const int cc = (c + 1) % (ld / AT), wv = __builtin_amdgcn_readfirstlane(wave);
                const double *brow = B + ((size_t)cc * AT + wv) * 128;
                const double *arow = A + (size_t)(blockIdx.x % 64) * ld * 128 + ((size_t)cc * AT + wv) * 128;       // A^T: ld rows of 128, a 1 MB panel per 8 workgroups
                glds(brow, &Bs[buf ^ 1][wv][0]);
                glds(brow + (size_t)8 * 128, &Bs[buf ^ 1][wv + 8][0]);
                glds(arow, &As[buf ^ 1][wv][0]);
                glds(arow + (size_t)8 * 128, &As[buf ^ 1][wv + 8][0]);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { frag(buf, ks); mm(); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else if (MODE == 8 || MODE == 9 || MODE == 10) {
            // 8: the next chunk's LDS stores in front of the last k-step's MFMAs (its fragments are loaded first), barrier behind them
            // 9: the stores in two halves, behind k-steps 1 and 2;  10: stores behind k-step 0 (the loads were issued a whole chunk earlier: distance 2)
            frag(buf, 0); mm();
            if (MODE == 10 && c + 1 < nchunk) lstore(buf ^ 1);
            frag(buf, 1); mm();
            if (MODE == 9 && c + 1 < nchunk) {
                const int il = tid >> 2, as = 4 * (tid & 3);
                As[buf ^ 1][as][il] = ra[0].x; As[buf ^ 1][as + 1][il] = ra[0].y; As[buf ^ 1][as + 2][il] = ra[1].x; As[buf ^ 1][as + 3][il] = ra[1].y;
            }
            frag(buf, 2); mm();
            if (MODE == 9 && c + 1 < nchunk) { for (int i = 0; i < 2; ++i) *reinterpret_cast<double2 *>(&Bs[buf ^ 1][(tid >> 6) + 8 * i][2 * lane]) = rb[i]; }
            frag(buf, 3);
            if (MODE == 8 && c + 1 < nchunk) lstore(buf ^ 1);
            mm();
            __syncthreads();
            if (MODE == 10 && c + 2 < nchunk) gload(c + 2);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { frag(buf, ks); mm(); }
            if (ST && c + 1 < nchunk) lstore(buf ^ 1);
            if (MODE != 0) __syncthreads();
        }
    }
    double s = 0;
    for (int x = 0; x < 4; ++x) for (int y = 0; y < 2; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
    out[(size_t)blockIdx.x * 512 + tid] = s + af[0] + bf[0] + ra[0].x + rb[0].x;
}

// The other shape: ONE 512-thread workgroup per CU on a 256 x 128 tile, 8 wavefronts of 64 x 64 (16 accumulators, 8 fragment reads per
// 16 MFMAs instead of 6 per 8), two wavefronts per SIMD with 256 registers each.  MODE 0: no barrier; 1: barrier per chunk; 3: + staging.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k3(double *out, const double *A, const double *B, int nchunk, int ld) {
    __shared__ double As[2][AT][256 + 17], Bs[2][AT][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4;
    for (int i = tid; i < 2 * AT * 273; i += 512) (&As[0][0][0])[i] = A[i % 4096] + 1.0;
    for (int i = tid; i < 2 * AT * LD; i += 512) (&Bs[0][0][0])[i] = B[i % 4096] - 1.0;
    __syncthreads();
    d4 acc[4][4];
    for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0, 0, 0, 0};
    const double *Ag = A + (size_t)blockIdx.x * 256 * ld + (size_t)(tid >> 1) * ld + 8 * (tid & 1);      // 256 rows x 16 k: 8 k-values per thread
    const double *Bg = B + (size_t)(tid >> 5) * 128 + 4 * (tid & 31);                                      // 16 rows x 128 columns: 4 per thread
    double2 ra[4] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}}, rb[2] = {{1, 2}, {3, 4}};
    for (int c = 0; c < nchunk; ++c) {
        const int buf = (MODE == 0) ? 0 : (c & 1);
        if (MODE == 3 && c + 1 < nchunk) {
            const int cc = (c + 1) % (ld / AT);
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const double2 *>(Ag + (size_t)cc * AT + 2 * i);
            for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const double2 *>(Bg + (size_t)cc * AT * 128 + 2 * i);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            double af[4], bf[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) af[x] = As[buf][4 * ks + lk][wr * 64 + 16 * x + lr];
#pragma unroll
            for (int y = 0; y < 4; ++y) bf[y] = Bs[buf][4 * ks + lk][wc * 64 + 16 * y + lr];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = MF(af[x], bf[y], acc[x][y]);
        }
        if (MODE == 3 && c + 1 < nchunk) {
            const int il = tid >> 1, as = 8 * (tid & 1);
            for (int i = 0; i < 4; ++i) { As[buf ^ 1][as + 2 * i][il] = ra[i].x; As[buf ^ 1][as + 2 * i + 1][il] = ra[i].y; }
            for (int i = 0; i < 2; ++i) *reinterpret_cast<double2 *>(&Bs[buf ^ 1][tid >> 5][4 * (tid & 31) + 2 * i]) = rb[i];
        }
        if (MODE != 0) __syncthreads();
    }
    double s = 0;
    for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
    out[(size_t)blockIdx.x * 512 + tid] = s + ra[0].x + rb[0].x;
}
template <int MODE> void run3(double *out, double *A, double *B, const char *name) {
    const int CU = 256, nchunk = 2048, ld = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k3<MODE>, dim3(CU), dim3(512), 0, 0, out, A, B, nchunk, ld);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k3<MODE>, dim3(CU), dim3(512), 0, 0, out, A, B, nchunk, ld);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double fl = (double)CU * 8 * (double)nchunk * 4 * 16 * 2048.0;
    printf("%-100s %.1f TFLOP/s (%.0f %% of 78.6)\n", name, fl / ms / 1e9, fl / ms / 1e9 / 78.6 * 100);
}
template <int MODE> void run(double *out, double *A, double *B, const char *name) {
    const int CU = 256, nchunk = 2048, ld = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(CU * 2), dim3(512), 0, 0, out, A, B, nchunk, ld);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(CU * 2), dim3(512), 0, 0, out, A, B, nchunk, ld);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double fl = (double)CU * 2 * 8 * (double)nchunk * 4 * 8 * 2048.0;
    printf("%-100s %.1f TFLOP/s (%.0f %% of 78.6)\n", name, fl / ms / 1e9, fl / ms / 1e9 / 78.6 * 100);
}
int main() {
    const int ld = 1024;
    setvbuf(stdout, nullptr, _IONBF, 0);
    double *out = nullptr, *A = nullptr, *B = nullptr;
    if (hipMalloc(&out, 512 * 512 * 8) != hipSuccess || hipMalloc(&A, (size_t)512 * 128 * ld * 8) != hipSuccess ||      // A: a 128 x 1024 panel per workgroup (512 MB)
        hipMalloc(&B, (size_t)ld * 128 * 8) != hipSuccess) { printf("allocation failed\n"); return 1; }                 // B: 1024 x 128
    if (hipMemset(A, 0, (size_t)512 * 128 * ld * 8) != hipSuccess || hipMemset(B, 0, (size_t)ld * 128 * 8) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("memset failed\n"); return 1; }
    run<0>(out, A, B, "0: fragments from LDS + MFMAs, no barrier");
    run<1>(out, A, B, "1: + a workgroup barrier per 16-deep chunk");
    run<2>(out, A, B, "2: barrier, next chunk's first fragments loaded behind this chunk's last MFMAs");
    run<3>(out, A, B, "3: barrier + register-staged global loads / LDS stores of the next chunk");
    run<4>(out, A, B, "4: as 3 with the across-barrier fragments of 2");
    run<5>(out, A, B, "5: barrier + the global loads only (nothing stored)");
    run<6>(out, A, B, "6: barrier + the LDS stores only (registers' constants)");
    run<7>(out, A, B, "7: as 3, four workgroups share an A panel (the column tiles of bwd_fused)");
    run<8>(out, A, B, "8: as 3, the stores in front of the last k-step's MFMAs, barrier behind them");
    run<9>(out, A, B, "9: as 3, the stores in two halves behind k-steps 1 and 2");
    run<10>(out, A, B, "10: as 3, the stores behind k-step 0, their loads issued a chunk earlier (same registers)");
    run<11>(out, A, B, "11: B chunk by LDS-DMA, A chunk in registers two chunks ahead, stored behind k-step 0");
    run<12>(out, A, B, "12: as 11, four workgroups share an A panel");
    run<13>(out, A, B, "13: both chunks by LDS-DMA one chunk ahead (operands K-major in memory, the Gram kernel's refill)");
    run3<0>(out, A, B, "one workgroup per CU, 8 wavefronts of 64 x 64 (16 accumulators): no barrier");
    run3<1>(out, A, B, "   + a workgroup barrier per chunk");
    run3<3>(out, A, B, "   + register-staged global loads / LDS stores of the next chunk");
    return 0;
}
