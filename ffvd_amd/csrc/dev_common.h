// Device-side helpers shared by the kernel sources of libffvd_hip.so (gfx950): the fp64 MFMA wrapper, the kernel-matrix
// element, wavefront-level LDS ordering, the Cholesky pivot, and the nll assembly.  Definitions moved here unchanged from
// kernels.hip (round 4) so that tiny.hip -- the one-launch iteration of the reference's own experiment size -- runs the SAME code.
#pragma once
#include "kernels.h"

namespace ffvd {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// D(16x16) += A(16x4) * B(4x16).  Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15];
// it owns D[(l >> 4) + 4 r][l & 15] in element r of the accumulator.
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}


// K(i,j) of one kernel from pre-scaled rows.  SE: variance * exp(-(-2 x.z + (|x|^2 + |z|^2)) / 2)
// (kernels_multi_output.py:180-181,247); LINEAR: sum_p (x_p * variance) * z_p (kernels.py:276).
// exp for the kernel matrices: the device library's algorithm and constants (argument reduction by ln 2 in two
// pieces, degree-11 polynomial, ldexp), minus its two range selects -- the argument -r^2/2 never comes near the
// overflow threshold, and ldexp already flushes results below the denormal range to zero.  In-range results are
// bit-identical to exp(); NaN propagates.  Six VALU instructions less per element of the 2.7e8-element K_fu build.
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

template <int KIND>
__device__ __forceinline__ double kernel_value(double dot, double xx, double zz, double variance) {
    if (KIND == 0) {
        double r2 = -2.0 * dot + (xx + zz);
        return variance * exp_kernel(-r2 / 2.0);
    }
    return dot;   // LINEAR: variance already folded into the x operand
}


__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// LDS hand-off between the lanes of ONE wavefront: order the LDS traffic for the compiler and the hardware.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}


// The same between LDS accesses of ONE wavefront that has global stores in flight: LDS serves a wavefront's accesses in order,
// so only the compiler needs telling (the fences above also wait for vmcnt(0), i.e. for the stores).
__device__ __forceinline__ void wave_lds_order() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// sqrt(x) and 1/sqrt(x) of tf.linalg.cholesky's pivot (conditionals_multi_output.py:28,162) from the hardware
// reciprocal square root (2^-24 accurate) and ONE cubic (Halley-type) step: with e = 1 - x y^2,
// y <- y (1 + e/2 + 3 e^2/8) is accurate to 1.4e-16 (tools/rsq_probe.hip; two Newton steps give 2.4e-16) in five
// dependent operations -- this chain sits 64 times on the critical path of every block step, the IEEE sqrt and
// divide expansions would be several times longer.  sqrt(x) = x y with one correction, off the critical path.
__device__ __forceinline__ void pivot_sqrt(const double ajj, double &piv, double &y) {
    y = __builtin_amdgcn_rsq(ajj);
    const double e = fma(-(ajj * y), y, 1.0);
    y = fma(y, e * fma(0.375, e, 0.5), y);
    piv = ajj * y;
    piv = fma(0.5 * y, fma(-piv, piv, ajj), piv);
}


// Priors + nll assembly (dgp_model.py:105-143, 259-297, 326-334) by ONE 256-thread workgroup: the body of finalize_kernel, shared with the
// one-launch small-problem iteration (tiny.hip), whose last workgroup runs it in place.  scratch: [4][10] doubles of LDS.
// NTHR: threads of the workgroup (256 = finalize_kernel; the small-problem kernel's workgroups have 256 or 512).
template <int N, int NWAVES>
__device__ __forceinline__ void block_sum_multi(double (&v)[N], double (*scratch)[N] /*[NWAVES][N]*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v[i] += __shfl_xor(v[i], m);
    }
    __syncthreads();                                   // a previous use of scratch is over
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) scratch[wave][i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double t = scratch[0][i];
#pragma unroll
        for (int w = 1; w < NWAVES; ++w) t += scratch[w][i];      // fixed order (NWAVES = 4: the order of block_sum_multi_256)
        v[i] = t;
    }
}

// Two halves: the ten parameter-only sums (finalize_priors: every thread returns with the totals), and the per-chain assembly on
// them (finalize_assemble).  finalize_kernel runs one after the other; the small-problem kernel forms the sums early, in a
// workgroup that is waiting anyway.
template <int NTHR>
__device__ __forceinline__ void finalize_priors(const FinalizeArgs &a, double (*scratch)[10] /*[NTHR / 64][10]*/, double (&sm)[10]) {
    const int tid = threadIdx.x;
    // shared priors and constants: ten small sums, every thread takes a strided share, one reduction for all
    // 0 |Z|^2  1 |U|^2  2 sum loglen^2  3 sum (logvar - log 0.05)^2  4 |log_Q|^2  5 |C|^2  6 |d|^2  7 |log_Rchols|^2
    // 8 sum_j log R_j  9 sum_d log sqrt(Q_d)
#pragma unroll
    for (int i = 0; i < 10; ++i) sm[i] = 0.0;
    if (a.shared_terms && a.prior_type == 1)
        for (int i = tid; i < a.M * a.P; i += NTHR) sm[0] += a.Z[i] * a.Z[i];
    if (a.branch == 0)
        for (int i = tid; i < a.M * a.Dl; i += NTHR) {
            const int m = i / a.Dl, dl = i % a.Dl;
            const double u = a.U[(size_t)m * a.D + a.d_begin + dl];
            sm[1] += u * u;
        }
    if (a.kind == 0)                    // Layer.prior_hyper dgp_model.py:123-130 over the local dims
        for (int i = tid; i < a.Dl * a.P; i += NTHR) {
            const double l = a.loglen[(size_t)a.d_begin * a.P + i];
            sm[2] += l * l;
        }
    for (int dl = tid; dl < a.Dl; dl += NTHR) {
        const double dv = a.logvar[a.d_begin + dl] - (a.kind == 0 ? LOG_PRIOR_VARIANCE_SE : LOG_PRIOR_VARIANCE_LIN);
        sm[3] += dv * dv;
        sm[9] += log(sqrt(exp(a.log_Q[a.d_begin + dl])));          // -sum_d log(Q_d ** 0.5) likelihoods.py:91 / :101
    }
    if (a.shared_terms) {               // hypaparameter_prior dgp_model.py:326-334
        for (int d = tid; d < a.D; d += NTHR) sm[4] += a.log_Q[d] * a.log_Q[d];
        for (int i = tid; i < a.D * a.Ydim; i += NTHR) sm[5] += a.CC[i] * a.CC[i];
        for (int j = tid; j < a.Ydim; j += NTHR) sm[6] += a.DD[j] * a.DD[j];
        for (int i = tid; i < a.Ydim * a.Ydim; i += NTHR) sm[7] += a.log_Rchols[i] * a.log_Rchols[i];
    }
    for (int j = tid; j < a.Ydim; j += NTHR) sm[8] += log(exp(a.log_Rchols[j]));    // -reduce_sum(log(Rchols)) likelihoods.py:101
    block_sum_multi<10, NTHR / 64>(sm, scratch);
}

template <int NTHR>
__device__ __forceinline__ void finalize_assemble(const FinalizeArgs &a, double (*scratch)[10] /*[NTHR / 64][10]*/, const double (&sm)[10]) {
    const int tid = threadIdx.x;
    const double Tn = (double)a.T;      // batch_size == Y_N == T for the full batch (dgp_model.py:261-262)
    const double prior_hyper = -sm[2] / 2.0 - sm[3] / 2.0;
    const double hyp = a.shared_terms ? (-sm[4] / 2.0 - sm[5] / 2.0 - sm[6] / 2.0 - sm[7] / 2.0) : 0.0;
    const double prior_z = (a.shared_terms && a.prior_type == 1) ? -sm[0] / 2.0 : 0.0;     // prior_Z dgp_model.py:108-109
    const double prior_u = (a.branch == 0) ? -0.5 * sm[1] : 0.0;                            // prior_U dgp_model.py:134-135
    const double logR = sm[8], logsqQ = sm[9];
    // per-chain assembly: one thread per chain (strided), then a fixed-order sum over chains
    double part[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int s = tid; s < a.S; s += NTHR) {
        const double *ct = a.chain_terms + (size_t)s * 8;
        double terms[7] = {0, 0, 0, 0, 0, 0, 0};
        double prior = prior_hyper + prior_u;
        if (a.shared_terms) {
            prior += prior_z + ct[3] + hyp;
            terms[1] = -(ct[0] + Tn * (-logR)) / Tn;                   // nll_log_likelihood :264
        }
        terms[0] = -prior / Tn;                                          // nll_part_prior :286 / :296
        terms[2] = -(ct[1] + Tn * (-logsqQ)) / Tn;                     // x_t_prior_Q :283-284 / :294
        terms[3] = -ct[2] / Tn;                                          // trace term :257 / :292
        if (a.branch == 1) {
            double term1 = 0.0, term2 = 0.0;
            for (int dl = 0; dl < a.Dl; ++dl) {
                const size_t bb = (size_t)s * a.Dl + dl;
                const double *ht = a.hterms + bb * 2;
                double logdet = ht[0];
                // log|I + L^-1 G L^-T / Q| = log|K + G/Q| - log|K|  with K = K_uu + jitter I
                if (a.route == 1 && !a.whitened) logdet -= a.kterms[2 * dl];
                if (a.route == 1 || a.fsq_from_trpart) {
                    // sum_t |F_t|^2 (= tr(K^-1 K_uf K_fu) on route 1): add it back to the trace term (:255)
                    double fsq = 0.0;
                    for (int t = 0; t < a.ntiles; ++t) fsq += a.trpart[bb * a.ntiles + t];
                    terms[3] += -(0.5 * fsq / exp(a.log_Q[a.d_begin + dl])) / Tn;
                }
                term1 += -0.5 * logdet;                                  // :253
                term2 += 0.5 * ht[1];                                    // :254
            }
            terms[4] = -term1 / Tn;                                      // :257
            terms[5] = -term2 / Tn;
        }
        terms[6] = terms[0] + terms[1] + terms[2] + terms[3] + terms[4] + terms[5];   // :288 / :297
        a.chain_nll[s] = terms[6];
        for (int i = 0; i < 7; ++i) part[i] += terms[i];
    }
    block_sum_multi<7, NTHR / 64>(part, reinterpret_cast<double(*)[7]>(&scratch[0][0]));
    // A failed or abandoned factorisation of THIS rank must be visible in the sums every rank receives (ffvd_elbo_allreduce,
    // ffvd_*_step_allreduce test them for finiteness): a bad pivot usually produces NaN by itself, an abandoned dataflow launch
    // (info = -1) leaves finite garbage.  The reference's counterpart is the error session.run raises (dgp_model.py:320-324).
    int bad = 0;
    for (int i = tid; i < a.ninfo; i += NTHR) bad |= (a.info[i] != 0);
    if (__syncthreads_or(bad))
#pragma unroll
        for (int i = 0; i < 7; ++i) part[i] = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
    for (int i = 0; i < 7; ++i)
        if (tid == i) a.out_terms[i] = part[i];
    if (tid == 7) a.out_terms[7] = a.shared_terms ? (double)a.S : 0.0;
}

template <int NTHR>
__device__ __forceinline__ void finalize_body(const FinalizeArgs &a, double (*scratch)[10] /*[NTHR / 64][10]*/) {
    double sm[10];
    finalize_priors<NTHR>(a, scratch, sm);
    finalize_assemble<NTHR>(a, scratch, sm);
}

}  // namespace ffvd
