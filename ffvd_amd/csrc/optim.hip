// Optimiser / sampler steps as fused multi-tensor elementwise kernels.
//
// One launch updates every parameter array of the model: blockIdx.y selects the tensor from a by-value table,
// blockIdx.x strides over its elements.  HBM-bound streaming work of a few hundred KB: the point of fusing is one
// launch instead of 8 x (5..9) elementwise TensorFlow ops per step.
#include "optim.h"

// keep every multiply and add a separate IEEE operation: the updates then agree with the NumPy restatement to the
// last bit instead of differing by fused-multiply-add roundings in cancelling sums such as b1*m + (1-b1)*g
#pragma clang fp contract(off)

namespace ffvd {

__global__ __launch_bounds__(256) void adam_kernel(OptTable tab, double lr_t, double b1, double b2, double eps) {
    const OptTensor t = tab.t[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (int64_t)gridDim.x * 256) {
        const double g = t.grad[i];
        const double m = b1 * t.s0[i] + (1.0 - b1) * g;
        const double v = b2 * t.s1[i] + (1.0 - b2) * (g * g);
        t.s0[i] = m;
        t.s1[i] = v;
        t.theta[i] -= lr_t * m / (sqrt(v) + eps);
    }
}

void launch_adam(hipStream_t stream, const OptTable &tab, double lr_t, double beta1, double beta2, double eps) {
    if (tab.count <= 0) return;
    int64_t nmax = 1;
    for (int i = 0; i < tab.count; ++i) nmax = tab.t[i].n > nmax ? tab.t[i].n : nmax;
    int64_t blocks = (nmax + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks, tab.count), dim3(256), 0, stream, tab, lr_t, beta1, beta2, eps);
}

__global__ __launch_bounds__(256) void sghmc_kernel(OptTable tab, double epsilon, double mdecay, double x_n, int burn_in) {
    const OptTensor t = tab.t[blockIdx.y];
    const double eps_scaled = epsilon / sqrt(x_n);                         // base_model.py:164
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (int64_t)gridDim.x * 256) {
        const double grad = t.grad[i], xi = t.s0[i], g = t.s1[i], g2 = t.s2[i], p = t.s3[i];
        const double r_t = 1.0 / (xi + 1.0);                               // :156
        const double g_t = (1.0 - r_t) * g + r_t * grad;                   // :157
        const double g2_t = (1.0 - r_t) * g2 + r_t * (grad * grad);        // :158
        const double xi_t = 1.0 + xi * (1.0 - g * g / (g2 + 1e-16));       // :159
        const double Minv = 1.0 / (sqrt(g2 + 1e-16) + 1e-16);              // :160
        const double noise_scale = 2.0 * (eps_scaled * eps_scaled) * mdecay * Minv;     // :167
        const double sigma = sqrt(fmax(noise_scale, 1e-16));               // :168
        const double sample_t = t.noise[i] * sigma;                        // :169
        const double p_t = p - (epsilon * epsilon) * Minv * grad - mdecay * p + sample_t;   // :170
        t.theta[i] += p_t;                                                 // :171
        t.s3[i] = p_t;
        if (burn_in) { t.s0[i] = xi_t; t.s1[i] = g_t; t.s2[i] = g2_t; }    // :178-179
    }
}

void launch_sghmc(hipStream_t stream, const OptTable &tab, double epsilon, double mdecay, double x_n, int burn_in) {
    if (tab.count <= 0) return;
    int64_t nmax = 1;
    for (int i = 0; i < tab.count; ++i) nmax = tab.t[i].n > nmax ? tab.t[i].n : nmax;
    int64_t blocks = (nmax + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sghmc_kernel, dim3((unsigned)blocks, tab.count), dim3(256), 0, stream, tab, epsilon, mdecay, x_n,
                       burn_in);
}

}  // namespace ffvd
