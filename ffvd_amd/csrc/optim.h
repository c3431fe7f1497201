// Optimiser / sampler steps: fused multi-tensor elementwise kernels; see optim.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffvd {

constexpr int OPT_MAX_TENSORS = 10;
struct OptTensor {
    double *theta;          // parameter, updated in place
    const double *grad;
    double *s0, *s1, *s2, *s3;   // Adam: m, v.  SG-HMC: xi, g, g2, p
    const double *noise;    // SG-HMC: standard-normal draw, same shape as theta (injected for parity)
    int64_t n;
};
struct OptTable {
    OptTensor t[OPT_MAX_TENSORS];
    int count;
};

// tf.compat.v1.train.AdamOptimizer.minimize (dgp_model.py:303-305): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) is formed
// on the host; m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2; theta <- theta - lr_t m / (sqrt(v) + eps).
void launch_adam(hipStream_t stream, const OptTable &tab, double lr_t, double beta1, double beta2, double eps);

// BaseModel.generate_update_step (base_model.py:143-179); burn_in != 0 also advances xi, g, g2 (burn_in_op),
// otherwise only theta and p move (sample_op).  Every right-hand side reads the OLD state.
void launch_sghmc(hipStream_t stream, const OptTable &tab, double epsilon, double mdecay, double x_n, int burn_in);

}  // namespace ffvd
