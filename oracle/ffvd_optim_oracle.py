"""CPU restatement of the optimiser / sampler steps of the FFVD training loop -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.  Parity status: unpinned
(the reference has no tests; see ffvd_oracle.py).

  sghmc_step   BaseModel.generate_update_step, /root/reference vfegpssm/base_model.py:143-179, op by op.
  adam_step    tf.compat.v1.train.AdamOptimizer(self.adam_lr).minimize(self.nll), dgp_model.py:303-305.  TensorFlow
               is a third-party dependency the reference does not vendor or pin (README.md:18-21); this restates the
               optimiser's published rule (Kingma & Ba 2015, Algorithm 1 with the "epsilon hat" form TensorFlow
               documents): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); m <- b1 m + (1 - b1) g; v <- b2 v + (1 - b2) g^2;
               theta <- theta - lr_t m / (sqrt(v) + eps); defaults b1 = 0.9, b2 = 0.999, eps = 1e-8.
"""
from __future__ import annotations

import numpy as np


def decayed_learning_rate(global_step=1):
    return 0.003 * (0.95 ** (global_step / 1000))             # base_model.py:190


def adam_step(theta, grad, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """Returns (theta_new, m_new, v_new); t is the 1-based step count of THIS update."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m_new = beta1 * m + (1.0 - beta1) * grad
    v_new = beta2 * v + (1.0 - beta2) * (grad * grad)
    return theta - lr_t * m_new / (np.sqrt(v_new) + eps), m_new, v_new


def sghmc_step(theta, grad, xi, g, g2, p, noise, epsilon, mdecay, X_N, burn_in):
    """Returns (theta, xi, g, g2, p) after one burn_in_op (burn_in=True) or sample_op (False).
    `noise` replaces tf.random.normal(tf.shape(theta)) (:169)."""
    r_t = 1.0 / (xi + 1.0)                                               # :156
    g_t = (1.0 - r_t) * g + r_t * grad                                   # :157
    g2_t = (1.0 - r_t) * g2 + r_t * grad ** 2                            # :158
    xi_t = 1.0 + xi * (1.0 - g * g / (g2 + 1e-16))                       # :159
    Minv = 1.0 / (np.sqrt(g2 + 1e-16) + 1e-16)                           # :160
    epsilon_scaled = epsilon / np.sqrt(float(X_N))                       # :164
    noise_scale = 2.0 * epsilon_scaled ** 2 * mdecay * Minv              # :167
    sigma = np.sqrt(np.maximum(noise_scale, 1e-16))                      # :168
    sample_t = noise * sigma                                             # :169
    p_t = p - epsilon ** 2 * Minv * grad - mdecay * p + sample_t         # :170
    theta_t = theta + p_t                                                # :171
    if burn_in:                                                          # :179 burn_in_updates + sample_updates
        return theta_t, xi_t, g_t, g2_t, p_t
    return theta_t, xi, g, g2, p_t                                       # :178 sample_updates only
