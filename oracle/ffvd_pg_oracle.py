"""Particle-Gibbs sweep for the latent trajectory -- TEST INFRASTRUCTURE ONLY (CPU restatement, parity unpinned).

Restates the INTENT of `BaseModel.PG_for_X_speedup` (vfegpssm/base_model.py:78-138).  As written the reference op is
a no-op: `TensorArray.write` results are discarded (:115), the final `tf.compat.v1.assign` is never run (:137) and the
function returns `tf.ones(1)` (:138), so `gp_x_sampling()` (models.py:156-158) leaves X unchanged (SURVEY.md 3.5,
Appendix B item 6).  What the code sets out to do, line by line, is restated here with every random draw injected:

  particles_0 ~ N(0, I), PG_particles - 1 of them                                          (:79, injected as x0)
  for tt in 0 .. X_N - 2:
      x_t   = particles_tt                                                                  (:91)
      f     = conditional_after_kernel_precalculation(Lm, [x_t, c_tt], Z, kern, U, white)   (:93-97)
      x_t+1 = x_t + f_mu + eps_tt * sqrt(f_var + Q)                                         (:99-101, eps injected)
      w_i   = logdensity_norm(Y[tt], predict_mean(x_t+1,i), Rchols)   for the new particles (:105-106)
      w_N   = logdensity_norm(Y[tt], predict_mean(X[tt+1]), Rchols)   for the reference     (:108-109)
      idx   ~ Categorical(logits = w), PG_particles - 1 draws                               (:113, uniforms injected)
      particles_tt+1 = [x_t+1,1 .. x_t+1,N-1, X[tt+1]][idx]                                 (:111,:114-115)
  final_index ~ uniform{0..PG_particles-1}; if < PG_particles - 1: X <- particles[:, final_index]   (:135-137)

The categorical draw is the inverse CDF of softmax(w) at the injected uniform u in [0, 1): the first k whose
cumulative probability exceeds u (cumulative sums formed sequentially, in index order).
"""
from __future__ import annotations

import numpy as np

from . import ffvd_oracle as orc


def categorical_from_uniform(logits, u):
    """Indices k(u) = min{k : cdf_k > u} of softmax(logits); cdf by sequential summation, last entry forced to cover 1."""
    w = np.exp(logits - np.max(logits))
    cdf = np.cumsum(w)                       # sequential, index order
    tot = cdf[-1]
    idx = np.searchsorted(cdf, np.asarray(u) * tot, side="right")
    return np.minimum(idx, len(logits) - 1)


def pg_sweep(Lm_inverse_seq, Z, kern, U_val, X_ref, Y, control_inputs, CC, DD, Rchols, Q, x0, eps, unif):
    """One sweep.  X_ref (X_N, D); Y (>= X_N - 1, Ydim); control_inputs (>= X_N - 1, C) or None; Rchols (Ydim, Ydim)
    lower-triangular matrix as `likelihood.Rchols` (= exp(log_Rchols)); Q (D,); x0 (N-1, D); eps (X_N-1, N-1, D);
    unif (X_N-1, N-1).  Returns particles (X_N, N-1, D) -- `resampled_X` of :133 -- and the indices drawn (X_N-1, N-1)."""
    X_ref = np.asarray(X_ref, dtype=np.float64)
    XN, D = X_ref.shape
    n1 = x0.shape[0]
    parts = np.zeros((XN, n1, D))
    idxs = np.zeros((XN - 1, n1), dtype=np.int64)
    parts[0] = x0                                                                         # :87
    has_c = control_inputs is not None and np.asarray(control_inputs).shape[1] > 0
    for tt in range(XN - 1):
        x_t = parts[tt]                                                                   # :91
        xc = np.concatenate((x_t, np.repeat(np.asarray(control_inputs)[tt][None, :], n1, axis=0)), axis=1) if has_c else x_t
        f_mu, f_var = orc.conditional_after_kernel_precalculation(Lm_inverse_seq, xc, Z, kern, U_val)   # :95-97
        f_mu = f_mu + x_t                                                                 # :99
        x_next = f_mu + eps[tt] * np.sqrt(f_var + Q[None, :])                             # :101
        w = np.empty(n1 + 1)
        w[:n1] = orc.logdensity_norm(Y[tt][None, :], orc.predict_mean(x_next, CC, DD), Rchols)          # :105-106
        w[n1] = orc.logdensity_norm(Y[tt][None, :], orc.predict_mean(X_ref[tt + 1][None, :], CC, DD), Rchols)[0]   # :108-109
        cand = np.concatenate((x_next, X_ref[tt + 1][None, :]), axis=0)                   # :103,:111
        idx = categorical_from_uniform(w, unif[tt])                                       # :113
        idxs[tt] = idx
        parts[tt + 1] = cand[idx]                                                         # :114-115
    return parts, idxs


def select_trajectory(X_ref, particles, final_index):
    """:135-137: the slot `final_index` of the stored per-time particle states replaces X unless it is the reference's."""
    n1 = particles.shape[1]
    return particles[:, final_index].copy() if final_index < n1 else np.asarray(X_ref).copy()
