"""Gaussian emission likelihood -- counterpart of vfegpssm/likelihoods.py (Gaussian part only).

Holds the emission parameters (C, d, log R-Cholesky) as NumPy arrays; the ELBO reductions that consume them
(predict_mean, logdensity_norm_diag*) run inside the fused HIP reduction kernel of the engine.
"""
from __future__ import annotations

import numpy as np


class Gaussian:
    """Gaussian.__init__ (likelihoods.py:12-61) for Y_dim == 1 style parameterisation:
    CC (X_output_dim, Y_dim), DD (Y_dim,), log_Rchols = log(RR_chol) (Y_dim, Y_dim)."""

    def __init__(self, Y_dim, X_output_dim, CC=None, DD=None, RR_chol=None, hyperparameter_sampling=False,
                 likelihood_traning=True):
        self.Y_dim, self.X_output_dim = int(Y_dim), int(X_output_dim)
        self.CC = np.ones((X_output_dim, Y_dim)) if CC is None else np.array(CC, dtype=np.float64)      # :17-19
        self.DD = np.zeros(Y_dim) if DD is None else np.array(DD, dtype=np.float64).reshape(Y_dim)      # :21-23
        if RR_chol is None:
            self.log_Rchols = np.full((Y_dim, Y_dim), np.log(0.1))                                      # :52
        else:
            self.log_Rchols = np.log(np.array(RR_chol, dtype=np.float64)).reshape(Y_dim, Y_dim)         # :54
        if self.CC.shape != (self.X_output_dim, self.Y_dim):
            raise ValueError(f"CC: expected {(self.X_output_dim, self.Y_dim)}, got {self.CC.shape}")
        self.trainable = bool(likelihood_traning) and not hyperparameter_sampling

    @property
    def Rchols(self):
        return np.exp(self.log_Rchols)                                                                   # :55
