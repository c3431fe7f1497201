"""Gaussian emission likelihood -- counterpart of vfegpssm/likelihoods.py (Gaussian part only).

Holds the emission parameters (C, d, log R-Cholesky) as NumPy arrays; the ELBO reductions that consume them
(predict_mean, logdensity_norm_diag*) run inside the fused HIP reduction kernel of the engine.
"""
from __future__ import annotations

import numpy as np

from . import _lib


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

    def predict_mean(self, X_end):
        """X_end @ CC + DD (likelihoods.py:76-79), on the GPU."""
        lib = _lib.load()
        X_end = _lib.as_f64(X_end)
        if X_end.ndim != 2 or X_end.shape[1] != self.X_output_dim:
            raise ValueError(f"X_end: expected (N, {self.X_output_dim}), got {X_end.shape}")
        out = np.empty((X_end.shape[0], self.Y_dim))
        CC, DD = _lib.as_f64(self.CC), _lib.as_f64(self.DD)
        _lib.check(lib.ffvd_op_predict_mean(_lib.dptr(X_end), X_end.shape[0], self.X_output_dim, _lib.dptr(CC),
                                            _lib.dptr(DD), self.Y_dim, _lib.dptr(out)), None, "predict_mean")
        return out


def _logdensity(nonvec, y, ymean, Rchols):
    lib = _lib.load()
    y = _lib.as_f64(y)
    ymean = _lib.as_f64(ymean, y.shape, "ymean")
    if y.ndim != 2:
        raise ValueError("y: expected (N, J)")
    N, J = y.shape
    R = _lib.as_f64(Rchols, (J,), "Rchols")
    out = np.empty((N, J) if nonvec else (N,))
    _lib.check(lib.ffvd_op_logdensity_norm_diag(int(nonvec), _lib.dptr(y), _lib.dptr(ymean), _lib.dptr(R), N, J,
                                                _lib.dptr(out)), None, "logdensity_norm_diag")
    return out


def logdensity_norm_diag_nonvec(y, ymean, Rchols):
    """Elementwise -0.5((y-ymean)/R)^2 - log R (likelihoods.py:89-93)."""
    return _logdensity(True, y, ymean, Rchols)


def logdensity_norm_diag(y, ymean, Rchols):
    """Per-row -0.5 sum_j((y-ymean)/R_j)^2 - sum_j log R_j (likelihoods.py:96-111)."""
    return _logdensity(False, y, ymean, Rchols)
