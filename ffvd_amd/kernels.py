"""Kernel objects mirroring vfegpssm/kernels.py and vfegpssm/kernels_multi_output.py (NumPy in/out, HIP compute).

`SquaredExponential(input_dim, variance, lengthscales, ARD=True)` follows Stationary.__init__
(kernels_multi_output.py:140-161): log-parameterised `logvariance` / `loglengthscales`.
`LinearK(input_dim, variance)` follows kernels.py:250-281 (scalar variance, ARD=False).
K / Kdiag dispatch to libffvd_hip.so (ffvd_op_kernel_matrix / ffvd_op_kernel_diag).
"""
from __future__ import annotations

import numpy as np

from . import _lib


class Kernel:
    kind = None

    def __init__(self, input_dim, active_dims=None, name=None):
        self.input_dim = int(input_dim)
        if active_dims is not None and list(active_dims) != list(range(self.input_dim)):
            raise NotImplementedError("active_dims other than slice(input_dim) are not used by the GP-SSM path")
        self.name = name

    def _X(self, X, name="X"):
        X = _lib.as_f64(X, name=name)
        if X.ndim != 2 or X.shape[1] != self.input_dim:
            raise ValueError(f"{name}: expected (N, {self.input_dim}), got {X.shape}")
        return X

    def _loglen_ptr(self):
        return None

    def K(self, X, X2=None, presliced=False):
        lib = _lib.load()
        X = self._X(X)
        N = X.shape[0]
        if X2 is None:
            N2, x2p = N, None
        else:
            X2 = self._X(X2, "X2")
            N2, x2p = X2.shape[0], _lib.dptr(X2)
        out = np.empty((N, N2))
        ll = self._loglen_ptr()
        rc = lib.ffvd_op_kernel_matrix(self.kind, _lib.dptr(X), N, x2p, N2, self.input_dim, float(self.logvariance),
                                       None if ll is None else _lib.dptr(ll), 0.0, _lib.dptr(out))
        _lib.check(rc, None, "kernel.K")
        return out

    def Kdiag(self, X, presliced=False):
        lib = _lib.load()
        X = self._X(X)
        out = np.empty(X.shape[0])
        rc = lib.ffvd_op_kernel_diag(self.kind, _lib.dptr(X), X.shape[0], self.input_dim, float(self.logvariance),
                                     _lib.dptr(out))
        _lib.check(rc, None, "kernel.Kdiag")
        return out


class SquaredExponential(Kernel):
    """The radial basis function (RBF) or squared exponential kernel, ARD lengthscales."""
    kind = 0

    def __init__(self, input_dim, variance=0.1, lengthscales=1.0, active_dims=None, ARD=None, name=None,
                 kernel_optimization=False, U_kernel_optimization=False):
        super().__init__(input_dim, active_dims, name)
        ls = np.asarray(lengthscales, dtype=np.float64)
        if ls.ndim == 0:
            ls = np.full(self.input_dim, float(ls))
        if ls.shape != (self.input_dim,):
            raise ValueError(f"lengthscales: expected ({self.input_dim},), got {ls.shape}")
        self.ARD = True if ARD is None else bool(ARD)
        self.logvariance = np.float64(np.log(variance))          # kernels_multi_output.py:156
        self.loglengthscales = np.log(ls)                        # kernels_multi_output.py:160

    @property
    def variance(self):
        return np.exp(self.logvariance)

    @property
    def lengthscales(self):
        return np.exp(self.loglengthscales)

    def _loglen_ptr(self):
        return np.ascontiguousarray(self.loglengthscales, dtype=np.float64)


class LinearK(Kernel):
    """The linear kernel K = (X * variance) X2^T with one scalar variance."""
    kind = 1

    def __init__(self, input_dim, variance=1.0, active_dims=None, ARD=None, name=None):
        super().__init__(input_dim, active_dims, name)
        v = np.asarray(variance, dtype=np.float64)
        if v.ndim != 0:
            raise ValueError("LinearK takes one scalar variance (ARD=False, models.py:62)")
        self.ARD = False
        self.logvariance = np.float64(np.log(v))                 # kernels.py:264

    @property
    def variance(self):
        return np.exp(self.logvariance)


def stack_hypers(kern):
    """(kind, kernel_type, logvariance (D,), loglengthscales (D,P) or None) of a list of D kernels of one type."""
    kinds = {k.kind for k in kern}
    if len(kinds) != 1:
        raise ValueError("all kernels in the list must have the same type")
    kind = kinds.pop()
    logvar = np.array([float(k.logvariance) for k in kern], dtype=np.float64)
    loglen = None
    if kind == 0:
        loglen = np.ascontiguousarray(np.stack([k.loglengthscales for k in kern]), dtype=np.float64)
    return kind, ("SquaredExponential" if kind == 0 else "LinearK"), logvar, loglen
