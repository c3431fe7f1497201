"""Optimiser / sampler steps of the FFVD training loop as operators, NumPy in/out -> C ABI.

  adam_step   one update of tf.compat.v1.train.AdamOptimizer.minimize(nll) (dgp_model.py:303-305)
  sghmc_step  one burn_in_op / sample_op of BaseModel.generate_update_step (base_model.py:143-179)
  AdamState / SghmcState  the per-variable state the reference keeps in tf.Variables

The device-resident training step (forward + backward + update without leaving the GPU) is
`ElboEngine.adam_step`; these wrappers move flat host arrays and exist for parity tests and for callers that
all-reduce gradients between the backward pass and the update.
"""
from __future__ import annotations

import numpy as np

from . import _lib

ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-8          # TensorFlow defaults (the reference passes only lr)


def decayed_learning_rate(global_step=1):
    """BaseModel.get_minibatch (base_model.py:188-194): always called with global_step = 1 by the reference."""
    return 0.003 * (0.95 ** (global_step / 1000))


class AdamState:
    def __init__(self, shape):
        self.m = np.zeros(shape)
        self.v = np.zeros(shape)
        self.t = 0


def adam_step(theta, grad, state, lr, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    """Returns the updated theta (new array); `state` (AdamState) advances in place."""
    th = _lib.as_f64(theta).copy()
    g = _lib.as_f64(grad, th.shape, "grad")
    if state.m.shape != th.shape:
        raise ValueError("adam_step: state shape does not match theta")
    state.t += 1
    m, v = np.ascontiguousarray(state.m), np.ascontiguousarray(state.v)
    _lib.check(_lib.load().ffvd_op_adam_step(_lib.dptr(th), _lib.dptr(g), _lib.dptr(m), _lib.dptr(v), th.size, float(lr),
                                             float(beta1), float(beta2), float(eps), state.t), None, "ffvd_op_adam_step")
    state.m, state.v = m, v
    return th


class SghmcState:
    """xi, g, g2 start at ones, the momentum p at zeros (base_model.py:151-154)."""

    def __init__(self, shape):
        self.xi = np.ones(shape)
        self.g = np.ones(shape)
        self.g2 = np.ones(shape)
        self.p = np.zeros(shape)


def sghmc_step(theta, grad, state, noise, epsilon=0.01, mdecay=0.05, X_N=1, burn_in=True):
    """Returns the updated theta; `state` advances in place.  `noise` is the standard-normal draw of
    base_model.py:169 (the reference draws it inside the graph; it is injected here so runs are reproducible)."""
    th = _lib.as_f64(theta).copy()
    g = _lib.as_f64(grad, th.shape, "grad")
    nz = _lib.as_f64(noise, th.shape, "noise")
    st = [np.ascontiguousarray(a, dtype=np.float64) for a in (state.xi, state.g, state.g2, state.p)]
    if any(a.shape != th.shape for a in st):
        raise ValueError("sghmc_step: state shape does not match theta")
    _lib.check(_lib.load().ffvd_op_sghmc_step(_lib.dptr(th), _lib.dptr(g), *[_lib.dptr(a) for a in st], _lib.dptr(nz),
                                              th.size, float(epsilon), float(mdecay), float(X_N), int(bool(burn_in))),
               None, "ffvd_op_sghmc_step")
    state.xi, state.g, state.g2, state.p = st
    return th
