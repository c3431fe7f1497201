"""Seeded synthetic workloads for the FFVD ELBO hot path.

The draw order below is part of the contract: golden vectors under
`tests/golden/` and `bench.py` both rely on it.  NumPy only; no device code.
"""
from __future__ import annotations

import numpy as np

SEED = 20230209

# name -> (T, D, C, M, S, kernel_type, U_collapse, dtype)
CONFIGS = {
    # BASELINE.json configs[1]/[2]: the headline workload
    "c2": dict(T=4096, D=4, C=1, M=512, S=32, kernel_type="SquaredExponential", U_collapse=True),
    # BASELINE.json configs[3]: MFMA stress (fp32 contractions in the reference plan)
    "c4": dict(T=16384, D=8, C=1, M=2048, S=64, kernel_type="SquaredExponential", U_collapse=True),
    # BASELINE.json configs[4]: LinearK + explicit-U branch, shard latent dims
    "c5": dict(T=4096, D=16, C=1, M=512, S=1, kernel_type="LinearK", U_collapse=False),
    # scaled-down shapes the CPU oracle finishes in seconds
    "tiny": dict(T=96, D=2, C=1, M=24, S=3, kernel_type="SquaredExponential", U_collapse=True),
    "small": dict(T=384, D=4, C=1, M=96, S=4, kernel_type="SquaredExponential", U_collapse=True),
    "ragged": dict(T=301, D=3, C=2, M=77, S=2, kernel_type="SquaredExponential", U_collapse=True),
    "small_lin": dict(T=256, D=6, C=1, M=64, S=1, kernel_type="LinearK", U_collapse=False),
}


def make_workload(T, D, C, M, S, kernel_type="SquaredExponential", U_collapse=True, Ydim=1, seed=SEED):
    """Return (params, Y, control_inputs, meta) exactly as SURVEY.md section 8(d) prescribes.

    params['X'] has shape (S, T+1, D): S latent trajectories X_s = mu + 0.1*eps_s,
    i.e. the Monte-Carlo latent-state draw of utils.py:11 with injected eps.
    """
    P = D + C
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.standard_normal((T, C))
    mu = np.empty((T + 1, D))
    mu[0] = rng.standard_normal(D)
    steps = rng.standard_normal((T, D))
    for t in range(T):
        mu[t + 1] = 0.95 * mu[t] + 0.3 * steps[t]
    eps = rng.standard_normal((S, T + 1, D))
    X = mu[None, :, :] + 0.1 * eps
    idx = rng.choice(T, M, replace=False)
    Z = np.concatenate((mu[idx], c[idx]), axis=1) + 0.05 * rng.standard_normal((M, P))
    U = rng.standard_normal((M, D))
    dd = np.arange(D, dtype=np.float64)
    if kernel_type == "SquaredExponential":
        variance = np.full(D, 0.5)
        lengthscales = np.repeat((2.0 + 0.1 * dd)[:, None], P, axis=1)
    else:
        variance = 0.05 * (1.0 + dd / D)
        lengthscales = np.ones((D, P))
    Q = (0.4 + 0.05 * dd) ** 2
    CC = (0.5 * (-0.5) ** dd)[:, None] * np.ones((1, Ydim))
    DD = np.full(Ydim, 0.05)
    R = np.full((Ydim, Ydim), 0.4)
    Y = mu[1:] @ CC + DD + 0.4 * rng.standard_normal((T, Ydim))
    params = dict(
        X=np.ascontiguousarray(X), Z=np.ascontiguousarray(Z), U=np.ascontiguousarray(U),
        logvariance=np.log(variance), loglengthscales=np.log(lengthscales),
        log_Q=np.log(Q), CC=CC, DD=DD, log_Rchols=np.log(R),
    )
    meta = dict(T=T, D=D, C=C, M=M, S=S, P=P, Ydim=Ydim, kernel_type=kernel_type,
                U_collapse=U_collapse, seed=seed)
    return params, Y, c, meta


def make_named(name, **overrides):
    cfg = dict(CONFIGS[name])
    cfg.update(overrides)
    return make_workload(**cfg)


def algorithmic_flops(T, D, M, S, P, U_collapse=True, **_):
    """W_alg of SURVEY.md section 8(d): structure-aware flop count per ELBO iteration."""
    if U_collapse:
        return S * D * (2 * T * M * M + M ** 3 / 3 + T * M * (2 * P + 4)) + D * (2 * M ** 3 / 3)
    return S * D * (T * M * M + 2 * T * M + T * M * (2 * P + 4)) + D * (M ** 3 / 3)
