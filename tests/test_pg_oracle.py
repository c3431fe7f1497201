"""CPU: the particle-Gibbs restatement (oracle/ffvd_pg_oracle.py) -- the intent of BaseModel.PG_for_X_speedup
(vfegpssm/base_model.py:78-138); the reference op itself never updates X, so these are property tests."""
import numpy as np

from ffvd_amd import synthetic
from oracle import ffvd_oracle as orc, ffvd_pg_oracle as pgo


def test_categorical_from_uniform_is_the_inverse_cdf():
    logits = np.array([0.0, np.log(2.0), -np.inf, np.log(1.0)])          # p = 1/4, 1/2, 0, 1/4
    u = np.array([0.0, 0.2499, 0.25, 0.7499, 0.75, 0.999999])
    np.testing.assert_array_equal(pgo.categorical_from_uniform(logits, u), [0, 0, 1, 1, 3, 3])
    rng = np.random.default_rng(0)
    logits = rng.standard_normal(7) * 3.0
    idx = pgo.categorical_from_uniform(logits, rng.random(200000))
    p = np.exp(logits - logits.max())
    p /= p.sum()
    np.testing.assert_allclose(np.bincount(idx, minlength=7) / idx.size, p, atol=5e-3)
    # shifting the logits changes nothing (tfp Categorical(logits=...) is shift invariant)
    uu = rng.random(1000)
    np.testing.assert_array_equal(pgo.categorical_from_uniform(logits, uu), pgo.categorical_from_uniform(logits + 123.0, uu))


def _setup(name="tiny", N=8, seed=2):
    params, Y, c, meta = synthetic.make_named(name)
    kern = orc.make_kernels(params)
    Lm = orc.kernel_pre_cal(params["Z"], kern)
    X = params["X"][0]
    rng = np.random.default_rng(seed)
    T, D = meta["T"], meta["D"]
    draws = dict(x0=rng.standard_normal((N - 1, D)), eps=rng.standard_normal((T, N - 1, D)), unif=rng.random((T, N - 1)))
    args = (Lm, params["Z"], kern, params["U"], X, Y, c, params["CC"], params["DD"], np.exp(params["log_Rchols"]),
            np.exp(params["log_Q"]))
    return args, draws, X, meta


def test_pg_sweep_shapes_and_reference_particle():
    args, d, X, meta = _setup()
    parts, idx = pgo.pg_sweep(*args, **d)
    N1 = d["x0"].shape[0]
    assert parts.shape == (meta["T"] + 1, N1, meta["D"]) and idx.shape == (meta["T"], N1)
    np.testing.assert_array_equal(parts[0], d["x0"])
    assert idx.min() >= 0 and idx.max() <= N1
    t, i = np.argwhere(idx == N1)[0]                       # an ancestor equal to the reference: its state is X[t+1]
    np.testing.assert_array_equal(parts[t + 1, i], X[t + 1])
    # conditioning on the reference: the trajectory that explains Y is drawn clearly more often than 1/N (0.22 vs 0.125 here)
    assert (idx == N1).mean() > 1.5 / (N1 + 1)
    # u -> 1 picks the last candidate, i.e. the reference, at every step: the sweep returns X
    p1, i1 = pgo.pg_sweep(*args, d["x0"], d["eps"], np.full_like(d["unif"], 1.0 - 1e-16))
    assert np.all(i1 == N1)
    np.testing.assert_array_equal(p1[1:], np.repeat(X[1:, None, :], N1, axis=1))
    np.testing.assert_array_equal(pgo.select_trajectory(X, parts, N1), X)
    np.testing.assert_array_equal(pgo.select_trajectory(X, parts, 2), parts[:, 2])


def test_pg_sweep_step_is_the_rollout_step():
    """Without resampling pressure (u spread so that particle i keeps ancestor i is not guaranteed) the propagation itself
    must be the rollout step of collect_samples_formal: check the first step by hand."""
    args, d, X, meta = _setup(N=4)
    Lm, Z, kern, U, X_ref, Y, c, CC, DD, R, Q = args
    parts, idx = pgo.pg_sweep(*args, **d)
    xc = np.concatenate((d["x0"], np.repeat(c[0][None, :], 3, axis=0)), axis=1)
    f_mu, f_var = orc.conditional_after_kernel_precalculation(Lm, xc, Z, kern, U)
    x1 = d["x0"] + f_mu + d["eps"][0] * np.sqrt(f_var + Q[None, :])
    cand = np.concatenate((x1, X[1][None, :]))
    np.testing.assert_allclose(parts[1], cand[idx[0]], rtol=0, atol=0)
