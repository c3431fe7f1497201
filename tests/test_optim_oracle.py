"""CPU checks of the optimiser / sampler restatement (oracle/ffvd_optim_oracle.py) -- no GPU."""
import numpy as np
import pytest

from oracle import ffvd_optim_oracle as oo


def test_adam_first_step_is_a_signed_lr_step():
    """With m = v = 0 and t = 1 the bias corrections cancel: theta moves by lr * g / (|g| + eps / sqrt(1 - b2))."""
    rng = np.random.default_rng(1)
    th, g = rng.standard_normal(50), rng.standard_normal(50)
    lr = oo.decayed_learning_rate()
    assert lr == pytest.approx(0.003 * 0.95 ** 0.001)
    th1, m1, v1 = oo.adam_step(th, g, np.zeros(50), np.zeros(50), 1, lr)
    np.testing.assert_allclose(m1, 0.1 * g, rtol=1e-15)
    np.testing.assert_allclose(v1, 0.001 * g * g, rtol=1e-12)
    np.testing.assert_allclose(th1, th - lr * g / (np.abs(g) + 1e-8 / np.sqrt(0.001)), rtol=1e-12)


def test_adam_three_steps_scalar_by_hand():
    th, m, v = 1.0, 0.0, 0.0
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    ref = th
    mm = vv = 0.0
    for t, g in enumerate((0.5, -0.25, 2.0), start=1):
        th, m, v = oo.adam_step(th, g, m, v, t, lr)
        mm = b1 * mm + (1 - b1) * g
        vv = b2 * vv + (1 - b2) * g * g
        ref = ref - lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * mm / (np.sqrt(vv) + eps)
        assert th == pytest.approx(ref, rel=1e-15)


def test_sghmc_ops_follow_the_reference_formulas():
    """base_model.py:150-179 on a scalar, by hand; sample_op must leave xi, g, g2 alone."""
    theta, grad, xi, g, g2, p, z = 0.3, -1.5, 1.0, 1.0, 1.0, 0.0, 0.7
    eps, md, XN = 0.01, 0.05, 513
    Minv = 1.0 / (np.sqrt(g2 + 1e-16) + 1e-16)
    sigma = np.sqrt(max(2.0 * (eps / np.sqrt(XN)) ** 2 * md * Minv, 1e-16))
    p_t = p - eps ** 2 * Minv * grad - md * p + z * sigma
    out = oo.sghmc_step(theta, grad, xi, g, g2, p, z, eps, md, XN, burn_in=True)
    assert out[0] == pytest.approx(theta + p_t, rel=1e-15) and out[4] == pytest.approx(p_t, rel=1e-15)
    assert out[1] == pytest.approx(1.0 + xi * (1.0 - g * g / (g2 + 1e-16)))       # xi_t
    assert out[2] == pytest.approx(0.5 * g + 0.5 * grad)                          # g_t with r_t = 1/2
    assert out[3] == pytest.approx(0.5 * g2 + 0.5 * grad ** 2)
    out_s = oo.sghmc_step(theta, grad, xi, g, g2, p, z, eps, md, XN, burn_in=False)
    assert out_s[1:4] == (xi, g, g2) and out_s[0] == out[0] and out_s[4] == out[4]


def test_sghmc_noise_floor():
    """:168 clamps the noise variance at 1e-16 (a huge g2 would otherwise underflow it)."""
    out = oo.sghmc_step(0.0, 0.0, 1.0, 1.0, 1e40, 0.0, 1.0, 0.01, 0.05, 100, burn_in=False)
    assert out[0] == pytest.approx(1e-8)
