"""CPU: the oracle against the committed golden vectors, its independent torch twin, the survey's anchor
values, and algebraic identities of the path.  (Parity is unpinned by the reference itself: it has no tests.)"""
import numpy as np
import pytest

from conftest import load_golden
from ffvd_amd import synthetic
from oracle import ffvd_oracle as orc
from oracle import ffvd_oracle_torch as orct

# SURVEY.md section 8(a) anchors (throw-away restatement of the survey session; detects a mis-reading)
ANCHOR_B = dict(nll_log_likelihood=-0.7574060071934727, later_term1=0.05602634235378841,
                later_term2=-0.02577171403304086, nll_reg_trace_inverse_Q_B=0.00035000219639542106,
                x_t_prior_Q=-2.5238253925875025, nll_part_prior=0.881323722504348, nll=-2.369303046759484)
ANCHOR_A = dict(nll_reg_trace_inverse_Q_B=0.00035000219639551896, x_t_prior_Q=-2.4575163628166186,
                nll_part_prior=0.9499730039810291, nll=-2.2645993638326667)


def test_actuator_anchors(actuator):
    params, Y, c = actuator
    tb = orc.nll_terms(params, Y, c, U_collapse=True)
    ta = orc.nll_terms(params, Y, c, U_collapse=False)
    for k, v in ANCHOR_B.items():
        assert tb[k] == pytest.approx(v, rel=1e-9, abs=1e-13), k
    for k, v in ANCHOR_A.items():
        assert ta[k] == pytest.approx(v, rel=1e-9, abs=1e-13), k


def test_prior_centre_is_the_float32_logarithm():
    """dgp_model.py:127: tf.cast(tf.math.log(0.05), tf.float64) takes the logarithm in float32 and widens it; the LinearK
    line (:130) uses np.log(0.05).  Both restatements and the HIP kernels (kernels.h) carry exactly these two values."""
    import re, os, torch
    assert orc.LOG_PRIOR_VARIANCE_SE == -2.995732307434082 == float(torch.log(torch.tensor(0.05, dtype=torch.float32)))
    assert orc.LOG_PRIOR_VARIANCE_LIN == -2.995732273553991 == float(np.log(0.05))
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "ffvd_amd", "csrc", "kernels.h")).read()
    se = float(re.search(r"LOG_PRIOR_VARIANCE_SE = ([-0-9.e]+);", hdr).group(1))
    lin = float(re.search(r"LOG_PRIOR_VARIANCE_LIN = ([-0-9.e]+);", hdr).group(1))
    assert se == orc.LOG_PRIOR_VARIANCE_SE and lin == orc.LOG_PRIOR_VARIANCE_LIN
    # the effect on the actuator fixture: 3.4e-8 in the constant, 6e-11 in the nll (SURVEY's anchors used the fp64 value)
    class K:                                        # noqa: D401
        def __init__(self, lv): self.logvariance = np.array(lv); self.loglengthscales = np.zeros(5)
    assert abs(orc.prior_hyper([K(-1.0)]) + 0.5 * (-1.0 - orc.LOG_PRIOR_VARIANCE_SE) ** 2) < 1e-16


def test_actuator_golden(actuator):
    params, Y, c = actuator
    g = load_golden("actuator")
    for branch, collapse in (("B", True), ("A", False)):
        t = orc.nll_terms(params, Y, c, U_collapse=collapse)
        for k, v in t.items():
            assert v == pytest.approx(float(g[f"{branch}_{k}"]), rel=1e-12, abs=1e-14)


@pytest.mark.parametrize("name", ["tiny", "small", "ragged", "small_lin"])
def test_synthetic_golden_and_torch_twin(name):
    params, Y, c, meta = synthetic.make_named(name)
    g = load_golden(name)
    for branch, collapse in (("B", True), ("A", False)):
        t = orc.nll_terms_chains(params, Y, c, U_collapse=collapse, kernel_type=meta["kernel_type"])
        for k, v in t.items():
            np.testing.assert_allclose(v, g[f"{branch}_{k}"], rtol=1e-11, atol=1e-13, err_msg=f"{name}/{branch}/{k}")
        p0 = dict(params)
        p0["X"] = params["X"][0]
        tp, tY, tc = orct.to_torch(p0, Y, c)
        tt = orct.nll_terms(tp, tY, tc, U_collapse=collapse, kernel_type=meta["kernel_type"])
        assert float(tt["nll"]) == pytest.approx(t["nll_per_chain"][0], rel=1e-10)


def test_workload_is_deterministic():
    a = synthetic.make_named("tiny")
    b = synthetic.make_named("tiny")
    for k in a[0]:
        np.testing.assert_array_equal(a[0][k], b[0][k])
    np.testing.assert_array_equal(a[1], b[1])
    assert synthetic.algorithmic_flops(**synthetic.make_named("c2", S=1, T=8, M=8)[3]) > 0
    # W_alg of config 2 = 2.847e11
    assert synthetic.algorithmic_flops(T=4096, D=4, M=512, S=32, P=5) == pytest.approx(2.847e11, rel=1e-3)


def test_gram_route_identity():
    """H~ = L^-1 (K_uf K_fu) L^-T equals F^T F; trace via trsm equals trace via inverse."""
    params, Y, c, meta = synthetic.make_named("tiny")
    kern = orc.make_kernels(params)
    X = params["X"][0]
    xc = np.concatenate((X[:-1], c[: meta["T"]]), axis=1)
    Z = params["Z"]
    for k in kern:
        Kuu = k.K(Z) + 1e-5 * np.eye(Z.shape[0])
        L = np.linalg.cholesky(Kuu)
        Kfu = k.K(xc, Z)
        F = Kfu @ np.linalg.inv(L).T
        G = Kfu.T @ Kfu
        Ht = np.linalg.solve(L, np.linalg.solve(L, G).T).T
        np.testing.assert_allclose(Ht, F.T @ F, rtol=1e-7, atol=1e-8)
        from scipy.linalg import solve_triangular
        A = solve_triangular(L, Kfu.T, lower=True)
        np.testing.assert_allclose(np.sum(A * A), np.sum(F * F), rtol=1e-12)


def test_whitened_factor_without_forming_h():
    """The identities behind the round-3 training forward: with K = L L^T and A = K + K_uf K_fu / Q = L_A L_A^T,
    the Cholesky factor of the whitened H = L^-1 A L^-T is L^-1 L_A (lower x lower, positive diagonal: the factor is unique), so
    L_H^-T = L^T L_A^-T -- what an extended factorisation of A leaves in extension rows that start as L^T -- and the solved row
    L_A^-1 c equals L_H^-1 (L^-1 c).  log|H| = log|A| - log|K|."""
    from scipy.linalg import solve_triangular
    params, Y, c, meta = synthetic.make_named("small")
    kern = orc.make_kernels(params)
    X = params["X"][0]
    xc = np.concatenate((X[:-1], c[: meta["T"]]), axis=1)
    Z, M = params["Z"], meta["M"]
    for d, k in enumerate(kern):
        K = k.K(Z) + 1e-5 * np.eye(M)
        Kfu = k.K(xc, Z)
        alpha = 1.0 / np.exp(params["log_Q"][d])
        A = K + alpha * (Kfu.T @ Kfu)
        cvec = alpha * (Kfu.T @ (X[1:, d] - X[:-1, d]))
        L, LA = np.linalg.cholesky(K), np.linalg.cholesky(A)
        Linv = solve_triangular(L, np.eye(M), lower=True)
        H = Linv @ A @ Linv.T
        LH = np.linalg.cholesky(0.5 * (H + H.T))
        np.testing.assert_allclose(Linv @ LA, LH, rtol=1e-7, atol=1e-9)
        rows = solve_triangular(LA, L, lower=True).T                       # L^T L_A^-T
        np.testing.assert_allclose(rows, np.linalg.inv(LH).T, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(solve_triangular(LA, cvec, lower=True), solve_triangular(LH, Linv @ cvec, lower=True), rtol=1e-7, atol=1e-10)
        assert 2 * np.sum(np.log(np.diag(LH))) == pytest.approx(2 * np.sum(np.log(np.diag(LA))) - 2 * np.sum(np.log(np.diag(L))), rel=1e-9)


def test_linear_kernel_projection_through_its_rank():
    """The identities behind config 5's forward: LinearK's K_fu = s2 X Z^T (kernels.py:276) has rank P, so
    F = K_fu L^-T = s2 X C with C = Z^T L^-T, F u = s2 X (C u) and sum_j F_tj^2 = s2^2 x_t^T (C C^T) x_t -- the two quantities the
    explicit-U branch reads of F (conditionals_multi_output.py:44-52)."""
    from scipy.linalg import solve_triangular
    params, Y, c, meta = synthetic.make_named("small_lin")
    kern = orc.make_kernels(params, kernel_type="LinearK")
    X = params["X"][0]
    xc = np.concatenate((X[:-1], c[: meta["T"]]), axis=1)
    Z, M = params["Z"], meta["M"]
    for d, k in enumerate(kern):
        s2 = np.exp(params["logvariance"][d])
        K = k.K(Z) + 1e-5 * np.eye(M)
        L = np.linalg.cholesky(K)
        F = solve_triangular(L, k.K(xc, Z).T, lower=True).T                # K_fu L^-T
        C = solve_triangular(L, Z, lower=True).T                           # Z^T L^-T  (P x M)
        u = params["U"][:, d]
        np.testing.assert_allclose(F, s2 * (xc @ C), rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(F @ u, s2 * (xc @ (C @ u)), rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(np.sum(F * F, axis=1), s2 * s2 * np.einsum("tp,pq,tq->t", xc, C @ C.T, xc), rtol=1e-7, atol=1e-12)


def test_conditional_routes_agree():
    """Branch-A conditional via trsm (:6-70) == via the pre-computed inverse (:324-387)."""
    params, Y, c, meta = synthetic.make_named("tiny")
    kern = orc.make_kernels(params)
    X = params["X"][0]
    xc = np.concatenate((X[:-1], c[: meta["T"]]), axis=1)
    m1, v1 = orc.conditional(xc, params["Z"], kern, params["U"], white=True)
    Linv = orc.kernel_pre_cal(params["Z"], kern)
    m2, v2 = orc.conditional_after_kernel_precalculation(Linv, xc, params["Z"], kern, params["U"])
    np.testing.assert_allclose(m1, m2, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(v1, v2, rtol=1e-7, atol=1e-10)


def test_torch_gradient_matches_finite_differences():
    params, Y, c, meta = synthetic.make_named("tiny")
    p0 = dict(params)
    p0["X"] = params["X"][0]
    _, g = orct.nll_and_grad(p0, Y, c, wrt=("log_Q", "logvariance"), U_collapse=True)
    eps = 1e-6
    for key in ("log_Q", "logvariance"):
        for i in range(meta["D"]):
            pp, pm = dict(p0), dict(p0)
            pp[key] = p0[key].copy(); pp[key][i] += eps
            pm[key] = p0[key].copy(); pm[key][i] -= eps
            fd = (orc.nll_terms(pp, Y, c)["nll"] - orc.nll_terms(pm, Y, c)["nll"]) / (2 * eps)
            assert g[key][i] == pytest.approx(fd, rel=2e-5, abs=1e-8)


def test_get_rand_is_the_workload_draw():
    """utils.py:11 with injected eps reproduces X_s = mu + 0.1 eps_s of the synthetic workload."""
    mean = np.arange(6.0).reshape(3, 2)
    var = np.full((3, 2), 0.01)
    eps = np.ones((3, 2))
    np.testing.assert_allclose(orc.get_rand(mean, var, eps), mean + 0.1)


def test_explicit_u_closed_form_gradient_matches_autograd():
    """oracle/ffvd_grad_oracle.nll_grad_explicit_u (Cholesky adjoint in closed form) against torch autograd of the
    independent restatement: the CPU twin of the HIP backward pass of the explicit-U branch."""
    from oracle import ffvd_grad_oracle as gorc
    from oracle import ffvd_oracle_torch as orct
    from ffvd_amd import synthetic
    for name, ov in (("tiny", {}), ("tiny", dict(C=0))):
        params, Y, c, meta = synthetic.make_named(name, **ov)
        p = dict(params)
        p["X"] = params["X"][0]
        keys = ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")
        _, ga = orct.nll_and_grad(p, Y, c, wrt=keys, U_collapse=False)
        g = gorc.nll_grad_explicit_u(p, Y, c)
        for k in keys:
            err = np.max(np.abs(g[k] - ga[k])) / (np.max(np.abs(ga[k])) + 1e-300)
            assert err < 1e-8, (k, err)
