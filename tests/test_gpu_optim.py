"""GPU parity of the optimiser / sampler steps through the C ABI against the CPU oracle."""
import numpy as np
import pytest

from ffvd_amd import optim, synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_grad_oracle as gorc
from oracle import ffvd_optim_oracle as oo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 257, 5000])
def test_adam_operator_matches_oracle(n):
    rng = np.random.default_rng(n)
    th = rng.standard_normal(n)
    st = optim.AdamState(th.shape)
    ref, m, v = th.copy(), np.zeros(n), np.zeros(n)
    lr = optim.decayed_learning_rate()
    for t in range(1, 5):
        g = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 3, n)
        th = optim.adam_step(th, g, st, lr)
        ref, m, v = oo.adam_step(ref, g, m, v, t, lr)
        # the kernel keeps mul/add unfused, so only lr_t (host pow) can differ from NumPy, by an ulp
        np.testing.assert_allclose(th, ref, rtol=1e-15, atol=1e-17)
        np.testing.assert_array_equal(st.m, m)
        np.testing.assert_array_equal(st.v, v)
    assert st.t == 4


@pytest.mark.parametrize("shape", [(3,), (40, 5)])
def test_sghmc_operator_matches_oracle(shape):
    """burn_in_op and sample_op of base_model.py:143-179, alternated as sghmc_step (:915-925) does."""
    rng = np.random.default_rng(7)
    th = rng.standard_normal(shape)
    st = optim.SghmcState(shape)
    ref = (th.copy(), np.ones(shape), np.ones(shape), np.ones(shape), np.zeros(shape))
    for it in range(6):
        g, z = rng.standard_normal(shape), rng.standard_normal(shape)
        burn = it % 2 == 0
        th = optim.sghmc_step(th, g, st, z, epsilon=0.01, mdecay=0.05, X_N=513, burn_in=burn)
        ref = oo.sghmc_step(ref[0], g, ref[1], ref[2], ref[3], ref[4], z, 0.01, 0.05, 513, burn)
        for got, want in zip((th, st.xi, st.g, st.g2, st.p), ref):
            np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-18)


def test_operator_argument_errors():
    with pytest.raises(ValueError):
        optim.adam_step(np.zeros(3), np.zeros(4), optim.AdamState((3,)), 0.01)
    with pytest.raises(ValueError):
        optim.sghmc_step(np.zeros(3), np.zeros(3), optim.SghmcState((3,)), np.zeros(3), X_N=0)


def _oracle_mean_grad(params, Y, c):
    S = params["X"].shape[0]
    tot = None
    for s in range(S):
        p = dict(params)
        p["X"] = params["X"][s]
        g = gorc.nll_grad(p, Y, c)
        if tot is None:
            tot = {k: (np.zeros((S,) + v.shape) if k == "X" else np.zeros_like(v)) for k, v in g.items()}
        tot["X"][s] = g["X"] / S
        for k in g:
            if k != "X":
                tot[k] += g[k] / S
    return tot


def test_device_resident_training_matches_oracle_loop():
    """train_hypers (base_model.py:944-950) x 4 on the device == closed-form gradient oracle + Adam oracle on the CPU."""
    params, Y, c, meta = synthetic.make_named("tiny")
    keys = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")
    lr = optim.decayed_learning_rate()
    ref = {k: np.array(params[k], dtype=np.float64) for k in keys}
    m = {k: np.zeros_like(ref[k]) for k in keys}
    v = {k: np.zeros_like(ref[k]) for k in keys}
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        e.set_params(params)
        nlls = []
        for t in range(1, 5):
            nlls.append(e.adam_step(lr)["nll"])
            g = _oracle_mean_grad(dict(params, **ref), Y, c)
            for k in keys:
                ref[k], m[k], v[k] = oo.adam_step(ref[k], g[k], m[k], v[k], t, lr)
        got = e.get_params()
        final = e.nll_terms()["nll"]
    assert nlls[0] > nlls[-1] > final                         # Adam is descending
    for k in keys:
        # a step is ~lr * sign(g): entries whose gradient is at the eps*cond noise level can flip late digits
        # (Adam divides by |g|: where a gradient entry is down at its own eps*cond noise the step direction is noise too)
        np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=2e-5 * lr + 1e-9 * np.max(np.abs(ref[k])), err_msg=k)
    np.testing.assert_array_equal(got["U"], params["U"])      # U is integrated out: untouched


def test_train_mask_freezes_arrays():
    params, Y, c, meta = synthetic.make_named("tiny")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        e.set_params(params)
        e.adam_step(0.01, train=("logvariance", "log_Q"))
        got = e.get_params()
    assert not np.array_equal(got["logvariance"], params["logvariance"])
    assert not np.array_equal(got["log_Q"], params["log_Q"])
    for k in ("X", "Z", "loglengthscales", "CC", "DD", "log_Rchols"):
        np.testing.assert_array_equal(got[k], np.asarray(params[k], dtype=np.float64).reshape(got[k].shape))


def test_failed_factorisation_leaves_parameters_untouched():
    params, Y, c, meta = synthetic.make_named("tiny")
    p = dict(params)
    p["Z"] = params["Z"].copy()
    p["Z"][5] = p["Z"][4]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True, jitter=0.0) as e:
        e.set_data(Y, c)
        e.set_params(p)
        with pytest.raises(np.linalg.LinAlgError):
            e.adam_step(0.01)
        np.testing.assert_array_equal(e.get_params()["Z"], p["Z"])


def test_fit_runs_the_training_loop(actuator):
    """RegressionModel.fit(iterations=k): models.py:142-182 with the default case 4 (collapsed U, Adam on everything)."""
    from ffvd_amd.models import RegressionModel
    params, Y, c = actuator
    m = RegressionModel("normal")
    A = m.ARGS
    A.CC, A.DD = params["CC"], params["DD"]
    A.QQ_chol = np.exp(0.5 * params["log_Q"])
    A.RR_chol = np.exp(params["log_Rchols"])
    A.lengthscales, A.variance = np.exp(params["loglengthscales"]), np.exp(params["logvariance"])
    A.UU_ini, A.XX_0_ini, A.x_initialization = params["U"], params["X"][0], params["X"][1:]
    A.control_inputs, A.num_inducing, A.x_dims, A.ZZ = c, 100, [4], params["Z"]
    A.U_collapse, A.kernel_optimization, A.case_val = True, True, 4
    m.fit(Y, kernel_type="SquaredExponential", iterations=5, route="gram", grad=True)
    assert len(m.nll_seq) == 6 and m.nll_seq[1] == pytest.approx(m.nll_seq[0], rel=1e-8)
    assert m.nll_seq[-1] < m.nll_seq[1]
    # the trained values came back into the reference-named attributes and re-evaluate to the same nll
    after = m.model.nll()
    m.model._resident = False
    assert m.model.nll() == pytest.approx(after, rel=1e-12)
    assert not np.allclose(m.model.layers[-1].Z, params["Z"])
    assert len(m.model.window) == 5


def test_collect_samples_formal_on_actuator(actuator, tmp_path):
    """Train a few steps, then predict: base_model.py:197-350 end to end (posterior U, rollouts, predictive y, RMSE),
    compared with the CPU restatement fed with the trained parameters and the same noise."""
    from ffvd_amd.models import RegressionModel
    from oracle import ffvd_oracle as orc
    params, Y, c = actuator
    n_train, test_len, num = 400, 40, 6
    m = RegressionModel("normal")
    A = m.ARGS
    A.CC, A.DD = params["CC"], params["DD"]
    A.QQ_chol = np.exp(0.5 * params["log_Q"])
    A.RR_chol = np.exp(params["log_Rchols"])
    A.lengthscales, A.variance = np.exp(params["loglengthscales"]), np.exp(params["logvariance"])
    A.UU_ini, A.XX_0_ini, A.x_initialization = params["U"], params["X"][0], params["X"][1:n_train + 1]
    A.control_inputs, A.num_inducing, A.x_dims, A.ZZ = c, 100, [4], params["Z"]
    A.U_collapse, A.kernel_optimization, A.case_val = True, True, 4
    m.fit(Y[:n_train], kernel_type="SquaredExponential", iterations=2, route="gram", grad=True)
    eps = np.random.default_rng(11).standard_normal((test_len, num, 4))
    out = m.model.collect_samples_formal(num, 50, c, test_len, U_collapse=True, Y_test=Y[n_train:n_train + test_len],
                                         Y_train_std=1.7, Y_train=Y[:n_train], eps=eps)
    # CPU restatement on the trained parameters
    p = m.model.parameters()
    X = p["X"][0]
    okern = orc.make_kernels(p)
    Q = np.exp(p["log_Q"])
    Lo = orc.kernel_pre_cal(p["Z"], okern)
    Uo, Ho = orc.collapse_u_mean_after_kernel_precalculation(Lo, np.concatenate((X[:-1], c[:n_train]), axis=1), X, p["Z"],
                                                             okern, Q)
    px, pv = orc.rollout(Lo, p["Z"], okern, Uo, Ho, X[-1], c, n_train, test_len, Q, eps)
    ref = orc.predict_y_summary(px, pv, p["CC"], p["DD"], p["log_Rchols"], Y[n_train:n_train + test_len], 1.7)
    np.testing.assert_allclose(out["U_val"], Uo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(out["predict_x"], px, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(out["predict_y"], ref["predict_y"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(out["predict_y_var"], ref["predict_y_var"], rtol=1e-7)
    assert out["RMSE"] == pytest.approx(ref["RMSE"], rel=1e-7)
    assert m.model.RMSE_val == out["RMSE"] and m.model.fit_y.shape == (n_train,)
    # results file with the reference's keys (base_model.py:512-517)
    from ffvd_amd import data_io
    name = data_io.save_results(str(tmp_path / "run"), m.model, Y[n_train:n_train + test_len], Y[:n_train], 1.7,
                                U_val=out["U_val"])
    z = np.load(name)
    for k in ("y_train_vfe", "y_test_vfe", "v_test_vfe_var", "Y_test_data", "Y_train_data", "Y_train_std", "CC_val", "DD_val",
              "log_R_cholesky", "log_QQ", "Z_val", "U_val", "X_val", "k_lengthscales", "k_log_variances", "case", "ll_seq",
              "running_time_seq", "PG_num", "mc_posterior_samples"):
        assert k in z.files, k
    np.testing.assert_array_equal(z["y_test_vfe"], out["predict_y"])
    assert z["X_val"].shape == (n_train, 4) and z["k_lengthscales"].shape == (4, 5)


def test_device_resident_sghmc_matches_oracle_loop():
    """burn_in_op, burn_in_op, sample_op (base_model.py:143-179) on the kernel hyper-parameters, on the device, against
    closed-form gradients + the SG-HMC restatement on the CPU with the same injected noise."""
    params, Y, c, meta = synthetic.make_named("tiny")
    keys = ("logvariance", "loglengthscales")
    rng = np.random.default_rng(3)
    ref = {k: np.array(params[k], dtype=np.float64) for k in keys}
    st = {k: [np.ones_like(ref[k]), np.ones_like(ref[k]), np.ones_like(ref[k]), np.zeros_like(ref[k])] for k in keys}
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        e.set_params(params)
        for burn in (True, True, False):
            noise = {k: rng.standard_normal(ref[k].shape) for k in keys}
            e.sghmc_step(noise, epsilon=0.01, mdecay=0.05, burn_in=burn)
            g = _oracle_mean_grad(dict(params, **ref), Y, c)
            for k in keys:
                out = oo.sghmc_step(ref[k], g[k], *st[k], noise[k], 0.01, 0.05, meta["T"] + 1, burn)
                ref[k], st[k] = out[0], list(out[1:])
        got = e.get_params()
        with pytest.raises(ValueError):
            e.sghmc_step({"X": np.zeros((meta["S"], meta["T"] + 1, meta["D"]))})
    for k in keys:
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
    np.testing.assert_array_equal(got["Z"], params["Z"])


def test_case5_training_loop(actuator):
    """FFVD_Main.py case 5: collapsed U, SG-HMC on the kernel hyper-parameters, Adam on everything else
    (models.py:142-182 -> sghmc_step 21 updates, then train_hypers with a window sample fed for that step only)."""
    from ffvd_amd.models import RegressionModel
    params, Y, c = actuator
    m = RegressionModel("normal")
    A = m.ARGS
    A.CC, A.DD = params["CC"], params["DD"]
    A.QQ_chol = np.exp(0.5 * params["log_Q"])
    A.RR_chol = np.exp(params["log_Rchols"])
    A.lengthscales, A.variance = np.exp(params["loglengthscales"]), np.exp(params["logvariance"])
    A.UU_ini, A.XX_0_ini, A.x_initialization = params["U"], params["X"][0], params["X"][1:]
    A.control_inputs, A.num_inducing, A.x_dims, A.ZZ = c, 100, [4], params["Z"]
    A.kernel_optimization, A.U_optimization, A.Z_optimization, A.U_collapse, A.case_val = False, False, True, True, 5
    m.fit(Y, kernel_type="SquaredExponential", iterations=0, route="gram", grad=True)
    mod = m.model
    assert mod.vars == ["logvariance", "loglengthscales"] and "Z" in mod._adam_train and "logvariance" not in mod._adam_train
    mod.seed(123)
    mod.sghmc_step()
    assert len(mod.window) == 1 and set(mod.window[0]) == {"logvariance", "loglengthscales"}
    chain = mod.engine.get_params()
    assert not np.allclose(chain["logvariance"], params["logvariance"])          # the sampler moved them
    np.testing.assert_array_equal(chain["Z"], params["Z"])                        # ... and nothing else
    mod.sghmc_step()
    chain = mod.engine.get_params()
    t = mod.train_hypers()
    after = mod.engine.get_params()
    assert np.isfinite(t["nll"])
    for k in mod.vars:                                                            # fed, not assigned (:948-949)
        np.testing.assert_array_equal(after[k], chain[k])
    assert not np.array_equal(after["Z"], chain["Z"])                             # Adam moved the rest
    mod.pull_parameters()
    assert mod.layers[-1].kernel[0].logvariance == pytest.approx(after["logvariance"][0])


def _actuator_args(m, params, c, n_train=None):
    A = m.ARGS
    A.CC, A.DD = params["CC"], params["DD"]
    A.QQ_chol = np.exp(0.5 * params["log_Q"])
    A.RR_chol = np.exp(params["log_Rchols"])
    A.lengthscales, A.variance = np.exp(params["loglengthscales"]), np.exp(params["logvariance"])
    A.UU_ini, A.XX_0_ini = params["U"], params["X"][0]
    A.x_initialization = params["X"][1:] if n_train is None else params["X"][1:n_train + 1]
    A.control_inputs, A.num_inducing, A.x_dims, A.ZZ = c, 100, [4], params["Z"]
    return A


def test_case1_trains_the_explicit_u_branch(actuator):
    """FFVD_Main.py case 1: explicit U, everything (U included) trained by Adam; the nll must go down and U must move."""
    from ffvd_amd.models import RegressionModel
    params, Y, c = actuator
    m = RegressionModel("normal")
    A = _actuator_args(m, params, c)
    A.kernel_optimization, A.U_optimization, A.Z_optimization, A.U_collapse, A.case_val = True, True, True, False, 1
    m.fit(Y, kernel_type="SquaredExponential", iterations=6, grad=True)
    assert m.model.vars == [] and "U" in m.model._adam_train
    assert m.nll_seq[1] == pytest.approx(m.nll_seq[0], rel=1e-9) and m.nll_seq[-1] < m.nll_seq[1]
    assert not np.allclose(m.model.layers[-1].U, params["U"])
    after = m.model.nll()
    m.model._resident = False
    assert m.model.nll() == pytest.approx(after, rel=1e-12)


def test_case2_samples_hypers_and_u(actuator):
    """FFVD_Main.py case 2: explicit U; SG-HMC on the kernel hyper-parameters and U, Adam on Z, X, log_Q, C, d, R."""
    from ffvd_amd.models import RegressionModel
    params, Y, c = actuator
    m = RegressionModel("normal")
    A = _actuator_args(m, params, c)
    A.kernel_optimization, A.U_optimization, A.Z_optimization, A.U_collapse, A.case_val = False, False, True, False, 2
    m.fit(Y, kernel_type="SquaredExponential", iterations=0, grad=True)
    mod = m.model
    assert mod.vars == ["logvariance", "loglengthscales", "U"] and "U" not in mod._adam_train
    mod.seed(7)
    mod.sghmc_step()
    chain = mod.engine.get_params()
    assert set(mod.window[0]) == {"logvariance", "loglengthscales", "U"}
    assert not np.allclose(chain["U"], params["U"]) and np.array_equal(chain["Z"], params["Z"])
    t = mod.train_hypers()
    after = mod.engine.get_params()
    assert np.isfinite(t["nll"]) and np.array_equal(after["U"], chain["U"]) and not np.array_equal(after["Z"], chain["Z"])


def test_case6_particle_gibbs_in_the_loop(actuator):
    """FFVD_Main.py case 6 (:317-324): explicit U, X_PG = True.  `gp_x_sampling()` runs between the SG-HMC step and the
    Adam step (models.py:150-168).  Mode "reference" reproduces what the reference's op does (nothing: Appendix B item 6),
    mode "intent" runs the particle-Gibbs sweep on the device and installs a new trajectory unless the uniformly drawn slot
    is the conditioned particle's; the training loop must keep working on the new X."""
    from ffvd_amd.models import RegressionModel
    params, Y, c = actuator
    m = RegressionModel("normal")
    A = _actuator_args(m, params, c)
    A.kernel_optimization, A.U_optimization, A.Z_optimization, A.U_collapse, A.case_val = True, True, True, False, 6
    A.X_PG, A.PG_particles = True, 6
    m.fit(Y, kernel_type="SquaredExponential", iterations=0, grad=True)
    mod = m.model
    assert mod.X_PG and mod.PG_particles == 6
    mod.seed(3)
    x_before = mod.layers[-1].X.copy()
    assert mod.gp_x_sampling(mode="reference") == 0
    np.testing.assert_array_equal(mod.layers[-1].X, x_before)
    replaced = 0
    for _ in range(6):                                   # P(conditioned slot six times in a row) = 6^-6
        replaced += mod.gp_x_sampling()
    assert replaced >= 1
    x_after = mod.layers[-1].X
    assert x_after.shape == x_before.shape and np.all(np.isfinite(x_after)) and not np.array_equal(x_after, x_before)
    np.testing.assert_array_equal(mod.engine.get_params()["X"][0] if mod._resident else x_after, x_after)
    t = mod.train_hypers()                               # the Adam step runs on the resampled trajectory
    assert np.isfinite(t["nll"])
    with pytest.raises(ValueError):
        mod.gp_x_sampling(mode="something")
    # the whole loop through fit(): sghmc_step (no-op in case 6), gp_x_sampling, train_hypers
    m2 = RegressionModel("normal")
    A2 = _actuator_args(m2, params, c)
    A2.kernel_optimization, A2.U_optimization, A2.Z_optimization, A2.U_collapse, A2.case_val = True, True, True, False, 6
    A2.X_PG, A2.PG_particles = True, 4
    m2.fit(Y, kernel_type="SquaredExponential", iterations=3, grad=True)
    assert len(m2.nll_seq) == 4 and np.all(np.isfinite(m2.nll_seq))


def test_fit_called_exactly_like_the_driver(actuator, tmp_path):
    """FFVD_Main.py:343 / :345-349 verbatim: `fit` without `iterations`, `route` or `grad` trains 2 * ARGS.iterations
    rounds (models.py:142); `collect_samples_formal` takes the driver's keyword set and writes the results file."""
    from ffvd_amd.models import RegressionModel
    params, Y, c = actuator
    m = RegressionModel("normal")
    A = _actuator_args(m, params, c, n_train=400)
    A.kernel_optimization, A.U_optimization, A.Z_optimization, A.U_collapse, A.case_val = True, False, True, True, 4
    A.iterations, A.hyperparameter_sampling, A.X_PG = 2, False, False
    Y_train, Y_test = Y[:400], Y[400:440]
    m.fit(Y_train, Y_test=Y_test, tensorboard_savepath="results", dataname="actuator", fileid="x",
          kernel_type="SquaredExponential", kernel_train_flag=True, epsilon=.01)
    assert len(m.nll_seq) == 1 + 2 * A.iterations and m.nll_seq[-1] < m.nll_seq[1]
    assert m.model.engine.grad and m.model.engine.route == "gram"
    path = str(tmp_path / "results" / "actuator" / "C4VFE_result")
    out = m.model.collect_samples_formal(5, 32, A.control_inputs, test_len=len(Y_test), sghmc_var_len=len(m.model.vars),
                                         U_collapse=A.U_collapse, Y_test=Y_test, Y_train_std=1.3, save_path_file=path,
                                         Y_train=Y_train, case="C4", ll_seq=m.ll_seq, running_time_seq=m.running_time_seq,
                                         PG_num=100)
    z = np.load(out["results_file"])
    np.testing.assert_array_equal(z["y_test_vfe"], out["predict_y"])
    assert str(z["case"]) == "C4" and int(z["PG_num"]) == 100 and np.isfinite(out["RMSE"])


def test_adam_set_follows_the_trainable_flags(actuator):
    """AdamOptimizer.minimize only touches trainable=True variables (dgp_model.py:62-69,176-184,
    likelihoods.py:14-55, kernels_multi_output.py:156,160).  Case 6 (X_PG): X is not trainable; likelihood_traning=False
    freezes C, d, R; kernel_optimization=False with kernel_train_flag=False freezes the kernel hyper-parameters."""
    from ffvd_amd.models import RegressionModel
    params, Y, c = actuator
    m = RegressionModel("normal")
    A = _actuator_args(m, params, c)
    A.kernel_optimization, A.U_optimization, A.Z_optimization, A.U_collapse, A.case_val = True, True, True, False, 6
    A.X_PG, A.PG_particles = True, 4
    m.fit(Y, kernel_type="SquaredExponential", iterations=0, grad=True)
    mod = m.model
    assert "X" not in mod._adam_train and "U" in mod._adam_train
    before = mod.parameters()["X"].copy()
    mod.train_hypers()
    after = mod.engine.get_params()
    np.testing.assert_array_equal(after["X"], before)                 # Adam leaves the PG-sampled trajectory alone
    assert not np.array_equal(after["Z"], params["Z"])
    m2 = RegressionModel("normal")
    A2 = _actuator_args(m2, params, c)
    A2.kernel_optimization, A2.U_optimization, A2.Z_optimization, A2.U_collapse, A2.case_val = False, False, True, True, 5
    m2.fit(Y, kernel_type="SquaredExponential", kernel_train_flag=False, likelihood_traning=False, iterations=0,
           route="gram", grad=True)
    mod2 = m2.model
    assert mod2.vars == [] and set(mod2._adam_train) == {"X", "Z", "log_Q"}
    mod2.train_hypers()
    after = mod2.engine.get_params()
    for k in ("logvariance", "loglengthscales", "CC", "DD", "log_Rchols"):
        np.testing.assert_array_equal(after[k], np.asarray(params[k]).reshape(after[k].shape), err_msg=k)
    assert not np.array_equal(after["log_Q"], params["log_Q"])


@pytest.mark.parametrize("mode", ["intent", "reference"])
def test_rollouts_interleaved_with_sghmc_sample_op(mode):
    """collect_samples_formal with sghmc_var_len > 0 (base_model.py:223-240; FFVD_Main cases 2/3/5): `spacing` x
    sample_op before each rollout.  Against a CPU loop built from the closed-form gradient, the SG-HMC restatement and
    the rollout restatement with the same injected noise.  "intent": rollout i sees the variables after its own
    sample_ops; "reference": every rollout sees the final values (the graph is evaluated once, at :326-327)."""
    from ffvd_amd.dgp_model import DGPSSM
    from ffvd_amd.kernels import SquaredExponential
    from ffvd_amd.likelihoods import Gaussian
    from oracle import ffvd_oracle as orc
    params, Y, c, meta = synthetic.make_named("tiny", S=1)
    T, D, M, P = meta["T"], meta["D"], meta["M"], meta["P"]
    n_train, test_len, num, spacing = T, 6, 3, 2
    cc = np.concatenate((c, np.random.default_rng(5).standard_normal((test_len, meta["C"]))))
    kern = [SquaredExponential(P, ARD=True, variance=np.exp(params["logvariance"][d]),
                               lengthscales=np.exp(params["loglengthscales"][d]), kernel_optimization=False) for d in range(D)]
    lik = Gaussian(1, D, CC=params["CC"], DD=params["DD"], RR_chol=np.exp(params["log_Rchols"]))
    X = params["X"][0]
    mod = DGPSSM(Y, [D], M, [kern], lik, QQ_chol=np.exp(0.5 * params["log_Q"]), ZZ=params["Z"], control_inputs=cc,
                 U_ini=params["U"], X_0_ini=X[0], X_train_ini=X[1:], kernel_optimization=False, U_optimization=False,
                 U_collapse=True, Z_optimization=True, case_val=5, prior_type="normal", route="gram", grad=True)
    assert mod.vars == ["logvariance", "loglengthscales"]
    mod.seed(42)
    eps = np.random.default_rng(9).standard_normal((test_len, num, D))
    out = mod.collect_samples_formal(num, spacing, cc, test_len, sghmc_var_len=2, U_collapse=True, Y_train=Y, eps=eps,
                                     rollout_mode=mode)
    # CPU loop
    rng = np.random.default_rng(42)
    keys = ("logvariance", "loglengthscales")
    cur = {k: np.array(params[k], dtype=np.float64) for k in keys}
    st = {k: [np.ones_like(cur[k]), np.ones_like(cur[k]), np.ones_like(cur[k]), np.zeros_like(cur[k])] for k in keys}
    Q = np.exp(params["log_Q"])

    def roll(p, e):
        ok = orc.make_kernels(p)
        Lo = orc.kernel_pre_cal(p["Z"], ok)
        Uo, Ho = orc.collapse_u_mean_after_kernel_precalculation(Lo, np.concatenate((X[:-1], cc[:T]), axis=1), X, p["Z"], ok, Q)
        return orc.rollout(Lo, p["Z"], ok, Uo, Ho, X[-1], cc, n_train, test_len, Q, e)

    px_ref, samples = [], {k: [] for k in keys}
    for i in range(num):
        for _ in range(spacing):
            noise = {k: rng.standard_normal(cur[k].shape) for k in keys}
            g = _oracle_mean_grad(dict(params, **cur), Y, c)
            for k in keys:
                o = oo.sghmc_step(cur[k], g[k], *st[k], noise[k], 0.01, 0.05, T + 1, False)
                cur[k], st[k] = o[0], list(o[1:])
        for k in keys:
            samples[k].append(cur[k].copy())
        if mode == "intent":
            px_ref.append(roll(dict(params, **cur), eps[:, i:i + 1])[0])
    px_ref = np.concatenate(px_ref, axis=0) if mode == "intent" else roll(dict(params, **cur), eps)[0]
    for k in keys:
        np.testing.assert_allclose(out["mc_posterior_samples"][k], np.stack(samples[k]), rtol=1e-8, atol=1e-11, err_msg=k)
    np.testing.assert_allclose(out["predict_x"], px_ref, rtol=1e-6, atol=1e-8)
    assert out["predict_x"].shape == (num, test_len, D)
    with pytest.raises(ValueError):
        mod.collect_samples_formal(num, spacing, cc, test_len, sghmc_var_len=1)


def test_driver_example_runs(tmp_path):
    """examples/ffvd_main_actuator.py = the logic of FFVD_Main.py:192-351 with the import swap, on the committed
    actuator fixture: the default case 4 and the SG-HMC case 5, a few iterations each."""
    import subprocess
    import sys
    from conftest import ROOT
    import os
    for case, extra in ((4, []), (5, ["--samples", "2", "--forced_spacing", "2"])):
        cmd = [sys.executable, os.path.join(ROOT, "examples", "ffvd_main_actuator.py"), "--case_val", str(case),
               "--iterations", "2", "--test_len", "40", "--results_dir", str(tmp_path / "results")] + extra
        proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
        assert "RMSE:" in proc.stdout
        files = list((tmp_path / "results" / "actuator").glob(f"C{case}VFE_result_actuator_*_results.npz"))
        assert len(files) == 1, files
        z = np.load(files[0])
        assert z["y_test_vfe"].shape == (40,) and np.all(np.isfinite(z["y_test_vfe"]))
        if case == 5:
            assert z["mc_posterior_samples_logvariance"].shape == (2, 4)


# ---- sharded, device-resident training step (dgp_model.py:303-305, base_model.py:944-950 across ranks) ----

def _single_engine_trajectory(params, Y, c, meta, steps, lr, collapse=True, noise=None):
    kw = dict(route="gram") if collapse else {}
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], U_collapse=collapse, grad=True, **kw) as e:
        e.set_data(Y, c)
        e.set_params(params)
        nlls = [e.adam_step(lr)["nll"] for _ in range(steps)]
        if noise is not None:
            nlls.append(e.sghmc_step(noise)["nll"])
        return nlls, e.get_params()


def test_sharded_training_step_one_rank_native_rccl():
    """ffvd_adam_step_allreduce / ffvd_sghmc_step_allreduce on a 1-rank communicator: the real binding, the real
    ncclAllReduce of the gradient block in HBM, the update from the reduced block -- bit-equal to the plain steps."""
    from ffvd_amd.distributed import ShardedElbo
    params, Y, c, meta = synthetic.make_named("tiny")
    lr = optim.decayed_learning_rate()
    rng = np.random.default_rng(5)
    noise = {"logvariance": rng.standard_normal(meta["D"]), "loglengthscales": rng.standard_normal((meta["D"], meta["P"]))}
    ref_nll, ref = _single_engine_trajectory(params, Y, c, meta, 4, lr, noise=noise)
    sh = ShardedElbo(params, Y, c, meta, rank=0, world=1, mode="chains", device=0, always_reduce=True, route="gram", grad=True)
    try:
        nlls = [sh.adam_step(lr)["nll"] for _ in range(4)]
        nlls.append(sh.sghmc_step(noise)["nll"])
        got = sh.engine.get_params()
    finally:
        sh.close()
    assert nlls == ref_nll
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


@pytest.mark.parametrize("collapse,mode", [(True, "chains"), (True, "dims"), (False, "dims"), (False, "chains")])
def test_two_shards_three_step_form_equals_single_engine(collapse, mode):
    """Two shard handles on the one GPU, the exchange carried by the host (ffvd_train_local / _exchange_get / _set /
    ffvd_adam_apply): after 4 steps every shard holds the single-engine parameters (1e-9), chain shards their own rows of X,
    latent-dim shards all of X; then one sharded sample_op."""
    from ffvd_amd import distributed as dm
    params, Y, c, meta = synthetic.make_named("small")
    S, D = meta["S"], meta["D"]
    lr = optim.decayed_learning_rate()
    rng = np.random.default_rng(6)
    noise = {"log_Q": rng.standard_normal(D)}
    ref_nll, ref = _single_engine_trajectory(params, Y, c, meta, 4, lr, collapse=collapse, noise=noise)
    kw = dict(route="gram") if collapse else {}
    engines, plans = [], []
    try:
        for r in range(2):
            pl = dm.plan(meta, 2, r, mode)
            e = ElboEngine(meta["T"], D, meta["C"], meta["M"], pl["s_count"], U_collapse=collapse, grad=True,
                           d_begin=pl["d_begin"], d_count=pl["d_count"], shared_terms=pl["shared_terms"], **kw)
            e.set_data(Y, c)
            e.set_params(dict(params, X=params["X"][pl["s_begin"]: pl["s_begin"] + pl["s_count"]]))
            engines.append(e)
            plans.append(pl)
        nlls = []
        for step in range(5):
            blocks = [e.train_local(S) for e in engines]
            assert blocks[0].shape == blocks[1].shape
            red = blocks[0] + blocks[1]
            if step < 4:
                outs = [e.adam_apply(red, lr) for e in engines]
            else:
                outs = [e.sghmc_apply(red, noise) for e in engines]
            np.testing.assert_array_equal(outs[0], outs[1])
            nlls.append(dm.finish(outs[0])["nll"])
        got = [e.get_params() for e in engines]
    finally:
        for e in engines:
            e.close()
    np.testing.assert_allclose(nlls, ref_nll, rtol=1e-9)
    for r, (g, pl) in enumerate(zip(got, plans)):
        for k in ref:
            want = ref[k][pl["s_begin"]: pl["s_begin"] + pl["s_count"]] if k == "X" else ref[k]
            np.testing.assert_allclose(g[k], want, rtol=0, atol=1e-9 * max(1.0, float(np.max(np.abs(want)))), err_msg=f"rank {r} {k}")


def test_plain_steps_refuse_a_shard():
    """A handle that was told it is a shard must not train on its share alone."""
    from ffvd_amd.distributed import ShardedElbo
    params, Y, c, meta = synthetic.make_named("tiny")
    sh = ShardedElbo(params, Y, c, meta, rank=0, world=1, mode="chains", device=0, always_reduce=True, route="gram", grad=True)
    try:
        sh.engine.shard_of = 2                  # what ShardedElbo(world=2) records on its engine
        with pytest.raises(ValueError, match="shard"):
            sh.engine.adam_step(0.01)
    finally:
        sh.close()
    pl = dict(d_begin=0, d_count=1)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True, **pl) as e:
        e.set_data(Y, c)
        e.set_params(params)
        with pytest.raises(ValueError, match="ffvd_adam_step_allreduce"):
            e.adam_step(0.01)
        with pytest.raises(ValueError, match="no pending backward pass"):
            e.adam_apply(np.zeros(int(e.lib.ffvd_train_exchange_count(e._h))), 0.01)


TRAIN_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["FFVD_ROOT"])
import numpy as np, torch.distributed as dist
from ffvd_amd import synthetic, optim
from ffvd_amd.distributed import ShardedElbo
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1]
dist.init_process_group("gloo", rank=rank, world_size=world)       # two processes share the one GPU: gloo carries the block
params, Y, c, meta = synthetic.make_named("small")
sh = ShardedElbo(params, Y, c, meta, rank=rank, world=world, mode=mode, device=0, route="gram", grad=True, collective="torch")
lr = optim.decayed_learning_rate()
nlls = [sh.adam_step(lr)["nll"] for _ in range(4)]
got = sh.engine.get_params()
np.savez(sys.argv[2] + f".rank{rank}.npz", nll=np.array(nlls), s_begin=sh.plan["s_begin"], s_count=sh.plan["s_count"], **got)
sh.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("mode,port", [("chains", "29551"), ("dims", "29552")])
def test_two_processes_sharded_adam_trajectory(tmp_path, mode, port):
    """Two real ranks on the one GPU of the test box run ShardedElbo.adam_step 4 times (forward + backward + ONE exchange of
    the gradient block + fused update per step; gloo carries the block because RCCL needs one GPU per rank): both ranks end
    on the single-engine ffvd_adam_step trajectory to 1e-9."""
    import os
    import subprocess
    import sys
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFVD_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), mode, str(tmp_path / "out")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    params, Y, c, meta = synthetic.make_named("small")
    ref_nll, ref = _single_engine_trajectory(params, Y, c, meta, 4, optim.decayed_learning_rate())
    for r in range(2):
        z = np.load(str(tmp_path / "out") + f".rank{r}.npz")
        np.testing.assert_allclose(z["nll"], ref_nll, rtol=1e-9)
        b, n = int(z["s_begin"]), int(z["s_count"])
        for k in ref:
            want = ref[k][b: b + n] if k == "X" else ref[k]
            np.testing.assert_allclose(z[k], want, rtol=0, atol=1e-9 * max(1.0, float(np.max(np.abs(want)))), err_msg=f"rank {r} {k}")
