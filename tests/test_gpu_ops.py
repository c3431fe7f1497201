"""GPU parity, operator level: each HIP entry point against the NumPy oracle on the same inputs, through the
C ABI.  Tolerances are stated per test (fp64; the north-star acceptance is rtol 1e-4 on the ELBO)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ffvd_amd import _lib, synthetic
from ffvd_amd import conditionals, conditionals_multi_output as cmo
from ffvd_amd import likelihoods, utils
from ffvd_amd.kernels import LinearK, SquaredExponential
from oracle import ffvd_oracle as orc

pytestmark = pytest.mark.gpu


def tiny():
    params, Y, c, meta = synthetic.make_named("tiny")
    X0 = params["X"][0]
    xc = np.concatenate((X0[:-1], c[: meta["T"]]), axis=1)
    kern = [SquaredExponential(meta["P"], variance=np.exp(params["logvariance"][d]),
                               lengthscales=np.exp(params["loglengthscales"][d])) for d in range(meta["D"])]
    return params, Y, c, meta, X0, xc, kern


def test_se_kernel_matrix_matches_golden():
    params, Y, c, meta, X0, xc, kern = tiny()
    g = np.load(os.path.join(GOLDEN, "ops_tiny.npz"))
    for d, k in enumerate(kern):
        np.testing.assert_allclose(k.K(params["Z"]), g["Kuu"][d], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(k.K(xc, params["Z"]), g["Kfu"][d], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(k.Kdiag(xc), g["Kdiag"][d], rtol=1e-15)
    lin = LinearK(meta["P"], variance=0.07)
    np.testing.assert_allclose(lin.K(xc, params["Z"]), g["Klin"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(lin.Kdiag(xc), g["Klin_diag"], rtol=1e-13)


@pytest.mark.parametrize("N,N2,P", [(1, 1, 1), (3, 130, 17), (257, 5, 9), (64, 64, 5)])
def test_kernel_matrix_shapes(N, N2, P):
    rng = np.random.default_rng(N * 1000 + N2)
    X, X2 = rng.standard_normal((N, P)), rng.standard_normal((N2, P))
    ls = 0.5 + rng.random(P)
    k = SquaredExponential(P, variance=0.7, lengthscales=ls)
    ok = orc.SquaredExponential(np.log(0.7), np.log(ls))
    np.testing.assert_allclose(k.K(X, X2), ok.K(X, X2), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(k.K(X), ok.K(X), rtol=1e-12, atol=1e-15)
    assert k.K(X[:0], X2).shape == (0, N2)          # empty input


@pytest.mark.parametrize("n,batch", [(1, 1), (5, 3), (64, 2), (65, 2), (200, 3), (512, 2)])
def test_cholesky_matches_numpy(n, batch):
    lib = _lib.load()
    rng = np.random.default_rng(n + batch)
    B = rng.standard_normal((batch, n, n + 3))
    A = B @ np.swapaxes(B, 1, 2) + 0.1 * np.eye(n)
    L = np.empty_like(A)
    info = np.zeros(batch, dtype=np.int32)
    rc = lib.ffvd_op_cholesky(_lib.dptr(A), n, batch, _lib.dptr(L), info.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int32)))
    assert rc == 0 and not info.any()
    ref = np.linalg.cholesky(A)
    np.testing.assert_allclose(L, ref, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(L @ np.swapaxes(L, 1, 2), A, rtol=1e-12, atol=1e-11)   # factorisation property


def test_cholesky_more_workgroups_than_the_chip_holds():
    """3000 matrices of 4 block columns: a block-column launch has up to 12000 workgroups, far more than are resident at
    once, so late workgroups start long after early ones have finished -- any in-place write another workgroup of the same
    launch still reads (the diagonal block, round 2's first left-looking draft) shows up here and nowhere in the small cases."""
    lib = _lib.load()
    n, batch = 200, 3000
    rng = np.random.default_rng(17)
    B = rng.standard_normal((batch, n, n + 2))
    A = B @ np.swapaxes(B, 1, 2) + 0.5 * np.eye(n)
    L = np.empty_like(A)
    info = np.zeros(batch, dtype=np.int32)
    rc = lib.ffvd_op_cholesky(_lib.dptr(A), n, batch, _lib.dptr(L), info.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int32)))
    assert rc == 0 and not info.any()
    np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("n", [512, 320])
def test_dataflow_cholesky_progress_modes_agree_bitwise(n, monkeypatch):
    """Few matrices (every block row on a compute unit of its own): a block row announces every column it has solved, a gather waits
    term by term, and the diagonal sums are formed column by column (FFVD_DF_FINE = 1, the default there); 2 = the progress words
    without the early sums; 0 = row-level progress as for large batches.  Same terms in the same order: identical bits."""
    lib = _lib.load()
    batch = 32                               # >= 32 matrices: the dataflow launch; 32 x 8 block rows = one per compute unit
    rng = np.random.default_rng(n)
    B = rng.standard_normal((batch, n, n + 2))
    A = B @ np.swapaxes(B, 1, 2) + 0.5 * np.eye(n)
    out = {}
    for mode in ("1", "2", "0"):
        monkeypatch.setenv("FFVD_DF_FINE", mode)
        L = np.empty_like(A)
        info = np.zeros(batch, dtype=np.int32)
        rc = lib.ffvd_op_cholesky(_lib.dptr(A), n, batch, _lib.dptr(L), info.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int32)))
        assert rc == 0 and not info.any()
        out[mode] = L
    assert np.array_equal(out["1"], out["0"]) and np.array_equal(out["2"], out["0"])
    np.testing.assert_allclose(out["1"], np.linalg.cholesky(A), rtol=1e-9, atol=1e-10)


_STALL_SCRIPT = r"""
import sys, time
import numpy as np
from ffvd_amd import _lib, synthetic
from ffvd_amd.engine import ElboEngine
lib = _lib.load()
n, batch = 200, 40                       # >= 32 matrices: the dataflow launch
rng = np.random.default_rng(5)
B = rng.standard_normal((batch, n, n + 2))
A = B @ np.swapaxes(B, 1, 2) + 0.5 * np.eye(n)
L = np.empty_like(A)
info = np.zeros(batch, dtype=np.int32)
t0 = time.perf_counter()
rc = lib.ffvd_op_cholesky(_lib.dptr(A), n, batch, _lib.dptr(L), info.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int32)))
el = time.perf_counter() - t0
msg = lib.ffvd_last_error(None).decode()
err = float(np.max(np.abs(L - np.linalg.cholesky(A))))
print("RC", rc, "ELAPSED", round(el, 2), "INFO0", int(info[0]), "ERR", err, "MSG", msg)
# the ELBO iteration and a training step through a handle: every dataflow launch of this build stalls, every call recovers
params, Y, c, meta = synthetic.make_named("small")
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
    e.set_data(Y, c)
    t0 = time.perf_counter()
    t = e.nll_terms(params)
    el = time.perf_counter() - t0
    n1 = int(lib.ffvd_stall_recoveries(e._h))
    w = lib.ffvd_last_error(e._h).decode()
    tg, g = e.nll_and_grad(params)
    n2 = int(lib.ffvd_stall_recoveries(e._h))
print("ELBO", repr(t["nll"]), repr(tg["nll"]), n1, n2, round(el, 2), "MSG", w)
np.save(sys.argv[1], g["Z"])
"""


def test_dataflow_cholesky_gives_up_instead_of_hanging(tmp_path):
    """Every wait of the one-launch Cholesky is bounded: in the test build `libffvd_hip_dfstall.so` (ffvd_amd/build.py) the first
    block row of matrix 0 never announces its diagonal block, so the rows below it can never proceed.  They must leave after
    the 1 s bound and the launch must end.  Round 3 (VERDICT r2 item 6 / ADVICE): the call then RECOVERS in process -- the batch
    (operator) or the whole iteration (handle) is enqueued once more with the launch-per-column Cholesky, which has no
    inter-workgroup waits -- returns OK with a warning in ffvd_last_error, and the recovered result equals the oracle's."""
    import subprocess
    import sys
    import time
    from ffvd_amd import build as fb
    # never skipped: the variant library is git-ignored, so a fresh checkout builds it here (one hipcc run of
    # kernels.hip; a no-op when its recorded source hash is current) and a box without hipcc FAILS the test
    lib_path = fb.build_variant("dfstall")
    assert os.path.exists(lib_path)
    env = dict(os.environ, FFVD_LIB=lib_path, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               FFVD_NO_TINY="1")           # the handle part exercises the dataflow Cholesky of the multi-kernel schedule
    env.pop("FFVD_CHOL", None)
    t0 = time.perf_counter()
    zpath = str(tmp_path / "dz.npy")
    out = subprocess.run([sys.executable, "-c", _STALL_SCRIPT, zpath], env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RC")][0]
    rc = int(line.split()[1])
    elapsed = float(line.split()[3])
    assert rc == 0, line                                   # recovered
    assert int(line.split()[5]) == 0 and float(line.split()[7]) < 1e-9, line       # info clean, factor = numpy's
    assert "gave up on a bounded wait" in line and "re-run" in line
    assert 0.5 < elapsed < 20.0, line                      # the bound is 1 s per wait; rows give up together via the abort word
    eline = [ln for ln in out.stdout.splitlines() if ln.startswith("ELBO")][0].split()
    from conftest import load_golden
    gold = float(load_golden("small")["B_nll"])
    assert float(eline[1]) == pytest.approx(gold, rel=1e-8) and float(eline[2]) == pytest.approx(gold, rel=1e-8)
    assert int(eline[3]) == 1 and int(eline[4]) == 1       # the stall is remembered (ffvd_stall_hold): the second call does not try the dataflow launch again
    assert 0.5 < float(eline[5]) < 30.0
    assert "re-run with the launch-per-column Cholesky" in " ".join(eline[6:])
    assert time.perf_counter() - t0 < 150.0
    # the product library still works on the same GPU afterwards, and gives the gradient the recovered run produced
    from ffvd_amd import synthetic
    from ffvd_amd.engine import ElboEngine
    params, Y, c, meta = synthetic.make_named("small")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        _, g = e.nll_and_grad(params)
        assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
    dz = np.load(zpath)
    np.testing.assert_allclose(dz, g["Z"], rtol=0, atol=1e-7 * float(np.max(np.abs(g["Z"]))))
    lib = _lib.load()
    rng = np.random.default_rng(6)
    B = rng.standard_normal((40, 100, 102))
    A = B @ np.swapaxes(B, 1, 2) + 0.5 * np.eye(100)
    L = np.empty_like(A)
    info = np.zeros(40, dtype=np.int32)
    rc = lib.ffvd_op_cholesky(_lib.dptr(A), 100, 40, _lib.dptr(L), info.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int32)))
    assert rc == 0 and not info.any()
    np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("n,m", [(1, 1), (7, 3), (64, 64), (100, 512), (200, 1), (513, 130)])
def test_trsm_matches_scipy(n, m):
    """ffvd_op_trsm = tf.linalg.triangular_solve(Lm, Kmn, lower=True) (conditionals_multi_output.py:34): the blocked
    substitution of the Cholesky panel step against scipy, on the factor of an SE K_uu + 1e-5 I (cond ~ 1e6: what the path
    actually solves against) with garbage above the diagonal."""
    from scipy.linalg import solve_triangular
    lib = _lib.load()
    rng = np.random.default_rng(n * 1000 + m)
    z = np.cumsum(0.1 * rng.standard_normal((n, 2)), axis=0)
    K = 0.5 * np.exp(-0.5 * ((z[:, None, :] - z[None, :, :]) ** 2).sum(-1) / 4.0) + 1e-5 * np.eye(n)
    Lc = np.linalg.cholesky(K)
    B = rng.standard_normal((n, m))
    ref = solve_triangular(Lc, B, lower=True)
    L = Lc + np.triu(rng.standard_normal((n, n)), 1)            # the upper triangle must be ignored
    X = np.empty((n, m))
    assert lib.ffvd_op_trsm(_lib.dptr(np.ascontiguousarray(L)), n, _lib.dptr(B), m, _lib.dptr(X)) == 0
    scale = np.max(np.abs(ref))
    np.testing.assert_allclose(X, ref, rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(Lc @ X, B, rtol=0, atol=1e-10 * max(1.0, scale))     # residual: backward stable
    assert lib.ffvd_op_trsm(None, n, _lib.dptr(B), m, _lib.dptr(X)) == _lib.FFVD_EINVAL


def test_cholesky_reports_non_pd():
    lib = _lib.load()
    A = np.eye(70)[None].repeat(2, 0).copy()
    A[1, 66, 66] = -1.0
    L = np.empty_like(A)
    info = np.zeros(2, dtype=np.int32)
    rc = lib.ffvd_op_cholesky(_lib.dptr(A), 70, 2, _lib.dptr(L), info.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int32)))
    assert rc == _lib.FFVD_ENOTPD
    assert info[0] == 0 and info[1] == 67
    assert b"matrix 1" in lib.ffvd_last_error(None)


def test_kernel_pre_cal_matches_oracle():
    params, Y, c, meta, X0, xc, kern = tiny()
    g = np.load(os.path.join(GOLDEN, "ops_tiny.npz"))
    W = cmo.kernel_pre_cal(params["Z"], kern)
    # L^-T of a kappa ~ 1e6 matrix: compare through the defining property too
    np.testing.assert_allclose(np.stack(W), g["Lm_inverse_seq"], rtol=1e-6, atol=1e-7)
    for d, k in enumerate(kern):
        Kuu = g["Kuu"][d] + 1e-5 * np.eye(meta["M"])
        np.testing.assert_allclose(W[d].T @ Kuu @ W[d], np.eye(meta["M"]), atol=1e-8)
        assert np.allclose(np.tril(W[d], -1), 0.0)


def test_conditional_matches_oracle():
    params, Y, c, meta, X0, xc, kern = tiny()
    g = np.load(os.path.join(GOLDEN, "ops_tiny.npz"))
    mean, var = cmo.conditional(xc, params["Z"], kern, params["U"], white=True)
    np.testing.assert_allclose(mean, g["cond_mean"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(var, g["cond_var"], rtol=1e-7, atol=1e-10)
    with pytest.raises(NotImplementedError):
        cmo.conditional(xc, params["Z"], kern, params["U"], white=False)
    # single-kernel variant (conditionals.py, jitter 1e-7): R GPs sharing one kernel
    ok = orc.SquaredExponential(params["logvariance"][0], params["loglengthscales"][0])
    m1, v1 = conditionals.conditional(xc[:10], params["Z"], kern[0], params["U"], white=True)
    m2, v2 = orc.conditional(xc[:10], params["Z"], [ok] * meta["D"], params["U"], white=True, jitter=1e-7)
    np.testing.assert_allclose(m1, m2, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(v1, v2, rtol=1e-5, atol=1e-9)


def test_collapse_matches_oracle():
    params, Y, c, meta, X0, xc, kern = tiny()
    g = np.load(os.path.join(GOLDEN, "ops_tiny.npz"))
    Q = np.exp(params["log_Q"])
    T = float(meta["T"])
    # reference call pattern dgp_model.py:273-280: pre-cal then collapse
    W = cmo.kernel_pre_cal(params["Z"], kern)
    out = cmo.collapse_after_kernel_precalculation(W, xc, X0, params["Z"], kern, Q, T, T)
    np.testing.assert_allclose(out, g["collapse"], rtol=1e-8)
    # and with the ORACLE's L^-T handed in: isolates projection + Gram + Cholesky(H)
    out2 = cmo.collapse_after_kernel_precalculation(list(g["Lm_inverse_seq"]), xc, X0, params["Z"], kern, Q, T, T)
    np.testing.assert_allclose(out2, g["collapse"], rtol=1e-9)
    # mini-batch rescaling factors stay parameterised (:246-248): batch_size != Y_N
    okern = orc.make_kernels(params)
    ref = orc.collapse_after_kernel_precalculation(list(g["Lm_inverse_seq"]), xc, X0, params["Z"], okern, Q, T, 2 * T)
    out3 = cmo.collapse_after_kernel_precalculation(list(g["Lm_inverse_seq"]), xc, X0, params["Z"], kern, Q, T, 2 * T)
    np.testing.assert_allclose(out3, ref, rtol=1e-9)


def test_likelihood_operators_match_oracle():
    """likelihoods.py:76-111 and utils.py:11 as standalone operators (rows a8-a10, a13 of SURVEY 8a)."""
    rng = np.random.default_rng(7)
    N, D, J = 301, 4, 2
    X = rng.standard_normal((N, D))
    lik = likelihoods.Gaussian(J, D, CC=rng.standard_normal((D, J)), DD=rng.standard_normal(J),
                               RR_chol=np.array([[0.4, 0.9], [1.0, 1.0]]))
    ym = lik.predict_mean(X)
    np.testing.assert_allclose(ym, orc.predict_mean(X, lik.CC, lik.DD), rtol=1e-14, atol=1e-15)
    y = rng.standard_normal((N, J))
    R = lik.Rchols[0]
    np.testing.assert_allclose(likelihoods.logdensity_norm_diag(y, ym, R), orc.logdensity_norm_diag(y, ym, R), rtol=1e-13)
    np.testing.assert_allclose(likelihoods.logdensity_norm_diag_nonvec(y, ym, R),
                               orc.logdensity_norm_diag_nonvec(y, ym, R), rtol=1e-13)
    assert likelihoods.logdensity_norm_diag(y[:0], ym[:0], R).shape == (0,)
    mean, var, eps = rng.standard_normal((5, 3)), rng.random((5, 3)), rng.standard_normal((5, 3))
    np.testing.assert_allclose(utils.get_rand((mean, var), eps), orc.get_rand(mean, var, eps), rtol=1e-15)
    with pytest.raises(ValueError):
        lik.predict_mean(X[:, :3])


def test_posterior_u_and_precalc_conditional_match_oracle():
    """collapse_u_mean_after_kernel_precalculation (:206-227) and
    conditional_after_kernel_precalculation (:306-387) incl. the q_sqrt d=0 quirk."""
    params, Y, c, meta, X0, xc, kern = tiny()
    g = np.load(os.path.join(GOLDEN, "ops_tiny.npz"))
    Q = np.exp(params["log_Q"])
    W = list(g["Lm_inverse_seq"])
    Um, Hinv = cmo.collapse_u_mean_after_kernel_precalculation(W, xc, X0, params["Z"], kern, Q)
    np.testing.assert_allclose(Um, g["U_mean"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Hinv, g["H_inv_sqrt"], rtol=1e-8, atol=1e-10)
    m, v = cmo.conditional_after_kernel_precalculation(W, xc[:7], params["Z"], kern, g["U_mean"], q_sqrt=g["H_inv_sqrt"],
                                                        white=True)
    np.testing.assert_allclose(m, g["precalc_mean"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(v, g["precalc_var"], rtol=1e-8, atol=1e-11)
    # without q_sqrt it equals conditional() (same L^-T, same arithmetic)
    m2, v2 = cmo.conditional_after_kernel_precalculation(W, xc, params["Z"], kern, params["U"], white=True)
    np.testing.assert_allclose(m2, g["cond_mean"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(v2, g["cond_var"], rtol=1e-7, atol=1e-10)
    # one-row input: the rollout call pattern (base_model.py:296)
    m1, v1 = cmo.conditional_after_kernel_precalculation(W, xc[:1], params["Z"], kern, g["U_mean"],
                                                          q_sqrt=g["H_inv_sqrt"], white=True)
    np.testing.assert_allclose(m1, g["precalc_mean"][:1], rtol=1e-9, atol=1e-11)
    with pytest.raises(NotImplementedError):
        cmo.conditional_after_kernel_precalculation(W, xc, params["Z"], kern, params["U"], white=False)


@pytest.mark.parametrize("name,R,steps,with_q", [("tiny", 3, 7, True), ("ragged", 70, 5, True), ("small", 5, 12, False)])
def test_rollout_matches_oracle(name, R, steps, with_q):
    """the prediction loop of collect_samples_formal (base_model.py:288-314), R rollouts side by side,
    against the CPU restatement with the same injected noise; keeps the d = 0 q_sqrt quirk (a14)."""
    from ffvd_amd import conditionals_multi_output as cmo
    from ffvd_amd.prediction import rollout
    from ffvd_amd.kernels import SquaredExponential
    params, Y, c, meta = synthetic.make_named(name)
    D, C, T = meta["D"], meta["C"], meta["T"]
    X = params["X"][0]
    Q = np.exp(params["log_Q"])
    okern = orc.make_kernels(params)
    kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]),
                               lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
    rng = np.random.default_rng(5)
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))          # control inputs continue past the training rows
    eps = rng.standard_normal((steps, R, D))
    xc = np.concatenate((X[:-1], c), axis=1)
    Lo = orc.kernel_pre_cal(params["Z"], okern)
    Uo, Ho = orc.collapse_u_mean_after_kernel_precalculation(Lo, xc, X, params["Z"], okern, Q)
    px_o, pv_o = orc.rollout(Lo, params["Z"], okern, Uo, Ho if with_q else None, X[-1], ctrl, T, steps, Q, eps)
    Lg = cmo.kernel_pre_cal(params["Z"], kern)
    Ug, Hg = cmo.collapse_u_mean_after_kernel_precalculation(Lg, xc, X, params["Z"], kern, Q)
    px, pv = rollout(Lg, params["Z"], kern, Ug, Hg if with_q else None, X[-1], ctrl, T, steps, Q, eps)
    # errors compound along the chain (each step feeds the next); 1e-8 after a dozen steps is eps * cond(K_uu)
    np.testing.assert_allclose(px, px_o, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(pv, pv_o, rtol=1e-8, atol=1e-10)
    assert px.shape == (R, steps, D) and np.all(pv > 0)
    # every rollout starts from the same state: with equal noise they coincide
    eps2 = np.repeat(eps[:, :1], R, axis=1)
    px2, _ = rollout(Lg, params["Z"], kern, Ug, None, X[-1], ctrl, T, steps, Q, eps2)
    np.testing.assert_array_equal(px2[0], px2[-1])


def test_step_kernels_with_more_than_eight_inputs():
    """The m-major K build of a step has a general form for P = D + C > 8 inputs (no zero-padded 8-component rows), the one-workgroup
    particle-Gibbs step one for D > 8 latent dims: rollouts (with q_sqrt) and a sweep at D = 9, C = 2 against the CPU restatement."""
    from ffvd_amd import conditionals_multi_output as cmo
    from ffvd_amd.prediction import rollout, pg_sweep
    from ffvd_amd.kernels import SquaredExponential
    from oracle import ffvd_pg_oracle as pgo
    params, Y, c, meta = synthetic.make_named("tiny", D=9, C=2)
    D, C, T = meta["D"], meta["C"], meta["T"]
    X = params["X"][0]
    Q = np.exp(params["log_Q"])
    okern = orc.make_kernels(params)
    kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]),
                               lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
    rng = np.random.default_rng(8)
    R, steps = 5, 6
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
    eps = rng.standard_normal((steps, R, D))
    xc = np.concatenate((X[:-1], c), axis=1)
    Lo = orc.kernel_pre_cal(params["Z"], okern)
    Uo, Ho = orc.collapse_u_mean_after_kernel_precalculation(Lo, xc, X, params["Z"], okern, Q)
    px_o, pv_o = orc.rollout(Lo, params["Z"], okern, Uo, Ho, X[-1], ctrl, T, steps, Q, eps)
    Lg = cmo.kernel_pre_cal(params["Z"], kern)
    Ug, Hg = cmo.collapse_u_mean_after_kernel_precalculation(Lg, xc, X, params["Z"], kern, Q)
    px, pv = rollout(Lg, params["Z"], kern, Ug, Hg, X[-1], ctrl, T, steps, Q, eps)
    np.testing.assert_allclose(px, px_o, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(pv, pv_o, rtol=1e-8, atol=1e-10)
    N = 7
    x0, epg, u = rng.standard_normal((N - 1, D)), rng.standard_normal((T, N - 1, D)), rng.random((T, N - 1))
    Rch = np.exp(params["log_Rchols"])
    po, io = pgo.pg_sweep(Lo, params["Z"], okern, params["U"], X, Y, c, params["CC"], params["DD"], Rch, Q, x0, epg, u)
    pg, ig = pg_sweep(Lg, params["Z"], kern, params["U"], X, Y, c, params["CC"], params["DD"], Rch, Q, x0, epg, u)
    np.testing.assert_array_equal(ig, io)
    np.testing.assert_allclose(pg, po, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("R", [20, 70])
def test_rollout_with_a_dense_q_sqrt(R):
    """The reference hands over q_sqrt = L_H^-T, upper triangular like W = L^-T: W q_sqrt is then triangular too and the step's second
    product skips the zero rows (round 5).  A caller's q_sqrt need not be triangular: with a dense one (slice d = 0, quirk a14) both
    forms of the loop -- the resident one at 20 rollouts, the per-step launches at 70 -- must take the full k range."""
    from ffvd_amd import conditionals_multi_output as cmo
    from ffvd_amd.prediction import rollout
    from ffvd_amd.kernels import SquaredExponential
    params, Y, c, meta = synthetic.make_named("small")
    D, C, T = meta["D"], meta["C"], meta["T"]
    X = params["X"][0]
    Q = np.exp(params["log_Q"])
    okern = orc.make_kernels(params)
    kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]),
                               lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
    rng = np.random.default_rng(17)
    steps = 9
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
    eps = rng.standard_normal((steps, R, D))
    xc = np.concatenate((X[:-1], c), axis=1)
    Lo = orc.kernel_pre_cal(params["Z"], okern)
    Uo, Ho = orc.collapse_u_mean_after_kernel_precalculation(Lo, xc, X, params["Z"], okern, Q)
    assert np.all(np.tril(Ho[0], -1) == 0.0)                       # what the reference passes IS upper triangular
    Hd = Ho + 0.05 * rng.standard_normal(Ho.shape) * np.abs(Ho).max()
    px_o, pv_o = orc.rollout(Lo, params["Z"], okern, Uo, Hd, X[-1], ctrl, T, steps, Q, eps)
    Lg = cmo.kernel_pre_cal(params["Z"], kern)
    px, pv = rollout(Lg, params["Z"], kern, Uo, Hd, X[-1], ctrl, T, steps, Q, eps)
    np.testing.assert_allclose(px, px_o, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(pv, pv_o, rtol=1e-8, atol=1e-10)
    # and the triangular one through the same call differs from the dense result (the inflation term is not a no-op)
    px_t, pv_t = rollout(Lg, params["Z"], kern, Uo, Ho, X[-1], ctrl, T, steps, Q, eps)
    px_to, pv_to = orc.rollout(Lo, params["Z"], okern, Uo, Ho, X[-1], ctrl, T, steps, Q, eps)
    np.testing.assert_allclose(pv_t, pv_to, rtol=1e-8, atol=1e-10)
    assert np.abs(pv_t - pv).max() > 1e-6


_RR_SCRIPT = r"""
import sys
import numpy as np
from ffvd_amd import synthetic, _lib, conditionals_multi_output as cmo
from ffvd_amd.prediction import rollout
from ffvd_amd.kernels import SquaredExponential
params, Y, c, meta = synthetic.make_named("small")
D, C, T = meta["D"], meta["C"], meta["T"]
X = params["X"][0]
Q = np.exp(params["log_Q"])
kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
rng = np.random.default_rng(5)
R, steps = 24, 40
ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
eps = rng.standard_normal((steps, R, D))
xc = np.concatenate((X[:-1], c), axis=1)
Lg = cmo.kernel_pre_cal(params["Z"], kern)
Ug, Hg = cmo.collapse_u_mean_after_kernel_precalculation(Lg, xc, X, params["Z"], kern, Q)
px, pv = rollout(Lg, params["Z"], kern, Ug, Hg, X[-1], ctrl, T, steps, Q, eps)
px2, pv2 = rollout(Lg, params["Z"], kern, Ug, Hg, X[-1], ctrl, T, steps, Q, eps)
assert np.array_equal(px, px2) and np.array_equal(pv, pv2)
print("FALLBACKS", int(_lib.load().ffvd_op_rollout_fallbacks()))
np.savez(sys.argv[1], px=px, pv=pv)
"""


def test_rollout_resident_loop_fenced_build_agrees(tmp_path):
    """the resident rollout loop's hand-offs carry no fences (agent-scope write-through stores and loads, gfx942 / gfx950
    semantics).  The build variant `rrfenced` puts an agent-scope release in front of every count and an acquire behind every wait --
    the construction the HIP memory model prescribes; the same seeded rollout through both builds must give the same bits (the
    fences change when data becomes visible, never what is computed), and neither may fall back to the per-step launches."""
    import subprocess
    import sys
    from ffvd_amd import build as fb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, lib in (("product", fb.build()), ("rrfenced", fb.build_variant("rrfenced"))):
        env = dict(os.environ, FFVD_LIB=lib, PYTHONPATH=root)
        env.pop("FFVD_STEP_LOOP", None)
        path = str(tmp_path / f"{name}.npz")
        out = subprocess.run([sys.executable, "-c", _RR_SCRIPT, path], env=env, capture_output=True, text=True, timeout=200)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "FALLBACKS 0" in out.stdout, out.stdout
        outs[name] = np.load(path)
    np.testing.assert_array_equal(outs["product"]["px"], outs["rrfenced"]["px"])
    np.testing.assert_array_equal(outs["product"]["pv"], outs["rrfenced"]["pv"])


_GRID_SCRIPT = r"""
import sys
import numpy as np
from ffvd_amd import synthetic, conditionals_multi_output as cmo
from ffvd_amd.prediction import rollout
from ffvd_amd.kernels import SquaredExponential
out = {}
for D, R in ((1, 37), (2, 70), (3, 33), (5, 100), (6, 64), (7, 37), (8, 129)):
    params, Y, c, meta = synthetic.make_named("small", D=D)
    C, T = meta["C"], meta["T"]
    X = params["X"][0]
    Q = np.exp(params["log_Q"])
    kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
    rng = np.random.default_rng(D)
    steps = 6
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
    eps = rng.standard_normal((steps, R, D))
    xc = np.concatenate((X[:-1], c), axis=1)
    Lg = cmo.kernel_pre_cal(params["Z"], kern)
    Ug, Hg = cmo.collapse_u_mean_after_kernel_precalculation(Lg, xc, X, params["Z"], kern, Q)
    for q in (0, 1):
        px, pv = rollout(Lg, params["Z"], kern, Ug, Hg if q else None, X[-1], ctrl, T, steps, Q, eps)
        out["px%d_%d" % (D, q)], out["pv%d_%d" % (D, q)] = px, pv
np.savez(sys.argv[1], **out)
"""


def test_skinny_product_grid_is_only_a_placement(tmp_path):
    """Round 5: the skinny product of a step runs on a 1-D grid that maps each latent dim to its own XCDs and walks every other block
    of 32 slots backwards (long and short triangular slabs share a CU).  Which workgroup computes which (slab, row group) must not
    change a bit: rollouts with 1 to 8 latent dims (1, 2, 3 XCDs per dim; with and without the q_sqrt right-hand side; 2 to 5 row
    groups) through the mapped grid and through the plain 3-D grid (FFVD_SKINNY_GRID3D=1)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, extra in (("mapped", {}), ("grid3d", {"FFVD_SKINNY_GRID3D": "1"})):
        env = dict(os.environ, PYTHONPATH=root, FFVD_STEP_LOOP="0", **extra)
        path = str(tmp_path / f"{name}.npz")
        out = subprocess.run([sys.executable, "-c", _GRID_SCRIPT, path], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        outs[name] = np.load(path)
    assert len(outs["mapped"].files) == 28
    for k in outs["mapped"].files:
        np.testing.assert_array_equal(outs["mapped"][k], outs["grid3d"][k], err_msg=k)


_CACHE_SCRIPT = r"""
import sys
import numpy as np
from ffvd_amd import synthetic, _lib, conditionals_multi_output as cmo
from ffvd_amd.prediction import rollout, pg_sweep
from ffvd_amd.kernels import SquaredExponential
out = {}
for name, R in (("small", 24), ("ragged", 70), ("tiny", 5), ("small", 24)):        # different shapes in turn: cached blocks change hands
    params, Y, c, meta = synthetic.make_named(name)
    D, C, T = meta["D"], meta["C"], meta["T"]
    X = params["X"][0]
    Q = np.exp(params["log_Q"])
    kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
    rng = np.random.default_rng(5)
    steps = 9
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
    eps = rng.standard_normal((steps, R, D))
    xc = np.concatenate((X[:-1], c), axis=1)
    Lg = cmo.kernel_pre_cal(params["Z"], kern)
    Ug, Hg = cmo.collapse_u_mean_after_kernel_precalculation(Lg, xc, X, params["Z"], kern, Q)
    px, pv = rollout(Lg, params["Z"], kern, Ug, Hg, X[-1], ctrl, T, steps, Q, eps)
    x0, e2, u = rng.standard_normal((R, D)), rng.standard_normal((T, R, D)), rng.random((T, R))
    parts, idx = pg_sweep(Lg, params["Z"], kern, params["U"], X, Y, c, params["CC"], params["DD"], np.exp(params["log_Rchols"]), Q, x0, e2, u)
    mean, var = cmo.conditional(xc, params["Z"], kern, params["U"], white=True)
    k = f"{name}{len(out)}"
    out.update({k + "px": px, k + "pv": pv, k + "parts": parts, k + "idx": idx, k + "mean": np.asarray(mean), k + "var": np.asarray(var), k + "U": Ug})
assert _lib.load().ffvd_op_release_cache() == 0
np.savez(sys.argv[1], **out)
"""


def test_operator_cache_hands_out_blocks_nobody_relies_on(tmp_path):
    """Round 5: the ffvd_op_* entry points keep their device temporaries in a per-thread cache instead of hipMalloc / hipFree per call.
    A cached block comes back with whatever its last user left in it; FFVD_OP_CACHE_POISON=1 makes every block start as NaN bytes.
    A sequence of operator calls over changing shapes (rollouts, a particle-Gibbs sweep, conditionals, the posterior U) must give
    the same bits with and without the poison -- nothing reads what it has not written."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, FFVD_OP_CACHE_POISON=mode, PYTHONPATH=root)
        path = str(tmp_path / f"cache{mode}.npz")
        out = subprocess.run([sys.executable, "-c", _CACHE_SCRIPT, path], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        outs[mode] = np.load(path)
    assert sorted(outs["0"].files) == sorted(outs["1"].files) and len(outs["0"].files) == 28
    for k in outs["0"].files:
        assert np.all(np.isfinite(outs["0"][k])), k
        np.testing.assert_array_equal(outs["0"][k], outs["1"][k], err_msg=k)


def test_rollout_argument_errors():
    from ffvd_amd.prediction import rollout
    from ffvd_amd.kernels import SquaredExponential
    kern = [SquaredExponential(3, variance=0.5, lengthscales=np.ones(3)) for _ in range(2)]
    Z = np.zeros((4, 3))
    with pytest.raises(ValueError):          # control inputs too short for the horizon
        rollout(np.zeros((2, 4, 4)), Z, kern, np.zeros((4, 2)), None, np.zeros(2), np.zeros((3, 1)), 2, 5, np.ones(2),
                np.zeros((5, 1, 2)))
    with pytest.raises(ValueError):          # eps must be (steps, R, D)
        rollout(np.zeros((2, 4, 4)), Z, kern, np.zeros((4, 2)), None, np.zeros(2), np.zeros((9, 1)), 2, 5, np.ones(2),
                np.zeros((4, 1, 2)))


@pytest.mark.parametrize("name,ov,N,Ydim", [("tiny", {}, 12, 1), ("small", {}, 5, 1), ("tiny", dict(C=0), 2, 1),
                                            ("ragged", {}, 101, 1), ("tiny", {}, 9, 3), ("ragged", {}, 700, 1)])
def test_pg_sweep_matches_oracle(name, ov, N, Ydim):
    """one particle-Gibbs sweep (the intent of PG_for_X_speedup, base_model.py:78-138) against the CPU
    restatement with the same injected draws: identical ancestor indices, particle states to 1e-9 (errors compound along
    the trajectory).  Ydim = 3 exercises the triangular solve of logdensity_norm with a full lower-triangular Rchols.
    700 particles: more than the skinny product takes (512 rows: the tiled projection and the row-major K) and N D > 2048 (the
    general form of the one-workgroup step instead of the fast one)."""
    from ffvd_amd import conditionals_multi_output as cmo
    from ffvd_amd.prediction import pg_sweep
    from ffvd_amd.kernels import SquaredExponential
    from oracle import ffvd_pg_oracle as pgo
    params, Y, c, meta = synthetic.make_named(name, **ov)
    D, C, T = meta["D"], meta["C"], meta["T"]
    X = params["X"][0]
    Q = np.exp(params["log_Q"])
    rng = np.random.default_rng(11)
    CC, DD, R = params["CC"], params["DD"], np.exp(params["log_Rchols"])
    if Ydim > 1:                                   # a multi-output observation model for the weights only
        CC = rng.standard_normal((D, Ydim)) * 0.5
        DD = rng.standard_normal(Ydim) * 0.1
        R = np.tril(rng.standard_normal((Ydim, Ydim)) * 0.2) + np.diag(0.4 + rng.random(Ydim))
        Y = X[1:] @ CC + DD + 0.4 * rng.standard_normal((T, Ydim))
    okern = orc.make_kernels(params)
    kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]),
                               lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
    x0, eps, u = rng.standard_normal((N - 1, D)), rng.standard_normal((T, N - 1, D)), rng.random((T, N - 1))
    Lo = orc.kernel_pre_cal(params["Z"], okern)
    po, io = pgo.pg_sweep(Lo, params["Z"], okern, params["U"], X, Y, c, CC, DD, R, Q, x0, eps, u)
    Lg = cmo.kernel_pre_cal(params["Z"], kern)
    pg, ig = pg_sweep(Lg, params["Z"], kern, params["U"], X, Y, c, CC, DD, R, Q, x0, eps, u)
    assert pg.shape == (T + 1, N - 1, D) and ig.shape == (T, N - 1)
    np.testing.assert_array_equal(ig, io)
    tol = dict(rtol=1e-7, atol=1e-8) if N > 512 else dict(rtol=1e-9, atol=1e-10)        # (the tiled projection sums a row in another order)
    np.testing.assert_allclose(pg, po, **tol)
    np.testing.assert_array_equal(pg[0], x0)
    # a particle that drew the reference's index carries the reference state
    t, i = np.argwhere(ig == N - 1)[0]
    np.testing.assert_array_equal(pg[t + 1, i], X[t + 1])
    # u -> 1 always selects the last candidate (the reference): the sweep then returns X itself from t = 1 on
    p1, i1 = pg_sweep(Lg, params["Z"], kern, params["U"], X, Y, c, CC, DD, R, Q, x0, eps, np.full_like(u, 1.0 - 1e-16))
    assert np.all(i1 == N - 1)
    np.testing.assert_array_equal(p1[1:], np.repeat(X[1:, None, :], N - 1, axis=1))


def test_pg_sweep_argument_errors():
    from ffvd_amd.prediction import pg_sweep
    from ffvd_amd.kernels import SquaredExponential
    kern = [SquaredExponential(3, variance=0.5, lengthscales=np.ones(3)) for _ in range(2)]
    Z, L, U = np.zeros((4, 3)), np.zeros((2, 4, 4)), np.zeros((4, 2))
    X, Y, c = np.zeros((6, 2)), np.zeros((5, 1)), np.zeros((5, 1))
    CC, DD, R, Q = np.ones((2, 1)), np.zeros(1), np.ones((1, 1)), np.ones(2)
    good = dict(x0=np.zeros((3, 2)), eps=np.zeros((5, 3, 2)), unif=np.zeros((5, 3)))
    with pytest.raises(ValueError):          # uniforms outside [0, 1)
        pg_sweep(L, Z, kern, U, X, Y, c, CC, DD, R, Q, good["x0"], good["eps"], np.ones((5, 3)))
    with pytest.raises(ValueError):          # eps must be (X_N - 1, PG_particles - 1, D)
        pg_sweep(L, Z, kern, U, X, Y, c, CC, DD, R, Q, good["x0"], np.zeros((4, 3, 2)), good["unif"])
    with pytest.raises(ValueError):          # control inputs too short
        pg_sweep(L, Z, kern, U, X, Y, c[:3], CC, DD, R, Q, **good)
    with pytest.raises(ValueError):          # non-positive diagonal of Rchols
        pg_sweep(L, Z, kern, U, X, Y, c, CC, DD, np.zeros((1, 1)), Q, **good)


def test_pg_sweep_linear_kernel():
    """The sweep with the D-kernel LinearK list of config 5 (kernels.py:270-281 through the multi-output conditional)."""
    from ffvd_amd.prediction import pg_sweep
    from oracle import ffvd_pg_oracle as pgo
    params, Y, c, meta = synthetic.make_named("small_lin")
    D, C, T = meta["D"], meta["C"], meta["T"]
    okern = orc.make_kernels(params, kernel_type=meta["kernel_type"])
    Lm = orc.kernel_pre_cal(params["Z"], okern)
    kern = [LinearK(D + C, variance=np.exp(params["logvariance"][d])) for d in range(D)]
    X = params["X"][0]
    rng = np.random.default_rng(3)
    R, Q = np.exp(params["log_Rchols"]), np.exp(params["log_Q"])
    for N in (7, 400):        # 400 x D = 6: the general form of the step behind the skinny product (the fast one takes N D <= 2048)
        x0, eps, u = rng.standard_normal((N - 1, D)), rng.standard_normal((T, N - 1, D)), rng.random((T, N - 1))
        po, io = pgo.pg_sweep(Lm, params["Z"], okern, params["U"], X, Y, c, params["CC"], params["DD"], R, Q, x0, eps, u)
        pg, ig = pg_sweep(Lm, params["Z"], kern, params["U"], X, Y, c, params["CC"], params["DD"], R, Q, x0, eps, u)
        np.testing.assert_array_equal(ig, io)
        np.testing.assert_allclose(pg, po, **(dict(rtol=1e-8, atol=1e-9) if N < 100 else dict(rtol=1e-7, atol=1e-8)))


def test_resident_rollout_loop_gives_up_and_the_launches_take_over(monkeypatch):
    """The rollout loop with resident operands is one launch whose 128 workgroups wait for each other: every wait is bounded, and a
    launch that cannot finish (here: one workgroup leaves at step 3) must end with its abort word set, after which the per-step launches
    run the loop from the initial rows -- same bits as FFVD_STEP_LOOP=0."""
    import time
    from ffvd_amd import conditionals_multi_output as cmo
    from ffvd_amd.prediction import rollout
    params, Y, c, meta = synthetic.make_named("small")
    D, C, T = meta["D"], meta["C"], meta["T"]
    X, Q = params["X"][0], np.exp(params["log_Q"])
    kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]), lengthscales=np.exp(params["loglengthscales"][d]))
            for d in range(D)]
    rng = np.random.default_rng(11)
    R, steps = 20, 12
    ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
    eps = rng.standard_normal((steps, R, D))
    L = cmo.kernel_pre_cal(params["Z"], kern)
    U, H = cmo.collapse_u_mean_after_kernel_precalculation(L, np.concatenate((X[:-1], c), axis=1), X, params["Z"], kern, Q)
    monkeypatch.setenv("FFVD_STEP_LOOP", "0")
    want = rollout(L, params["Z"], kern, U, H, X[-1], ctrl, T, steps, Q, eps)
    monkeypatch.setenv("FFVD_STEP_LOOP", "2")
    monkeypatch.setenv("FFVD_RR_TEST_STALL", "1")
    t0 = time.perf_counter()
    got = rollout(L, params["Z"], kern, U, H, X[-1], ctrl, T, steps, Q, eps)
    el = time.perf_counter() - t0
    assert 0.09 < el < 5.0, el                     # the bounded wait (0.1 s) fired, nothing hung
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])


def test_step_loops_equal_the_per_step_launches(monkeypatch):
    """Round 4: the step loops of the rollouts and of the particle-Gibbs sweep can run as ONE persistent launch whose
    workgroups walk the phases of every step and meet at a grid-wide barrier (loops.hip, FFVD_STEP_LOOP=1 -- opt-in: measured 2-4 x
    SLOWER than the three / four dependent launches per step, a 256-workgroup barrier costs more than a kernel boundary on this chip).
    Both forms call the same kernel bodies (step_bodies.h): the results must be bit-identical, with and without the q_sqrt inflation,
    at 7 and at 100 rollouts (one and several row blocks per slab), LinearK included."""
    from ffvd_amd import conditionals_multi_output as cmo
    from ffvd_amd.prediction import rollout, pg_sweep
    from ffvd_amd.kernels import SquaredExponential, LinearK
    rng = np.random.default_rng(3)
    for name, R, steps in (("small", 7, 40), ("ragged", 100, 25), ("small_lin", 33, 30)):
        params, Y, c, meta = synthetic.make_named(name)
        D, C, T = meta["D"], meta["C"], meta["T"]
        X = params["X"][0]
        Q = np.exp(params["log_Q"])
        if meta["kernel_type"] == "LinearK":
            kern = [LinearK(D + C, variance=np.exp(params["logvariance"][d])) for d in range(D)]
        else:
            kern = [SquaredExponential(D + C, variance=np.exp(params["logvariance"][d]),
                                       lengthscales=np.exp(params["loglengthscales"][d])) for d in range(D)]
        ctrl = np.concatenate((c, rng.standard_normal((steps, C))))
        eps = rng.standard_normal((steps, R, D))
        xc = np.concatenate((X[:-1], c), axis=1)
        L = cmo.kernel_pre_cal(params["Z"], kern)
        U, H = cmo.collapse_u_mean_after_kernel_precalculation(L, xc, X, params["Z"], kern, Q)
        x0, epg, u = rng.standard_normal((R, D)), rng.standard_normal((T, R, D)), rng.random((T, R))
        Rch = np.exp(params["log_Rchols"])
        out = {}
        for mode, env in (("loop", "1"), ("launches", "0"), ("resident", "2")):
            monkeypatch.setenv("FFVD_STEP_LOOP", env)
            out[mode] = (rollout(L, params["Z"], kern, U, H, X[-1], ctrl, T, steps, Q, eps),
                         rollout(L, params["Z"], kern, U, None, X[-1], ctrl, T, steps, Q, eps),
                         pg_sweep(L, params["Z"], kern, params["U"], X, Y, c, params["CC"], params["DD"], Rch, Q, x0, epg, u))
        monkeypatch.delenv("FFVD_STEP_LOOP", raising=False)
        for a, b in zip(out["loop"], out["launches"]):
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
        # the rollout loop with resident operands (FFVD_STEP_LOOP=2; the default up to 32 rollouts, on request up to 64, with M <= 512 and 8 latent dims;
        # 100 rollouts fall back to the launches): other summation order inside a row of F, same values to rounding
        for a, b in zip(out["resident"][:2], out["launches"][:2]):
            np.testing.assert_allclose(a[0], b[0], rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(a[1], b[1], rtol=1e-9, atol=1e-10)
