"""GPU parity, model level: the full ELBO (`nll` + component terms) through the C ABI against the committed
golden vectors, the oracle on the same seeded inputs, and size-independent properties at BASELINE's full size."""
import os

import numpy as np
import pytest

from conftest import load_golden
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from ffvd_amd.dgp_model import DGPSSM
from ffvd_amd.kernels import SquaredExponential
from ffvd_amd.likelihoods import Gaussian
from ffvd_amd.models import RegressionModel
from oracle import ffvd_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-9      # fp64 GPU vs fp64 CPU oracle; north-star acceptance is 1e-4
TERMS_B = ("nll_part_prior", "nll_log_likelihood", "x_t_prior_Q", "nll_reg_trace_inverse_Q_B", "later_term1",
           "later_term2", "nll")
TERMS_A = ("nll_part_prior", "nll_log_likelihood", "x_t_prior_Q", "nll_reg_trace_inverse_Q_B", "nll")


@pytest.fixture(params=["one_launch", "multi_kernel"])
def schedule(request, monkeypatch):
    """Shapes of the reference's own experiment size take the one-launch path of tiny.hip by default; the tests that take this
    fixture also run on the multi-kernel schedule of rounds 1-3 (FFVD_NO_TINY=1, read when a handle is created)."""
    if request.param == "multi_kernel":
        monkeypatch.setenv("FFVD_NO_TINY", "1")
    return request.param


def run_engine(params, Y, c, meta, collapse, **kw):
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], params["X"].shape[0], Ydim=Y.shape[1],
                    kernel_type=meta["kernel_type"], U_collapse=collapse, **kw) as eng:
        eng.set_data(Y, c)
        return eng.nll_terms(params)


def assert_terms(got, ref, names, rtol=RTOL, prefix=""):
    for n in names:
        r = float(ref[prefix + n])
        # the trace term is a cancellation (T*sigma^2 - |F|^2): its absolute floor is set by |F|^2 * eps
        atol = 1e-11 if n != "nll_reg_trace_inverse_Q_B" else 1e-10
        assert got[n] == pytest.approx(r, rel=rtol, abs=atol), (n, got[n], r)


@pytest.mark.parametrize("name", ["tiny", "small", "ragged", "small_lin"])
@pytest.mark.parametrize("branch", ["B", "A"])
def test_synthetic_golden(name, branch, schedule):
    params, Y, c, meta = synthetic.make_named(name)
    g = load_golden(name)
    got = run_engine(params, Y, c, meta, collapse=(branch == "B"))
    assert_terms(got, g, TERMS_B if branch == "B" else TERMS_A, prefix=branch + "_")
    np.testing.assert_allclose(got["nll_per_chain"], g[branch + "_nll_per_chain"], rtol=RTOL)


@pytest.mark.parametrize("branch", ["B", "A"])
def test_actuator_config1(actuator, branch):
    """BASELINE config 1: actuator, M=100, D=4, T=512 -- golden values reproduce SURVEY's anchors."""
    params, Y, c = actuator
    meta = dict(T=512, D=4, C=1, M=100, kernel_type="SquaredExponential")
    p = dict(params)
    p["X"] = params["X"][None]
    got = run_engine(p, Y, c, meta, collapse=(branch == "B"))
    assert_terms(got, load_golden("actuator"), TERMS_B if branch == "B" else TERMS_A, prefix=branch + "_")


def test_model_facade_matches_engine(actuator):
    """RegressionModel/DGPSSM (reference-named surface) give the same nll as the raw engine."""
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
    m.fit(Y, kernel_type="SquaredExponential", iterations=0)
    g = load_golden("actuator")
    assert m.nll_seq[0] == pytest.approx(float(g["B_nll"]), rel=RTOL)
    t = m.model.nll_terms()
    assert t["later_term1"] == pytest.approx(float(g["B_later_term1"]), rel=RTOL)


@pytest.mark.parametrize("name", ["tiny", "small", "ragged", "small_lin"])
def test_gram_route_matches_golden(name):
    """route="gram": log|K_uu + K_uf K_fu/Q| - log|K_uu| form of the collapsed bound.  Same algebra, different
    rounding (error ~ eps * cond(K_uu) instead of eps * sqrt(cond)): nll to 1e-8, terms to 1e-7 relative (the
    trace term is a cancellation, absolute 1e-9)."""
    params, Y, c, meta = synthetic.make_named(name)
    g = load_golden(name)
    got = run_engine(params, Y, c, meta, collapse=True, route="gram")
    for n in TERMS_B:
        r = float(g["B_" + n])
        assert got[n] == pytest.approx(r, rel=(1e-8 if n == "nll" else 1e-7), abs=1e-9), (n, got[n], r)
    np.testing.assert_allclose(got["nll_per_chain"], g["B_nll_per_chain"], rtol=1e-8)
    ref = run_engine(params, Y, c, meta, collapse=True)
    assert got["nll"] == pytest.approx(ref["nll"], rel=1e-8)


def test_gram_route_actuator(actuator):
    params, Y, c = actuator
    meta = dict(T=512, D=4, C=1, M=100, kernel_type="SquaredExponential")
    p = dict(params)
    p["X"] = params["X"][None]
    got = run_engine(p, Y, c, meta, collapse=True, route="gram")
    g = load_golden("actuator")
    assert got["nll"] == pytest.approx(float(g["B_nll"]), rel=1e-8)
    for n in TERMS_B:
        assert got[n] == pytest.approx(float(g["B_" + n]), rel=1e-6, abs=1e-9)
    with pytest.raises(ValueError):
        run_engine(p, Y, c, meta, collapse=False, route="gram")     # explicit-U branch has no Gram form


GRAD_KEYS = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")


@pytest.mark.parametrize("name", ["tiny", "ragged", "small"])
def test_gradient_matches_autograd(name, schedule):
    """d nll / d (X, Z, kernel hypers, Q, C, d, R) from the HIP backward pass against torch autograd of
    the independent oracle restatement (what tf.gradients(nll, vars), base_model.py:148, returns)."""
    from oracle import ffvd_oracle_torch as orct
    params, Y, c, meta = synthetic.make_named(name)
    S = params["X"].shape[0]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram", grad=True) as e:
        e.set_data(Y, c)
        terms, g = e.nll_and_grad(params)
        terms2, g2 = e.nll_and_grad(params)
    for k in GRAD_KEYS:
        np.testing.assert_array_equal(g[k], g2[k])            # deterministic
    ref = {k: np.zeros_like(g[k]) for k in GRAD_KEYS}
    nll_ref = 0.0
    for s in range(S):
        p = dict(params)
        p["X"] = params["X"][s]
        t, ga = orct.nll_and_grad(p, Y, c, wrt=GRAD_KEYS, U_collapse=True)
        nll_ref += t["nll"] / S
        ref["X"][s] = ga["X"] / S
        for k in GRAD_KEYS[1:]:
            ref[k] += ga[k] / S
    assert terms["nll"] == pytest.approx(nll_ref, rel=1e-8)
    for k in GRAD_KEYS:
        scale = np.max(np.abs(ref[k])) + 1e-300
        err = np.max(np.abs(g[k] - ref[k])) / scale
        # Z and the lengthscales go through K_uu^-1 - A^-1: the GPU and the closed-form oracle both evaluate it in
        # whitened variables (W (I - H^-1) W^T) and sit 1e-9..1e-8 apart on these shapes; torch autograd of the
        # restatement, which the oracle itself is checked against, carries 1e-7..1e-6 there
        tol = 1e-6 if k in ("Z", "loglengthscales", "logvariance") else 1e-7
        assert err < tol, (k, err)


def test_sharded_gradients_sum_to_the_whole():
    """SURVEY 8(e) for the backward pass: gradients of chain shards / latent-dim shards (each scaled by
    1/S_total, priors weighted by the shard's share) add up to the unsharded gradient."""
    params, Y, c, meta = synthetic.make_named("small")
    S, D = meta["S"], meta["D"]

    def grads(s0, s1, d0, dc, shared):
        p = dict(params)
        p["X"] = params["X"][s0:s1]
        with ElboEngine(meta["T"], D, meta["C"], meta["M"], s1 - s0, d_begin=d0, d_count=dc, shared_terms=shared,
                        route="gram", grad=True) as e:
            e.set_data(Y, c)
            return e.nll_and_grad(p, S_total=S)[1]

    whole = grads(0, S, 0, D, True)
    a, b = grads(0, 1, 0, D, True), grads(1, S, 0, D, True)
    for k in GRAD_KEYS[1:]:
        # regrouping the chains changes summation orders only: 1e-8 on the K_uu-side gradients (5e-6 before the backward
        # pass moved to whitened variables)
        tol = 1e-7 if k in ("Z", "loglengthscales", "logvariance") else 1e-10
        np.testing.assert_allclose(a[k] + b[k], whole[k], rtol=0, atol=tol * np.max(np.abs(whole[k])))
    np.testing.assert_allclose(np.concatenate((a["X"], b["X"])), whole["X"], rtol=1e-9, atol=1e-15)
    a, b = grads(0, S, 0, 1, True), grads(0, S, 1, D - 1, False)
    for k in GRAD_KEYS:
        np.testing.assert_allclose(a[k] + b[k], whole[k], rtol=1e-9, atol=1e-10 * np.max(np.abs(whole[k])) + 1e-300)


def test_chain_and_dim_sharding_sum_to_the_whole():
    """partial sums of chain shards / latent-dim shards add up to the unsharded sums."""
    params, Y, c, meta = synthetic.make_named("small")
    S, D = meta["S"], meta["D"]

    def sums(collapse, s0, s1, d0, dc, shared):
        p = dict(params)
        p["X"] = params["X"][s0:s1]
        with ElboEngine(meta["T"], D, meta["C"], meta["M"], s1 - s0, U_collapse=collapse, d_begin=d0, d_count=dc,
                        shared_terms=shared) as e:
            e.set_data(Y, c)
            return e.elbo_sums(p)

    for collapse in (True, False):
        whole = sums(collapse, 0, S, 0, D, True)
        chains = sums(collapse, 0, 1, 0, D, True) + sums(collapse, 1, S, 0, D, True)
        np.testing.assert_allclose(chains, whole, rtol=1e-12, atol=1e-13)
        dims = sums(collapse, 0, S, 0, 1, True) + sums(collapse, 0, S, 1, D - 1, False)
        np.testing.assert_allclose(dims, whole, rtol=1e-12, atol=1e-13)
        assert whole[7] == S


def test_chains_per_pass_is_invisible():
    """Evaluating the chains in several passes changes nothing but rounding: the Gram kernel cuts its rows into
    ranges according to the units per pass (split-K), so the summation order -- not the arithmetic -- depends on it."""
    params, Y, c, meta = synthetic.make_named("small")
    a = run_engine(params, Y, c, meta, True, chains_per_pass=1)
    b = run_engine(params, Y, c, meta, True, chains_per_pass=3)
    d = run_engine(params, Y, c, meta, True)
    for n in TERMS_B:
        assert a[n] == pytest.approx(d[n], rel=1e-13, abs=1e-15) and b[n] == pytest.approx(d[n], rel=1e-13, abs=1e-15)
    np.testing.assert_allclose(b["nll_per_chain"], d["nll_per_chain"], rtol=1e-13)


def test_resident_parameters_and_repeatability():
    params, Y, c, meta = synthetic.make_named("small")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"]) as e:
        e.set_data(Y, c)
        e.set_params(params)
        a, b = e.elbo_sums(), e.elbo_sums()
        np.testing.assert_array_equal(a, b)           # bitwise reproducible: no atomics anywhere
        ms = e.time_elbo(3)
        assert ms > 0
        np.testing.assert_array_equal(e.elbo_sums(), a)
        st = e.profile_stages()
        if int(e.lib.ffvd_single_launch(e._h)):
            assert st["gram_H"] > 0                       # the one launch of tiny.hip is booked there
        else:
            assert st["project_F"] > 0 and st["gram_H"] > 0


def test_wide_inputs_and_multi_output():
    """P up to 17 (config-5 shape) and Ydim > 1."""
    params, Y, c, meta = synthetic.make_workload(T=200, D=16, C=1, M=70, S=2, kernel_type="LinearK", U_collapse=False)
    got = run_engine(params, Y, c, meta, collapse=False)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=False, kernel_type="LinearK")
    assert_terms(got, ref, TERMS_A)
    params, Y, c, meta = synthetic.make_workload(T=130, D=3, C=2, M=40, S=2, Ydim=2)
    params["log_Rchols"] = np.log(np.array([[0.4, 0.7], [9.0, 9.0]]))     # only row 0 is used (dgp_model.py:250)
    got = run_engine(params, Y, c, meta, collapse=True)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
    assert_terms(got, ref, TERMS_B)


@pytest.mark.parametrize("route", ["reference", "gram"])
def test_more_than_512_inducing_points(route):
    """M = 600 -> Mp = 640: two column groups in the projection kernel, 15 Gram tiles, 10 Cholesky block steps."""
    params, Y, c, meta = synthetic.make_workload(T=700, D=2, C=1, M=600, S=2)
    got = run_engine(params, Y, c, meta, collapse=True, route=route)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
    tol = 1e-9 if route == "reference" else 1e-7
    # with 600 inducing points for 700 transitions the trace term is ~2e-5, a cancellation of two O(1) numbers:
    # the Gram route's absolute error (~ eps * cond(K_uu)) is 1e-9 there, i.e. large only relative to that term
    atol = 1e-10 if route == "reference" else 1e-8
    for n in TERMS_B:
        assert got[n] == pytest.approx(ref[n], rel=tol, abs=atol), (n, got[n], ref[n])
    assert got["nll"] == pytest.approx(ref["nll"], rel=1e-8)
    if route == "reference":
        got = run_engine(params, Y, c, meta, collapse=False)
        ref = orc.nll_terms_chains(params, Y, c, U_collapse=False)
        assert_terms(got, ref, TERMS_A)


def gram_route_tolerance(params, meta):
    """Predicted absolute error of the Gram route's nll terms (terms of order 1): the route forms log|K_uu + G/Q| - log|K_uu| and
    tr(K_uu^-1 G), whose rounding error is a small multiple of eps * cond(K_uu + jitter I) of the terms (SURVEY section 7
    "Conditioning"): 4 eps cond.  Measured in units of eps * cond (tools/factor_acc.py, both forms of the Cholesky's diagonal factor):
    0.003 ... 0.15 at M = 512, 0.1 ... 0.7 at M = 768, 0.45 ... 0.7 at M = 1024, 1.1 ... 2.1 at M = 2048."""
    kappa = 0.0
    for d in range(meta["D"]):
        w = np.linalg.eigvalsh(orc.SquaredExponential(params["logvariance"][d], params["loglengthscales"][d]).K(params["Z"])
                               + orc.JITTER_MULTI_OUTPUT * np.eye(meta["M"]))
        kappa = max(kappa, w[-1] / w[0])
    return 4.0 * np.finfo(np.float64).eps * kappa


def test_config4_shape_m2048():
    """BASELINE configs[3] shape along M (M = 2048: 4 column groups, 32 Cholesky block steps, 136 Gram tiles) at a
    T the CPU oracle finishes in seconds; fp64 (the reference's dtype)."""
    params, Y, c, meta = synthetic.make_workload(T=2304, D=2, C=1, M=2048, S=1)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
    got = run_engine(params, Y, c, meta, collapse=True)
    for n in TERMS_B:
        assert got[n] == pytest.approx(ref[n], rel=1e-8, abs=1e-9), (n, got[n], ref[n])
    got = run_engine(params, Y, c, meta, collapse=True, route="gram")
    # Gram route at M = 2048, T = 2304: the predicted bound 4 eps cond(K_uu) = 5.3e-8 (cond 6e7) of an nll of -0.113; measured
    # 0.7e-8 ... 2.9e-8 with the different summation orders the factorisation has had
    tol = gram_route_tolerance(params, meta)
    assert 2e-8 < tol < 1e-7
    assert abs(got["nll"] - ref["nll"]) <= tol


def test_config5_linear_kernel_dim_shards():
    """BASELINE configs[4]: LinearK, explicit-U branch, T=4096, x_dim=16, M=512, latent dims sharded 8 ways
    (here: eight engines on one GPU, partial sums added on the host as the all-reduce would)."""
    params, Y, c, meta = synthetic.make_named("c5")
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=False, kernel_type="LinearK")
    from ffvd_amd.distributed import plan, finish
    total = np.zeros(8)
    for r in range(8):
        pl = plan(meta, 8, r, "dims")
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], kernel_type="LinearK", U_collapse=False,
                        d_begin=pl["d_begin"], d_count=pl["d_count"], shared_terms=pl["shared_terms"]) as e:
            e.set_data(Y, c)
            total += e.elbo_sums(params)
    got = finish(total)
    # K_uu of a linear kernel has rank P = 17 << M: only the 1e-5 jitter makes it positive definite, so the
    # whitened solve is conditioned like 1e5 * |K|; the terms still agree to 1e-7
    for n in TERMS_A:
        assert got[n] == pytest.approx(ref[n], rel=1e-7, abs=1e-9), (n, got[n], ref[n])


def test_native_rccl_all_reduce_one_rank():
    """The multi-GPU step on one rank through the C ABI alone (no torch): ffvd_comm_unique_id / ffvd_comm_init build a
    1-rank ncclComm_t, ffvd_elbo_allreduce runs kernels -> finalize -> ncclAllReduce on the handle's stream -> copy back
    (world_size 1 exercises the RCCL binding, the communicator and the collective call on the result block)."""
    from ffvd_amd.distributed import ShardedElbo, finish
    params, Y, c, meta = synthetic.make_named("small")
    g = load_golden("small")
    sh = ShardedElbo(params, Y, c, meta, rank=0, world=1, mode="chains", device=0, always_reduce=True)
    try:
        t = finish(sh.step())
        assert t["nll"] == pytest.approx(float(g["B_nll"]), rel=RTOL)
        assert finish(sh.step()) == t
    finally:
        sh.close()
    # backward pass through the same collective path (8 sums + packed shared-parameter gradients in one ncclAllReduce)
    shg = ShardedElbo(params, Y, c, meta, rank=0, world=1, mode="chains", device=0, always_reduce=True, route="gram", grad=True)
    try:
        tg, grads = shg.nll_and_grad()
    finally:
        shg.close()
    assert tg["nll"] == pytest.approx(float(g["B_nll"]), rel=1e-8)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        _, gl = e.nll_and_grad(params)
    for k in GRAD_KEYS:
        np.testing.assert_array_equal(grads[k], gl[k])


def test_caller_owned_communicator_and_async_form():
    """ffvd_elbo_allreduce with a communicator the CALLER owns (`ffvd_elbo_allreduce(h, rccl_comm)`): here the
    ncclComm_t of a second handle; and the enqueue-only form followed by ffvd_sync."""
    import ctypes as ct
    from ffvd_amd import _lib
    params, Y, c, meta = synthetic.make_named("tiny")
    g = load_golden("tiny")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"]) as owner, \
            ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"]) as e:
        owner.comm_init(1, 0, owner.comm_unique_id())
        comm = owner.lib.ffvd_comm_get(owner._h)
        assert comm
        e.set_data(Y, c)
        e.set_params(params)
        sums = e.elbo_allreduce(comm)
        assert sums[6] / sums[7] == pytest.approx(float(g["B_nll"]), rel=RTOL)
        with pytest.raises(ValueError, match="communicator"):
            e.elbo_allreduce(None)                     # this handle has none of its own
        _lib.check(e.lib.ffvd_elbo_allreduce_async(e._h, comm, None), e._h, "ffvd_elbo_allreduce_async")
        e.sync()


def test_async_step_reports_failed_factorisation():
    """the _async forms cannot report a non-PD matrix; ffvd_sync() must (include/ffvd_abi.h), and so must the
    collective step built on them -- never a silent NaN."""
    from ffvd_amd.distributed import ShardedElbo
    params, Y, c, meta = synthetic.make_named("tiny")
    bad = dict(params)
    bad["X"] = params["X"].copy()
    bad["X"][1, 7, 0] = np.nan
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"]) as e:
        e.set_data(Y, c)
        e.set_params(bad)
        e.elbo_async()
        with pytest.raises(np.linalg.LinAlgError, match="chain 1"):
            e.sync()
        e.set_params(params)
        e.elbo_async()
        e.sync()
    sh = ShardedElbo(bad, Y, c, meta, rank=0, world=1, device=0, always_reduce=True)
    try:
        with pytest.raises(np.linalg.LinAlgError):
            sh.step()
    finally:
        sh.close()


def test_not_positive_definite_is_reported():
    params, Y, c, meta = synthetic.make_named("tiny")
    p = dict(params)
    p["Z"] = params["Z"].copy()
    p["Z"][5] = p["Z"][4]                                # duplicate inducing point, no jitter => singular K_uu
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], jitter=0.0) as e:
        e.set_data(Y, c)
        with pytest.raises(np.linalg.LinAlgError, match="K_uu"):
            e.nll_terms(p)
        good = e.nll_terms(params)                       # the handle stays usable after a numerical error
        assert np.isfinite(good["nll"])


def test_usage_errors():
    params, Y, c, meta = synthetic.make_named("tiny")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"]) as e:
        with pytest.raises(ValueError, match="ffvd_set_data"):
            e.elbo_sums()
        e.set_data(Y, c)
        with pytest.raises(ValueError):
            e.set_params(dict(params, Z=params["Z"][:-1]))
        with pytest.raises(ValueError):
            e.set_data(Y[:-1], c)


def test_full_size_config2_properties():
    """BASELINE config 2 shape (T=4096, M=512, D=4) with S=4: two chains checked against the oracle
    (seconds on CPU), all chains through permutation equivariance and linearity of the partial sums."""
    params, Y, c, meta = synthetic.make_named("c2", S=4)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], 4) as e:
        e.set_data(Y, c)
        t = e.nll_terms(params)
        per = t["nll_per_chain"].copy()
        perm = [2, 0, 3, 1]
        t2 = e.nll_terms(dict(params, X=params["X"][perm]))
        np.testing.assert_allclose(t2["nll_per_chain"], per[perm], rtol=1e-13)
        assert t2["nll"] == pytest.approx(t["nll"], rel=1e-13)
    assert t["nll"] == pytest.approx(per.mean(), rel=1e-13)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], 4, route="gram") as e:
        e.set_data(Y, c)
        tg = e.nll_terms(params)
    np.testing.assert_allclose(tg["nll_per_chain"], per, rtol=1e-8)
    for s in (0, 3):
        p = dict(params)
        p["X"] = params["X"][s]
        ref = orc.nll_terms(p, Y, c, U_collapse=True)
        assert per[s] == pytest.approx(ref["nll"], rel=1e-8)


def test_full_batch_schedule_config2(monkeypatch):
    """The headline launch shape itself (T=4096, M=512, D=4, S=32: 128 units in one unsplit pass).  At this size the
    Gram kernel keeps its raw tiles and the trace partials tr(K^-1 K_uf K_fu) are formed later on the side stream
    (DESIGN.md section 5); smaller tests never take that schedule.  Checked without the CPU oracle: every chain's nll
    must equal the one a 4-chain engine (split-K schedule, verified against the oracle in the test above) computes for
    it, the reference route must agree, and so must the same engine with the trace back in the Gram epilogue."""
    params, Y, c, meta = synthetic.make_named("c2")
    S = meta["S"]
    assert S == 32

    def chains(route, sel=None):
        X = params["X"] if sel is None else params["X"][sel]
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], X.shape[0], route=route) as e:
            e.set_data(Y, c)
            t = e.nll_terms(dict(params, X=X))
            t_again = e.nll_terms(dict(params, X=X))
        np.testing.assert_array_equal(t["nll_per_chain"], t_again["nll_per_chain"])     # side-stream work is ordered
        return t

    full = chains("gram")
    assert full["nll"] == pytest.approx(full["nll_per_chain"].mean(), rel=1e-13)
    for sel in ([0, 1, 2, 3], [28, 29, 30, 31]):
        part = chains("gram", sel)
        np.testing.assert_allclose(full["nll_per_chain"][sel], part["nll_per_chain"], rtol=1e-8)
    ref = chains("reference")
    np.testing.assert_allclose(full["nll_per_chain"], ref["nll_per_chain"], rtol=1e-8)
    for name in ("nll_log_likelihood", "x_t_prior_Q", "later_term1", "later_term2", "nll_part_prior"):
        assert full[name] == pytest.approx(ref[name], rel=1e-7), name
    monkeypatch.setenv("FFVD_NO_DEFER_TRACE", "1")
    plain = chains("gram")
    np.testing.assert_allclose(full["nll_per_chain"], plain["nll_per_chain"], rtol=1e-9)


def test_full_batch_gradient_config2(monkeypatch):
    """Backward pass at the headline shape (the schedule of test_full_batch_schedule_config2 with the L_A^-T rows and the
    side-stream K_uu chain of the backward pass).  (1) The same 32-chain gradient with every side-stream schedule switched
    off must be BITWISE equal wherever the arithmetic is the same (everything but log_Q, which reads the trace partials):
    a missing event wait shows up here.  (2) The sum of two 16-chain shards, which run the split-K schedule, must
    reproduce it to the accuracy cond(K_uu) = 1.2e7 allows."""
    params, Y, c, meta = synthetic.make_named("c2")
    S = meta["S"]

    def grads(s0, s1):
        p = dict(params, X=params["X"][s0:s1])
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], s1 - s0, route="gram", grad=True) as e:
            e.set_data(Y, c)
            t, g = e.nll_and_grad(p, S_total=S)
            t2, g2 = e.nll_and_grad(p, S_total=S)
        for k in GRAD_KEYS:
            np.testing.assert_array_equal(g[k], g2[k])
        return t, g

    tw, whole = grads(0, S)
    assert all(np.all(np.isfinite(whole[k])) for k in GRAD_KEYS)
    ta, a = grads(0, S // 2)
    tb, b = grads(S // 2, S)
    assert tw["nll"] == pytest.approx(0.5 * (ta["nll"] + tb["nll"]), rel=1e-9)
    # the two schedules sum the Gram matrices in different orders: with the backward pass in whitened variables that
    # moves dZ / dlengthscales / dvariance by less than 1e-5 of the largest entry at cond(K_uu) = 1.2e7 (it was 2e-3 when
    # K^-1 - A^-1 was formed from two explicit inverses), dX by 1e-7
    np.testing.assert_allclose(np.concatenate((a["X"], b["X"])), whole["X"], rtol=0, atol=1e-6 * np.max(np.abs(whole["X"])))
    for k in GRAD_KEYS[1:]:
        tol = 1e-5 if k in ("Z", "loglengthscales", "logvariance") else 1e-7
        np.testing.assert_allclose(a[k] + b[k], whole[k], rtol=0, atol=tol * np.max(np.abs(whole[k])), err_msg=k)
    monkeypatch.setenv("FFVD_NO_DEFER_TRACE", "1")
    monkeypatch.setenv("FFVD_GRAD_SERIAL", "1")
    monkeypatch.setenv("FFVD_NO_KFU_FIRST", "1")
    ts, serial = grads(0, S)
    for k in GRAD_KEYS:
        if k == "log_Q":
            np.testing.assert_allclose(serial[k], whole[k], rtol=1e-7)
        else:
            np.testing.assert_array_equal(serial[k], whole[k], err_msg=k)
    assert ts["nll"] == pytest.approx(tw["nll"], rel=1e-9)


def test_training_forward_variants_agree(monkeypatch):
    """The Gram-route training forward factorises A with L^T in the extension rows (L_H^-T = L^T L_A^-T comes out, DESIGN.md
    section 7) and the dataflow kernel reads those rows straight from L.  Two longer ways to the same numbers stay selectable and
    must agree: the rows written to memory first (what the launch-per-column Cholesky variants need), and H = W^T A W formed by
    two products and factorised with identity rows (rounds 1-2).  Odd M (padding), two passes' worth of shapes."""
    for name, kw in (("small", {}), ("ragged", {}), ("c2", dict(S=2, T=512, M=200))):
        params, Y, c, meta = synthetic.make_named(name, **kw)

        def grads():
            with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
                e.set_data(Y, c)
                return e.nll_and_grad(params)

        t0, g0 = grads()
        monkeypatch.setenv("FFVD_GRAD_LT_ARMED", "1")
        t1, g1 = grads()
        monkeypatch.delenv("FFVD_GRAD_LT_ARMED")
        monkeypatch.setenv("FFVD_GRAD_WHITEN_PRODUCTS", "1")
        t2, g2 = grads()
        monkeypatch.delenv("FFVD_GRAD_WHITEN_PRODUCTS")
        assert t1["nll"] == t0["nll"]
        assert t2["nll"] == pytest.approx(t0["nll"], rel=1e-9, abs=1e-10)
        for k in GRAD_KEYS:
            np.testing.assert_array_equal(g1[k], g0[k], err_msg=name + " " + k)         # same arithmetic, rows via memory
            scale = np.max(np.abs(g0[k])) + 1e-300
            tol = 1e-6 if k in ("Z", "loglengthscales", "logvariance") else 1e-8
            np.testing.assert_allclose(g2[k], g0[k], rtol=0, atol=tol * scale, err_msg=name + " " + k)


def test_linear_kernel_through_its_rank(monkeypatch):
    """Explicit-U branch with LinearK, forward only: K_fu = sigma^2 X Z^T has rank P, so fmean and sum_j F^2 are P x P forms per row
    of C = Z^T L^-T, which the K_uu chain leaves in ONE extension block.  Against the oracle, against
    the M-wide projection (FFVD_NO_LINEAR_LOWRANK=1), with several chains in several passes, ragged M and no control inputs."""
    for name, kw, cpp in (("small_lin", dict(S=3), 1), ("small_lin", dict(S=2, M=45, T=200), 0), ("small_lin", dict(C=0, S=1), 0)):
        params, Y, c, meta = synthetic.make_named(name, **kw)
        ref = orc.nll_terms_chains(params, Y, c, U_collapse=False, kernel_type="LinearK")
        got = run_engine(params, Y, c, meta, collapse=False, chains_per_pass=cpp)
        assert_terms(got, ref, TERMS_A)
        monkeypatch.setenv("FFVD_NO_LINEAR_LOWRANK", "1")
        wide = run_engine(params, Y, c, meta, collapse=False, chains_per_pass=cpp)
        monkeypatch.delenv("FFVD_NO_LINEAR_LOWRANK")
        assert_terms(wide, ref, TERMS_A)
        for n in TERMS_A:
            assert got[n] == pytest.approx(wide[n], rel=1e-10, abs=1e-11), n


_DEFER_SCRIPT = r"""
import sys
import numpy as np
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2", S=57, T=512, M=256)
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
    e.set_data(Y, c)
    t, g = e.nll_and_grad(params)
np.savez(sys.argv[1], nll=t["nll"], **g)
"""


def test_identity_structured_rows_behind_all_main_rows(tmp_path):
    """228 matrices x (4 main + 4 identity-structured + 1 vector) block rows: the dataflow Cholesky runs them in two groups (120 and
    108 matrices, the second one ragged) and dispatches the identity-structured rows of both groups behind the main rows of both
    (DfArgs::defer_ext).  Which workgroup runs when must not change a bit: the same training forward + backward with the
    round-2 block order (FFVD_DF_DEFER=0, read once per process, hence the child process) gives identical numbers."""
    import subprocess
    import sys
    outs = []
    for mode in ("1", "0"):
        path = str(tmp_path / ("g%s.npz" % mode))
        env = dict(os.environ, FFVD_DF_DEFER=mode, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, "-c", _DEFER_SCRIPT, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    assert np.isfinite(float(outs[0]["nll"])) and float(outs[0]["nll"]) == float(outs[1]["nll"])
    for k in GRAD_KEYS:
        assert np.all(np.isfinite(outs[0][k])), k
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)


@pytest.mark.parametrize("branch", ["B", "A"])
def test_no_control_inputs(branch):
    """C = 0: the reference concatenates control inputs only when they exist (dgp_model.py:268-271, base_model.py:243-246);
    P = D, X_combine = X[:-1].  Compared with the oracle on the same seeded inputs (no golden file for this shape)."""
    params, Y, c, meta = synthetic.make_named("tiny", C=0)
    assert c.shape == (meta["T"], 0) and params["Z"].shape[1] == meta["D"]
    collapse = branch == "B"
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=collapse, kernel_type=meta["kernel_type"])
    got = run_engine(params, Y, c, meta, collapse=collapse)
    assert_terms(got, ref, TERMS_B if collapse else TERMS_A)
    if collapse:
        g = run_engine(params, Y, c, meta, collapse=True, route="gram")
        assert g["nll"] == pytest.approx(ref["nll"], rel=1e-8)
        from oracle import ffvd_grad_oracle as gorc
        with ElboEngine(meta["T"], meta["D"], 0, meta["M"], meta["S"], route="gram", grad=True) as e:
            e.set_data(Y, c)
            _, grads = e.nll_and_grad(params)
        S = meta["S"]
        want = np.zeros_like(grads["Z"])
        for s in range(S):
            p = dict(params)
            p["X"] = params["X"][s]
            want += gorc.nll_grad(p, Y, c)["Z"] / S
        # cond(K_uu) is 1e6 for these 2-D inducing inputs; GPU and closed form, both in whitened variables, agree to 1e-7
        # (1e-4 was needed when both formed K^-1 - A^-1 from explicit inverses)
        np.testing.assert_allclose(grads["Z"], want, rtol=0, atol=1e-6 * np.max(np.abs(want)))


@pytest.mark.parametrize("ov", [dict(T=170, M=150, S=2, D=3, C=1),      # Mp = 192, Tp = 192: half-empty 128-tiles
                                dict(T=130, M=40, S=2, D=4, C=2),       # P = 6: last shape of the fused backward epilogue
                                dict(T=130, M=40, S=2, D=5, C=2)])      # P = 7: materialised-E fallback
def test_gradient_on_awkward_shapes(ov):
    """Backward pass against the closed-form oracle where tiles are partial and on both sides of the P <= 6 switch
    between the fused and the two-kernel E reduction."""
    from oracle import ffvd_grad_oracle as gorc
    params, Y, c, meta = synthetic.make_named("tiny", **ov)
    S = meta["S"]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram", grad=True) as e:
        e.set_data(Y, c)
        _, g = e.nll_and_grad(params)
    ref = None
    for s in range(S):
        p = dict(params)
        p["X"] = params["X"][s]
        a = gorc.nll_grad(p, Y, c)
        if ref is None:
            ref = {k: (np.zeros((S,) + v.shape) if k == "X" else np.zeros_like(v)) for k, v in a.items()}
        ref["X"][s] = a["X"] / S
        for k in a:
            if k != "X":
                ref[k] += a[k] / S
    for k in GRAD_KEYS:
        err = np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)
        # whitened backward on both sides: 150 points in 4-D give 1e-8 on dZ between GPU and CPU (7e-5 with the explicit
        # inverses)
        tol = 1e-6 if k == "Z" else (1e-7 if k in ("loglengthscales", "logvariance") else 1e-8)
        assert err < tol, (k, err)


@pytest.mark.parametrize("route", ["reference", "gram"])
def test_non_finite_inputs_are_reported_not_propagated(route):
    """A NaN in a latent trajectory must surface as a numerical error (first non-positive pivot), never as a silent
    NaN nll or a hang; the handle stays usable."""
    params, Y, c, meta = synthetic.make_named("tiny")
    bad = dict(params)
    bad["X"] = params["X"].copy()
    bad["X"][1, 7, 0] = np.nan
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route=route) as e:
        e.set_data(Y, c)
        with pytest.raises(np.linalg.LinAlgError, match="chain 1"):
            e.nll_terms(bad)
        ok = e.nll_terms(params)
        assert np.isfinite(ok["nll"])


@pytest.mark.parametrize("name,ov,nshard", [("small", dict(S=1, D=2), 3), ("ragged", dict(S=2, D=1), 4),
                                             ("small_lin", dict(S=1, D=2, U_collapse=True), 2)])
def test_time_shards_sum_to_the_single_engine_nll(name, ov, nshard, monkeypatch):
    """SURVEY 8e fallback (S * D < ranks): T-shard engines evaluate disjoint row ranges; their exchange buffers (raw Gram
    tiles K_uf K_fu, delta^T K_fu rows, chain sums) are added as the all-reduce would, every shard finishes on the sum
    and must reproduce the unsharded Gram-route nll and the oracle's.  Ragged shard sizes (T not divisible)."""
    from ffvd_amd.distributed import shard_range
    params, Y, c, meta = synthetic.make_named(name, **ov)
    T, S = meta["T"], meta["S"]
    monkeypatch.setenv("FFVD_NO_TINY", "1")           # the unsharded GRAM-route engine (the one-launch path computes in the reference's op order)
    whole = run_engine(params, Y, c, meta, collapse=True, route="gram")
    monkeypatch.delenv("FFVD_NO_TINY")
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True, kernel_type=meta["kernel_type"])
    engines, bufs = [], []
    try:
        for r in range(nshard):
            t0, tc = shard_range(T, nshard, r)
            e = ElboEngine(tc, meta["D"], meta["C"], meta["M"], S, kernel_type=meta["kernel_type"], route="gram",
                           t_shard=(t0, T))
            e.set_data(Y[t0: t0 + tc], c[t0: t0 + tc])
            e.set_params(dict(params, X=np.ascontiguousarray(params["X"][:, t0: t0 + tc + 1])))
            engines.append(e)
            bufs.append(e.tshard_local())
        total = np.sum(bufs, axis=0)
        for e in engines:
            sums = e.tshard_finish(total)
            assert sums[7] == S
            # (not the same arithmetic: the plain handle's K_uu chain is one dataflow launch that forms K^-1 itself, the T-shard handle
            #  multiplies L^-T L^-1 in a launch of its own -- eps * cond(K_uu) on the trace term, 1.5e-10 measured)
            assert sums[6] / S == pytest.approx(whole["nll"], rel=1e-9)
            assert sums[6] / S == pytest.approx(ref["nll"], rel=1e-7)
            for i, n in enumerate(TERMS_B[:-1]):
                assert sums[i] / S == pytest.approx(whole[n], rel=1e-9, abs=1e-9), n      # (the trace term is the remainder of a cancellation)
    finally:
        for e in engines:
            e.close()


def test_time_shard_through_native_rccl_one_rank():
    """ffvd_elbo_tshard end to end on one rank: local rows -> ncclAllReduce of the exchange buffer on the handle's
    stream -> finish (world 1: the collective is the identity, everything else is the real path)."""
    from ffvd_amd.distributed import ShardedElbo, finish
    params, Y, c, meta = synthetic.make_named("small", S=1, D=2)
    whole = run_engine(params, Y, c, meta, collapse=True, route="gram")
    sh = ShardedElbo(params, Y, c, meta, rank=0, world=1, mode="time", device=0)
    try:
        t = finish(sh.step())
        assert finish(sh.step()) == t
    finally:
        sh.close()
    # not the same arithmetic since round 3: at this size the plain handle's K_uu chain is one dataflow launch that forms K^-1
    # itself, the T-shard handle multiplies L^-T L^-1 in a launch of its own -- eps * cond(K_uu) on the trace term
    assert t["nll"] == pytest.approx(whole["nll"], rel=1e-9)
    with pytest.raises(ValueError):
        ElboEngine(64, 2, 1, 16, 1, route="reference", t_shard=(0, 128))      # the Gram form is what makes T additive
    with pytest.raises(ValueError):
        ElboEngine(64, 2, 1, 16, 1, route="gram", t_shard=(100, 128))         # shard outside the job


GRAD_NAMES = ("Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")


@pytest.mark.parametrize("name,ov,nshard", [("small", dict(S=1, D=2), 3), ("ragged", dict(S=2, D=1), 4),
                                             ("small_lin", dict(S=1, D=2, U_collapse=True), 2), ("small", dict(S=3, D=3), 2)])
def test_time_shards_sum_to_the_single_engine_gradient(name, ov, nshard, monkeypatch):
    """The backward pass of a T-sharded job: after the exchange of the raw tiles every shard differentiates
    its own rows against the job's factorisation; the gradient blocks are added as the second all-reduce would, and the sum must
    be the unsharded Gram-route gradient -- to 1e-7 of each array's largest entry (measured 2e-8 at worst: the tiles are summed in
    another order than the unsharded pass, eps * cond(A) in the factor, and the gradient sees it amplified once more; the nll itself
    agrees to 1e-10), dX row by row with the rows two shards share taking a part from each."""
    from ffvd_amd.distributed import shard_range
    params, Y, c, meta = synthetic.make_named(name, **ov)
    T, S = meta["T"], meta["S"]
    monkeypatch.setenv("FFVD_NO_TINY", "1")           # the unsharded GRAM-route engine
    with ElboEngine(T, meta["D"], meta["C"], meta["M"], S, kernel_type=meta["kernel_type"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        whole, gw = e.nll_and_grad(params)
    monkeypatch.delenv("FFVD_NO_TINY")
    engines, bufs = [], []
    try:
        for r in range(nshard):
            t0, tc = shard_range(T, nshard, r)
            e = ElboEngine(tc, meta["D"], meta["C"], meta["M"], S, kernel_type=meta["kernel_type"], route="gram",
                           t_shard=(t0, T), grad=True)
            e.set_data(Y[t0: t0 + tc], c[t0: t0 + tc])
            e.set_params(dict(params, X=np.ascontiguousarray(params["X"][:, t0: t0 + tc + 1])))
            engines.append(e)
            bufs.append(e.tshard_local())
        total = np.sum(bufs, axis=0)
        blocks = [e.tshard_finish_grad(total) for e in engines]
        assert [bool(np.any(b[:8])) for b in blocks] == [True] + [False] * (nshard - 1)      # the terms travel once
        block = np.sum(blocks, axis=0)
        dX = np.zeros_like(params["X"])
        for r, e in enumerate(engines):
            t0, tc = shard_range(T, nshard, r)
            sums, g = e.tshard_grad_fetch(block)
            assert sums[7] == S and sums[6] / S == pytest.approx(whole["nll"], rel=1e-9)
            for n in GRAD_NAMES:
                scale = max(np.abs(gw[n]).max(), 1e-12)
                assert np.abs(g[n] - gw[n]).max() <= 1e-7 * scale, (n, r)
            dX[:, t0: t0 + tc + 1] += g["X"]
        assert np.abs(dX - gw["X"]).max() <= 1e-7 * np.abs(gw["X"]).max()
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("name,ov,nshard", [("small", dict(S=2, D=2), 3), ("ragged", dict(S=1, D=2), 2)])
def test_time_shards_adam_trajectory(name, ov, nshard, monkeypatch):
    """T-shard handles train (dgp_model.py:303-305 trains every variable).  `nshard` handles on the one GPU, the
    exchanges carried by a fake collective (sums in Python), the product's own assembly per shard (`distributed.tshard_adam_step`:
    tiles, gradient block, boundary rows of dX, ffvd_tshard_adam_apply).  After every step: (i) each shard's parameters equal a
    host-side Adam (TensorFlow semantics, oracle/ffvd_optim_oracle.py) applied to the job's gradient to 1e-13; (ii) the shards' copies
    of the shared parameters are bit-identical, and so are the two copies of every row neighbouring shards share; (iii) the first
    step's nll is the unsharded engine's (1e-9) and four steps lower it."""
    from ffvd_amd import distributed as dm
    from oracle import ffvd_optim_oracle as oo
    params, Y, c, meta = synthetic.make_named(name, **ov)
    T, S, D = meta["T"], meta["S"], meta["D"]
    lr = 0.003
    monkeypatch.setenv("FFVD_NO_TINY", "1")
    with ElboEngine(T, D, meta["C"], meta["M"], S, route="gram", grad=True) as e:
        e.set_data(Y, c)
        whole = e.nll_terms(params)["nll"]
    monkeypatch.delenv("FFVD_NO_TINY")
    engines, ranges = [], []
    try:
        for r in range(nshard):
            t0, tc = dm.shard_range(T, nshard, r)
            e = ElboEngine(tc, D, meta["C"], meta["M"], S, route="gram", t_shard=(t0, T), grad=True)
            e.set_data(Y[t0: t0 + tc], c[t0: t0 + tc])
            e.set_params(dict(params, X=np.ascontiguousarray(params["X"][:, t0: t0 + tc + 1])))
            engines.append(e)
            ranges.append((t0, tc))
        host = {k: np.array(params[k], dtype=np.float64) for k in list(GRAD_NAMES) + ["X"]}
        hm = {k: np.zeros_like(v) for k, v in host.items()}
        hv = {k: np.zeros_like(v) for k, v in host.items()}
        nlls = []
        for step in range(1, 5):
            # a collective all shards take part in, faked: phase 1 collects every shard's contribution, phase 2 hands out the sum
            class Fake:
                def __init__(self):
                    self.parts, self.total = [], None
                def collect(self, a):
                    self.parts.append(np.array(a, dtype=np.float64).ravel())
                    return np.zeros_like(self.parts[-1])
                def give(self, a):
                    return self.total
            f3 = Fake()
            locs = [e.tshard_local() for e in engines]
            total = np.sum(locs, axis=0)
            blocks = [e.tshard_finish_grad(total, S_total=S) for e in engines]
            block = np.sum(blocks, axis=0)
            fetched = [e.tshard_grad_fetch(block) for e in engines]
            for r, (sums, g) in enumerate(fetched):
                dm.tshard_boundary_rows(g["X"], r, nshard, f3.collect)
            f3.total = np.sum(f3.parts, axis=0) if nshard > 1 else None
            # the job's gradient as the shards hold it: shared arrays from the block, dX assembled from the completed rows
            gX = np.zeros_like(host["X"])
            outs = []
            for r, (e, (sums, g)) in enumerate(zip(engines, fetched)):
                rows = dm.tshard_boundary_rows(g["X"], r, nshard, f3.give)
                t0, tc = ranges[r]
                gX[:, t0: t0 + tc + 1] = rows
                outs.append(e.tshard_adam_apply(rows, lr))
            nlls.append(dm.finish(outs[0])["nll"])
            assert all(np.array_equal(o, outs[0]) for o in outs)
            shared = fetched[0][1]
            for k in GRAD_NAMES:
                host[k], hm[k], hv[k] = oo.adam_step(host[k], shared[k], hm[k], hv[k], step, lr)
            host["X"], hm["X"], hv["X"] = oo.adam_step(host["X"], gX, hm["X"], hv["X"], step, lr)
            got = [e.get_params() for e in engines]
            for r, gp in enumerate(got):
                t0, tc = ranges[r]
                for k in GRAD_NAMES:
                    np.testing.assert_array_equal(gp[k], got[0][k], err_msg=f"step {step} shard {r} {k}: copies diverged")
                    np.testing.assert_allclose(gp[k], host[k], rtol=0, atol=1e-13 * max(1.0, float(np.max(np.abs(host[k])))), err_msg=k)
                np.testing.assert_allclose(gp["X"], host["X"][:, t0: t0 + tc + 1], rtol=0, atol=1e-13)
                if r > 0:
                    np.testing.assert_array_equal(gp["X"][:, 0], got[r - 1]["X"][:, -1], err_msg=f"step {step}: boundary row {r}")
        assert nlls[0] == pytest.approx(whole, rel=1e-9)
        assert nlls[-1] < nlls[0]
        # ---- then two SG-HMC steps (burn_in_op, sample_op: base_model.py:143-179) on the kernel hyper-parameters: the same noise on
        # every shard, X_N of the step size = the JOB's rows (T + 1), the shards' copies bit-identical, equal to the host-side oracle
        rng = np.random.default_rng(9)
        keys = ("logvariance", "loglengthscales")
        st = {k: [np.ones_like(host[k]), np.ones_like(host[k]), np.ones_like(host[k]), np.zeros_like(host[k])] for k in keys}
        for burn in (True, False):
            noise = {k: rng.standard_normal(host[k].shape) for k in keys}
            locs = [e.tshard_local() for e in engines]
            total = np.sum(locs, axis=0)
            block = np.sum([e.tshard_finish_grad(total, S_total=S) for e in engines], axis=0)
            fetched = [e.tshard_grad_fetch(block) for e in engines]
            outs = [e.tshard_sghmc_apply(noise, 0.01, 0.05, burn) for e in engines]
            assert all(np.array_equal(o, outs[0]) for o in outs)
            for k in keys:
                th, xi, g1, g2, pm = oo.sghmc_step(host[k], fetched[0][1][k], st[k][0], st[k][1], st[k][2], st[k][3], noise[k], 0.01, 0.05, T + 1, burn)
                host[k], st[k] = th, [xi, g1, g2, pm]
            got = [e.get_params() for e in engines]
            for gp in got:
                for k in keys:
                    np.testing.assert_array_equal(gp[k], got[0][k], err_msg=f"sghmc {k}: copies diverged")
                    np.testing.assert_allclose(gp[k], host[k], rtol=0, atol=1e-13 * max(1.0, float(np.max(np.abs(host[k])))), err_msg=k)
    finally:
        for e in engines:
            e.close()


def test_time_shard_gradient_through_native_rccl_one_rank(monkeypatch):
    """ffvd_elbo_tshard_grad end to end on one rank (both collectives are the identity, everything else is the real path), and
    what a T-shard handle refuses."""
    from ffvd_amd.distributed import ShardedElbo
    params, Y, c, meta = synthetic.make_named("small", S=2, D=2)
    monkeypatch.setenv("FFVD_NO_TINY", "1")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        whole, gw = e.nll_and_grad(params)
    monkeypatch.delenv("FFVD_NO_TINY")
    sh = ShardedElbo(params, Y, c, meta, rank=0, world=1, mode="time", device=0, grad=True)
    try:
        t, g = sh.nll_and_grad()
        t2, g2 = sh.nll_and_grad()
        assert t == t2 and all(np.array_equal(g[k], g2[k]) for k in g)
        # (round 5: T-shard handles train -- ffvd_tshard_adam_apply; on one rank the boundary exchange is empty)
        before = sh.engine.get_params()
        t3 = sh.adam_step(1e-3)
        after = sh.engine.get_params()
        assert t3["nll"] == t["nll"]
        for k in ("X", "Z", "log_Q", "loglengthscales"):
            assert np.max(np.abs(after[k] - before[k])) > 0 and np.max(np.abs(after[k] - before[k])) <= 1e-3 * (1 + 1e-6), k
        assert sh.nll_terms()["nll"] < t["nll"]
    finally:
        sh.close()
    assert t["nll"] == pytest.approx(whole["nll"], rel=1e-9)
    for n in GRAD_NAMES + ("X",):
        assert np.abs(g[n] - gw[n]).max() <= 1e-7 * max(np.abs(gw[n]).max(), 1e-12), n
    with pytest.raises(ValueError, match="every latent dim"):
        ElboEngine(64, 2, 1, 16, 1, route="gram", t_shard=(0, 128), grad=True, d_begin=1, d_count=1)


SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FFVD_ROOT"])
import numpy as np, torch, torch.distributed as dist
from ffvd_amd import synthetic
from ffvd_amd.distributed import ShardedElbo, finish, all_reduce_grads
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1]
dist.init_process_group("gloo", rank=rank, world_size=world)       # two processes share the one GPU: gloo moves the CUDA tensor
if mode == "time":                                                  # S * D = 1 < world: shard the transitions
    params, Y, c, meta = synthetic.make_named("small", S=1, D=1)
    sh = ShardedElbo(params, Y, c, meta, rank=rank, world=world, mode="auto", device=0, collective="torch")
    assert sh.time_shard
    t = finish(sh.step())
    t2 = finish(sh.step())
    shg = ShardedElbo(params, Y, c, meta, rank=rank, world=world, mode="auto", device=0, collective="torch", grad=True)
    tg, g = shg.nll_and_grad()
    np.save(os.path.join(os.environ["FFVD_OUT"], "tgrad%d.npy" % rank), np.concatenate([g[k].ravel() for k in sorted(g) if k != "U"]))
    print("RESULT", rank, repr(t["nll"]), repr(t2["nll"]), repr(tg["nll"]), flush=True)
    dist.destroy_process_group()
    sys.exit(0)
params, Y, c, meta = synthetic.make_named("small")
sh = ShardedElbo(params, Y, c, meta, rank=rank, world=world, mode=mode, device=0, route="gram", grad=True, collective="torch")
t = finish(sh.step())
t2 = finish(sh.step())
tg, g = sh.nll_and_grad()
# explicit-U branch through the same collective path (dU is all-reduced with the other shared gradients)
meta_a = dict(meta, U_collapse=False)
sha = ShardedElbo(params, Y, c, meta_a, rank=rank, world=world, mode=mode, device=0, grad=True, collective="torch")
ta, ga = sha.nll_and_grad()
print("RESULT", rank, repr(t["nll"]), repr(t2["nll"]), repr(tg["nll"]), repr(float(np.abs(g["Z"]).sum())),
      repr(ta["nll"]), repr(float(np.abs(ga["U"]).sum())), flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ["chains", "dims"])
def test_two_processes_share_the_gpu_and_reduce(tmp_path, mode):
    """SURVEY 8(e) end to end with real kernels: two ranks (chain shards / latent-dim shards) on the one GPU of the test
    box, partial sums and gradients all-reduced (gloo carries the CUDA tensors here; RCCL needs one GPU per rank), and
    every rank ends up with the single-process nll and gradient."""
    import os
    import subprocess
    import sys
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFVD_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541" if mode == "chains" else "29542",
               WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), mode], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    params, Y, c, meta = synthetic.make_named("small")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        t, g = e.nll_and_grad(params)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], U_collapse=False, grad=True) as e:
        e.set_data(Y, c)
        ta, ga = e.nll_and_grad(params)
    for out in outs:
        line = [l for l in out.splitlines() if l.startswith("RESULT")][0].split()
        nll1, nll2, nllg, zsum, nlla, usum = (float(x) for x in line[2:8])
        assert nll1 == nll2
        assert nll1 == pytest.approx(t["nll"], rel=1e-12) and nllg == pytest.approx(t["nll"], rel=1e-12)
        assert zsum == pytest.approx(float(np.abs(g["Z"]).sum()), rel=1e-4)
        assert nlla == pytest.approx(ta["nll"], rel=1e-12)
        assert usum == pytest.approx(float(np.abs(ga["U"]).sum()), rel=1e-9)


def test_two_processes_time_shard(tmp_path, monkeypatch):
    """The T-shard fallback with two real ranks on the one GPU of the test box (S * D = 1): the exchange buffer travels
    over gloo, both ranks end up with the single-process nll."""
    import os
    import subprocess
    import sys
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFVD_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", WORLD_SIZE="2", FFVD_OUT=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, str(script), "time"], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    params, Y, c, meta = synthetic.make_named("small", S=1, D=1)
    whole = run_engine(params, Y, c, meta, collapse=True, route="gram")
    for out in outs:
        line = [l for l in out.splitlines() if l.startswith("RESULT")][0].split()
        assert float(line[2]) == float(line[3])
        # two T-ranges summed, then factorised: a different summation order than the unsharded Gram pass (eps * cond(K_uu))
        assert float(line[2]) == pytest.approx(whole["nll"], rel=1e-9)
        assert float(line[4]) == pytest.approx(whole["nll"], rel=1e-9)
    # ... and with the job's gradient (second exchange: the gradient block; third: dX rows at their global position)
    monkeypatch.setenv("FFVD_NO_TINY", "1")           # the unsharded GRAM-route engine
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        _, gw = e.nll_and_grad(params)
    gw.pop("U", None)
    want = np.concatenate([gw[k].ravel() for k in sorted(gw)])
    for r in range(2):
        got = np.load(tmp_path / ("tgrad%d.npy" % r))
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-7 * np.abs(want).max()


@pytest.mark.parametrize("ov,env", [(dict(M=256, T=320, S=3, D=2), {"FFVD_GSPLIT": "1"}),       # one off-diagonal tile + one pair combo per unit
                                    (dict(M=768, T=832, S=1, D=2), {"FFVD_GSPLIT": "1"}),       # 15 + 3
                                    (dict(M=512, T=576, S=34, D=4), {}),                        # 1088 workgroups: 64 of them cut in row halves
                                    (dict(M=512, T=4096, S=1, D=4), {}),                        # 4 units in 8 row ranges: a unit's ranges on two XCDs
                                    (dict(M=512, T=4096, S=1, D=2), {})])                       # 2 units: on four XCDs each
def test_gram_pair_combos_other_shapes(ov, env, monkeypatch):
    """The 16-granular diagonal workgroups of the Gram kernel (two diagonal 128-tiles per workgroup, nine MFMA tiles per wavefront,
    LDS-DMA staging) away from config 2: other panel counts, and a launch whose last workgroups -- pair combos among them -- run as
    two row halves that meet through memory.  Gram route against the reference op order (which has no such kernel) and the oracle."""
    params, Y, c, meta = synthetic.make_named("c2", **ov)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("FFVD_NO_TINY", "1")
    S = meta["S"]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram") as e:
        e.set_data(Y, c)
        tg = e.nll_terms(params)
        assert e.nll_terms(params)["nll"] == tg["nll"]
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S) as e:
        e.set_data(Y, c)
        tr = e.nll_terms(params)
    # (T = 576 leaves the nll of a chain at 0.04-0.11, the remainder of terms of order 1: the absolute floor is that of the terms.
    #  The Gram route's error is eps * cond(K_uu) of those terms; measured against the oracle with two forms of the Cholesky's diagonal
    #  factor, tools/factor_acc.py: M = 512: 1e-11 / 4e-10, M = 768: 5e-10 / 5e-10 (4e-9 unsplit), M = 1024: 3e-9 / 5e-9 -- while the
    #  reference op order on the same kernels agrees with the oracle to 1e-14)
    # The tolerance is predicted, not fitted: 4 eps cond(K_uu + jitter I) of terms of order 1 (cond = 0.7e7 / 1.3e7 / 2.5e7 at M = 256 /
    # 512 / 768: 6e-9 / 1.2e-8 / 2.2e-8).
    tol = gram_route_tolerance(params, meta)
    assert 2e-9 < tol < 4e-8
    np.testing.assert_allclose(tg["nll_per_chain"], tr["nll_per_chain"], rtol=0, atol=tol)
    p0 = dict(params, X=params["X"][S - 1])
    ref = orc.nll_terms(p0, Y, c, U_collapse=True)
    assert abs(tg["nll_per_chain"][S - 1] - ref["nll"]) <= tol
    if S >= 34:          # training forward + backward on the same launch shape: the Gram kernel also writes the symmetric copy of A
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram", grad=True) as e:
            e.set_data(Y, c)
            t1, g1 = e.nll_and_grad(params)
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, grad=True) as e:
            e.set_data(Y, c)
            t2, g2 = e.nll_and_grad(params)
        assert t1["nll"] == pytest.approx(t2["nll"], rel=1e-8, abs=2e-9)
        for n in ("Z", "loglengthscales", "log_Q"):
            assert np.abs(g1[n] - g2[n]).max() <= 1e-6 * np.abs(g2[n]).max(), n


def _random_shapes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        T = int(rng.integers(5, 200))
        M = int(rng.integers(3, min(T, 140) + 1))
        out.append(dict(T=T, M=M, D=int(rng.integers(1, 6)), C=int(rng.integers(0, 4)), S=int(rng.integers(1, 4))))
    return out


@pytest.mark.parametrize("ov", _random_shapes(10, 20230209), ids=lambda o: "T{T}_M{M}_D{D}_C{C}_S{S}".format(**o))
def test_random_shapes_against_oracle(ov):
    """Ragged sizes on every axis (T, M not multiples of the 64/128 tiles, one to five latent dims, zero to three
    control inputs): all three evaluation paths against the oracle on the same seeded inputs."""
    params, Y, c, meta = synthetic.make_named("tiny", **ov)
    refB = orc.nll_terms_chains(params, Y, c, U_collapse=True, kernel_type=meta["kernel_type"])
    refA = orc.nll_terms_chains(params, Y, c, U_collapse=False, kernel_type=meta["kernel_type"])
    gotB = run_engine(params, Y, c, meta, collapse=True)
    gotA = run_engine(params, Y, c, meta, collapse=False)
    gotG = run_engine(params, Y, c, meta, collapse=True, route="gram")
    assert_terms(gotB, refB, TERMS_B)
    assert_terms(gotA, refA, TERMS_A)
    assert gotG["nll"] == pytest.approx(refB["nll"], rel=1e-7)          # Gram route: eps * cond(K_uu)
    np.testing.assert_allclose(gotB["nll_per_chain"], refB["nll_per_chain"], rtol=RTOL)


@pytest.mark.parametrize("name", ["tiny", "ragged", "small"])
def test_explicit_u_gradient_matches_autograd(name):
    """SURVEY 8f-1, "(and U in branch A)": d nll / d (X, Z, U, kernel hypers, Q, C, d, R) of the explicit-U branch from
    the HIP backward pass (Cholesky adjoint in closed form) against torch autograd of the independent restatement."""
    from oracle import ffvd_oracle_torch as orct
    params, Y, c, meta = synthetic.make_named(name)
    S = params["X"].shape[0]
    keys = GRAD_KEYS + ("U",)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, U_collapse=False, grad=True) as e:
        e.set_data(Y, c)
        terms, g = e.nll_and_grad(params)
        _, g2 = e.nll_and_grad(params)
    ref = {k: np.zeros_like(g[k]) for k in keys}
    nll_ref = 0.0
    for s in range(S):
        p = dict(params)
        p["X"] = params["X"][s]
        t, ga = orct.nll_and_grad(p, Y, c, wrt=keys, U_collapse=False)
        nll_ref += t["nll"] / S
        ref["X"][s] = ga["X"] / S
        for k in keys[1:]:
            ref[k] += ga[k] / S
    assert terms["nll"] == pytest.approx(nll_ref, rel=1e-9)
    for k in keys:
        np.testing.assert_array_equal(g[k], g2[k])
        err = np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)
        assert err < 1e-8, (k, err)          # no K^-1 - A^-1 cancellation in this branch: everything is at 1e-10


@pytest.mark.parametrize("kernel_type,ov", [
    ("LinearK", dict()),                                         # small_lin: P = 7 -> E materialised, generic-P reductions
    ("LinearK", dict(T=301, M=77, D=3, C=2, S=2)),               # P = 5 -> fused epilogue without the Hadamard product
    ("LinearK", dict(T=200, M=40, D=16, C=1, S=1)),              # P = 17: BASELINE config 5's input dimension
    ("SquaredExponential", dict(T=160, M=48, D=6, C=2, S=2)),    # SE kernel with P = 8 > 6 in the explicit-U branch
], ids=["lin_P7", "lin_P5_fused", "lin_P17", "se_P8"])
def test_explicit_u_gradient_linear_kernel_and_large_p(kernel_type, ov):
    """the LinearK chain rule (kernels.py:270-281: K = (X s2) X2^T, Kdiag = s2 |x|^2 -- no Hadamard
    product with K, d/dx = s2 z, Kdiag feeds the trace term) in the explicit-U backward pass, and the generic-P (> 6)
    path of that branch for both kernels, against torch autograd of the independent restatement."""
    from oracle import ffvd_oracle_torch as orct
    params, Y, c, meta = synthetic.make_named("small_lin", kernel_type=kernel_type, **ov)
    if kernel_type == "SquaredExponential":
        params = dict(params, loglengthscales=np.log(2.0 + 0.1 * np.arange(meta["D"]))[:, None] * np.ones((1, meta["P"])))
    S = params["X"].shape[0]
    keys = GRAD_KEYS + ("U",) if kernel_type == "SquaredExponential" else tuple(k for k in GRAD_KEYS if k != "loglengthscales") + ("U",)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, kernel_type=kernel_type, U_collapse=False, grad=True) as e:
        e.set_data(Y, c)
        terms, g = e.nll_and_grad(params)
        _, g2 = e.nll_and_grad(params)
    ref = {k: np.zeros_like(g[k]) for k in keys}
    nll_ref = 0.0
    for s in range(S):
        p = dict(params)
        p["X"] = params["X"][s]
        t, ga = orct.nll_and_grad(p, Y, c, wrt=keys, U_collapse=False, kernel_type=kernel_type)
        nll_ref += t["nll"] / S
        ref["X"][s] = ga["X"] / S
        for k in keys[1:]:
            ref[k] += ga[k] / S
    assert terms["nll"] == pytest.approx(nll_ref, rel=1e-7)
    for k in keys:
        np.testing.assert_array_equal(g[k], g2[k])
        err = np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)
        # LinearK: K_uu has rank P << M, only the 1e-5 jitter makes it positive definite (cond ~ 1e5 |K| / 1e-5)
        assert err < (1e-5 if kernel_type == "LinearK" else 1e-8), (k, err)
    if kernel_type == "LinearK":
        assert not np.any(g["loglengthscales"])                   # no lengthscales: neither a data nor a prior term


@pytest.mark.parametrize("route", ["gram", "reference"])
@pytest.mark.parametrize("ov", [dict(U_collapse=True), dict(U_collapse=True, T=301, M=77, D=3, C=2, S=2),
                                dict(U_collapse=True, T=200, M=40, D=16, C=1, S=1)], ids=["lin_P7", "lin_P5_fused", "lin_P17"])
def test_collapsed_gradient_linear_kernel(route, ov):
    """Round 3 (VERDICT r2 Missing 3): the backward pass of the collapsed branch with LinearK (kernels.py:270-281) -- no Hadamard
    factor in the chain rule through K_fu and K_uu, Kdiag_t = s2 |x_t|^2 in the trace term (so X, logvariance and log_Q see it) --
    on both routes, fused (P <= 6) and generic-P reductions, against torch autograd of the independent restatement."""
    from oracle import ffvd_oracle_torch as orct
    params, Y, c, meta = synthetic.make_named("small_lin", **ov)
    S = params["X"].shape[0]
    keys = tuple(k for k in GRAD_KEYS if k != "loglengthscales")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, kernel_type="LinearK", U_collapse=True, route=route, grad=True) as e:
        e.set_data(Y, c)
        terms, g = e.nll_and_grad(params)
        _, g2 = e.nll_and_grad(params)
    ref = {k: np.zeros_like(g[k]) for k in keys}
    nll_ref = 0.0
    for s in range(S):
        p = dict(params)
        p["X"] = params["X"][s]
        t, ga = orct.nll_and_grad(p, Y, c, wrt=keys, U_collapse=True, kernel_type="LinearK")
        nll_ref += t["nll"] / S
        ref["X"][s] = ga["X"] / S
        for k in keys[1:]:
            ref[k] += ga[k] / S
    assert terms["nll"] == pytest.approx(nll_ref, rel=1e-6)
    for k in keys:
        np.testing.assert_array_equal(g[k], g2[k])
        err = np.max(np.abs(g[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)
        # K_uu has rank P << M, only the 1e-5 jitter makes it positive definite (cond ~ |K| / 1e-5): the explicit-U test's bound
        assert err < 1e-5, (route, k, err)
    assert not np.any(g["loglengthscales"])                       # no lengthscales: neither a data nor a prior term


def test_collapsed_linear_kernel_trains():
    """Device-resident Adam steps in the collapsed branch with LinearK (config 5's shape of inputs, collapsed U) lower the nll on
    both routes; the lengthscales (LinearK has none) stay where they are."""
    params, Y, c, meta = synthetic.make_named("small_lin", U_collapse=True, T=512, M=96, D=8, S=2)
    for route in ("gram", "reference"):
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], kernel_type="LinearK", U_collapse=True, route=route,
                        grad=True) as e:
            e.set_data(Y, c)
            e.set_params(params)
            first = e.adam_step(0.003)["nll"]
            for _ in range(8):
                last = e.adam_step(0.003)["nll"]
            got = e.get_params()
        assert np.isfinite(first) and last < first, (route, first, last)
        np.testing.assert_array_equal(got["loglengthscales"], params["loglengthscales"])
        assert np.max(np.abs(got["Z"] - params["Z"])) > 0


def test_config5_training_step_linear_kernel():
    """BASELINE configs[4] can now be trained: a few device-resident Adam steps on the LinearK / explicit-U workload at its
    full shape (T=4096, x_dim=16, M=512) lower the nll."""
    params, Y, c, meta = synthetic.make_named("c5")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], kernel_type="LinearK", U_collapse=False,
                    grad=True) as e:
        e.set_data(Y, c)
        e.set_params(params)
        first = e.adam_step(0.003)["nll"]
        for _ in range(5):
            last = e.adam_step(0.003)["nll"]
    assert np.isfinite(first) and last < first


def test_explicit_u_gradient_dim_shards_add_up():
    params, Y, c, meta = synthetic.make_named("small")
    S, D = meta["S"], meta["D"]

    def grads(d0, dc, shared):
        with ElboEngine(meta["T"], D, meta["C"], meta["M"], S, U_collapse=False, grad=True, d_begin=d0, d_count=dc,
                        shared_terms=shared) as e:
            e.set_data(Y, c)
            return e.nll_and_grad(params, S_total=S)[1]

    whole, a, b = grads(0, D, True), grads(0, 1, True), grads(1, D - 1, False)
    for k in GRAD_KEYS + ("U",):
        np.testing.assert_allclose(a[k] + b[k], whole[k], rtol=0, atol=1e-10 * np.max(np.abs(whole[k])) + 1e-300)


def test_explicit_inverse_backward_switch(monkeypatch):
    """FFVD_GRAD_EXPLICIT=1 (read when the handle is created) selects the older backward pass that forms K^-1 - A^-1 from
    two explicit inverses: 20 % faster at the headline shape, eps * cond(K_uu) less accurate on dZ.
    Both forms must agree with the oracle on a well-conditioned case, the default one more closely."""
    from oracle import ffvd_grad_oracle as gorc
    monkeypatch.setenv("FFVD_NO_TINY", "1")           # a switch of the multi-kernel backward pass
    params, Y, c, meta = synthetic.make_named("tiny")
    S = meta["S"]
    want = None
    for s in range(S):
        a = gorc.nll_grad(dict(params, X=params["X"][s]), Y, c)
        want = a["Z"] / S if want is None else want + a["Z"] / S

    def dz():
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S, route="gram", grad=True) as e:
            e.set_data(Y, c)
            t, g = e.nll_and_grad(params)
        return t["nll"], g["Z"]

    nll_w, z_w = dz()
    monkeypatch.setenv("FFVD_GRAD_EXPLICIT", "1")
    nll_e, z_e = dz()
    scale = np.max(np.abs(want))
    assert nll_e == pytest.approx(nll_w, rel=1e-9)
    assert np.max(np.abs(z_w - want)) < 1e-7 * scale
    assert np.max(np.abs(z_e - want)) < 1e-5 * scale
    assert not np.array_equal(z_w, z_e)              # the switch really selects another code path


def test_combo_workgroups_and_tail_split_at_m1024():
    """Round 3 Gram schedule away from the headline shape: M = 1024 (two groups of four panels: 28 off-diagonal tiles + 6 combo
    workgroups per unit) with 32 units in one unsplit launch = 1088 workgroups, of which the last 64 are cut into row halves (32 chunks
    each at T = 1024; T >= M because the synthetic inducing inputs are drawn from the trajectory).  Every chain must equal what a one-chain engine (2 units: the split-K schedule with the legacy diagonal tiles)
    computes for it, two chains are checked against the oracle, forward and gradient are bit-reproducible."""
    params, Y, c, meta = synthetic.make_workload(T=1024, D=2, C=1, M=1024, S=16)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], 16, route="gram") as e:
        e.set_data(Y, c)
        full = e.nll_terms(params)
        again = e.nll_terms(params)
    np.testing.assert_array_equal(full["nll_per_chain"], again["nll_per_chain"])
    for s in (0, 7, 15):
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], 1, route="gram") as e:
            e.set_data(Y, c)
            one = e.nll_terms(dict(params, X=params["X"][s:s + 1]))
        # as many inducing points as transitions: cond(K_uu) ~ 1e7 and an nll of -0.03; the two schedules sum the Gram matrices in
        # different orders, which moves the Gram route by eps * cond (measured 1.3e-9 absolute)
        assert full["nll_per_chain"][s] == pytest.approx(one["nll"], rel=1e-8, abs=2e-8)
    for s in (0, 15):
        ref = orc.nll_terms(dict(params, X=params["X"][s]), Y, c, U_collapse=True)
        assert full["nll_per_chain"][s] == pytest.approx(ref["nll"], rel=1e-7, abs=2e-8)
    # training forward + backward through the same launch shape (the Gram kernel also stores the symmetric copy of A)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], 16, route="gram", grad=True) as e:
        e.set_data(Y, c)
        t1, g1 = e.nll_and_grad(params)
        t2, g2 = e.nll_and_grad(params)
    assert t1["nll"] == pytest.approx(full["nll"], rel=1e-8, abs=2e-8)
    for k in GRAD_KEYS:
        np.testing.assert_array_equal(g1[k], g2[k])
        assert np.all(np.isfinite(g1[k]))


def test_tail_exchange_bit_identity_soak():
    """The half-block exchange of the Gram kernel's tail split (`gram_tail_exchange`: relaxed write-through stores, no release
    fence -- sound on gfx942 / gfx950 only, and the build refuses other targets) as a TEST rather than only a tool (ADVICE r3;
    tools/soak.py, tools/soak_train.py run longer): at the headline shape every iteration cuts its last 128 workgroups into 256
    halves that meet through that exchange.  250 forward iterations and 25 forward + backward passes, every result
    bit-identical to the first, no stall recovery."""
    params, Y, c, meta = synthetic.make_named("c2")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram") as e:
        e.set_data(Y, c)
        e.set_params(params)
        first = e.nll_terms()
        for i in range(250):
            t = e.nll_terms()
            assert t["nll"] == first["nll"], i
            assert np.array_equal(t["nll_per_chain"], first["nll_per_chain"]), i
        assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True) as e:
        e.set_data(Y, c)
        t0, g0 = e.nll_and_grad(params)
        for i in range(25):
            t, g = e.nll_and_grad(params)
            assert t["nll"] == t0["nll"], i
            for k in GRAD_KEYS:
                assert np.array_equal(g[k], g0[k]), (i, k)
        assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
    assert t0["nll"] == pytest.approx(first["nll"], rel=1e-9)


STALLED_RANK_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FFVD_ROOT"])
import numpy as np, torch, torch.distributed as dist
from ffvd_amd import synthetic
from ffvd_amd.distributed import ShardedElbo, finish
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
params, Y, c, meta = synthetic.make_named("small")
sh = ShardedElbo(params, Y, c, meta, rank=rank, world=world, mode="chains", device=0, route="gram", grad=True, collective="torch")
before = sh.engine.get_params()
out = []
for what in ("step", "adam"):
    try:
        if what == "step":
            sh.step()
        else:
            sh.adam_step(1e-2)
        out.append("OK")
    except Exception as exc:                     # noqa: BLE001 -- the test asserts on the type and text
        out.append(type(exc).__name__ + ":" + str(exc).replace(" ", "_")[:240])
after = sh.engine.get_params()
same = all(np.array_equal(before[k], after[k]) for k in before)
digest = float(sum(np.abs(after[k]).sum() for k in ("Z", "logvariance", "loglengthscales", "log_Q")))
print("RESULT", rank, out[0], out[1], int(same), repr(digest), flush=True)
dist.destroy_process_group()
'''


def test_an_abandoned_factorisation_fails_on_every_rank(tmp_path):
    """VERDICT r3 W8 / ADVICE r3 (medium): a dataflow Cholesky that gives up on a bounded wait (info = -1) leaves FINITE garbage in
    that rank's sums; collective calls do not retry.  `finalize_kernel` now turns the seven sums into NaN whenever an info flag of
    its rank is non-zero, so the all-reduce carries the failure to every rank.  Two ranks share the GPU; rank 1 runs the `dfstall`
    build, whose every dataflow launch stalls: BOTH ranks must raise from the forward step and from the sharded Adam step, and
    both must leave their (replicated) parameters untouched and equal -- the reference's only failure mode is likewise an
    error at `session.run` (dgp_model.py:320-324)."""
    import subprocess
    import sys
    from ffvd_amd import build as fb
    stall_lib = fb.build_variant("dfstall")
    script = tmp_path / "stalled_rank_worker.py"
    script.write_text(STALLED_RANK_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFVD_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2",
               FFVD_NO_TINY="1")           # the dfstall build stalls the dataflow Cholesky of the multi-kernel schedule
    env.pop("FFVD_CHOL", None)
    env.pop("FFVD_LIB", None)
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r))
        if r == 1:
            e["FFVD_LIB"] = stall_lib
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    lines = [[l for l in out.splitlines() if l.startswith("RESULT")][0].split() for out in outs]
    for line in lines:
        assert line[2] != "OK" and line[3] != "OK", line            # forward step and Adam step raise on BOTH ranks
        assert int(line[4]) == 1, line                              # parameters untouched
    assert "abandoned" in lines[1][2] or "bounded" in lines[1][2] or "abandoned" in lines[1][3], lines[1]     # the stalled rank names its own failure
    assert "another_rank" in lines[0][2] and "another_rank" in lines[0][3], lines[0]
    assert lines[0][5] == lines[1][5]                               # replicas still equal


def _schedule_case(case):
    if case == "c2_full":
        params, Y, c, meta = synthetic.make_named("c2")
        return params, Y, c, meta, dict(route="gram"), "full unsplit"
    if case == "c2_rank4":        # (forward at 1, 2, 4, 8 chains: the side chain behind the tile pass; with grad and at other counts: beside it)
        params, Y, c, meta = synthetic.make_named("c2", S=4)
        return params, Y, c, meta, dict(route="gram"), "side late"
    if case == "c2_rank3":
        params, Y, c, meta = synthetic.make_named("c2", S=3)
        return params, Y, c, meta, dict(route="gram"), "split-K one pass"
    if case == "c2_rank8":
        params, Y, c, meta = synthetic.make_named("c2", S=8)
        return params, Y, c, meta, dict(route="gram"), "side late"
    if case == "c2_rank2":
        params, Y, c, meta = synthetic.make_named("c2", S=2)
        return params, Y, c, meta, dict(route="gram"), "side late"
    if case == "c2_16":           # (16 chains = 64 units = 512 workgroups, one whole round: unsplit, below the 128 units of the full-batch schedule)
        params, Y, c, meta = synthetic.make_named("c2", S=16)
        return params, Y, c, meta, dict(route="gram"), "side late: unsplit pass with raw tiles"
    if case == "c2_reference":
        params, Y, c, meta = synthetic.make_named("c2", S=8)
        return params, Y, c, meta, dict(route="reference"), "projection route, K_uu chain as one dataflow launch on the side stream"
    if case == "actuator_multi_kernel":
        params, Y, c, meta = synthetic.make_named("small", S=1)
        return params, Y, c, meta, dict(route="gram"), "small side"
    raise KeyError(case)


@pytest.mark.parametrize("S", [1, 4])
def test_inverse_tiles_by_column_equal_the_tail_form(S, monkeypatch):
    """Round 5 (lead 2): in the few-chain schedules the K_uu chain's dataflow launch also leaves K^-1 = L^-T L^-1; its
    tiles can be accumulated column by column as the rows of L^-T arrive (df_inverse_column, FFVD_DF_KACC=1: the accumulators travel
    through memory between columns; measured slower, opt-in) instead of all at once behind the last column (df_inverse_tiles).  The same products on the same
    accumulators in the same order: every term and the per-chain nll bit for bit, at the shape of a 1- and a 4-chain rank of config 2."""
    params, Y, c, meta = synthetic.make_named("c2", S=S)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FFVD_DF_KACC", mode)
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram") as e:
            assert "side late" in e.lib.ffvd_schedule_name(e._h).decode()
            e.set_data(Y, c)
            outs[mode] = (e.nll_terms(params), e.nll_terms(params))
            assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
    monkeypatch.delenv("FFVD_DF_KACC")
    for n in TERMS_B:
        assert outs["1"][0][n] == outs["0"][0][n] == outs["1"][1][n], n
    np.testing.assert_array_equal(outs["1"][0]["nll_per_chain"], outs["0"][0]["nll_per_chain"])
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
    assert outs["1"][0]["nll"] == pytest.approx(ref["nll"], rel=1e-8)


@pytest.mark.parametrize("passes,mode", [(2, 0), (3, 0), (4, 1), (5, 2)])
def test_pipelined_passes_are_bit_identical(passes, mode, monkeypatch):
    """the pass-pipelined forward iteration (enqueue_elbo_pipe, FFVD_PIPE / FFVD_PIPE_MODE: K_fu build of pass p+1 |
    Gram kernel of pass p | Cholesky(A) of pass p-1 on separate streams; measured slower, kept opt-in -- DESIGN.md section 5 "Round 5")
    runs the same kernels on the same units with another launch partition: at BASELINE configs[1]'s full shape its terms and per-chain
    nll equal the full-batch schedule's bit for bit -- uneven partitions (32 chains in 3 or 5 passes) and delayed streams included."""
    params, Y, c, meta = synthetic.make_named("c2")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram") as e:
        assert "full unsplit" in e.lib.ffvd_schedule_name(e._h).decode()
        e.set_data(Y, c)
        base = e.nll_terms(params)
    monkeypatch.setenv("FFVD_PIPE", str(passes))
    monkeypatch.setenv("FFVD_PIPE_MODE", str(mode))
    for env in ({}, {"FFVD_DEBUG_SIDE_DELAY_US": "300"}, {"FFVD_DEBUG_MAIN_DELAY_US": "300"}):
        for k in ("FFVD_DEBUG_SIDE_DELAY_US", "FFVD_DEBUG_MAIN_DELAY_US"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram") as e:
            assert "pipelined passes" in e.lib.ffvd_schedule_name(e._h).decode()
            e.set_data(Y, c)
            got = e.nll_terms(params)
            again = e.nll_terms(params)
            assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
        for n in TERMS_B:
            assert got[n] == base[n] and again[n] == base[n], (passes, mode, env, n)
        np.testing.assert_array_equal(got["nll_per_chain"], base["nll_per_chain"])


@pytest.mark.parametrize("case,grad", [("c2_full", False), ("c2_full", True), ("c2_rank4", False), ("c2_rank4", True), ("c2_rank3", False), ("c2_rank8", False), ("c2_rank2", False), ("c2_16", False),
                                       ("c2_reference", False), ("actuator_multi_kernel", False), ("actuator_multi_kernel", True)])
def test_results_do_not_depend_on_which_stream_is_late(case, grad, monkeypatch):
    """the iteration runs on two streams tied by events, and a missing wait would not crash -- stale progress words or
    a half-written K^-1 give a finite wrong nll.  `plan_schedule` (abi.hip) now decides every schedule flag in one place and names
    the schedule; here each schedule of the multi-kernel path runs three times -- as it is, with a 300 us spin kernel at the head of
    the side stream at every fork, and with one on the main stream -- and the three results must be BIT-identical (forward terms,
    per-chain nll, and with grad the whole gradient)."""
    monkeypatch.setenv("FFVD_NO_TINY", "1")
    params, Y, c, meta, kw, name = _schedule_case(case)
    outs = []
    for env in ({}, {"FFVD_DEBUG_SIDE_DELAY_US": "300"}, {"FFVD_DEBUG_MAIN_DELAY_US": "300"}):
        for k in ("FFVD_DEBUG_SIDE_DELAY_US", "FFVD_DEBUG_MAIN_DELAY_US"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], grad=grad, **kw) as e:
            sched = e.lib.ffvd_schedule_name(e._h).decode()
            assert sched != "INVALID"
            if not grad:
                assert name in sched, (case, sched)
            e.set_data(Y, c)
            if grad:
                t, g = e.nll_and_grad(params)
            else:
                t, g = e.nll_terms(params), {}
            assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
        outs.append((t, g))
    t0, g0 = outs[0]
    for t, g in outs[1:]:
        for n in TERMS_B:
            assert t[n] == t0[n], (case, n)
        np.testing.assert_array_equal(t["nll_per_chain"], t0["nll_per_chain"])
        for k in g0:
            np.testing.assert_array_equal(g[k], g0[k])
