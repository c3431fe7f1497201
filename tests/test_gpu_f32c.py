"""GPU parity of the fp32-contraction path (ffvd_config.dtype = FFVD_F32C, BASELINE configs[3]) and BASELINE
configs[3] at its FULL shape (T=16384, x_dim=8, M=2048, S=64) in both arithmetics.

Tolerances (stated here, measured on MI355X): the fp32 path rounds K_fu, L^-1 (as GEMM
operand) and F to fp32 and sums the two T x M x M products in fp32 chains of at most 4096 terms; against the fp64
oracle every component term and the nll agree to 1e-5 ABSOLUTE (the terms are O(1e-2..1), the nll O(0.1..3); the trace
term is a cancellation).  The error grows with |L^-T|, i.e. with how densely the inducing points cover the data:
measured worst case 5.5e-6 (M = 1100 inducing points for T = 1400 transitions), <= 1.2e-6 at the BASELINE shapes.
The north-star acceptance is rtol 1e-4 on the nll."""
import numpy as np
import pytest

from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_oracle as orc

pytestmark = pytest.mark.gpu

TERMS_B = ("nll_part_prior", "nll_log_likelihood", "x_t_prior_Q", "nll_reg_trace_inverse_Q_B", "later_term1",
           "later_term2", "nll")
NLL_RTOL_F32C = 5e-6          # relative part, for per-chain comparisons at the BASELINE shapes
TERM_ATOL_F32C = 1e-5


def run_engine(params, Y, c, meta, **kw):
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], params["X"].shape[0], Ydim=Y.shape[1],
                    kernel_type=meta["kernel_type"], U_collapse=True, **kw) as eng:
        eng.set_data(Y, c)
        return eng.nll_terms(params)


def check_f32c(got, ref, label=""):
    errs = {n: abs(got[n] - ref[n]) for n in TERMS_B}
    print(f"f32c vs fp64 oracle {label}: nll rel.err {errs['nll'] / abs(ref['nll']):.2e}; abs term errors "
          + ", ".join(f"{n}={e:.1e}" for n, e in errs.items()))
    for n in TERMS_B:
        assert errs[n] <= TERM_ATOL_F32C, (n, got[n], ref[n])
    assert errs["nll"] <= 1e-4 * abs(ref["nll"])                      # the north-star acceptance, with a wide margin


@pytest.mark.parametrize("ov", [dict(), dict(T=301, M=77, D=3, C=2, S=2), dict(T=700, M=150, D=2, C=0, S=3),
                                dict(T=40, M=9, D=1, C=1, S=1), dict(T=1000, M=600, D=2, C=1, S=2),
                                dict(T=257, M=130, D=5, C=8, S=2), dict(T=1400, M=1100, D=1, C=1, S=2)],
                         ids=["small", "ragged", "Mp192_C0", "tiny_D1", "M600", "P13", "M1100"])
def test_f32c_against_oracle(ov):
    """Seeded shapes incl. ragged T/M (Mp = 128, 192, 640, 1152), no control input, P = 13 > 12 (the generic K_fu build)."""
    params, Y, c, meta = synthetic.make_named("small", **ov)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
    got = run_engine(params, Y, c, meta, dtype="f32c")
    check_f32c(got, ref, str(ov))
    np.testing.assert_allclose(got["nll_per_chain"], ref["nll_per_chain"], rtol=0, atol=TERM_ATOL_F32C)


def test_f32c_linear_kernel():
    """LinearK through the fp32 K_fu build (K_uu of rank P << M: the whitened products are conditioned like 1e5 |K|)."""
    params, Y, c, meta = synthetic.make_named("small_lin", U_collapse=True, S=2)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True, kernel_type="LinearK")
    got = run_engine(params, Y, c, meta, dtype="f32c")
    print("f32c LinearK nll", got["nll"], ref["nll"])
    assert got["nll"] == pytest.approx(ref["nll"], rel=1e-3)


def test_f32c_headline_shape_vs_fp64_engine():
    """config 2 shape (T=4096, M=512, D=4, S=32) in fp32 contractions vs the fp64 engine, chain by chain; two runs are
    bitwise equal (no atomics); the accumulator flush period does not matter beyond rounding."""
    params, Y, c, meta = synthetic.make_named("c2")
    ref = run_engine(params, Y, c, meta, route="reference")
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], dtype="f32c") as e:
        e.set_data(Y, c)
        got = e.nll_terms(params)
        again = e.nll_terms(params)
    np.testing.assert_array_equal(got["nll_per_chain"], again["nll_per_chain"])
    check_f32c(got, ref, "c2")
    np.testing.assert_allclose(got["nll_per_chain"], ref["nll_per_chain"], rtol=NLL_RTOL_F32C)


def test_f32c_several_passes_and_shards():
    """chains_per_pass < S (the workspace holds one chain at a time), a latent-dim shard and a chain subset: every layout
    gives the same per-chain values as the one-pass evaluation (bitwise: the per-unit arithmetic does not depend on the
    batch), and the dim shards add up."""
    params, Y, c, meta = synthetic.make_named("ragged", S=3)
    one = run_engine(params, Y, c, meta, dtype="f32c")
    many = run_engine(params, Y, c, meta, dtype="f32c", chains_per_pass=1)
    np.testing.assert_array_equal(one["nll_per_chain"], many["nll_per_chain"])
    total = np.zeros(8)
    for d0, dc, shared in ((0, 1, True), (1, 2, False)):
        with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], dtype="f32c", d_begin=d0, d_count=dc,
                        shared_terms=shared) as e:
            e.set_data(Y, c)
            total += e.elbo_sums(params)
    assert total[7] == meta["S"]
    assert total[6] / total[7] == pytest.approx(one["nll"], rel=1e-12)


def test_f32c_usage_errors():
    with pytest.raises(ValueError):
        ElboEngine(64, 2, 1, 16, 1, dtype="f32c", route="gram")
    with pytest.raises(ValueError):
        ElboEngine(64, 2, 1, 16, 1, dtype="f32c", U_collapse=False)
    with pytest.raises(ValueError):
        ElboEngine(64, 2, 1, 16, 1, dtype="bf16")


def test_f32c_not_positive_definite_is_reported():
    params, Y, c, meta = synthetic.make_named("tiny")
    bad = dict(params)
    bad["X"] = params["X"].copy()
    bad["X"][1, 7, 0] = np.nan
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], dtype="f32c") as e:
        e.set_data(Y, c)
        with pytest.raises(np.linalg.LinAlgError, match="chain 1"):
            e.nll_terms(bad)
        assert np.isfinite(e.nll_terms(params)["nll"])


# ---- BASELINE configs[3] at its full shape ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def c4():
    return synthetic.make_named("c4")


@pytest.fixture(scope="module")
def c4_chain0_oracle(c4):
    params, Y, c, meta = c4
    p = dict(params, X=params["X"][:1])
    return orc.nll_terms_chains(p, Y, c, U_collapse=True)          # ~30-60 s of CPU: one chain, eight latent dims


def test_config4_full_shape_one_chain_vs_oracle(c4, c4_chain0_oracle):
    """T=16384, x_dim=8, M=2048: chain 0 through all three evaluation paths against the fp64 CPU oracle."""
    params, Y, c, meta = c4
    p = dict(params, X=params["X"][:1])
    ref = c4_chain0_oracle
    got = run_engine(p, Y, c, meta, route="reference")
    for n in TERMS_B:
        assert got[n] == pytest.approx(ref[n], rel=1e-8, abs=1e-9), (n, got[n], ref[n])
    got = run_engine(p, Y, c, meta, route="gram")
    assert got["nll"] == pytest.approx(ref["nll"], rel=1e-7)
    for n in TERMS_B:
        assert got[n] == pytest.approx(ref[n], rel=1e-6, abs=1e-8), (n, got[n], ref[n])
    got = run_engine(p, Y, c, meta, dtype="f32c")
    check_f32c(got, ref, "c4 chain 0")


@pytest.mark.parametrize("arith", ["f32c", "f64-gram", "f64-reference"])
def test_config4_full_shape_all_chains(c4, c4_chain0_oracle, arith):
    """All 64 chains at the full shape: the batch's chain 0 is the oracle's; a chain's nll does not depend on what else
    is in the batch (sub-batches, reversed order: bitwise for equal pass layouts, else to rounding); the mean is the mean."""
    params, Y, c, meta = c4
    kw = dict(dtype="f32c") if arith == "f32c" else dict(route=arith.split("-")[1])
    tol = NLL_RTOL_F32C if arith == "f32c" else 1e-7
    S = meta["S"]
    full = run_engine(params, Y, c, meta, **kw)
    assert np.all(np.isfinite(full["nll_per_chain"]))
    assert full["nll_per_chain"][0] == pytest.approx(c4_chain0_oracle["nll"], rel=tol)
    assert full["nll"] == pytest.approx(full["nll_per_chain"].mean(), rel=1e-13)
    sel = [63, 1, 40, 0]
    sub = run_engine(dict(params, X=params["X"][sel]), Y, c, meta, **kw)
    np.testing.assert_allclose(sub["nll_per_chain"], full["nll_per_chain"][sel], rtol=1e-9 if arith != "f32c" else 1e-7)
    if arith == "f32c":
        rev = run_engine(dict(params, X=params["X"][::-1].copy()), Y, c, meta, **kw)
        np.testing.assert_allclose(rev["nll_per_chain"][::-1], full["nll_per_chain"], rtol=1e-7)


# ---- backward pass in the reference's op order: fp64 and fp32 contractions (base_model.py:148) ------------

GRAD_KEYS = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")


def _oracle_gradient(params, Y, c):
    """closed-form gradient of the mean-over-chains nll (oracle/ffvd_grad_oracle.py, checked against torch autograd)"""
    from oracle import ffvd_grad_oracle as gorc
    S = params["X"].shape[0]
    ref = None
    for s in range(S):
        g = gorc.nll_grad(dict(params, X=params["X"][s]), Y, c)
        if ref is None:
            ref = {k: (np.zeros((S,) + g[k].shape) if k == "X" else np.zeros_like(g[k])) for k in GRAD_KEYS}
        ref["X"][s] = g["X"] / S
        for k in GRAD_KEYS[1:]:
            ref[k] += g[k] / S
    return ref


def _grad_errors(g, ref):
    return {k: float(np.max(np.abs(np.asarray(g[k]).reshape(ref[k].shape) - ref[k])) / (np.max(np.abs(ref[k])) + 1e-300)) for k in GRAD_KEYS}


SHAPES = [dict(), dict(T=301, M=77, D=3, C=2, S=2), dict(T=700, M=150, D=2, C=0, S=3), dict(T=40, M=9, D=1, C=1, S=1),
          dict(T=1000, M=600, D=2, C=1, S=2), dict(T=257, M=130, D=5, C=8, S=2)]
SHAPE_IDS = ["small", "ragged", "Mp192_C0", "tiny_D1", "M600", "P13"]


@pytest.mark.parametrize("ov", SHAPES, ids=SHAPE_IDS)
def test_reference_route_gradient_fp64(ov):
    """grad = 1 on FFVD_ROUTE_REFERENCE (F = K_fu L^-T, H = F^T F / Q + I factorised as the reference does): the backward pass
    reads the factor of H, L_H^-T, L^-1 and sum_s H_s from that forward pass; every array against the closed-form oracle."""
    params, Y, c, meta = synthetic.make_named("small", **ov)
    ref = _oracle_gradient(params, Y, c)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="reference", grad=True) as e:
        e.set_data(Y, c)
        t, g = e.nll_and_grad(params)
        t2, g2 = e.nll_and_grad(params)
    for k in GRAD_KEYS:
        np.testing.assert_array_equal(g[k], g2[k])
    assert t["nll"] == pytest.approx(orc.nll_terms_chains(params, Y, c, U_collapse=True)["nll"], rel=1e-9)
    errs = _grad_errors(g, ref)
    print("reference-route fp64 gradient vs oracle", ov, {k: f"{v:.1e}" for k, v in errs.items()})
    for k, v in errs.items():
        assert v < (1e-6 if k in ("Z", "loglengthscales", "logvariance") else 1e-7), (k, v)


@pytest.mark.parametrize("ov", SHAPES, ids=SHAPE_IDS)
def test_f32c_gradient_against_oracle(ov):
    """dtype = f32c with grad = 1: K_fu and the products K_fu L^-T, F^T F AND K_fu Gamma in fp32 on the matrix cores, the M x M
    side and every reduction in fp64.  Stated tolerance, relative to the largest entry of each array: 1e-2 for Z and the kernel
    hyper-parameters, 1e-3 for X, 1e-4 for log_Q, 1e-9 for C / d / R (which never see the fp32 products).  Measured on MI355X over
    these shapes: X <= 9e-5, Z <= 3.5e-3, logvariance <= 9.4e-4, loglengthscales <= 5.7e-4, log_Q <= 3.5e-6.  VERDICT r2 asked for
    1e-4 throughout; that is not what ONE fp32 product gives here: dl/dK_fu = 2 K_fu Gamma with |Gamma| ~ alpha |L^-T|^2 up to 1e4,
    so rounding K_fu and Gamma to fp32 (6e-8 relative) leaves 1e-4..1e-3 of the largest entry of the K_fu-side sums
    (tools/f32_backward_sim.py reproduces 4e-5..9e-4 on the CPU; a hi + lo split of Gamma -- two fp32 products, the time of the
    fp64 one -- only halves it; an F accurate to fp32 rounding would give 1e-5 but IS the fp64 product), and dZ is a difference
    of that side and the fp64 K_uu side.  The step direction of Adam tolerates it: the next test trains at the config-4 shape."""
    params, Y, c, meta = synthetic.make_named("small", **ov)
    ref = _oracle_gradient(params, Y, c)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], dtype="f32c", grad=True) as e:
        e.set_data(Y, c)
        t, g = e.nll_and_grad(params)
        t2, g2 = e.nll_and_grad(params)
    for k in GRAD_KEYS:
        np.testing.assert_array_equal(g[k], g2[k])                      # no atomics: bitwise reproducible
    errs = _grad_errors(g, ref)
    print("f32c gradient vs fp64 oracle", ov, {k: f"{v:.1e}" for k, v in errs.items()})
    tol = dict(X=1e-3, Z=1e-2, logvariance=1e-2, loglengthscales=1e-2, log_Q=1e-4, CC=1e-9, DD=1e-9, log_Rchols=1e-9)
    for k, v in errs.items():
        assert v < tol[k], (k, v)


def test_f32c_adam_step_lowers_the_nll_at_the_config4_shape():
    """BASELINE configs[3] trains: device-resident Adam steps in fp32-contraction arithmetic at T=16384, x_dim=8, M=2048 (4 of
    the 64 chains: the shape of every launch is config 4's, the batch is what the test box's time allows)."""
    from ffvd_amd import optim
    params, Y, c, meta = synthetic.make_named("c4", S=4)
    with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], 4, dtype="f32c", grad=True) as e:
        e.set_data(Y, c)
        e.set_params(params)
        nlls = [e.adam_step(10 * optim.decayed_learning_rate())["nll"] for _ in range(3)]
        nlls.append(e.nll_terms()["nll"])
    print("c4-shape f32c training nll:", nlls)
    assert all(np.isfinite(nlls)) and nlls[-1] < nlls[0] and nlls[1] < nlls[0]
