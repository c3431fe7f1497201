"""GPU parity of the one-launch iteration (ffvd_amd/csrc/tiny.hip): the whole ELBO evaluation -- and its backward pass -- of the
reference's own experiment size (FFVD_Main.py:356-369: T <= 512, M = 100, D = 4; models.py:142-182) as ONE kernel launch, against
the CPU oracle (nll + the six named terms of dgp_model.py:264-297, the closed-form gradient of oracle/ffvd_grad_oracle.py), the
committed goldens, and the multi-kernel schedule of the same library (FFVD_NO_TINY=1)."""
import os

import numpy as np
import pytest

from conftest import load_golden
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
from oracle import ffvd_oracle as orc
from oracle import ffvd_grad_oracle as gorc

pytestmark = pytest.mark.gpu

TERMS = ("nll_part_prior", "nll_log_likelihood", "x_t_prior_Q", "nll_reg_trace_inverse_Q_B", "later_term1", "later_term2", "nll")
GRAD_KEYS = ("X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")


def engine(meta, S=None, **kw):
    return ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], S if S is not None else meta["S"], Ydim=meta["Ydim"], **kw)


def single_launch(e):
    return int(e.lib.ffvd_single_launch(e._h))


def oracle_grad(params, Y, c, **kw):
    S = params["X"].shape[0]
    want = None
    for s in range(S):
        g = gorc.nll_grad(dict(params, X=params["X"][s]), Y, c, **kw)
        if want is None:
            want = {k: np.zeros_like(v) for k, v in g.items() if k != "X"}
            want["X"] = np.zeros_like(params["X"])
        for k in g:
            if k == "X":
                want["X"][s] = g["X"] / S
            else:
                want[k] += g[k] / S
    return want


def actuator(S):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "actuator_slim.npz"), allow_pickle=False)
    params = {k: z[k] for k in ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")}
    Y, c = z["Y"], z["control_inputs"]
    T, D = params["X"].shape[0] - 1, params["X"].shape[1]
    meta = dict(T=T, D=D, C=c.shape[1], M=params["Z"].shape[0], S=S, Ydim=Y.shape[1])
    X = np.repeat(params["X"][None], S, axis=0)
    if S > 1:
        X = X + 1e-3 * np.random.default_rng(0).standard_normal(X.shape)
        X[0] = params["X"]
    params["X"] = np.ascontiguousarray(X)
    return params, Y, c, meta, z


@pytest.mark.parametrize("name", ["tiny", "small", "ragged"])
def test_forward_matches_golden_and_oracle(name):
    """The seeded synthetic shapes of tests/golden (M = 24, 96, 77: padded to 32, 96, 80 -- multiples of 16, not of 64; T = 96, 384,
    301: ragged last strip) through ONE launch: every named term against the golden vector and the oracle to 1e-10."""
    params, Y, c, meta = synthetic.make_named(name)
    g = load_golden(name)
    with engine(meta) as e:
        assert single_launch(e) in (4, 8)
        e.set_data(Y, c)
        got = e.nll_terms(params)
        again = e.nll_terms(params)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
    for n in TERMS:
        assert got[n] == pytest.approx(float(g["B_" + n]), rel=1e-10, abs=1e-11), n
        assert got[n] == pytest.approx(ref[n], rel=1e-10, abs=1e-11), n
        assert got[n] == again[n], n                    # bit-reproducible: fixed summation orders whoever closes
    np.testing.assert_allclose(got["nll_per_chain"], ref["nll_per_chain"], rtol=1e-10)


@pytest.mark.parametrize("S", [1, 10])
def test_actuator_config1(S):
    """BASELINE configs[0] (actuator, x_dim = 4, M = 100, S = 10): the fixture's nll and terms, one launch.  Chain 0 carries the
    fixture's own X: its nll is the golden value (SURVEY 8a anchor -2.369303046...)."""
    params, Y, c, meta, z = actuator(S)
    with engine(meta) as e:
        assert single_launch(e) in (4, 8)
        e.set_data(Y, c)
        got = e.nll_terms(params)
    assert got["nll_per_chain"][0] == pytest.approx(float(load_golden("actuator")["B_nll"]), rel=1e-10)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
    for n in TERMS:
        assert got[n] == pytest.approx(ref[n], rel=1e-10, abs=1e-11), n


@pytest.mark.parametrize("shape", [dict(T=64, D=1, C=1, M=16, S=1), dict(T=17, D=2, C=0, M=5, S=3), dict(T=512, D=4, C=1, M=128, S=2),
                                   dict(T=1000, D=3, C=2, M=113, S=2), dict(T=200, D=6, C=2, M=40, S=5), dict(T=130, D=2, C=1, M=97, S=7),
                                   dict(T=512, D=4, C=1, M=100, S=10), dict(T=512, D=4, C=1, M=100, S=6), dict(T=700, D=4, C=1, M=128, S=5)])
def test_shapes_forward_and_gradient(shape):
    """Edge shapes: one tile, no control input, M = 128 (the largest), M = 113 (padded to 128), P = 8 (the largest), T just over a
    strip boundary; the actuator shape at 10 chains (8 wavefronts per workgroup, 128-row strips, no workgroups of their own for the
    K_uu side: the strips and the head share its row blocks) and at 6 chains (4 wavefronts, likewise no side workgroups); M = 128 without
    side workgroups.  Forward against the oracle, gradient against the closed form."""
    params, Y, c, meta = synthetic.make_workload(**shape)
    with engine(meta, grad=True, route="gram") as e:
        assert single_launch(e) in (4, 8), shape
        e.set_data(Y, c)
        t, g = e.nll_and_grad(params)
        f = e.nll_terms(params)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True)
    for n in TERMS:
        assert t[n] == pytest.approx(ref[n], rel=1e-9, abs=1e-10), n
        assert f[n] == t[n], n
    want = oracle_grad(params, Y, c)
    for k in GRAD_KEYS:
        scale = float(np.max(np.abs(want[k]))) or 1.0
        tol = 2e-6 if k == "Z" else 1e-7           # dZ: eps * cond(K_uu) in BOTH closed forms
        assert np.max(np.abs(g[k] - want[k])) < tol * scale, (k, np.max(np.abs(g[k] - want[k])) / scale)


def test_equals_the_multi_kernel_schedule(monkeypatch):
    """The same library with FFVD_NO_TINY=1 (rounds 1-3: 17 / 53 dependent launches on two streams) and the one-launch path give the
    same nll and the same gradient; four device-resident Adam steps from either path end at the same parameters."""
    params, Y, c, meta = synthetic.make_named("small")
    out = {}
    for mode in ("one", "multi"):
        if mode == "multi":
            monkeypatch.setenv("FFVD_NO_TINY", "1")
        with engine(meta, grad=True, route="gram") as e:
            assert (single_launch(e) != 0) == (mode == "one")
            e.set_data(Y, c)
            t, g = e.nll_and_grad(params)
            e.set_params(params)
            for _ in range(4):
                e.adam_step(0.003)
            out[mode] = (t, g, e.get_params())
    monkeypatch.delenv("FFVD_NO_TINY")
    assert out["one"][0]["nll"] == pytest.approx(out["multi"][0]["nll"], rel=1e-10)
    for k in GRAD_KEYS:
        a, b = out["one"][1][k], out["multi"][1][k]
        assert np.max(np.abs(a - b)) < (5e-6 if k == "Z" else 1e-7) * np.max(np.abs(b)), k
    for k in GRAD_KEYS:
        a, b = out["one"][2][k], out["multi"][2][k]
        assert np.max(np.abs(a - b)) < 1e-6 * max(1.0, np.max(np.abs(b))), k


def test_dim_shards_and_uniform_prior():
    """Latent-dim shards (d_begin / d_count, shared_terms on one shard only) and the chain-count divisor S_total through the one-launch
    path: the shards' sums and gradient shares add up to the single handle's; prior_type 'uniform' drops prior_Z."""
    params, Y, c, meta = synthetic.make_workload(T=150, D=4, C=1, M=50, S=3)
    with engine(meta, grad=True) as e:
        e.set_data(Y, c)
        whole_t, whole_g = e.nll_and_grad(params)
    sums = np.zeros(8)
    gsum = {k: 0.0 for k in GRAD_KEYS}
    for d0, dc, shared in ((0, 1, True), (1, 3, False)):
        with engine(meta, grad=True, d_begin=d0, d_count=dc, shared_terms=shared) as e:
            assert single_launch(e)
            e.set_data(Y, c)
            t, g = e.nll_and_grad(params)
            sums += t["sums8"]
            for k in GRAD_KEYS:
                gsum[k] = gsum[k] + g[k]
    assert sums[6] / sums[7] == pytest.approx(whole_t["nll"], rel=1e-11)
    for k in GRAD_KEYS:
        assert np.max(np.abs(gsum[k] - whole_g[k])) < 1e-9 * max(1e-30, np.max(np.abs(whole_g[k]))), k
    with engine(meta, grad=True, prior_type="uniform") as e:
        e.set_data(Y, c)
        t, g = e.nll_and_grad(params)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=True, prior_type="uniform")
    assert t["nll"] == pytest.approx(ref["nll"], rel=1e-10)
    want = oracle_grad(params, Y, c, prior_type="uniform")
    assert np.max(np.abs(g["Z"] - want["Z"])) < 2e-6 * np.max(np.abs(want["Z"]))


def test_non_positive_definite_is_reported():
    """A failed pivot inside the launch surfaces as LinAlgError with the factorisation named (dgp_model.py:320-324 is the reference's
    counterpart: an error at session.run), and the handle stays usable."""
    params, Y, c, meta = synthetic.make_named("tiny")
    bad = dict(params, Z=np.repeat(params["Z"][:1], meta["M"], axis=0))       # identical inducing points ...
    with engine(meta, jitter=0.0) as e:                                       # ... and no jitter: K_uu is singular
        e.set_data(Y, c)
        with pytest.raises(np.linalg.LinAlgError):
            e.nll_terms(bad)
    with engine(meta) as e:
        e.set_data(Y, c)
        assert np.isfinite(e.nll_terms(params)["nll"])


def test_model_loop_runs_on_one_launch_per_step():
    """RegressionModel.fit -- the loop of models.py:142-182 -- on the actuator fixture lowers the nll; every train_hypers step is one
    launch + the fused Adam update."""
    params, Y, c, meta, z = actuator(1)
    with engine(meta, grad=True) as e:
        assert single_launch(e)
        e.set_data(Y, c)
        e.set_params(params)
        first = e.adam_step(0.003)["nll"]
        for _ in range(30):
            last = e.adam_step(0.003)["nll"]
    assert last < first


_STALL_SCRIPT = r"""
import sys, time
import numpy as np
from ffvd_amd import synthetic, _lib
from ffvd_amd.engine import ElboEngine
from ffvd_amd.distributed import ShardedElbo
lib = _lib.load()
params, Y, c, meta = synthetic.make_named("small")
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], grad=True) as e:
    assert int(lib.ffvd_single_launch(e._h))
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
# a collective step does not retry: it must fail, and the sums it would have contributed are NaN (never the previous iteration's)
sh = ShardedElbo(params, Y, c, meta, rank=0, world=1, mode="chains", device=0, always_reduce=True)
try:
    sh.step()
    print("COLLECTIVE OK")
except Exception as exc:
    print("COLLECTIVE", type(exc).__name__, str(exc).replace(" ", "_")[:200])
sh.close()
"""


def test_one_launch_gives_up_instead_of_hanging(tmp_path):
    """Every wait of the one-launch iteration is bounded (the roles wait for each other inside ONE launch, and their progress rests
    on all workgroups being resident).  In the test build `libffvd_hip_tinystall.so` the head of unit 0 never publishes W: its
    strips give up after the 1 s bound, set the abort word, every waiting workgroup leaves, the launch ends with info = -1.  The
    synchronous entry points then run the iteration ONCE more on the multi-kernel schedule (launch-per-column Cholesky), return OK
    with a warning, and the result equals the golden value; a collective step does not retry and must raise."""
    import subprocess
    import sys
    import time
    from ffvd_amd import build as fb
    lib_path = fb.build_variant("tinystall")
    env = dict(os.environ, FFVD_LIB=lib_path, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("FFVD_CHOL", None)
    env.pop("FFVD_NO_TINY", None)
    zpath = str(tmp_path / "dz.npy")
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, "-c", _STALL_SCRIPT, zpath], env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stderr[-2000:]
    eline = [ln for ln in out.stdout.splitlines() if ln.startswith("ELBO")][0].split()
    gold = float(load_golden("small")["B_nll"])
    assert float(eline[1]) == pytest.approx(gold, rel=1e-8) and float(eline[2]) == pytest.approx(gold, rel=1e-8)
    assert int(eline[3]) == 1 and int(eline[4]) == 1       # the stall is remembered: the second call does not try the one launch again
    assert 0.5 < float(eline[5]) < 30.0                    # the bound is 1 s per wait; workgroups give up together via the abort word
    assert "re-run" in " ".join(eline[6:])
    cline = [ln for ln in out.stdout.splitlines() if ln.startswith("COLLECTIVE")][0]
    assert "OK" not in cline.split()[1:2] and ("abandoned" in cline or "non-finite" in cline), cline
    assert time.perf_counter() - t0 < 150.0
    # the product library still works on the same GPU afterwards, and gives the gradient the recovered run produced (multi-kernel
    # backward pass there, one launch here: eps * cond(K_uu) apart on dZ)
    params, Y, c, meta = synthetic.make_named("small")
    with engine(meta, grad=True) as e:
        e.set_data(Y, c)
        _, g = e.nll_and_grad(params)
        assert int(e.lib.ffvd_stall_recoveries(e._h)) == 0
    dz = np.load(zpath)
    np.testing.assert_allclose(dz, g["Z"], rtol=0, atol=5e-6 * float(np.max(np.abs(g["Z"]))))


_BACKOFF_SCRIPT = r"""
import time
import numpy as np
from ffvd_amd import synthetic, _lib
from ffvd_amd.engine import ElboEngine
lib = _lib.load()
params, Y, c, meta = synthetic.make_named("small")
with ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"]) as e:
    assert int(lib.ffvd_single_launch(e._h))
    e.set_data(Y, c)
    e.set_params(params)
    name0 = lib.ffvd_schedule_name(e._h).decode()
    t0 = time.perf_counter()
    vals = [e.nll_terms()["nll"] for _ in range(10)]
    el = time.perf_counter() - t0
    name1 = lib.ffvd_schedule_name(e._h).decode()
    n = int(lib.ffvd_stall_recoveries(e._h))
    hold = int(lib.ffvd_stall_hold(e._h))
    # run the hold out: the next call after it probes the one launch again (this build stalls again: the hold doubles)
    for _ in range(hold):
        e.nll_terms()
    assert int(lib.ffvd_stall_hold(e._h)) == 0
    e.nll_terms()
    n2 = int(lib.ffvd_stall_recoveries(e._h))
    hold2 = int(lib.ffvd_stall_hold(e._h))
print("BACKOFF", round(el, 2), n, hold, n2, hold2, len(set(vals)), repr(vals[0]), "|", name0, "|", name1)
"""


def test_a_stall_is_remembered():
    """a co-tenant that keeps compute units busy used to cost EVERY call the 1 s bounded wait before the
    recovery ran (FFVD_OK each time).  Now the handle stays on the schedule without inter-workgroup waits for 16 calls after a
    recovery, probes the one launch again, and doubles the hold when the probe stalls.  `tinystall` build (the one launch always
    stalls): ten consecutive calls take < 3 s in total (>= 10 s before), all return the golden nll, one recovery is counted, the
    schedule name says so; after the hold has run out the next call probes, stalls, and the hold is 32."""
    import subprocess
    import sys
    from ffvd_amd import build as fb
    lib_path = fb.build_variant("tinystall")
    env = dict(os.environ, FFVD_LIB=lib_path, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("FFVD_CHOL", None)
    env.pop("FFVD_NO_TINY", None)
    out = subprocess.run([sys.executable, "-c", _BACKOFF_SCRIPT], env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("BACKOFF")][0]
    f = line.split()
    assert float(f[1]) < 3.0, line
    assert int(f[2]) == 1 and int(f[3]) == 16 - 9, line           # one recovery; nine of the sixteen held calls used
    assert int(f[4]) == 2 and int(f[5]) == 32, line               # the probe stalled again: second recovery, doubled hold
    assert int(f[6]) == 1, line                                   # (recovered and held calls run the same multi-kernel schedule: same bits)
    gold = float(load_golden("small")["B_nll"])
    assert float(f[7]) == pytest.approx(gold, rel=1e-8)
    names = line.split("|")
    assert "one launch" in names[1] and "stall back-off" in names[2], line


def test_async_calls_with_alternating_output_buffers():
    """the one launch reads its argument block from device memory, uploaded by an asynchronous copy out of pinned
    host memory whenever the block changes.  Two back-to-back ffvd_elbo_async calls with DIFFERENT output buffers change it twice with
    no synchronisation in between; the first launch must still see ITS block (a ring of pinned slots guarded by events), i.e. both
    buffers end up holding the sums -- before the fix the first buffer could stay untouched."""
    torch = pytest.importorskip("torch")
    params, Y, c, meta = synthetic.make_named("small")
    with engine(meta) as e:
        assert single_launch(e)
        e.set_data(Y, c)
        want = e.elbo_sums(params)                       # synchronous call: the handle's own buffer
        up0 = int(e.lib.ffvd_debug_tiny_uploads(e._h, 0))
        bufs = [torch.full((8,), float("nan"), dtype=torch.float64, device="cuda:0") for _ in range(6)]
        torch.cuda.synchronize()
        for b in bufs:                                   # six different blocks through a ring of four slots, no sync in between
            e.elbo_async(b.data_ptr())
        e.sync()
        assert int(e.lib.ffvd_debug_tiny_uploads(e._h, 0)) == up0 + 6
        for b in bufs:
            np.testing.assert_array_equal(b.cpu().numpy()[:7], np.asarray(want)[:7])
        e.elbo_async(bufs[-1].data_ptr())                # unchanged block: nothing travels
        e.sync()
        assert int(e.lib.ffvd_debug_tiny_uploads(e._h, 0)) == up0 + 6


def test_one_launch_private_memory_and_bit_repeatability():
    """VERDICT r4 W5 / item 2.  (i) ffvd_create checks the private (scratch) memory the loaded one-launch kernel reports against what
    the file was validated with (a plan that needs every workgroup resident must know what a resident wave costs); (ii) the probe that
    caught round 4's corrupted argument block (tools/dbg_tiny.py), as a test: forward and forward + backward launches in turn, for
    chain counts that exercise both workgroup sizes and the no-side plan, must repeat bit for bit and agree with each other."""
    lib = None
    for S in (6, 10, 1, 3):
        params, Y, c, meta = synthetic.make_workload(T=512, D=4, C=1, M=100, S=S)
        with ElboEngine(512, 4, 1, 100, S, grad=True) as e:
            lib = e.lib
            assert single_launch(e)
            pb = int(lib.ffvd_debug_tiny_private_bytes(e._h))
            assert 0 < pb <= 1024, pb
            e.set_data(Y, c)
            e.set_params(params)
            f = e.nll_terms()["nll"]
            vals = []
            for _ in range(3):
                t, g = e.nll_and_grad(params)
                vals.append((t["nll"], float(np.abs(g["Z"]).sum()), float(np.abs(g["X"]).sum()), g["Z"].tobytes(), g["X"].tobytes()))
                assert e.nll_terms()["nll"] == f
            assert len(set(vals)) == 1 and vals[0][0] == f, (S, [v[:3] for v in vals])


# ---- explicit-U branch (cases 1 / 2 / 3 / 6 of FFVD_Main.py:273-324; dgp_model.py:289-297, regularizer :337-359) in ONE launch (round 5) -----
GRAD_KEYS_A = GRAD_KEYS + ("U",)
TERMS_A = ("nll_part_prior", "nll_log_likelihood", "x_t_prior_Q", "nll_reg_trace_inverse_Q_B", "nll")      # (no later_term1 / 2 in this branch: dgp_model.py:297)


def oracle_grad_a(params, Y, c, **kw):
    S = params["X"].shape[0]
    want = None
    for s in range(S):
        g = gorc.nll_grad_explicit_u(dict(params, X=params["X"][s]), Y, c, **kw)
        if want is None:
            want = {k: np.zeros_like(v) for k, v in g.items() if k != "X"}
            want["X"] = np.zeros_like(params["X"])
        for k in g:
            if k == "X":
                want["X"][s] = g["X"] / S
            else:
                want[k] += g[k] / S
    return want


@pytest.mark.parametrize("name", ["tiny", "small", "ragged"])
def test_explicit_u_forward_matches_golden_and_oracle(name):
    """Branch A through ONE launch: F = K_fu L^-T on the strips, mean = F u, var = sigma^2 - |F_t|^2 (conditionals_multi_output.py:33-48),
    the transition term on the residual x_{t+1} - x_t - mean (dgp_model.py:346-351), no H and no second factorisation; every named
    term against the golden vector and the oracle to 1e-10, bit-reproducible."""
    params, Y, c, meta = synthetic.make_named(name)
    g = load_golden(name)
    with engine(meta, U_collapse=False) as e:
        assert single_launch(e) in (4, 8)
        e.set_data(Y, c)
        got = e.nll_terms(params)
        again = e.nll_terms(params)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=False)
    for n in TERMS_A:
        assert got[n] == pytest.approx(float(g["A_" + n]), rel=1e-10, abs=1e-11), n
        assert got[n] == pytest.approx(ref[n], rel=1e-10, abs=1e-11), n
        assert got[n] == again[n], n
    np.testing.assert_allclose(got["nll_per_chain"], ref["nll_per_chain"], rtol=1e-10)


@pytest.mark.parametrize("S", [1, 10])
def test_explicit_u_actuator_config1(S):
    """BASELINE configs[0] with U_collapse = False (FFVD_Main.py cases 1 / 2 / 3 / 6 at the reference's own size): chain 0 carries the
    fixture's X, its nll is the golden value (SURVEY 8a anchor -2.2645993638...)."""
    params, Y, c, meta, z = actuator(S)
    with engine(meta, U_collapse=False) as e:
        assert single_launch(e) in (4, 8)
        e.set_data(Y, c)
        got = e.nll_terms(params)
    assert got["nll_per_chain"][0] == pytest.approx(float(load_golden("actuator")["A_nll"]), rel=1e-10)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=False)
    for n in TERMS_A:
        assert got[n] == pytest.approx(ref[n], rel=1e-10, abs=1e-11), n


@pytest.mark.parametrize("shape", [dict(T=64, D=1, C=1, M=16, S=1), dict(T=17, D=2, C=0, M=5, S=3), dict(T=512, D=4, C=1, M=128, S=2),
                                   dict(T=1000, D=3, C=2, M=113, S=2), dict(T=200, D=6, C=2, M=40, S=5), dict(T=130, D=2, C=1, M=97, S=7),
                                   dict(T=512, D=4, C=1, M=100, S=10), dict(T=512, D=4, C=1, M=100, S=6), dict(T=512, D=4, C=1, M=100, S=1)])
def test_explicit_u_shapes_forward_and_gradient(shape):
    """The edge shapes of the collapsed branch's test through the explicit-U launch: forward against the oracle, gradient -- dU and the
    Cholesky adjoint of K_uu included (dl/dK_uu = W Phi W^T, Phi from L^T tril(-W (dl/dW)^T W)) -- against the closed form of
    oracle/ffvd_grad_oracle.py (itself checked against torch autograd in tests/test_oracle.py)."""
    params, Y, c, meta = synthetic.make_workload(**shape)
    with engine(meta, grad=True, U_collapse=False) as e:
        assert single_launch(e) in (4, 8), shape
        e.set_data(Y, c)
        t, g = e.nll_and_grad(params)
        f = e.nll_terms(params)
        t2, g2 = e.nll_and_grad(params)
    ref = orc.nll_terms_chains(params, Y, c, U_collapse=False)
    for n in TERMS_A:
        assert t[n] == pytest.approx(ref[n], rel=1e-9, abs=1e-10), n
        assert f[n] == t[n], n
    want = oracle_grad_a(params, Y, c)
    for k in GRAD_KEYS_A:
        scale = float(np.max(np.abs(want[k]))) or 1.0
        tol = 2e-6 if k == "Z" else 1e-7
        assert np.max(np.abs(g[k] - want[k])) < tol * scale, (k, np.max(np.abs(g[k] - want[k])) / scale)
        assert np.array_equal(g[k], g2[k]), k                                  # launches repeat bit for bit


def test_explicit_u_equals_the_multi_kernel_schedule(monkeypatch):
    """FFVD_NO_TINY_A=1 (rounds 1-4: the explicit-U branch on 17 / 53 dependent launches) against the one launch: same nll, same
    gradient, and four device-resident Adam steps (all nine arrays, U included) end at the same parameters."""
    params, Y, c, meta = synthetic.make_named("small")
    out = {}
    for mode in ("one", "multi"):
        if mode == "multi":
            monkeypatch.setenv("FFVD_NO_TINY_A", "1")
        with engine(meta, grad=True, U_collapse=False) as e:
            assert (single_launch(e) != 0) == (mode == "one")
            e.set_data(Y, c)
            t, g = e.nll_and_grad(params)
            e.set_params(params)
            nlls = [e.adam_step(0.003)["nll"] for _ in range(4)]
            out[mode] = (t, g, e.get_params(), nlls)
    monkeypatch.delenv("FFVD_NO_TINY_A")
    assert out["one"][0]["nll"] == pytest.approx(out["multi"][0]["nll"], rel=1e-10)
    for k in GRAD_KEYS_A:
        a, b = out["one"][1][k], out["multi"][1][k]
        assert np.max(np.abs(a - b)) < (5e-6 if k == "Z" else 1e-7) * np.max(np.abs(b)), k
    for k in GRAD_KEYS_A:
        a, b = out["one"][2][k], out["multi"][2][k]
        assert np.max(np.abs(a - b)) < 1e-6 * max(1.0, np.max(np.abs(b))), k
    assert out["one"][3][-1] < out["one"][3][0]


def test_explicit_u_dim_shards():
    """Latent-dim shards of the explicit-U launch: sums and gradient shares (dU: each shard its own columns) add up to the single
    handle's."""
    params, Y, c, meta = synthetic.make_workload(T=150, D=4, C=1, M=50, S=3)
    with engine(meta, grad=True, U_collapse=False) as e:
        e.set_data(Y, c)
        whole_t, whole_g = e.nll_and_grad(params)
    sums = np.zeros(8)
    gsum = {k: 0.0 for k in GRAD_KEYS_A}
    for d0, dc, shared in ((0, 1, True), (1, 3, False)):
        with engine(meta, grad=True, U_collapse=False, d_begin=d0, d_count=dc, shared_terms=shared) as e:
            assert single_launch(e)
            e.set_data(Y, c)
            t, g = e.nll_and_grad(params)
            sums += t["sums8"]
            for k in GRAD_KEYS_A:
                gsum[k] = gsum[k] + g[k]
    assert sums[6] / sums[7] == pytest.approx(whole_t["nll"], rel=1e-11)
    for k in GRAD_KEYS_A:
        assert np.max(np.abs(gsum[k] - whole_g[k])) < 1e-9 * max(1e-30, np.max(np.abs(whole_g[k]))), k
