"""bench.py's contract with the driver: defaults, ONE JSON line on stdout with the agreed keys, the self-spawning --gpus N form."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_defaults_without_flags():
    import bench
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.workload, a.dtype, a.route) == (1, 60, 10, "c2", "f64", "gram")
    c4 = bench.parse_args(["--workload", "c4"])
    assert (c4.dtype, c4.route, c4.steps, c4.warmup) == ("f32c", "reference", 5, 1)      # BASELINE configs[3]: fp32 contractions
    assert bench.parse_args(["--workload", "c5"]).route == "reference"
    assert bench.PEAK_TFLOPS == {"f64": 78.6, "f32c": 157.3}


def test_parallelism_text_says_what_the_step_does():
    """with one rank and no communicator the line must not claim an ncclAllReduce."""
    import bench
    assert "no collective" in bench.parallelism_text(1, "rccl", False) and "ncclAllReduce" not in bench.parallelism_text(1, "rccl", False)
    t1 = bench.parallelism_text(1, "rccl", True)
    assert "ncclAllReduce" in t1 and "1-rank communicator" in t1
    t8 = bench.parallelism_text(8, "rccl", True)
    assert "ncclAllReduce" in t8 and "1-rank" not in t8
    assert "gloo" in bench.parallelism_text(2, "torch", True)


REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("FFVD_LIB", None)
    if env:
        e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must be exactly one JSON line, got: %r" % lines[:3]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_one_json_line_with_the_agreed_keys():
    d = _run(["--steps", "4", "--warmup", "1"])
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"].startswith("ELBO iters/sec (T=4096, M=512, x_dim=4, S=32)") and d["unit"] == "iters/sec"
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-9                        # value = 1 / time per step
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["peak"] == 78.6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic_source"].startswith("profiles/traffic.json")
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0.0
    # the N = 1 line times the sequence N > 1 times: the 1-rank ncclAllReduce is inside the step (or the line says it is not)
    par = d["config"]["parallelism"]
    if "ncclAllReduce" in par:
        assert "1-rank communicator" in par and d["ms_per_step_without_collective"] > 0.0
    else:
        assert "no collective" in par


def test_c1_workload_is_the_actuator_fixture():
    """--workload c1 = BASELINE configs[0]: the actuator fixture with S = 10 chains, chain 0 the fixture's own trajectory (its nll is
    the golden value), the others perturbed with the seeded generator of SURVEY 8(d)."""
    import numpy as np
    import bench
    from oracle import ffvd_oracle as orc
    assert bench.parse_args(["--workload", "c1"]).workload == "c1"
    params, Y, c, meta = bench.load_c1()
    assert (meta["T"], meta["M"], meta["D"], meta["C"], meta["S"], meta["P"]) == (512, 100, 4, 1, 10, 5)
    assert params["X"].shape == (10, 513, 4) and Y.shape == (512, 1)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_actuator.npz"))
    got = orc.nll_terms(dict(params, X=params["X"][0]), Y, c, U_collapse=True)["nll"]
    assert got == pytest.approx(float(gold["B_nll"]), rel=1e-12)
    assert np.max(np.abs(params["X"][1] - params["X"][0])) > 0.1       # the other chains are draws, not copies
    p2 = bench.load_c1()[0]
    np.testing.assert_array_equal(p2["X"], params["X"])                 # seeded


@pytest.mark.gpu
def test_c1_line_has_gpu_and_cpu_side_by_side():
    """VERDICT r3 Missing 2: configs[0] -- the one configuration a CPU can run in milliseconds -- with the CPU baseline of ALL chains
    beside it, forward and training step, the whole iteration in one launch."""
    d = _run(["--workload", "c1", "--steps", "20", "--warmup", "5"])
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"].startswith("ELBO iters/sec (T=512, M=100, x_dim=4, S=10)")
    assert "actuator" in d["config"]["workload"] or "c1" in d["config"]["workload"]
    assert "one launch" in d["config"]["route"]
    assert d["train_ms_per_step"] > 0.0 and d["train_single_launch"] in (4, 8)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0.0 and "10 of 10 chains" in cb["sample"]
    assert cb["train_seconds_per_step"] > 0.0
    assert 0.0 < d["roofline"]["frac"] < 1.0
    assert d["nll"] == pytest.approx(d["nll"])          # finite


@pytest.mark.gpu
def test_plain_invocation_with_two_gpus_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the parent spawns the ranks itself.  On the one-GPU test box both ranks share
    device 0 and the eight sums travel over gloo (FFVD_BENCH_REHEARSAL=1); the JSON line says which exchange was timed."""
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env={"FFVD_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["chains_per_gpu"] == 16
    assert "gloo" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d
