"""CPU: the C-ABI library loads, exports every symbol include/ffvd_abi.h declares, and rejects bad
configurations without touching a GPU.  (No compute calls here.)"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from ffvd_amd import _lib, build


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ffvd_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ffvd_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    build.build()
    assert os.path.exists(_lib.LIB_PATH)
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, "ffvd_amd")


def test_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in ffvd_abi.h but not exported"
    assert set(_lib.exported_symbols()) == set(syms), "ctypes signature table and header disagree"


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.FfvdConfig) == 20 * 4 + 8
    assert C.sizeof(_lib.FfvdParams) == 9 * C.sizeof(C.c_void_p)


def test_create_rejects_bad_config_without_gpu():
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.FfvdConfig(T=0, D=4, C=1, M=16, S_local=1, Ydim=1, jitter=1e-5)
    assert lib.ffvd_create(C.byref(cfg), C.byref(h)) == _lib.FFVD_EINVAL
    assert b"bad shape" in lib.ffvd_last_error(None)
    cfg = _lib.FfvdConfig(T=8, D=30, C=5, M=16, S_local=1, Ydim=1, jitter=1e-5)
    assert lib.ffvd_create(C.byref(cfg), C.byref(h)) == _lib.FFVD_EINVAL
    assert b"exceeds" in lib.ffvd_last_error(None)
    cfg = _lib.FfvdConfig(T=8, D=4, C=1, M=16, S_local=1, Ydim=1, d_begin=3, d_count=2, jitter=1e-5)
    assert lib.ffvd_create(C.byref(cfg), C.byref(h)) == _lib.FFVD_EINVAL
    cfg = _lib.FfvdConfig(T=8, D=4, C=1, M=16, S_local=1, Ydim=1, prior_type=7, jitter=1e-5)
    assert lib.ffvd_create(C.byref(cfg), C.byref(h)) == _lib.FFVD_EINVAL
    assert lib.ffvd_create(None, C.byref(h)) == _lib.FFVD_EINVAL
    assert not h.value


def test_error_mapping():
    with pytest.raises(ValueError):
        from ffvd_amd.engine import ElboEngine
        ElboEngine(T=0, D=4, C=1, M=16, S=1)
    with pytest.raises(ValueError):
        from ffvd_amd.engine import ElboEngine
        ElboEngine(T=8, D=4, C=1, M=16, S=1, prior_type="strauss")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libffvd_hip.so"))
    with pytest.raises(_lib.FfvdError, match="no CPU fallback"):
        _lib.load()


def test_op_argument_validation_without_gpu():
    lib = _lib.load()
    out = np.zeros(4)
    x = np.zeros((2, 3))
    # SE kernel without lengthscales is a usage error and must be rejected before any device work
    rc = lib.ffvd_op_kernel_matrix(0, _lib.dptr(x), 2, None, 2, 3, 0.0, None, 0.0, _lib.dptr(out))
    assert rc == _lib.FFVD_EINVAL


def _compile_c_demo(tmp_path):
    import subprocess
    build.build()                              # no-op when the library is already there
    exe = str(tmp_path / "abi_demo")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "abi_demo.c"), "-o", exe, "-L" + os.path.join(ROOT, "ffvd_amd"), "-lffvd_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "ffvd_amd"), "-lm"]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    return exe


def test_header_is_plain_c99_and_a_c_host_links(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 with warnings as errors, and a C program that includes
    nothing else must link against the shared library (examples/abi_demo.c)."""
    import subprocess
    proc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "ffvd_abi.h")], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    _compile_c_demo(tmp_path)


@pytest.mark.gpu
def test_c_host_runs(tmp_path):
    """examples/abi_demo.c end to end on the GPU: both routes of the collapsed bound from a C main()."""
    import subprocess
    exe = _compile_c_demo(tmp_path)
    proc = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    lines = [l.split() for l in proc.stdout.splitlines() if l.startswith("route")]
    assert len(lines) == 2 and lines[0][1] == "0" and lines[1][1] == "1"
    nll_ref, nll_gram = float(lines[0][3]), float(lines[1][3])
    assert np.isfinite(nll_ref) and nll_gram == pytest.approx(nll_ref, rel=1e-8)
    assert lines[0][4:6] == ["(chains", "2)"]


SANITIZER_SCRIPT = r'''
import ctypes as C, sys
import numpy as np
from ffvd_amd import _lib
lib = _lib.load()
assert _lib.LIB_PATH.endswith("libffvd_hip_asan.so")
E = _lib.FFVD_EINVAL
h = C.c_void_p()
# ffvd_create: every rejection path that needs no device, plus the no-device path itself
bad = [dict(T=0), dict(D=0), dict(M=0), dict(S_local=0), dict(Ydim=0), dict(C=-1), dict(D=30, C=5), dict(dtype=7),
       dict(kernel_kind=9), dict(branch=5), dict(prior_type=7), dict(d_begin=3, d_count=2), dict(jitter=-1.0),
       dict(route=3), dict(grad=1, kernel_kind=1, branch=1, route=0, dtype=1), dict(route=1, branch=0), dict(dtype=1, route=1, branch=1),
       dict(dtype=1, branch=0)]
base = dict(T=8, D=4, C=1, M=16, S_local=1, Ydim=1, jitter=1e-5, branch=1, prior_type=1)
for ov in bad:
    cfg = _lib.FfvdConfig(**dict(base, **ov))
    assert lib.ffvd_create(C.byref(cfg), C.byref(h)) == E, ov
    assert lib.ffvd_last_error(None), ov
assert lib.ffvd_create(None, C.byref(h)) == E and lib.ffvd_create(C.byref(_lib.FfvdConfig(**base)), None) == E
rc = lib.ffvd_create(C.byref(_lib.FfvdConfig(**base)), C.byref(h))
assert rc == _lib.FFVD_EDEVICE and not h.value, rc            # this box has no GPU: the handle must not leak out
# handle-taking calls with a null handle
out8 = np.zeros(8); nll = C.c_double(); ms = C.c_float(); n8 = np.zeros(8, dtype=np.int32)
p = _lib.FfvdParams(); g = _lib.FfvdGrads()
assert lib.ffvd_destroy(None) == 0
for rc in (lib.ffvd_sync(None), lib.ffvd_set_data(None, None, None, 0), lib.ffvd_set_params(None, C.byref(p), 0),
           lib.ffvd_elbo(None, None, 0, _lib.dptr(out8), C.byref(nll)), lib.ffvd_elbo_async(None, None),
           lib.ffvd_elbo_grad(None, None, 0, 1, _lib.dptr(out8), C.byref(nll), C.byref(g)),
           lib.ffvd_adam_step(None, 0.1, 0.9, 0.999, 1e-8, 511, _lib.dptr(out8), C.byref(nll)),
           lib.ffvd_sghmc_step(None, 0.01, 0.05, 2, 1, None, _lib.dptr(out8), C.byref(nll)),
           lib.ffvd_optimizer_reset(None), lib.ffvd_get_params(None, None), lib.ffvd_update_params(None, None),
           lib.ffvd_chain_nll(None, None), lib.ffvd_time_elbo(None, 1, C.byref(ms)), lib.ffvd_profile_stages(None, None),
           lib.ffvd_stage_timing(None, 1), lib.ffvd_stage_times(None, _lib.dptr(out8), n8.ctypes.data_as(C.POINTER(C.c_int32))),
           lib.ffvd_comm_unique_id(None), lib.ffvd_comm_init(None, 1, 0, None), lib.ffvd_comm_destroy(None),
           lib.ffvd_elbo_allreduce(None, None, _lib.dptr(out8), C.byref(nll)), lib.ffvd_elbo_allreduce_async(None, None, None),
           lib.ffvd_allreduce_sum_async(None, None, None, 8), lib.ffvd_allreduce_sum(None, None, None, 8),
           lib.ffvd_adam_step_allreduce(None, None, 4, 0.1, 0.9, 0.999, 1e-8, 511, _lib.dptr(out8), C.byref(nll)),
           lib.ffvd_sghmc_step_allreduce(None, None, 4, 0.01, 0.05, 2, 1, None, _lib.dptr(out8), C.byref(nll)),
           lib.ffvd_train_local(None, 4), lib.ffvd_train_exchange_get(None, None), lib.ffvd_train_exchange_set(None, None),
           lib.ffvd_adam_apply(None, 0.1, 0.9, 0.999, 1e-8, 511, _lib.dptr(out8), C.byref(nll)),
           lib.ffvd_sghmc_apply(None, 0.01, 0.05, 2, 1, None, _lib.dptr(out8), C.byref(nll))):
    assert rc == E, rc
assert lib.ffvd_get_stream(None) is None and lib.ffvd_comm_get(None) is None and lib.ffvd_workspace_bytes(None) == 0
assert lib.ffvd_train_exchange_count(None) == 0 and lib.ffvd_train_exchange_ptr(None) is None
# operator entry points: argument validation happens before any device work
x = np.zeros((2, 3)); o = np.zeros(4); i32 = np.zeros(2, dtype=np.int32)
dp = _lib.dptr
assert lib.ffvd_op_kernel_matrix(0, dp(x), 2, None, 2, 3, 0.0, None, 0.0, dp(o)) == E         # SE without lengthscales
assert lib.ffvd_op_kernel_matrix(5, dp(x), 2, None, 2, 3, 0.0, dp(x), 0.0, dp(o)) == E
assert lib.ffvd_op_kernel_diag(0, None, 2, 3, 0.0, dp(o)) == E
assert lib.ffvd_op_cholesky(None, 2, 1, dp(o), i32.ctypes.data_as(C.POINTER(C.c_int32))) == E
assert lib.ffvd_op_cholesky(dp(o), 0, 1, dp(o), None) == E
assert lib.ffvd_op_trsm(None, 2, None, 1, None) == E
assert lib.ffvd_op_kernel_pre_cal(0, None, 2, 3, 1, dp(o), dp(o), 1e-5, dp(o)) == E
assert lib.ffvd_op_collapse(0, None, None, None, None, 4, 2, 3, 1, None, None, None, 4.0, 4.0, dp(o)) == E
assert lib.ffvd_op_conditional(0, None, 2, None, 2, 3, 1, None, None, None, 1e-5, None, None) == E
assert lib.ffvd_op_collapse_u_mean(0, None, None, None, None, 4, 2, 3, 1, None, None, None, None, None) == E
assert lib.ffvd_op_conditional_precalc(0, None, None, 2, None, 2, 3, 1, None, None, None, None, None, None) == E
assert lib.ffvd_op_predict_mean(None, 2, 3, None, None, 1, None) == E
assert lib.ffvd_op_logdensity_norm_diag(0, None, None, None, 2, 1, None) == E
assert lib.ffvd_op_get_rand(None, None, None, 4, None) == E
assert lib.ffvd_op_adam_step(None, None, None, None, 4, 0.1, 0.9, 0.999, 1e-8, 1) == E
assert lib.ffvd_op_sghmc_step(None, None, None, None, None, None, None, 4, 0.01, 0.05, 5.0, 1) == E
assert lib.ffvd_op_rollout(0, None, None, 2, 3, 2, None, None, None, None, None, 1, None, 1, 1, None, None, None, None) == E
assert lib.ffvd_op_pg_sweep(0, None, None, 2, 3, 2, None, None, None, None, 3, None, 1, None, 1, None, None, None, None, 2,
                            None, None, None, None, None) == E
print("SANITIZER-OK")
'''


def test_host_sanitizer_build_of_the_abi(tmp_path):
    """SURVEY section 5 / VERDICT r1 item 10: the C-ABI's host code (argument validation, handle lifetime, error paths)
    built with -fsanitize=address,undefined and driven through every entry point that can be reached without a GPU.
    CPU box only; the process fails on the first sanitizer report."""
    import subprocess
    import sys
    lib = build.build_asan()
    script = tmp_path / "san.py"
    script.write_text(SANITIZER_SCRIPT)
    env = dict(os.environ, FFVD_LIB=lib, LD_PRELOAD=build.asan_runtime(), PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    proc = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "SANITIZER-OK" in proc.stdout
    assert "AddressSanitizer" not in proc.stderr and "runtime error" not in proc.stderr, proc.stderr[-3000:]
