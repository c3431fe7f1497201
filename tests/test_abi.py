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
    assert C.sizeof(_lib.FfvdConfig) == 18 * 4 + 8
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
