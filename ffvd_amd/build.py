"""Build libffvd_hip.so (gfx950) in-tree with hipcc.  `python -m ffvd_amd.build [--force]`.

The library is rebuilt whenever the SHA-256 of its sources, headers and flags differs from the one recorded next to
it (`libffvd_hip.so.hash`) -- not by mtime, so a fresh checkout, or a snapshot whose files were re-stamped, really
compiles.  Each source is compiled to an object in its own hipcc process (they run side by side), then linked."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libffvd_hip.so")
HASH = LIB + ".hash"
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["kernels.hip", "kernels_f32.hip", "grad.hip", "optim.hip", "tiny.hip", "loops.hip", "abi.hip"]
HEADERS = ["kernels.h", "dev_common.h", "step_bodies.h", "tiny.h", "kernels_f32.h", "grad.h", "optim.h", os.path.join("..", "..", "include", "ffvd_abi.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value"]
LINK = ["-shared", "-fPIC", "--offload-arch=gfx950", "-ldl"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the ROCm toolchain is required to build libffvd_hip.so")


def source_hash():
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + LINK).encode())
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(HASH):
        return True
    with open(HASH) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into ffvd_amd/libffvd_hip.so; returns the path."""
    if not force and not needs_build():
        if verbose:
            print(f"{LIB}: up to date (source hash {source_hash()[:16]})", flush=True)
        return LIB
    cc = hipcc_path()
    os.makedirs(OBJDIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJDIR, src + ".o")
        cmd = [cc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + proc.stdout + proc.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [cc] + objs + LINK + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + proc.stdout + proc.stderr)
    os.replace(LIB + ".tmp", LIB)
    with open(HASH, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


ASAN_LIB = os.path.join(HERE, "libffvd_hip_asan.so")
ASAN_FLAGS = ["-O1", "-g", "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer",
              "-Xarch_host", "-fno-sanitize-recover=undefined"]


def asan_runtime():
    """Path of the AddressSanitizer runtime that must be LD_PRELOADed into an uninstrumented host (python)."""
    out = subprocess.run([hipcc_path(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = out.stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        raise RuntimeError("libclang_rt.asan-x86_64.so not found next to hipcc's clang")
    return path


def build_asan(verbose=False):
    """Host-side sanitizer build: the SAME sources with AddressSanitizer + UBSan on the HOST code only
    (handle lifetime, argument validation, staging buffers, error paths of the C ABI); device code is compiled as usual
    (GPU ASan is not available on this pool).  CPU-box job: tests/test_abi.py runs the no-GPU ABI calls against it."""
    cc = hipcc_path()
    flags = [f for f in FLAGS if f != "-O3"] + ASAN_FLAGS
    os.makedirs(OBJDIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJDIR, src + ".asan.o")
        proc = subprocess.run([cc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc (asan) failed on {src}:\n" + proc.stdout + proc.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [cc] + objs + LINK + ["-fsanitize=address,undefined", "-shared-libsan", "-o", ASAN_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc (asan) link failed:\n" + proc.stdout + proc.stderr)
    return ASAN_LIB


def variant_path(name):
    return os.path.join(HERE, f"libffvd_hip_{name}.so")


VARIANTS = {
    # the dataflow Cholesky with matrix 0 never announcing its first diagonal block: every other block row of that matrix
    # must give up after the bounded wait instead of hanging (tests/test_gpu_ops.py)
    "dfstall": ["-DFFVD_DF_TEST_STALL"],
    # wall-clock stamps inside the dataflow Cholesky (tools/df_trace.py)
    "dftrace": ["-DFFVD_DF_TRACE"],
    # A/B builds (tools/ab.sh) of the round-3 Gram schedule: without the tail split / without combos and tail split (= round 2)
    "notail": ["-DGRAM_TAIL_SPLIT=0"],
    "r2gram": ["-DGRAM_COMBO=0", "-DGRAM_TAIL_SPLIT=0"],
    # ... and round 3's Gram schedule (three 64 x 32-sub-block combos per four diagonal tiles) against round 4's pair combos
    "r3gram": ["-DGRAM_COMBO=1"],
    # the 64-pivot diagonal factor of the dataflow Cholesky by wavefront 0 alone (round 3) against all four wavefronts (round 4)
    "factor1w": ["-DDF_FACTOR_4W=0"],
    "factortiles": ["-DDF_FACTOR_ROWS=0"],
    "dftrace_factortiles": ["-DFFVD_DF_TRACE", "-DDF_FACTOR_ROWS=0"],
    "offdiagglds": ["-DGRAM_GLDS_OFFDIAG=1"],       # off-diagonal Gram tiles staged by LDS-DMA like the pair combos (measured neutral)
    "dftrace_notail": ["-DFFVD_DF_TRACE", "-DGRAM_TAIL_SPLIT=0"],
    # tiny.hip (the one-launch iteration): wall-clock stamps of every workgroup's phases (tools/tiny_trace.py); a build whose
    # unit-0 head never publishes W, so that every bounded wait of that launch must give up (tests/test_gpu_tiny.py)
    "tinytrace": ("tiny.hip", ["-DFFVD_TINY_TRACE"]),
    "tinystall": ("tiny.hip", ["-DFFVD_TINY_TEST_STALL"]),
    # A/B build: the head does not take row blocks of the K_uu side (tools/dbg_cmp.py diffs the scratch block of two builds)
    "tinynohelp": ("tiny.hip", ["-DFFVD_TINY_NO_HEAD_HELP"]),
    # A/B build: the blocked Cholesky of the one-launch iteration with tile solves behind every pivot chain (round 4, first form)
    "tinytiles": ("tiny.hip", ["-DTINY_CHOL_ROWS=0"]),
    # the resident rollout loop with release / acquire fences at its hand-offs (inside the HIP memory model; tests compare both forms)
    "rrfenced": ("loops.hip", ["-DFFVD_RR_FENCED"]),
    # wall-clock stamps of every workgroup of the skinny product of a rollout / particle-Gibbs step (tools/step_trace.py)
    # wall-clock stamps of every workgroup of the fused backward product (tools/bwd_trace.py)
    "bwdtrace": ("grad.hip", ["-DFFVD_BWD_TRACE"]),
    "bwdtrace_base": ("grad.hip", ["-DFFVD_BWD_TRACE", "-DBWD_EPI_XWF=0"]),
    "bwdtrace_vsf": ("grad.hip", ["-DFFVD_BWD_TRACE", "-DBWD_EPI_VSF=1"]),
    "steptrace": ["-DFFVD_STEP_TRACE"], "steptrace4": ["-DFFVD_STEP_TRACE", "-DFFVD_SKINNY_CHUNK=4"],
}


def build_variant(name, verbose=False):
    """Test / diagnostic build: kernels.hip recompiled with the variant's defines, every other object shared with the
    product library (which is built first).  Load it with FFVD_LIB=<path>."""
    build(force=False, verbose=verbose)
    out = variant_path(name)
    stamp = out + ".hash"
    spec = VARIANTS[name]
    source, defines = spec if isinstance(spec, tuple) else ("kernels.hip", spec)
    want = source_hash() + " " + source + " " + " ".join(defines)
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return out
    cc = hipcc_path()
    obj = os.path.join(OBJDIR, f"{source}.{name}.o")
    cmd = [cc] + FLAGS + defines + ["-c", os.path.join(CSRC, source), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed on {source} ({name}):\n" + proc.stdout + proc.stderr)
    objs = [obj] + [os.path.join(OBJDIR, src + ".o") for src in SOURCES if src != source]
    proc = subprocess.run([cc] + objs + LINK + ["-o", out], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc link failed ({name}):\n" + proc.stdout + proc.stderr)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return out


if __name__ == "__main__":
    for v in VARIANTS:
        if "--" + v in sys.argv:
            print(build_variant(v, verbose=True))
            sys.exit(0)
    if "--asan" in sys.argv:
        print(build_asan(verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
