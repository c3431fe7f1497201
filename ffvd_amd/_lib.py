"""ctypes binding of libffvd_hip.so (include/ffvd_abi.h).  Fails loudly when the library is missing:
there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# FFVD_LIB: load another build of the same ABI instead (the host-sanitizer build of ffvd_amd/build.py --asan)
LIB_PATH = os.environ.get("FFVD_LIB") or os.path.join(HERE, "libffvd_hip.so")

FFVD_OK, FFVD_EINVAL, FFVD_ENOMEM, FFVD_EDEVICE, FFVD_ENOTPD = 0, -1, -2, -3, 1
KERNEL_KIND = {"SquaredExponential": 0, "LinearK": 1}
BRANCH_A, BRANCH_B = 0, 1
PRIOR_TYPE = {"uniform": 0, "normal": 1}
ROUTE = {"reference": 0, "gram": 1}
DTYPE = {"f64": 0, "f32c": 1}
PARAMS_ON_DEVICE = 1
TERM_NAMES = ("nll_part_prior", "nll_log_likelihood", "x_t_prior_Q", "nll_reg_trace_inverse_Q_B",
              "later_term1", "later_term2", "nll")


class FfvdConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "T", "D", "C", "M", "S_local", "Ydim", "d_begin", "d_count", "shared_terms", "dtype",
        "kernel_kind", "branch", "prior_type", "device_id", "chains_per_pass", "route", "grad", "T_total", "t_begin",
        "reserved")] + [("jitter", C.c_double)]


class FfvdParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")]


class FfvdGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "X", "Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols", "U")]


_dp = C.POINTER(C.c_double)
_SIGNATURES = {
    "ffvd_create": (C.c_int, [C.POINTER(FfvdConfig), C.POINTER(C.c_void_p)]),
    "ffvd_destroy": (C.c_int, [C.c_void_p]),
    "ffvd_last_error": (C.c_char_p, [C.c_void_p]),
    "ffvd_sync": (C.c_int, [C.c_void_p]),
    "ffvd_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "ffvd_set_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ffvd_set_params": (C.c_int, [C.c_void_p, C.POINTER(FfvdParams), C.c_int]),
    "ffvd_elbo": (C.c_int, [C.c_void_p, C.POINTER(FfvdParams), C.c_uint32, _dp, _dp]),
    "ffvd_elbo_grad": (C.c_int, [C.c_void_p, C.POINTER(FfvdParams), C.c_uint32, C.c_int, _dp, _dp, C.POINTER(FfvdGrads)]),
    "ffvd_elbo_async": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ffvd_chain_nll": (C.c_int, [C.c_void_p, _dp]),
    "ffvd_time_elbo": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "ffvd_profile_stages": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ffvd_stage_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ffvd_stage_times": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int32)]),
    "ffvd_op_kernel_matrix": (C.c_int, [C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_double, _dp, C.c_double, _dp]),
    "ffvd_op_kernel_diag": (C.c_int, [C.c_int, _dp, C.c_int, C.c_int, C.c_double, _dp]),
    "ffvd_op_cholesky": (C.c_int, [_dp, C.c_int, C.c_int, _dp, C.POINTER(C.c_int32)]),
    "ffvd_op_trsm": (C.c_int, [_dp, C.c_int, _dp, C.c_int, _dp]),
    "ffvd_op_kernel_pre_cal": (C.c_int, [C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_double, _dp]),
    "ffvd_op_collapse": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                   C.c_double, C.c_double, _dp]),
    "ffvd_op_collapse_u_mean": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                          _dp, _dp]),
    "ffvd_op_conditional_precalc": (C.c_int, [C.c_int, _dp, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                              _dp, _dp, _dp]),
    "ffvd_op_predict_mean": (C.c_int, [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]),
    "ffvd_op_logdensity_norm_diag": (C.c_int, [C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp]),
    "ffvd_op_get_rand": (C.c_int, [_dp, _dp, _dp, C.c_int64, _dp]),
    "ffvd_adam_step": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, _dp,
                                 C.POINTER(C.c_double)]),
    "ffvd_adam_step_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.c_uint32, _dp, C.POINTER(C.c_double)]),
    "ffvd_sghmc_step_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_int,
                                            C.c_void_p, _dp, C.POINTER(C.c_double)]),
    "ffvd_train_local": (C.c_int, [C.c_void_p, C.c_int]),
    "ffvd_train_exchange_count": (C.c_int64, [C.c_void_p]),
    "ffvd_train_exchange_ptr": (C.c_void_p, [C.c_void_p]),
    "ffvd_train_exchange_get": (C.c_int, [C.c_void_p, _dp]),
    "ffvd_train_exchange_set": (C.c_int, [C.c_void_p, _dp]),
    "ffvd_adam_apply": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, _dp,
                                  C.POINTER(C.c_double)]),
    "ffvd_tshard_adam_apply": (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, _dp,
                                         C.POINTER(C.c_double)]),
    "ffvd_tshard_sghmc_apply": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_uint32, C.c_int, C.c_void_p, _dp,
                                          C.POINTER(C.c_double)]),
    "ffvd_sghmc_apply": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_uint32, C.c_int, C.c_void_p, _dp,
                                   C.POINTER(C.c_double)]),
    "ffvd_stall_recoveries": (C.c_int, [C.c_void_p]),
    "ffvd_stall_hold": (C.c_int, [C.c_void_p]),
    "ffvd_op_release_cache": (C.c_int, []),
    "ffvd_op_rollout_fallbacks": (C.c_int, []),
    "ffvd_single_launch": (C.c_int, [C.c_void_p]),
    "ffvd_schedule_name": (C.c_char_p, [C.c_void_p]),
    "ffvd_get_stream": (C.c_void_p, [C.c_void_p]),
    "ffvd_optimizer_reset": (C.c_int, [C.c_void_p]),
    "ffvd_update_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ffvd_sghmc_step": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_uint32, C.c_int, C.c_void_p, _dp,
                                  C.POINTER(C.c_double)]),
    "ffvd_get_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ffvd_op_adam_step": (C.c_int, [_dp, _dp, _dp, _dp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                    C.c_int64]),
    "ffvd_op_sghmc_step": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int64, C.c_double, C.c_double, C.c_double,
                                     C.c_int]),
    "ffvd_op_rollout": (C.c_int, [C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp,
                                  C.c_int, C.c_int, _dp, _dp, _dp, _dp]),
    "ffvd_op_pg_sweep": (C.c_int, [C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_int,
                                   _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_void_p]),
    "ffvd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ffvd_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ffvd_comm_destroy": (C.c_int, [C.c_void_p]),
    "ffvd_comm_get": (C.c_void_p, [C.c_void_p]),
    "ffvd_elbo_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, _dp, _dp]),
    "ffvd_elbo_allreduce_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ffvd_allreduce_sum_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "ffvd_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, _dp, C.c_int64]),
    "ffvd_elbo_tshard": (C.c_int, [C.c_void_p, C.c_void_p, _dp, _dp]),
    "ffvd_tshard_local": (C.c_int, [C.c_void_p]),
    "ffvd_tshard_count": (C.c_int64, [C.c_void_p]),
    "ffvd_tshard_get": (C.c_int, [C.c_void_p, _dp]),
    "ffvd_tshard_set": (C.c_int, [C.c_void_p, _dp]),
    "ffvd_tshard_finish": (C.c_int, [C.c_void_p, _dp, _dp]),
    "ffvd_tshard_finish_grad": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    "ffvd_tshard_grad_fetch": (C.c_int, [C.c_void_p, _dp, C.POINTER(FfvdGrads)]),
    "ffvd_elbo_tshard_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _dp, _dp, C.POINTER(FfvdGrads)]),
    "ffvd_op_conditional": (C.c_int, [C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                      C.c_double, _dp, _dp]),
}

TRAIN_BITS = {"X": 1, "Z": 2, "logvariance": 4, "loglengthscales": 8, "log_Q": 16, "CC": 32, "DD": 64,
              "log_Rchols": 128, "U": 256}
TRAIN_ALL = 511

_lib = None


class FfvdError(RuntimeError):
    pass


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch's ROCm wheel bundles its own libamdhip64.so.7 (same SONAME as
    /opt/rocm's); whichever copy is mapped first serves both libraries.  If libffvd_hip.so pulled in the system copy
    first, a later `import torch` (RCCL all-reduce of the partial sums, torch CUDA tensors as output buffers) ends up
    on a runtime it was not built for and reports "no GPUs found".  Mapping torch's copy first -- without importing
    torch -- makes the order of imports irrelevant."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load():
    """Load libffvd_hip.so (once).  Raises if it has not been built: `python -m ffvd_amd.build`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FfvdError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (run `python -m ffvd_amd.build` "
            "or `__graft_entry__.build()`); ffvd_amd has no CPU fallback")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI/header mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _DEBUG_SIGNATURES.items():      # not part of the ABI (absent from ffvd_abi.h): bound when present
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


# diagnostic exports of the library used by tools/ and tests/ (not declared in include/ffvd_abi.h)
_DEBUG_SIGNATURES = {
    "ffvd_debug_tiny_private_bytes": (C.c_int64, [C.c_void_p]),
    "ffvd_debug_tiny_uploads": (C.c_int, [C.c_void_p, C.c_int]),
    "ffvd_debug_enqueue_us": (C.c_double, [C.c_void_p]),
}


def exported_symbols():
    return tuple(_SIGNATURES)


def as_f64(a, shape=None, name="array"):
    """Contiguous fp64 view/copy; validates the shape when given."""
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None and tuple(arr.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(arr.shape)}")
    return arr


def dptr(arr):
    return arr.ctypes.data_as(_dp)


def check(rc, handle=None, what="ffvd"):
    if rc == FFVD_OK:
        return
    msg = load().ffvd_last_error(handle)
    msg = msg.decode() if msg else ""
    if rc == FFVD_ENOTPD:
        raise np.linalg.LinAlgError(f"{what}: {msg}")
    if rc == FFVD_EINVAL:
        raise ValueError(f"{what}: {msg}")
    if rc == FFVD_ENOMEM:
        raise MemoryError(f"{what}: {msg}")
    raise FfvdError(f"{what}: {msg} (status {rc})")
