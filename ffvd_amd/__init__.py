"""ffvd_amd: MI355X-native (gfx950) evaluation of FFVD's per-iteration ELBO hot path.

Host code is plain Python over a ctypes C ABI (include/ffvd_abi.h -> ffvd_amd/libffvd_hip.so); the kernels
are hand-written HIP.  Importing the package does not load the library; the first operator call does, and it
raises if the library has not been built (no CPU fallback).
"""
__all__ = ["engine", "kernels", "kernels_multi_output", "conditionals", "conditionals_multi_output",
           "likelihoods", "dgp_model", "models", "synthetic", "distributed"]
