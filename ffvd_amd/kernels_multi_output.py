"""vfegpssm/kernels_multi_output.py counterpart: the SE/ARD kernel used one-per-latent-dim."""
from .kernels import Kernel, SquaredExponential  # noqa: F401
