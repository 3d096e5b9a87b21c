"""vbq_amd -- MI355X-native hot path of Variational Bayesian Quantization.

A Python host over a C-ABI HIP extension (include/vbq.h, vbq_amd/csrc).  The package has no
CPU implementation of its kernels: on a machine without the built extension or without a
ROCm device the ops raise.
"""
from ._lib import VBQError, lib, library_path  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["VBQError", "lib", "library_path", "ops"]
