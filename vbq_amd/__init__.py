"""vbq_amd -- MI355X-native hot path of Variational Bayesian Quantization.

A Python host over a C-ABI HIP extension (include/vbq.h, vbq_amd/csrc).  The package has no
CPU implementation of its kernels: on a machine without the built extension or without a
ROCm device the ops raise.

    vbq_amd.quantize(mu, sigma, lmbda, table=...)          one-call surface (api.py)
    vbq_amd.ChannelwisePriorCDFQuantizer                   img-compression/quantizer.py:13-256
    vbq_amd.utils                                          img-compression/utils.py solvers
    vbq_amd.embeddings                                     word-embeddings notebook cells 25-30
    vbq_amd.priors                                         vae_models.py:14-43, learned_prior.py:6-334
    vbq_amd.dist                                           element-axis sharding + histogram all-reduce
    vbq_amd.LazyArray / vbq_amd.device_tensor              what compress_latents / compress hand out per lambda: ndarray-like views
                                                           over device tensors (lazy.py); device_tensor(x) = the tensor, no copy
"""
from ._lib import VBQError, lib, library_path  # noqa: F401
from . import ops  # noqa: F401
from .api import gaussian_table, quantize  # noqa: F401
from .lazy import LazyArray, device_tensor  # noqa: F401
from .quantizer import ChannelwisePriorCDFQuantizer  # noqa: F401

__all__ = ["VBQError", "lib", "library_path", "ops", "quantize", "gaussian_table", "ChannelwisePriorCDFQuantizer", "LazyArray",
           "device_tensor"]
