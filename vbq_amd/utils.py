"""Drop-in counterparts of the solver functions of img-compression/utils.py.

Same names and argument meaning as the reference (utils.py:23-24, 307-327, 330-360, 363-423);
`backend` is accepted for source compatibility and ignored -- the arithmetic always runs in the
HIP kernel K1c (vbq_argmax_candidates_f32).  The distortion `fun` must come from
`curry_normal_logpdf` below (the only distortion the reference ever passes, quantizer.py:185):
the kernel hard-wires  -0.5*((z-loc)/scale)**2  in separately rounded f32 operations.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import VBQError


def n_bit_binary_floats(n):
    """utils.py:23-24."""
    return [i * 2 ** (-n) + 2 ** (-n - 1) for i in range(2 ** n)]


class NormalNegHalfSqErr:
    """Callable returned by curry_normal_logpdf(ignore_const=True): f(z) = -0.5*((z-loc)/scale)**2
    (utils.py:319-320).  Carries loc/scale so that the solvers can hand them to the kernel."""

    def __init__(self, loc, scale, const=None):
        self.loc, self.scale, self.const = loc, scale, const

    def __call__(self, z):
        v = -0.5 * ((z - self.loc) / self.scale) ** 2
        return v if self.const is None else v + self.const


def curry_normal_logpdf(loc, scale, ignore_const=False, backend=np):
    """utils.py:307-327.  With ignore_const=False the additive constant does not depend on the
    candidate, so the argmax (and therefore every solver result) is the same."""
    if ignore_const:
        return NormalNegHalfSqErr(loc, scale)
    s = scale.detach().cpu().numpy() if isinstance(scale, torch.Tensor) else np.asarray(scale)
    return NormalNegHalfSqErr(loc, scale, const=-np.log(s) - 0.5 * np.log(2 * np.pi))


def _dev_f32(a, device):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
    return t.to(device, torch.float32).contiguous()


def _device():
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: vbq_amd.utils has no CPU implementation")
    return torch.device("cuda", torch.cuda.current_device())


def _mode(code_lengths, lambs):
    """The NumPy backend of the reference computes f64 scores when the lengths are integers
    (utils.py:388 leaves L uncast); the TF backend casts them to f32.  Mirror that by dtype."""
    dt = code_lengths.dtype
    is_int = (dt in (torch.int32, torch.int64, torch.int16)) if isinstance(code_lengths, torch.Tensor) \
        else np.issubdtype(np.asarray(code_lengths).dtype, np.integer)
    return "f64" if is_int else "f32"


def batch_quantize_indep_dims(Z_shape, code_points, code_lengths, fun, lambs, backend=np, return_np=True, mode=None):
    """utils.py:363-423.  code_points: K x M, or M x B x K; code_lengths likewise, or L x M x B x K."""
    if not isinstance(fun, NormalNegHalfSqErr):
        raise TypeError("fun must be created by vbq_amd.utils.curry_normal_logpdf (Gaussian distortion)")
    dev = _device()
    B, K = Z_shape
    mode = mode or _mode(code_lengths, lambs)
    P = _dev_f32(code_points, dev)
    Lt = _dev_f32(code_lengths, dev)
    if P.dim() == 2:                                                                # utils.py:384-386
        assert P.shape[0] == K
        P = P.t()[:, None, :].expand(-1, B, -1).contiguous()
        Lt = Lt.t()[:, None, :].expand(-1, B, -1).contiguous()
    mu = _dev_f32(fun.loc, dev).expand(B, K).contiguous()
    sg = _dev_f32(fun.scale, dev).expand(B, K).contiguous()
    lambs = list(lambs)
    zh, bt = ops.argmax_candidates(P, Lt, mu, sg, [float(l) for l in lambs], mode=mode)
    int_len = (not torch.is_tensor(code_lengths) and np.issubdtype(np.asarray(code_lengths).dtype, np.integer)) or \
              (torch.is_tensor(code_lengths) and not code_lengths.dtype.is_floating_point)
    Z_hat_dict, num_bits_dict = {}, {}
    for i, lamb in enumerate(lambs):
        z = zh[i]
        b = bt[i].to(torch.int32) if int_len else bt[i]
        Z_hat_dict[lamb] = z.cpu().numpy() if return_np else z
        num_bits_dict[lamb] = b.cpu().numpy() if return_np else b
    return Z_hat_dict, num_bits_dict


def quantize_indep_dims(z, code_points, code_lengths, fun, lamb, backend=np, mode=None):
    """utils.py:330-360: one K-vector against K x M code points (every point is scored)."""
    K = len(z)
    Zd, Bd = batch_quantize_indep_dims((1, K), code_points, code_lengths, fun, [lamb], return_np=True, mode=mode)
    return Zd[lamb][0], Bd[lamb][0]
