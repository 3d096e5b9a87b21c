"""quantize(mu, sigma, lmbda): the one-call surface named by BASELINE.json's north_star."""
from __future__ import annotations

from typing import Sequence, Union

import numpy as np
import torch

from . import ops
from ._lib import VBQError
from .tables import dyadic_xi, level_major_to_sorted, table_size


def gaussian_table(scale, N: int = 10, loc=0.0) -> np.ndarray:
    """Level-major f32 code-point table(s) [C, T] of a (factored) Gaussian prior:
    norm.ppf(xi, loc, scale) as FactoredGaussianPrior.inverse_cdf (vae_models.py:40-43)."""
    from scipy.stats import norm
    scale = np.atleast_1d(np.asarray(scale, dtype=np.float64))
    loc = np.broadcast_to(np.asarray(loc, dtype=np.float64), scale.shape)
    return norm.ppf(dyadic_xi(N)[None, :], loc=loc[:, None], scale=scale[:, None]).astype(np.float32)


def quantize(mu, sigma, lmbda: Union[float, Sequence[float]], *, table=None, prior=None, N: int = 10,
             lengths=None, layout: str = "bc", return_values: bool = False, return_bits: bool = False):
    """argmin over the 2^(N+1)-1 code points of  (z-mu)^2/(2 sigma^2) + lambda * R(z)  for every
    element and every lambda, with the arithmetic and tie rules of the reference
    (quantizer.py:156-188 + utils.py:363-423).

    mu, sigma : f32 torch (device) or NumPy arrays, [rows, C] ('bc'), [C, rows] ('cb') or [n].
    lmbda     : a scalar or a list; with a list the leading axis of the outputs is the lambda axis.
    table     : level-major f32 [C, T] code points (ChannelwisePriorCDFQuantizer.all_code_points);
                or `prior` with .inverse_cdf(xi[T, C]) from which the table is built.
    lengths   : optional f32 [L, C, N+1] code length per bit level (default: the level itself).
    Returns rank indices (uint16; position in the sorted table) and, if asked, the code-point
    values / code lengths (f32).  Torch in -> device tensors out; NumPy in -> NumPy out."""
    scalar = np.isscalar(lmbda)
    lambdas = [float(lmbda)] if scalar else [float(v) for v in lmbda]
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: vbq_amd.quantize has no CPU implementation")
    dev = torch.device("cuda", torch.cuda.current_device())
    was_np = not isinstance(mu, torch.Tensor)
    mu_t = (torch.from_numpy(np.ascontiguousarray(mu)) if was_np else mu).to(dev, torch.float32)
    sg_t = (torch.from_numpy(np.ascontiguousarray(sigma)) if not isinstance(sigma, torch.Tensor) else sigma).to(dev, torch.float32)
    C = 1 if mu_t.dim() == 1 else (mu_t.shape[1] if layout == "bc" else mu_t.shape[0])
    if table is None:
        if prior is None:
            raise ValueError("pass table=... or prior=...")
        xi = np.repeat(dyadic_xi(N)[:, None], C, axis=1)
        pts = prior.inverse_cdf(xi)
        pts = pts.cpu().numpy() if isinstance(pts, torch.Tensor) else np.asarray(pts)
        table = np.ascontiguousarray(pts.astype(np.float32).T)
    tab_t = (torch.from_numpy(np.ascontiguousarray(table)) if not isinstance(table, torch.Tensor) else table)
    tab_t = tab_t.to(dev, torch.float32).reshape(C, table_size(N))
    if not bool(torch.all(torch.diff(torch.from_numpy(level_major_to_sorted(tab_t.cpu().numpy())), dim=1) >= 0)):
        raise ValueError("table is not monotone in xi")
    len_t = None
    if lengths is not None:
        len_t = (torch.from_numpy(np.ascontiguousarray(lengths)) if not isinstance(lengths, torch.Tensor) else lengths)
        len_t = len_t.to(dev, torch.float32)
    res = ops.quantize(mu_t, sg_t, tab_t, lambdas, N=N, level_len=len_t, layout=layout,
                       want_zhat=return_values, want_bits=return_bits)
    res = res if isinstance(res, tuple) else (res,)
    if scalar:
        res = tuple(r[0] for r in res)
    if was_np:
        res = tuple(r.cpu().numpy() for r in res)
    return res if len(res) > 1 else res[0]
