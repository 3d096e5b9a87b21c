"""quantize(mu, sigma, lmbda): the one-call surface named by BASELINE.json's north_star."""
from __future__ import annotations

import weakref
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


# Tables that already passed the monotonicity check (and, for host tables, their device copies): a sweep calls quantize()
# many times with ONE table, and the check costs a device-to-host copy of it.  Keyed by the object's identity and, for
# tensors, its storage pointer and version counter, so an in-place edit or a new table is checked again.
_CHECKED = {}


def _table_key(table):
    if isinstance(table, torch.Tensor):
        return ("t", id(table), table.data_ptr(), table._version, tuple(table.shape), str(table.device))
    return ("n", id(table), table.__array_interface__["data"][0], tuple(table.shape))


def _device_table(table, C: int, N: int, dev, validate: bool) -> torch.Tensor:
    """Level-major f32 [C, T] on `dev`; raises ValueError unless every channel is non-decreasing in xi (tf.searchsorted is
    undefined otherwise, quantizer.py:135).  Checked once per table object."""
    T = table_size(N)
    if not isinstance(table, torch.Tensor):
        table = np.asarray(table)
    key = _table_key(table)
    hit = _CHECKED.get(key)
    if hit is not None and hit[0]() is table and hit[2] == (C, N, str(dev)):
        return hit[1]
    tab_t = (torch.from_numpy(np.ascontiguousarray(table, dtype=np.float32)) if not isinstance(table, torch.Tensor) else table)
    tab_t = tab_t.to(dev, torch.float32).reshape(C, T).contiguous()
    if validate:
        host = table if not isinstance(table, torch.Tensor) else tab_t.cpu().numpy()
        host = np.asarray(host, dtype=np.float32).reshape(C, T)
        if not bool(np.all(np.diff(level_major_to_sorted(host), axis=1) >= 0)):
            raise ValueError("table is not monotone in xi")
        if len(_CHECKED) > 64:
            _CHECKED.clear()
        try:
            _CHECKED[key] = (weakref.ref(table), tab_t, (C, N, str(dev)))
        except TypeError:                                   # objects without weak references are checked every time
            pass
    return tab_t


def quantize(mu, sigma, lmbda: Union[float, Sequence[float]], *, table=None, prior=None, N: int = 10,
             lengths=None, layout: str = "bc", return_values: bool = False, return_bits: bool = False,
             validate: bool = True):
    """argmin over the 2^(N+1)-1 code points of  (z-mu)^2/(2 sigma^2) + lambda * R(z)  for every
    element and every lambda, with the arithmetic and tie rules of the reference
    (quantizer.py:156-188 + utils.py:363-423).

    mu, sigma : f32 torch (device) or NumPy arrays, [rows, C] ('bc', how the latents arrive: quantizer.py:90-91),
                [C, rows] ('cb') or [n].
    lmbda     : a scalar or a list; with a list the leading axis of the outputs is the lambda axis.
    table     : level-major f32 [C, T] code points (ChannelwisePriorCDFQuantizer.all_code_points);
                or `prior` with .inverse_cdf(xi[T, C]) from which the table is built.
    lengths   : optional f32 [L, C, N+1] code length per bit level (default: the level itself).
    validate  : check (once per table object) that the table is monotone in xi; False skips the check.
    Returns rank indices (uint16; position in the sorted table) shaped like the input with the lambda axis in front and,
    if asked, the code-point values / code lengths (f32).  Torch in -> device tensors out; NumPy in -> NumPy out.

    Channel-last input takes the plane kernels: two input transposes (vbq_transpose_f32), the solve on [C, rows] planes
    (K1e for raw-length sweeps of 16-32 lambdas, K1 otherwise), and one batched transpose of the results back into the
    caller's layout (vbq_transpose_planes) -- 1.6x faster than the channel-last kernel it replaces."""
    scalar = np.isscalar(lmbda)
    lambdas = [float(lmbda)] if scalar else [float(v) for v in lmbda]
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: vbq_amd.quantize has no CPU implementation")
    if layout not in ("bc", "cb"):
        raise ValueError(f"layout must be 'bc' or 'cb', got {layout!r}")
    dev = torch.device("cuda", torch.cuda.current_device())
    was_np = not isinstance(mu, torch.Tensor)
    mu_t = (torch.from_numpy(np.ascontiguousarray(mu)) if was_np else mu).to(dev, torch.float32)
    sg_t = (torch.from_numpy(np.ascontiguousarray(sigma)) if not isinstance(sigma, torch.Tensor) else sigma).to(dev, torch.float32)
    if mu_t.shape != sg_t.shape:
        raise ValueError(f"mu {tuple(mu_t.shape)} and sigma {tuple(sg_t.shape)} differ in shape")
    C = 1 if mu_t.dim() == 1 else (mu_t.shape[1] if layout == "bc" else mu_t.shape[0])
    if table is None:
        if prior is None:
            raise ValueError("pass table=... or prior=...")
        xi = np.repeat(dyadic_xi(N)[:, None], C, axis=1)
        pts = prior.inverse_cdf(xi)
        pts = pts.cpu().numpy() if isinstance(pts, torch.Tensor) else np.asarray(pts)
        table = np.ascontiguousarray(pts.astype(np.float32).T)
    tab_t = _device_table(table, C, N, dev, validate)
    len_t = None
    if lengths is not None:
        len_t = (torch.from_numpy(np.ascontiguousarray(lengths)) if not isinstance(lengths, torch.Tensor) else lengths)
        len_t = len_t.to(dev, torch.float32)
    planes = layout == "bc" and mu_t.dim() == 2 and C > 1 and mu_t.shape[0] > 0
    if planes:
        res = ops.quantize(ops.transpose(mu_t), ops.transpose(sg_t), tab_t, lambdas, N=N, level_len=len_t, layout="cb",
                           want_zhat=return_values, want_bits=return_bits)
        res = res if isinstance(res, tuple) else (res,)
        res = tuple(ops.transpose_planes(r) for r in res)          # [L, C, rows] -> [L, rows, C]
    else:
        res = ops.quantize(mu_t, sg_t, tab_t, lambdas, N=N, level_len=len_t, layout=layout,
                           want_zhat=return_values, want_bits=return_bits)
        res = res if isinstance(res, tuple) else (res,)
    if scalar:
        res = tuple(r[0] for r in res)
    if was_np:
        res = tuple(r.cpu().numpy() for r in res)
    return res if len(res) > 1 else res[0]
