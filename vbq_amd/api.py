"""quantize(mu, sigma, lmbda): the one-call surface named by BASELINE.json's north_star."""
from __future__ import annotations

import weakref
from typing import Sequence, Union

import numpy as np
import torch

from . import ops
from ._lib import VBQError
from .tables import dyadic_xi, level_major_to_sorted, level_of_rank, rank_of_slot, table_size


def gaussian_table(scale, N: int = 10, loc=0.0) -> np.ndarray:
    """Level-major f32 code-point table(s) [C, T] of a (factored) Gaussian prior:
    norm.ppf(xi, loc, scale) as FactoredGaussianPrior.inverse_cdf (vae_models.py:40-43)."""
    from scipy.stats import norm
    scale = np.atleast_1d(np.asarray(scale, dtype=np.float64))
    loc = np.broadcast_to(np.asarray(loc, dtype=np.float64), scale.shape)
    return norm.ppf(dyadic_xi(N)[None, :], loc=loc[:, None], scale=scale[:, None]).astype(np.float32)


# Tables already on the device (level-major and rank order), with the verdict of the monotonicity check: a sweep calls quantize()
# many times with ONE table, and preparing it costs an upload, and the check a pass over it.
#   torch tensor   keyed by identity, storage pointer and version counter: an in-place edit or a new tensor is prepared again.
#   NumPy array    has no version counter -- an edit in place is invisible to any identity key.  Keyed by (data pointer, shape,
#                  dtype); on a key hit a strided 256-element sample is compared first and then the WHOLE content against what was
#                  uploaded (xxh3 digest when xxhash is importable, else np.array_equal against a kept host copy: 0.1-0.2 ms for
#                  the 2 MB of a [256, 2047] table -- never a cryptographic hash, 3.4 ms).  An array that cannot have been edited --
#                  read-only all the way down its .base chain, and the very object that was prepared -- skips the content check:
#                  `table.setflags(write=False)` makes a one-lambda call cost microseconds of table handling.
_CHECKED = {}


def _xxh3():
    try:
        import xxhash
        return xxhash.xxh3_128_digest
    except ImportError:
        return None


def _immutable(a: np.ndarray) -> bool:
    while isinstance(a, np.ndarray):
        if a.flags.writeable:
            return False
        a = a.base
    return a is None                                         # (a buffer of another kind behind it: cannot tell)


def _sample(a: np.ndarray) -> bytes:
    flat = a.reshape(-1)
    return flat[:: max(1, flat.size // 256)][:256].tobytes()


class _NumpyEntry:
    __slots__ = ("ref", "sample", "digest", "copy", "pair", "meta", "validated", "immutable")

    def same_content(self, a: np.ndarray) -> bool:
        if _sample(a) != self.sample:
            return False
        if self.digest is not None:
            return _xxh3()(a.reshape(-1).view(np.uint8).data) == self.digest
        # bytes, not values: -0.0 == 0.0 and NaN != NaN would both be wrong answers to "is this the table that was uploaded"
        return np.array_equal(a.reshape(-1).view(np.uint8), self.copy.reshape(-1).view(np.uint8))


def _check_monotone(host: np.ndarray, C: int, T: int):
    if not bool(np.all(np.diff(level_major_to_sorted(np.asarray(host, dtype=np.float32).reshape(C, T)), axis=1) >= 0)):
        raise ValueError("table is not monotone in xi")


def _upload(table, C: int, N: int, dev):
    T = table_size(N)
    tab_t = (torch.from_numpy(np.ascontiguousarray(table, dtype=np.float32)) if not isinstance(table, torch.Tensor) else table)
    tab_t = tab_t.to(dev, torch.float32).reshape(C, T).contiguous()
    slot_of_rank = torch.from_numpy(np.argsort(rank_of_slot(N))).to(dev)
    return tab_t, tab_t[:, slot_of_rank].contiguous()        # rank order == sorted order for a monotone table


def _device_table(table, C: int, N: int, dev, validate: bool):
    """(level-major, rank-order) f32 [C, T] copies on `dev`; raises ValueError unless every channel is non-decreasing in xi
    (tf.searchsorted is undefined otherwise, quantizer.py:135).  Uploaded once per table tensor / table content, checked once
    when `validate` asks for it."""
    T = table_size(N)
    meta = (C, N, str(dev))
    if len(_CHECKED) > 64:
        _CHECKED.clear()
    if isinstance(table, torch.Tensor):
        key = ("t", id(table), table.data_ptr(), table._version, tuple(table.shape), str(table.device))
        hit = _CHECKED.get(key)
        if hit is not None and hit[0]() is table and hit[2] == meta and (hit[3] or not validate):
            return hit[1]
        pair = hit[1] if hit is not None and hit[0]() is table and hit[2] == meta else _upload(table, C, N, dev)
        if validate:
            _check_monotone(pair[0].cpu().numpy(), C, T)
        _CHECKED[key] = (weakref.ref(table), pair, meta, bool(validate))
        return pair
    table = np.asarray(table)
    if not table.flags.c_contiguous:
        table = np.ascontiguousarray(table)                  # (a fresh array per call: takes the content check, never the identity path)
    key = ("n", table.__array_interface__["data"][0], table.shape, table.dtype.str)
    e = _CHECKED.get(key)
    if e is not None and e.meta == meta:
        # "cannot have been edited": read-only down its .base chain and the very object prepared before.  A writable view
        # taken BEFORE the owner froze it (row = table[3]; table.setflags(write=False)) still aliases the memory, and an owner
        # can thaw, edit and freeze again -- so even this path compares the strided 256-element sample (a microsecond);
        # only the full-content comparison is skipped.
        trusted = (e.immutable and e.ref is not None and e.ref() is table and _immutable(table) and _sample(table) == e.sample)
        if trusted or e.same_content(table):
            if validate and not e.validated:
                _check_monotone(table, C, T)
                e.validated = True
            return e.pair
    e = _NumpyEntry()
    e.pair = _upload(table, C, N, dev)
    if validate:
        _check_monotone(table, C, T)
    e.meta, e.validated, e.sample, e.immutable = meta, bool(validate), _sample(table), _immutable(table)
    try:
        e.ref = weakref.ref(table)
    except TypeError:
        e.ref = None
    h = _xxh3()
    e.digest = h(table.reshape(-1).view(np.uint8).data) if h is not None else None
    e.copy = None if h is not None else table.copy()
    _CHECKED[key] = e
    return e.pair


def quantize(mu, sigma, lmbda: Union[float, Sequence[float]], *, table=None, prior=None, N: int = 10,
             lengths=None, layout: str = "bc", out_layout: str = "same", return_values: bool = False,
             return_bits: bool = False, return_indices: bool = True, validate: bool = True):
    """argmin over the 2^(N+1)-1 code points of  (z-mu)^2/(2 sigma^2) + lambda * R(z)  for every
    element and every lambda, with the arithmetic and tie rules of the reference
    (quantizer.py:156-188 + utils.py:363-423).

    mu, sigma : f32 torch (device) or NumPy arrays, [rows, C] ('bc', how the latents arrive: quantizer.py:90-91),
                [C, rows] ('cb') or [n].
    lmbda     : a scalar or a list; with a list the leading axis of the outputs is the lambda axis.
    table     : level-major f32 [C, T] code points (ChannelwisePriorCDFQuantizer.all_code_points);
                or `prior` with .inverse_cdf(xi[T, C]) from which the table is built.
    lengths   : optional f32 [L, C, N+1] code length per bit level (default: the level itself).
    out_layout: 'same' (results shaped like the input) or 'planes': channel-major [L, C, rows] results whatever the input
                layout -- what the kernels write; a caller that goes on to a histogram, a lookup or the entropy coder (all of
                which take planes) saves the transpose back, half the time of a channel-last sweep.
    return_indices=False with return_values / return_bits: only those are produced (and returned).
    validate  : check (once per table tensor / table content) that the table is monotone in xi; False skips the check.
    Returns rank indices (uint16; position in the sorted table) with the lambda axis in front and, if asked, the code-point
    values / code lengths (f32).  Torch in -> device tensors out; NumPy in -> NumPy out.

    Channel-last input takes the plane kernels: one launch that turns means and sigmas into [C, rows] planes
    (vbq_prep_planes_f32), the solve on planes (K1e for raw-length sweeps of 16-32 lambdas, K1p for one or two lambdas, K1
    otherwise; indices only), then ONE pass over the index planes that writes whatever was asked for in the caller's layout
    (vbq_transpose_planes for indices alone; vbq_gather_latents_u16 for values / lengths, which are table lookups on the
    indices, with the indices transposed in the same pass).  Peak device memory of that route: inputs + their planes +
    index planes + results."""
    scalar = np.isscalar(lmbda)
    lambdas = [float(lmbda)] if scalar else [float(v) for v in lmbda]
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: vbq_amd.quantize has no CPU implementation")
    if layout not in ("bc", "cb"):
        raise ValueError(f"layout must be 'bc' or 'cb', got {layout!r}")
    if out_layout not in ("same", "planes"):
        raise ValueError(f"out_layout must be 'same' or 'planes', got {out_layout!r}")
    if not (return_indices or return_values or return_bits):
        raise ValueError("nothing to return")
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
    tab_t, srt_t = _device_table(table, C, N, dev, validate)
    len_t = None
    if lengths is not None:
        len_t = (torch.from_numpy(np.ascontiguousarray(lengths)) if not isinstance(lengths, torch.Tensor) else lengths)
        len_t = len_t.to(dev, torch.float32)
    L = len(lambdas)
    channel_last = layout == "bc" and mu_t.dim() == 2 and C > 1 and mu_t.shape[0] > 0
    if channel_last:
        mu_p, sg_p = ops.prep_planes(mu_t, sg_t)
    elif mu_t.dim() == 2 and layout == "bc":                   # C == 1 (or no rows): planes and channel-last coincide
        mu_p, sg_p = mu_t.reshape(C, -1), sg_t.reshape(C, -1)
    else:
        mu_p, sg_p = mu_t, sg_t
    idx_p = ops.quantize(mu_p, sg_p, tab_t, lambdas, N=N, level_len=len_t, layout="cb")        # [L, C, rows] / [L, n]
    to_caller = channel_last and out_layout == "same"
    if to_caller and (return_values or return_bits):
        z, b, _, qi = ops.gather_latents(idx_p, N=N, table_sorted=srt_t, level_len=len_t, want_zhat=return_values,
                                         want_raw_bits=return_bits, want_idx=return_indices)
        if b is not None and b.dtype != torch.float32:
            b = b.to(torch.float32)
        res = tuple(r for r in (qi, z, b) if r is not None)
    elif to_caller:
        res = (ops.transpose_planes(idx_p),)                    # [L, C, rows] -> [L, rows, C]
    else:
        shape = (L,) + tuple(mu_p.shape)
        if out_layout == "same":
            shape = (L,) + tuple(mu_t.shape)
        elif mu_t.dim() == 2 and layout == "bc":
            shape = (L, C, mu_t.shape[0])
        planes3 = idx_p.reshape(L, C, -1)
        res = (idx_p.reshape(shape),) if return_indices else ()
        if return_values:
            res += (ops.gather(planes3, srt_t, C, N=N, layout="cb").reshape(shape),)
        if return_bits:
            lev = torch.from_numpy(level_of_rank(N)).to(dev)
            if len_t is None:
                lt = lev.to(torch.float32).expand(C, -1).contiguous()                        # [C, T]: level of every rank
            else:
                lt = torch.gather(len_t, 2, lev.expand(L, C, -1)).contiguous()               # [L, C, T]: length of every rank
            res += (ops.gather(planes3, lt, C, N=N, layout="cb").reshape(shape),)
    if scalar:
        res = tuple(r[0] for r in res)
    if was_np:
        res = tuple(r.cpu().numpy() for r in res)
    return res if len(res) > 1 else res[0]
