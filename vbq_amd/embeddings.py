"""Word-embedding VBQ with the notebook's call surface
(word-embeddings/compress-trained-word-embeddings.ipynb cells 25-30, JSON lines 373-473).

    empirical_std(vecs)                          ipynb:373-374   (K3 moment pass)
    make_code_book(std, max_codepoint_length)    ipynb:383-390   (host: scipy ppf, as the notebook)
    compress_coordinates(means, stds, beta, ...) ipynb:429-443   (K1n)
    empirical_entropy(values)                    ipynb:452-455   (K2 histogram when indices are given)
    prediction_ranks(emb, analogies_id)          ipynb:199-209   (fused f32 MFMA GEMM + count, vbq_ranks.hip)
    test_beta / test_betas / quantize_coordinates / test_quantization   ipynb:464-473, cells 32, 36-37
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import ops
from ._lib import VBQError


def _device():
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: vbq_amd.embeddings has no CPU implementation")
    return torch.device("cuda", torch.cuda.current_device())


def _dev(a, dtype=torch.float32):
    from .lazy import device_tensor
    t = device_tensor(a)                     # a result of this package that still lives on the device: no round trip
    if t is None:
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
    return t.to(_device(), dtype).contiguous()


_STAGER = None


def _stager():
    global _STAGER
    if _STAGER is None:
        from .lazy import HostStager
        _STAGER = HostStager()
    return _STAGER


def empirical_std(vecs, exact: bool = True) -> np.float32:
    """np.sqrt(np.mean(vecs.ravel()**2)) (ipynb:374), float32.  exact=True reproduces NumPy's float32 summation order
    on the GPU (vbq_numpy_sum_sq_f32), so the code book built from it is the notebook's bit for bit; exact=False is
    the f64-accumulating moment kernel K3 (order-free, agrees to 1e-6 relative, a little faster)."""
    x = _dev(vecs).reshape(-1)
    if exact:
        s = np.float32(ops.numpy_sum_sq(x).cpu().numpy()[0])
        # np.mean: the f32 sum divided by the count (a float64 division under NumPy 1.17's scalar rules, rounded back
        # to f32 -- the same value as an f32 division whenever the count is a float32 number), then the root in f32
        return np.sqrt(np.float32(float(s) / x.numel()))
    m = ops.moments(x)
    return np.float32(np.sqrt(float(m[0, 1].item()) / x.numel()))


def make_code_book(std, max_codepoint_length: int = 10):
    """ipynb:383-390: (codepoints f64 [T] level-major, lengths int64 [T])."""
    import scipy.stats
    pts, lens = [], []
    for length in range(max_codepoint_length + 1):
        xi = np.arange(0.5 ** (length + 1), 1, 0.5 ** length)
        pts.append(scipy.stats.norm.ppf(xi, scale=std))
        lens.append(np.full(xi.shape, length, dtype=np.int64))
    return np.concatenate(pts), np.concatenate(lens)


def compress_coordinates_sweep(means, stds, betas: Sequence[float], codepoints, *, want_values=True):
    """All betas in one launch.  Returns (idx u16 [n_beta, *shape] device tensor, values f32 or None)."""
    N = int(np.log2(len(codepoints) + 1)) - 1
    return ops.quantize_notebook(_dev(means), _dev(stds), _dev(codepoints, torch.float64), [float(b) for b in betas],
                                 N=N, want_values=want_values)


def compress_coordinates(means, stds, beta, bitlengths=None, codepoints=None):
    """ipynb:429-443.  `codepoints` replaces the notebook's global of the same name; `bitlengths`
    must be the level of every slot (the only table the notebook ever passes) and is validated.
    Returns the quantized array (f32, shaped like `means`).  Torch in -> a device tensor; NumPy in -> an ndarray-like lazy view
    (vbq_amd.lazy): the notebook hands the result straight to prediction_ranks / empirical_entropy (ipynb:466-470), which
    take it on the device -- the 40 MB of a 100000 x 100 vocabulary cross PCIe only if somebody reads them on the host."""
    if codepoints is None:
        raise ValueError("pass codepoints=... (the notebook reads a global; make_code_book() builds it)")
    N = int(np.log2(len(codepoints) + 1)) - 1
    if bitlengths is not None:
        want = np.concatenate([np.full(2 ** n, n) for n in range(N + 1)])
        if not np.array_equal(np.asarray(bitlengths), want):
            raise ValueError("bitlengths must equal the bit level of each level-major slot")
    _, val = compress_coordinates_sweep(means, stds, [beta], codepoints)
    if isinstance(means, torch.Tensor):
        return val[0]
    from .lazy import DeviceStack
    return DeviceStack("coordinates", val, _stager()).rows()[0]


def entropy_from_counts(counts) -> float:
    c = np.asarray(counts.cpu().numpy() if isinstance(counts, torch.Tensor) else counts, dtype=np.float64).ravel()
    c = c[c > 0]
    tot = c.sum()
    return float(tot * np.log2(tot) - c.dot(np.log2(c)))


def entropy_from_indices(idx: torch.Tensor, N: int = 10):
    """ipynb:452-455 on rank indices: one K2 histogram per beta.  idx: u16 [n_beta, ...]."""
    cnt = ops.histogram(idx.reshape(idx.shape[0], -1), 1, N=N)
    return [entropy_from_counts(cnt[i]) for i in range(cnt.shape[0])]


def empirical_entropy(values) -> float:
    """ipynb:452-455 on an arbitrary value array (generic multiset count via torch.unique)."""
    v = _dev(values).reshape(-1)
    _, counts = torch.unique(v, return_counts=True)
    return entropy_from_counts(counts)


def prediction_ranks(emb, analogies_id):
    """ipynb cell 14 (ipynb:199-209).  `analogies_id` ([Q, 4] word ids) replaces the notebook's global.
    emb: [V, K] array or device tensor.  Returns int64 ranks (NumPy in -> NumPy out)."""
    import ctypes as C
    from . import _lib
    e = _dev(emb)
    if e.dim() != 2:
        raise ValueError("emb must be [V, K]")
    an_np = analogies_id.cpu().numpy() if isinstance(analogies_id, torch.Tensor) else np.asarray(analogies_id)
    if an_np.ndim != 2 or an_np.shape[1] != 4:
        raise ValueError("analogies_id must be [Q, 4]")
    V, K = e.shape
    if an_np.size and (an_np.min() < 0 or an_np.max() >= V):
        raise IndexError("analogy word id out of range")          # NumPy fancy indexing raises too
    an = torch.from_numpy(np.ascontiguousarray(an_np, dtype=np.int32)).to(e.device)
    Q = an.shape[0]
    out = torch.empty(Q, dtype=torch.int64, device=e.device)
    if Q == 0:
        return out if isinstance(emb, torch.Tensor) else out.cpu().numpy()
    h = _lib.lib()
    nbytes = h.vbq_analogy_ranks_workspace_bytes(V, K, Q)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=e.device)
    _lib.check(h.vbq_analogy_ranks_f32(ops._ptr(e), V, K, ops._ptr(an), Q, ops._ptr(out), ops._ptr(ws), C.c_size_t(nbytes),
                                       ops._stream(e)), "vbq_analogy_ranks_f32")
    return out if isinstance(emb, torch.Tensor) else out.cpu().numpy()


def analogy_metrics(ranks):
    """(mrr, acc, hits10) as computed in ipynb cells 30 and 37."""
    r = ranks.cpu().numpy() if isinstance(ranks, torch.Tensor) else np.asarray(ranks)
    return np.average(1 / (1 + r)), np.sum(r == 0) / len(r), np.sum(r < 10) / len(r)


def quantize_coordinates(means, quantization_max):
    """ipynb cell 36, the uniform-rounding baseline (torch on the device: two elementwise ops and a max)."""
    m = _dev(means)
    scale = (quantization_max + 0.5) / m.abs().max()
    q = torch.round(torch.clamp(scale * m, -quantization_max, quantization_max))
    return q if isinstance(means, torch.Tensor) else q.cpu().numpy()


def test_quantization(means, quantization_max, analogies_id):
    """ipynb cell 37: (mrr, acc, hits10, entropy, gzip, bz2, lzma bit lengths) of the rounding baseline.
    The three general-purpose compressors run on the host, as in the notebook."""
    import bz2
    import gzip
    import io
    import lzma
    quantized = np.asarray(quantize_coordinates(np.asarray(means), quantization_max))
    mrr, acc, hits10 = analogy_metrics(prediction_ranks(quantized, analogies_id))
    bits = empirical_entropy(quantized)
    raw = bytes(quantized.astype(np.int8 if quantization_max <= 127 else np.int16).data)
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", compresslevel=9) as f:
        f.write(raw)
    gz_bits = len(buf.getbuffer()) * 8
    bz_bits = len(bz2.compress(raw, 9)) * 8
    lz_bits = len(lzma.compress(raw, format=lzma.FORMAT_ALONE, preset=9)) * 8
    return mrr, acc, hits10, bits, gz_bits, bz_bits, lz_bits


def test_betas(means, stds, betas: Sequence[float], codepoints, analogies_id):
    """The notebook's sweep `[test_beta(vecs_u, stds_u, beta) for beta in betas]` (ipynb cell 32) with the work
    batched the way the device wants it: one K1n launch solves every beta, one K2 launch counts every beta's
    indices, then one fused rank GEMM per beta.  Returns float64 [n_beta, 4] = (mrr, acc, hits10, bits) rows."""
    idx, val = compress_coordinates_sweep(means, stds, betas, codepoints)
    bits = entropy_from_indices(idx, N=int(np.log2(len(codepoints) + 1)) - 1)
    shape = tuple(np.shape(means))
    rows = []
    for i in range(len(betas)):
        ranks = prediction_ranks(val[i].reshape(shape), analogies_id)
        rows.append(analogy_metrics(ranks) + (bits[i],))
    return np.array(rows, dtype=np.float64)


def test_beta(means, stds, beta, codepoints, analogies_id=None):
    """ipynb:464-473 (the notebook passes the (array, None) tuple on; the array is used here).  With
    `analogies_id` returns (mrr, acc, hits10, bits) like the notebook; without it (compressed, bits).
    Everything stays on the device: K1n -> K2 entropy -> fused rank GEMM."""
    idx, val = compress_coordinates_sweep(means, stds, [beta], codepoints)
    compressed = val[0]
    bits = entropy_from_indices(idx, N=int(np.log2(len(codepoints) + 1)) - 1)[0]
    if analogies_id is None:
        return compressed, bits
    if callable(analogies_id):                                       # a user-supplied prediction_ranks(emb)
        ranks = np.asarray(analogies_id(compressed.cpu().numpy()))
    else:
        ranks = prediction_ranks(compressed.reshape(np.shape(means)), analogies_id)
    return analogy_metrics(ranks) + (bits,)
