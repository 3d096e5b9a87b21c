"""Entropy-model bookkeeping of ChannelwisePriorCDFQuantizer.build_entropy_models
(img-compression/quantizer.py:99-110, 138-146) on histogram counts.

The heavy part (counting 1e7..1e9 indices) is K2 on the GPU; what is left here is O(L*C*T)
arithmetic on the count tensors, done with the same float32 operations as the reference so
that the resulting code-length tables are bit-identical to NumPy's:
    counts(f32) += n;  freqs = counts / sum(counts, axis=1);  model = -log2(freqs)
"""
from __future__ import annotations

import numpy as np
import torch


def rank_levels(N: int) -> np.ndarray:
    """Bit length of every rank index q: N - ctz(q + 1)."""
    k = np.arange(1, 2 ** (N + 1), dtype=np.int64)
    return (N - np.log2(k & -k).astype(np.int64)).astype(np.int64)


def level_counts_from_counts(counts: torch.Tensor, N: int) -> torch.Tensor:
    """[L, C, T] rank histogram -> [L, C, N+1] histogram of raw bit lengths (the
    np.bincount(raw_num_bits) of quantizer.py:104): integer sums, exact."""
    lv = torch.as_tensor(rank_levels(N), device=counts.device)
    out = torch.zeros(counts.shape[:-1] + (N + 1,), dtype=counts.dtype, device=counts.device)
    out.index_add_(-1, lv, counts)
    return out


def neg_log2_freq(counts, add_n_smoothing) -> np.ndarray:
    """quantizer.py:105-109 / 140-144 on an integer count array [..., K] -> float32 [..., K].
    Host NumPy float32, exactly the reference's operations (the arrays are tiny)."""
    c = np.array(counts.cpu().numpy() if isinstance(counts, torch.Tensor) else counts, dtype=np.float32)
    c += add_n_smoothing
    lead = c.shape[:-1]
    c2 = c.reshape(-1, c.shape[-1])
    freqs = c2 / np.sum(c2, axis=1)[:, None]
    return (-np.log2(freqs)).reshape(lead + (c.shape[-1],))


def neg_log2_lut(total_count: int, K: int, add_n_smoothing, max_entries: int = 1 << 24):
    """Table form of `neg_log2_freq` for rows that all hold the same number of samples: lut[k] =
    -log2(f32(k + n) / f32(total_count + K n)) for k = 0..total_count, computed with the reference's own NumPy float32
    operations (quantizer.py:105-109 / 140-144).  Valid -- and returned -- only when those operations do not depend
    on anything but k: n a non-negative integer and total_count + K n < 2^24, so that counts + n and every partial
    sum of np.sum are exact integers in float32 and every row's sum is the same number.  Otherwise None (the caller
    then runs `neg_log2_freq` on the counts themselves)."""
    n = float(add_n_smoothing)
    if not (n >= 0 and n.is_integer()) or total_count < 0 or total_count + K * n >= 2 ** 24 or total_count + 1 > max_entries:
        return None
    k = np.arange(total_count + 1, dtype=np.float32)
    k += add_n_smoothing
    total = np.float32(total_count + K * n)
    with np.errstate(divide="ignore"):
        return (-np.log2(k / total)).astype(np.float32)


def level_lengths_from_counts(counts: torch.Tensor, N: int, add_n_smoothing=1) -> torch.Tensor:
    """Pass-1 rank histogram [L, C, T] -> corrected level lengths f32 [L, C, N+1]
    = n + raw_code_length_entropy_model (quantizer.py:104-110, 171-175)."""
    model = neg_log2_freq(level_counts_from_counts(counts, N), add_n_smoothing)      # [L, C, N+1] f32
    lv = np.arange(N + 1, dtype=np.int32).astype(np.float32)
    return torch.from_numpy((lv + model).astype(np.float32)).to(counts.device)
