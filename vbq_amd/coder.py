"""Entropy coder for the rank indices: static-model rANS on the GPU.

The reference stops at ESTIMATING the rate as sum(-log2 freq) with add-n smoothed frequencies
(quantizer.py:138-146, 226-228; utils.py:547).  This module turns the same per-(lambda, channel)
histograms into integer frequency tables and the indices into a bitstream, and decodes it back.
Format: include/vbq.h (vbq_rans_encode_u16).  Nothing here changes the quantization path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

PROB_BITS = 15
DEFAULT_SEGMENT = 1024


def quantize_frequencies(counts, add_n_smoothing=1, prob_bits: int = PROB_BITS) -> np.ndarray:
    """Histogram counts [..., T] -> uint16 frequencies [..., T], every entry >= 1, every row summing
    to 2**prob_bits.  Deterministic largest-remainder rounding of the add-n smoothed frequencies
    the reference's entropy model uses (quantizer.py:141-143)."""
    c = np.asarray(counts.cpu().numpy() if isinstance(counts, torch.Tensor) else counts, dtype=np.float64)
    lead, T = c.shape[:-1], c.shape[-1]
    M = 1 << prob_bits
    if T > M:
        raise ValueError("more symbols than probability slots")
    c = c.reshape(-1, T) + float(add_n_smoothing)
    p = c / c.sum(axis=1, keepdims=True)
    out = np.empty(c.shape, dtype=np.int64)
    for r in range(c.shape[0]):
        ideal = p[r] * M
        f = np.maximum(1, np.floor(ideal).astype(np.int64))
        diff = M - int(f.sum())
        if diff > 0:                                    # hand the missing slots to the largest remainders
            order = np.argsort(-(ideal - np.floor(ideal)), kind="stable")
            f[order[:diff]] += 1 if diff <= T else 0
            if diff > T:
                f[order] += diff // T
                f[order[: diff % T]] += 1
        elif diff < 0:                                  # take the surplus from the most probable symbols
            need = -diff
            while need > 0:
                order = np.argsort(-f, kind="stable")
                for j in order:
                    if need == 0:
                        break
                    take = min(need, int(f[j]) - 1, max(1, int(f[j]) // 64))
                    f[j] -= take
                    need -= take
        assert f.sum() == M and f.min() >= 1
        out[r] = f
    return out.reshape(lead + (T,)).astype(np.uint16)


def ideal_bits(counts, freq, prob_bits: int = PROB_BITS) -> float:
    """Cross-entropy of the data under the quantised table: sum counts * -log2(freq / 2^PB)."""
    c = np.asarray(counts.cpu().numpy() if isinstance(counts, torch.Tensor) else counts, dtype=np.float64)
    return float(np.sum(c * (prob_bits - np.log2(np.asarray(freq, dtype=np.float64)))))


class RansCodec:
    """Encoder / decoder for u16 rank indices laid out as streams [S, n] (S = L*C planes of K1)."""

    def __init__(self, freq, N: int = 10, segment: int = DEFAULT_SEGMENT):
        f = freq if isinstance(freq, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(freq, dtype=np.uint16)))
        T = ops.table_size(N)
        self.freq_host = f.cpu().reshape(-1, T)
        sums = self.freq_host.to(torch.int64).sum(dim=1)
        if not bool(torch.all(sums == (1 << PROB_BITS))) or int(self.freq_host.to(torch.int64).min()) < 1:
            raise ValueError("every frequency row must be >= 1 and sum to 2**15")
        self.N, self.segment, self.T = N, int(segment), T
        self._freq_dev: Optional[torch.Tensor] = None

    def _freq(self, device):
        if self._freq_dev is None or self._freq_dev.device != device:
            self._freq_dev = self.freq_host.to(device).contiguous()
        return self._freq_dev

    def encode(self, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """idx: u16 device tensor [..., n] with prod(leading dims) == number of frequency rows.
        Returns (words u16 [S, nseg, segment+2], sizes u32 [S, nseg])."""
        idx = ops._dev(idx, torch.uint16, "idx")
        n = idx.shape[-1]
        S = idx.numel() // max(n, 1)
        if S != self.freq_host.shape[0]:
            raise ValueError(f"{S} index streams but {self.freq_host.shape[0]} frequency rows")
        nseg = (n + self.segment - 1) // self.segment
        words = torch.zeros((S, nseg, self.segment + 2), dtype=torch.uint16, device=idx.device)
        sizes = torch.zeros((S, nseg), dtype=torch.uint32, device=idx.device)
        check(_lib.lib().vbq_rans_encode_u16(ops._ptr(idx), S, n, self.N, self.segment, ops._ptr(self._freq(idx.device)),
                                             ops._ptr(words), ops._ptr(sizes), ops._stream(idx)), "vbq_rans_encode_u16")
        return words, sizes

    def decode(self, words: torch.Tensor, sizes: torch.Tensor, n: int) -> torch.Tensor:
        """words / sizes are untrusted (they may come from a file): shapes are checked here, segment sizes and
        word counts in the kernel; a damaged stream raises VBQError instead of returning garbage."""
        words = ops._dev(words, torch.uint16, "words")
        sizes = ops._dev(sizes, torch.uint32, "sizes")
        S = self.freq_host.shape[0]
        nseg = (n + self.segment - 1) // self.segment
        if words.numel() != S * nseg * (self.segment + 2) or sizes.numel() != S * nseg:
            raise ValueError(f"expected words [{S}, {nseg}, {self.segment + 2}] and sizes [{S}, {nseg}] for {n} symbols per "
                             f"stream, got {tuple(words.shape)} and {tuple(sizes.shape)}")
        idx = torch.empty((S, n), dtype=torch.uint16, device=words.device)
        status = torch.zeros(1, dtype=torch.uint32, device=words.device)
        check(_lib.lib().vbq_rans_decode_u16(ops._ptr(words), ops._ptr(sizes), S, n, self.N, self.segment,
                                             ops._ptr(self._freq(words.device)), ops._ptr(idx), ops._ptr(status),
                                             ops._stream(words)), "vbq_rans_decode_u16")
        st = int(status.cpu().item())
        if st:
            what = [m for b, m in ((1, "segment size out of range"), (2, "segment ran out of words"),
                                   (4, "left-over words / wrong final state"), (8, "invalid frequency table")) if st & b]
            raise _lib.VBQError("rANS bitstream rejected: " + ", ".join(what))
        return idx

    @staticmethod
    def compressed_bits(sizes: torch.Tensor) -> int:
        return int(sizes.to(torch.int64).sum().item()) * 16

    @staticmethod
    def pack(words: torch.Tensor, sizes: torch.Tensor) -> bytes:
        """Contiguous byte string: the valid words of every segment, stream-major (host side)."""
        w = words.cpu().numpy()
        sz = sizes.cpu().numpy().astype(np.int64)
        keep = np.arange(w.shape[-1])[None, None, :] < sz[..., None]
        return w[keep].tobytes()
