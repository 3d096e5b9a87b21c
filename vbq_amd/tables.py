"""Index arithmetic of the dyadic code book (host side).

Level-major slot (n, i)  <->  xi = (2i+1)/2**(n+1)  <->  rank k = (2i+1)*2**(N-n) in the merged
sorted table  <->  rank index q = k-1 (what the kernels emit; `qidx` of quantizer.py:135)."""
from __future__ import annotations

import numpy as np


def table_size(N: int) -> int:
    return 2 ** (N + 1) - 1


def dyadic_xi(N: int) -> np.ndarray:
    """All xi of levels 0..N, level-major (utils.py:23-24 stacked as in quantizer.py:30)."""
    return np.concatenate([(np.arange(2 ** n, dtype=np.float64) + 0.5) / 2 ** n for n in range(N + 1)])


def rank_of_slot(N: int) -> np.ndarray:
    """q = k-1 of every level-major slot."""
    out = []
    for n in range(N + 1):
        i = np.arange(2 ** n, dtype=np.int64)
        out.append((2 * i + 1) * 2 ** (N - n) - 1)
    return np.concatenate(out)


def level_of_rank(N: int) -> np.ndarray:
    """Bit length of rank index q: N - ctz(q+1)."""
    k = np.arange(1, 2 ** (N + 1), dtype=np.int64)
    return (N - np.log2(k & -k).astype(np.int64)).astype(np.int64)


def level_major_to_sorted(table_lm: np.ndarray) -> np.ndarray:
    """[..., T] level-major -> rank order (== sorted order for a table monotone in xi)."""
    N = int(np.log2(table_lm.shape[-1] + 1)) - 1
    out = np.empty_like(table_lm)
    out[..., rank_of_slot(N)] = table_lm
    return out


def sorted_to_level_major(table_sorted: np.ndarray) -> np.ndarray:
    N = int(np.log2(table_sorted.shape[-1] + 1)) - 1
    return np.ascontiguousarray(table_sorted[..., rank_of_slot(N)])
