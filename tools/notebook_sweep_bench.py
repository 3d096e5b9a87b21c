#!/usr/bin/env python3
"""The word-embedding notebook's whole evaluation sweep (cell 32: 50 betas x [compress, entropy, analogy ranks])
at its own size (100 000 x 100 embedding, 19 544 questions) on one GPU.   python tools/notebook_sweep_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vbq_amd import embeddings as E

rng = np.random.default_rng(0)
V, K, Q = 100_000, 100, 19_544
means = torch.from_numpy(rng.normal(-0.08, 1.23, (V, K)).astype(np.float32)).cuda()
stds = torch.from_numpy(np.clip(np.exp(rng.normal(-2, 0.7, (V, K))), 1e-4, 10).astype(np.float32)).cuda()
an = rng.integers(0, V, (Q, 4)).astype(np.int32)
betas = np.exp(np.linspace(np.log(0.01), np.log(100000), 50))           # ipynb cell 32
cp, _ = E.make_code_book(E.empirical_std(means))
E.test_betas(means, stds, betas[:2], cp, an)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = E.test_betas(means, stds, betas, cp, an)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"50-beta sweep, {V} x {K} embedding, {Q} analogy questions: {dt * 1e3:.0f} ms  (bits/coordinate {res[0, 3] / (V * K):.2f} -> {res[-1, 3] / (V * K):.3f})")
