#!/usr/bin/env python3
"""64-bit indexing check: one 1.1e9-element plane (idx offsets beyond 2^31 elements, byte offsets beyond 2^32)
through K1 -> K2, compared with the C oracle on windows at the start, across the 2^31 boundary and at the end.
    python tools/big_shard_check.py [--n 1100000000]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_inputs, N_BITS
from oracle import c_oracle as CO
from vbq_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_100_000_000)
a = ap.parse_args()
dev = torch.device("cuda")
_, _, tab = make_inputs(16, 1, 0)
g = torch.Generator(device=dev).manual_seed(5)
mu = torch.randn(a.n, device=dev, generator=g) * 1.2329 - 0.0799
sg = torch.exp(torch.randn(a.n, device=dev, generator=g) * 0.7 - 2.0).clamp_(1e-4, 10)
lam = [0.05, 3.0, 40.0]
L = len(lam)
idx = ops.quantize(mu, sg, torch.from_numpy(tab).to(dev), lam, N=N_BITS)
torch.cuda.synchronize()
assert idx.shape == (L, a.n)
bad = 0
W = 1_000_000
for start in (0, a.n // 2 - W // 2, (1 << 30) - W // 2, a.n - W):
    start = max(0, min(start, a.n - W))
    want = CO.quantize(mu[start:start + W].cpu().numpy(), sg[start:start + W].cpu().numpy(), tab, lam, N=N_BITS, threads=32)[:, :, 0]
    got = idx[:, start:start + W].cpu().numpy()
    bad += int((got != want).sum())
    print(f"window @{start}: {int((got != want).sum())} mismatches of {L * W}")
cnt = ops.histogram(idx, 1, N=N_BITS)
tot = cnt.sum(dim=-1).cpu().numpy().ravel()
print("histogram totals", tot, "expected", a.n)
assert bad == 0 and np.all(tot == a.n)
# per-lambda histogram of the last window equals a bincount of it
sub = ops.histogram(idx[:, a.n - W:].contiguous(), 1, N=N_BITS).cpu().numpy()[:, 0]
for l in range(L):
    assert np.array_equal(sub[l], np.bincount(idx[l, a.n - W:].cpu().numpy().astype(np.int64), minlength=2047))
print("OK")
