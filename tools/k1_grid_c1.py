#!/usr/bin/env python3
"""Developer tool: K1 (corrected lengths) on C = 1 tensors under the grid shapes of vbq_quantize_rows_f32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import LAMBDAS, N_BITS, gaussian_tables
from vbq_amd import ops
from tools.abtime import timeit

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
for n in (10_000_000, 9_437_184, 100_000_000):
    mu = torch.randn(n, device=dev, generator=g).mul_(1.2329).sub_(0.0799)
    sg = torch.randn(n, device=dev, generator=g).mul_(0.7).sub_(2.0).exp_().clamp_(1e-4, 10)
    tab = torch.from_numpy(gaussian_tables([1.2355])).to(dev)
    ll = (torch.arange(N_BITS + 1, device=dev, dtype=torch.float32)[None, None, :] + torch.rand((32, 1, N_BITS + 1), device=dev, generator=g)).contiguous()
    idx = torch.empty((32, n), dtype=torch.uint16, device=dev)
    for rep in range(2):
        for w in (0, 5, 4):
            med, best = timeit(lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, out_idx=idx, level_len=ll, rows=(0, n), workgroups_per_cu=w))
            print(f"n = {n:>10d} workgroups_per_cu = {w}: {med*1e3:8.1f} us ({med*1e3/(n/1e6):.2f} per 1e6)", flush=True)
    del idx
