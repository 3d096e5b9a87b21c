#!/usr/bin/env python3
"""Developer tool: K1t (vbq_level_counts_f32, raw lengths) time per element for C = 1 tensors of several sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import LAMBDAS, N_BITS, gaussian_tables
from vbq_amd import ops
from tools.abtime import timeit

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
for n in [int(v) for v in (sys.argv[1:] or [9437184, 10_000_000, 10_485_760, 20_000_000, 100_000_000])]:
    mu = torch.randn(n, device=dev, generator=g).mul_(1.2329).sub_(0.0799)
    sg = torch.randn(n, device=dev, generator=g).mul_(0.7).sub_(2.0).exp_().clamp_(1e-4, 10)
    tab = torch.from_numpy(gaussian_tables([float(torch.sqrt(torch.mean(mu.double() ** 2)))])).to(dev)
    lc = torch.zeros((32, 1, N_BITS + 1), dtype=torch.int64, device=dev)
    med, best = timeit(lambda: ops.level_counts(mu, sg, tab, LAMBDAS, N=N_BITS, out=lc))
    print(f"n = {n:>11d}: K1t {med*1e3:8.1f} us  ({med*1e3/(n/1e6):.2f} us per 1e6 elements)", flush=True)
