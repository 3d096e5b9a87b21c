#!/usr/bin/env python3
"""Developer tool: time per 1e6 elements of K1t, K1, K1e, K2 and K1nt for C = 1 tensors of several sizes -- grid-sizing
anomalies show up as a size that is slower per element than its neighbours."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import LAMBDAS, N_BITS, gaussian_tables
from vbq_amd import ops
from tools.abtime import timeit

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
for n in [int(v) for v in (sys.argv[1:] or [9437184, 10_000_000, 10_485_760, 13_000_001, 20_000_000, 50_000_000, 100_000_000, 120_000_000, 125_000_000])]:
    mu = torch.randn(n, device=dev, generator=g).mul_(1.2329).sub_(0.0799)
    sg = torch.randn(n, device=dev, generator=g).mul_(0.7).sub_(2.0).exp_().clamp_(1e-4, 10)
    tab = torch.from_numpy(gaussian_tables([float(torch.sqrt(torch.mean(mu.double() ** 2)))])).to(dev)
    lc = torch.zeros((32, 1, N_BITS + 1), dtype=torch.int64, device=dev)
    per = lambda t: f"{t*1e3/(n/1e6):6.2f}"
    t_k1t, _ = timeit(lambda: ops.level_counts(mu, sg, tab, LAMBDAS, N=N_BITS, out=lc))
    idx = torch.empty((32, n), dtype=torch.uint16, device=dev)
    ll = (torch.arange(N_BITS + 1, device=dev, dtype=torch.float32)[None, None, :] + torch.rand((32, 1, N_BITS + 1), device=dev, generator=g)).contiguous()
    t_k1, _ = timeit(lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, level_len=ll, out_idx=idx))
    t_k1e, _ = timeit(lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, out_idx=idx))
    cnt = torch.zeros((32, 1, 2047), dtype=torch.int32, device=dev)
    t_k2, _ = timeit(lambda: ops.histogram_models(idx.view(32, 1, n), 1, cnt, N=N_BITS))
    from vbq_amd import embeddings as Emb
    pts, _ = Emb.make_code_book(Emb.empirical_std(mu), N_BITS)
    cb = torch.from_numpy(pts).to(dev)
    betas = [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(1e5), 32))]
    t_nt, _ = timeit(lambda: ops.quantize_notebook(mu, sg, cb, betas, N=N_BITS, want_values=False, out_idx=idx))
    print(f"n = {n:>11d}: us per 1e6 elements  K1t {per(t_k1t)}  K1 {per(t_k1)}  K1e {per(t_k1e)}  K2 {per(t_k2)}  K1nt {per(t_nt)}", flush=True)
    del idx
