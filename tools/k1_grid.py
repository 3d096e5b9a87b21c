#!/usr/bin/env python3
"""Developer tool: K1 (corrected lengths, Kodak-24 planes) under the grid shapes of vbq_quantize_rows_f32: default (4 rounds
of 5 workgroups per CU) against persistent grids of 1..5 workgroups per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import LAMBDAS, N_BITS, make_inputs_with_table
from vbq_amd import ops
from tools.abtime import timeit

dev = torch.device("cuda")
rows, C = 36864, 256
mu_h, sg_h, tab_h = make_inputs_with_table(rows, C, 1000)
mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev)
sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev)
tab = torch.from_numpy(tab_h).to(dev)
rng = np.random.default_rng(5)
ll = torch.from_numpy((np.arange(N_BITS + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1.0, (32, C, N_BITS + 1)))).astype(np.float32)).to(dev)
idx = torch.empty((32, C, rows), dtype=torch.uint16, device=dev)
for rep in range(2):
    for w in (0, 5, 4, 3):
        med, best = timeit(lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx, level_len=ll,
                                                rows=(0, rows), workgroups_per_cu=w))
        print(f"K1 Kodak-24, workgroups_per_cu = {w}: {med*1e3:7.1f} us (min {best*1e3:7.1f})", flush=True)
