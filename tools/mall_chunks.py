#!/usr/bin/env python3
"""Developer experiment: does K2 read its indices faster when it runs right behind the K1 chunk that wrote them (the chunk
small enough to stay in the 256 MB memory-side cache)?  K1 and K2 on ONE stream, the rows cut into n chunks; event-timed.
    python tools/mall_chunks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import LAMBDAS, N_BITS, make_inputs_with_table as make_inputs
from vbq_amd import ops
from vbq_amd.pipeline import chunk_bounds
from tools.abtime import timeit

dev = torch.device("cuda")
rows, C = 36864, 256
mu_h, sg_h, tab_h = make_inputs(rows, C, 1000)
mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev)
sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev)
tab = torch.from_numpy(tab_h).to(dev)
L = len(LAMBDAS)
rng = np.random.default_rng(5)
ll = torch.from_numpy((np.arange(N_BITS + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1.0, (L, C, N_BITS + 1)))).astype(np.float32)).to(dev)
idx = torch.empty((L, C, rows), dtype=torch.uint16, device=dev)
cnt = torch.zeros((L, C, 2047), dtype=torch.int32, device=dev)
ref = None
for n in (1, 2, 3, 4, 6, 9):
    chunks = chunk_bounds(rows, n)
    def both():
        cnt.zero_()
        for r0, r1 in chunks:
            ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, level_len=ll, layout="cb", out_idx=idx, rows=(r0, r1))
            ops.histogram(idx, C, N=N_BITS, layout="cb", out=cnt, rows=(r0, r1))
    def k1_only():
        for r0, r1 in chunks:
            ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, level_len=ll, layout="cb", out_idx=idx, rows=(r0, r1))
    def k2_only():
        cnt.zero_()
        for r0, r1 in chunks:
            ops.histogram(idx, C, N=N_BITS, layout="cb", out=cnt, rows=(r0, r1))
    b, _ = timeit(both)
    a, _ = timeit(k1_only)
    c, _ = timeit(k2_only)
    both()
    torch.cuda.synchronize()
    got = cnt.cpu().numpy().copy()
    if ref is None:
        ref = got
    print(f"{n} chunks of {chunks[0][1] - chunks[0][0]} rows ({604 / n:.0f} MB of indices each): K1 then K2 per chunk {b * 1e3:.1f} us; "
          f"K1 chunks alone {a * 1e3:.1f}; K2 chunks alone (cold) {c * 1e3:.1f}; K2 behind its K1 = {1e3 * (b - a):.1f}; same counts: {np.array_equal(got, ref)}", flush=True)
