#!/usr/bin/env python3
"""the whole prior-fit alternation on one GPU's shard, stage by stage:
moment pass (K3) -> Gaussian code book (host ppf) -> solve with raw lengths (K1) -> histogram (K2) ->
corrected lengths -> solve again (K1) -> histogram (K2) -> entropy models (quantizer.py:96-146; ipynb:373-390).
    python tools/alternation_bench.py [--n 100000000]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import LAMBDAS, N_BITS
from vbq_amd import embeddings as E, entropy, ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000_000)
a = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
mu = torch.randn(a.n, device=dev, generator=g) * 1.2329 - 0.0799
sg = torch.exp(torch.randn(a.n, device=dev, generator=g) * 0.7 - 2.0).clamp_(1e-4, 10)
L = len(LAMBDAS)
idx = torch.empty((L, a.n), dtype=torch.uint16, device=dev)
ws = torch.empty(ops._lib.lib().vbq_quantize_workspace_bytes(1, L, N_BITS), dtype=torch.uint8, device=dev)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


rows = []
t, std = timed(lambda: E.empirical_std(mu))
rows.append(("moment pass (K3, 4 B/element)", t, f"{4 * a.n / t / 1e6:.0f} GB/s"))
t, (pts, _) = timed(lambda: E.make_code_book(std, N_BITS), reps=1)
tab = torch.from_numpy(pts.astype(np.float32)[None]).to(dev)
rows.append(("code book: scipy ppf of 2047 points (host)", t, ""))
t, _ = timed(lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, out_idx=idx, workspace=ws))
rows.append((f"solve, raw lengths (K1, {L} lambdas)", t, f"{a.n * L / t / 1e6:.0f} G latents/s"))
t, c1 = timed(lambda: ops.histogram(idx, 1, N=N_BITS))
rows.append(("histogram (K2)", t, f"{2 * a.n * L / t / 1e6:.0f} GB/s"))
t, level_len = timed(lambda: entropy.level_lengths_from_counts(c1, N_BITS, add_n_smoothing=1), reps=1)
rows.append(("corrected lengths from the bit-length histogram (host)", t, ""))
t, _ = timed(lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, level_len=level_len, out_idx=idx, workspace=ws))
rows.append((f"solve, corrected lengths (K1, {L} lambdas)", t, f"{a.n * L / t / 1e6:.0f} G latents/s"))
t, c2 = timed(lambda: ops.histogram(idx, 1, N=N_BITS))
rows.append(("histogram (K2)", t, f"{2 * a.n * L / t / 1e6:.0f} GB/s"))
t, models = timed(lambda: entropy.neg_log2_freq(c2, 1), reps=1)
rows.append(("entropy models -log2(freq) (host)", t, ""))
print(f"alternation on {a.n:.3g} elements, one code book, {L} lambdas")
for name, t, extra in rows:
    print(f"  {name:58s} {t:9.3f} ms  {extra}")
print(f"  {'total':58s} {sum(r[1] for r in rows):9.3f} ms")
