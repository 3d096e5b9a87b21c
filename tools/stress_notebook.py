"""Parity stress of the notebook kernel K1n vs the literal 2047-point brute force (C oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import c_oracle as CO, vbq_oracle as O
from vbq_amd import ops
N = 10
rng = np.random.default_rng(11)
dev = torch.device("cuda")
rank_of_slot = O.level_major_to_rank(N)
tot = bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    scale = float(np.exp(rng.uniform(np.log(0.05), np.log(20))))
    means = (scale * rng.standard_t(4, n)).astype(np.float32)
    stds = (np.exp(rng.normal(-2, 1.5, n)) * scale).astype(np.float32)
    pts, lens = O.notebook_code_book(np.float32(scale), N)
    srt = np.sort(pts)
    k = n // 10
    means[:k] = srt[rng.integers(0, 2047, k)].astype(np.float32)
    j = rng.integers(0, 2046, k)
    means[k:2 * k] = (0.5 * (srt[j] + srt[j + 1])).astype(np.float32)
    nbeta = [10, 50, 33][rnd % 3]
    betas = list(np.exp(np.linspace(np.log(0.01), np.log(1e5), nbeta)))
    if rnd % 3 == 2:                                   # unsorted, with duplicates and a second launch chunk boundary
        betas = list(rng.permutation(betas + betas[:7] + betas[10:40]))
    idx, val = ops.quantize_notebook(torch.from_numpy(means).to(dev), torch.from_numpy(stds).to(dev), torch.from_numpy(pts).to(dev), betas, N=N)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    for i, b in enumerate(betas):
        v, slot = CO.compress_coordinates(means, stds, b, pts, lens, threads=CO.max_threads())
        nb = int(np.count_nonzero(idx[i].astype(np.int64) != rank_of_slot[slot])) + int(np.count_nonzero(val[i] != v))
        tot += n; bad += nb
    print(f"round {rnd}: scale {scale:.3g}, {len(betas)} betas, {len(betas)*n:.3g} latents, mismatches so far {bad}", flush=True)
print("TOTAL", tot, "mismatches", bad)
sys.exit(1 if bad else 0)
