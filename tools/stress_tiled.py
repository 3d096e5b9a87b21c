"""Parity stress of the channel-last (tiled) and plain kernels vs the C oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.stats import norm
from oracle import c_oracle as CO
from vbq_amd import ops
N = 10
XI = np.concatenate([(np.arange(2 ** n) + 0.5) / 2 ** n for n in range(N + 1)])
LAM = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 32)]
rng = np.random.default_rng(7)
dev = torch.device("cuda")
tot = bad = 0
for rows, C in ((50000, 16), (20000, 40), (100000, 3)):
    scale = np.exp(rng.uniform(np.log(0.1), np.log(10.0), C))
    tab = norm.ppf(XI[None], scale=scale[:, None]).astype(np.float32)
    mu = (scale * rng.standard_t(4, (rows, C))).astype(np.float32)
    sg = (np.exp(rng.normal(-2, 1.5, (rows, C))) * scale).astype(np.float32)
    ll = (np.arange(N + 1, dtype=np.float32)[None, None] + np.abs(rng.normal(0, 2, (32, C, N + 1)))).astype(np.float32)
    for level_len in (None, ll):
        want = CO.quantize(mu, sg, tab, LAM, N=N, level_len=level_len, threads=CO.max_threads())
        got = ops.quantize(torch.from_numpy(mu).to(dev), torch.from_numpy(sg).to(dev), torch.from_numpy(tab).to(dev), LAM, N=N,
                           level_len=None if level_len is None else torch.from_numpy(level_len).to(dev)).cpu().numpy()
        nb = int(np.count_nonzero(got != want)); tot += got.size; bad += nb
        print(f"tiled BC {rows}x{C} {'corr' if level_len is not None else 'raw '}: {got.size:.3g} latents, mismatches {nb}", flush=True)
print("TOTAL", tot, "mismatches", bad)
sys.exit(1 if bad else 0)
