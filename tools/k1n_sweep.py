import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vbq_amd import embeddings as E, ops
from tools.kbench import timeit
dev = torch.device("cuda")
rng = np.random.default_rng(0)
n = 10_000_000
means = torch.from_numpy(rng.normal(-0.0799, 1.2329, n).astype(np.float32)).to(dev)
stds = torch.from_numpy(np.clip(np.exp(rng.normal(-2, 0.7, n)), 1e-4, 10).astype(np.float32)).to(dev)
pts, lens = E.make_code_book(E.empirical_std(means), 10)
cb = torch.from_numpy(pts).to(dev)
for nb in (8, 16, 32, 64):
    betas = list(np.exp(np.linspace(np.log(0.01), np.log(1e5), nb)))
    idx = torch.empty((nb, n), dtype=torch.uint16, device=dev)
    med, best = timeit(lambda: ops.quantize_notebook(means, stds, cb, betas, N=10, want_values=False, out_idx=idx), 5)
    print(f"nb={nb}: {med:.3f} ms")
