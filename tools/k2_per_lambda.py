import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import LAMBDAS, make_inputs, N_BITS
from vbq_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda")
n = 10_000_000
mu, sg, tab = make_inputs(n, 1, 0)
mu, sg, tab = (torch.from_numpy(a).to(dev) for a in (mu.ravel(), sg.ravel(), tab))
idx = ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS)
for l in range(0, 32, 3):
    one = idx[l:l + 1].contiguous()
    cnt = torch.zeros((1, 1, 2047), dtype=torch.int64, device=dev)
    med, best = timeit(lambda: ops.histogram(one, 1, N=N_BITS, out=cnt), 10)
    c = ops.histogram(one, 1, N=N_BITS)[0, 0].cpu().numpy()
    top = np.sort(c)[::-1]
    print(f"lambda[{l}]={LAMBDAS[l]:.4g}: {med*1e3:.1f} us  distinct bins={np.count_nonzero(c)}  top1={top[0]/n:.3f} top4={top[:4].sum()/n:.3f}")
