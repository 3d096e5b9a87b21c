import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from bench import LAMBDAS, N_BITS, make_inputs_with_table as make_inputs
from vbq_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda")
rows, C = 36864, 256
mu, sg, tab = make_inputs(rows, C, 0)
mu_t, sg_t = (torch.from_numpy(np.ascontiguousarray(a.T)).to(dev) for a in (mu, sg))
tab = torch.from_numpy(tab).to(dev)
idx = ops.quantize(mu_t, sg_t, tab, LAMBDAS, N=N_BITS, layout="cb")
cnt = torch.zeros((32, C, 2047), dtype=torch.int32, device=dev)
for l in (0, 4, 8, 12, 16, 20, 24, 28, 31):
    rep = idx[l:l+1].expand(32, C, rows).contiguous()
    med, best = timeit(lambda: ops.histogram(rep, C, N=N_BITS, layout="cb", out=cnt), 10)
    print(f"lambda[{l}]={LAMBDAS[l]:.3g}: K2 on 32 copies {med*1e3:.0f} us")
med, best = timeit(lambda: ops.histogram(idx, C, N=N_BITS, layout="cb", out=cnt), 10)
print(f"real sweep: {med*1e3:.0f} us")
