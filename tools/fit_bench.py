"""Timing of the prior-fit step (f1): one NLL+gradient pass over the validation latents
(post_process.py:68-81 sizes: ~500 images x 1536 positions x 256 channels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from vbq_amd import ops, priors
from tools.kbench import timeit
dev = torch.device("cuda")
C, n = 256, 500 * 1536
rng = np.random.default_rng(0)
scale = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C)).astype(np.float32)
x = torch.from_numpy((scale * rng.standard_normal((n, C), dtype=np.float32))).to(dev)
x_cb = ops.transpose(x)
p = priors.BMSHJ2018Prior(C, init_scale=1.0, seed=0)
params = p._params()
out = torch.zeros((C, 44), dtype=torch.float64, device=dev)
med, best = timeit(lambda: ops.bmshj_nll_grad(params, x_cb, out=out), 10)
E = n * C
print(f"nll_grad pass: {med:.3f} ms for {E:.3g} elements -> {E/med/1e6:.1f} G elements/s, {4*E/med/1e6:.0f} GB/s read")
t0 = time.perf_counter()
rec = p.fit(x, lr=0.1, its=20, tol=1e-2, logging_freq=5)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"fit: 20 Adam iterations in {dt*1e3:.1f} ms ({dt/21*1e3:.2f} ms per pass incl. host Adam); loss {rec[0]['loss']:.4f} -> {p.last_loss:.4f}")
