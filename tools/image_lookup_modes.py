#!/usr/bin/env python3
"""Developer experiment: the per-image call (one Kodak image x 16 lambdas) with the lookups through L2 (default for one image) or out
of LDS-resident tables (VBQ_LOOKUP_LDS=1 both, 2 models only, 3 sorted table only).  Run once per mode (the switch is read once)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vbq_amd
from bench import LAMBDAS_16, N_BITS, make_inputs, empirical_tables
dev = torch.device("cuda")
rows, C = 36864, 256
mu_h, sg_h = make_inputs(rows, C, seed=1000)
mu_bc, sg_bc = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)
tab_h = empirical_tables(mu_bc, rows, C, "bc")
class _Table:
    def inverse_cdf(self, xi):
        return np.ascontiguousarray(tab_h.T)
q = vbq_amd.ChannelwisePriorCDFQuantizer(C, N_BITS)
q.build_code_points(_Table())
q.build_entropy_models_from_latents(mu_bc, sg_bc, LAMBDAS_16, 1)
for imgs in (1, 2, 4, 8, 24):
    B = 1536 * imgs
    m = mu_bc[:B].reshape(imgs, 32, 48, C).contiguous()
    lv = (2.0 * torch.log(sg_bc[:B])).reshape(imgs, 32, 48, C).contiguous()
    f = lambda: q.compress_latents(m, lv, LAMBDAS_16, return_np=False)
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize()
    print(f"VBQ_LOOKUP_LDS={os.environ.get('VBQ_LOOKUP_LDS', 'default')}: {imgs} image(s): {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per call")
