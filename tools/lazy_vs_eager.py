#!/usr/bin/env python3
"""Developer tool: compress_latents for one image with everything read on the host -- round 4's eager form (three staging copies, one
pool thread per quantity) against round 5's lazy views, alternating in ONE process."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vbq_amd
from bench import LAMBDAS_16, N_BITS, make_inputs_with_table

dev = torch.device("cuda")
rows, C = 36864, 256
mu_h, sg_h, tab_h = make_inputs_with_table(rows, C, 1000)
mu_bc, sg_bc = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)


class _Table:
    def inverse_cdf(self, xi):
        return np.ascontiguousarray(tab_h.T)


q = vbq_amd.ChannelwisePriorCDFQuantizer(C, N_BITS)
q.build_code_points(_Table())
q.build_entropy_models_from_latents(mu_bc, sg_bc, LAMBDAS_16, 1)
m_np = mu_h[:1536].reshape(1, 32, 48, C).copy()
lv_np = (2 * np.log(sg_h[:1536])).astype(np.float32).reshape(1, 32, 48, C)
KEYS = ("Z_hat", "raw_num_bits", "num_bits")
stage, pool = {}, ThreadPoolExecutor(3)


def eager_r4():
    o = q.compress_latents(m_np, lv_np, LAMBDAS_16, return_np=False)
    host, done = {}, {}
    st = torch.cuda.current_stream()
    for key in KEYS:
        t = o[key][LAMBDAS_16[0]]._base              # the [L, ...] tensor the per-lambda views are rows of
        h = stage.get(key)
        if h is None or h.numel() < t.numel():
            h = stage[key] = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
        hv = h[:t.numel()].view(t.shape)
        hv.copy_(t, non_blocking=True)
        host[key] = hv
        done[key] = torch.cuda.Event()
        done[key].record(st)

    def copy_out(key):
        done[key].synchronize()
        return np.array(host[key].numpy())
    return dict(zip(KEYS, pool.map(copy_out, KEYS)))


def lazy_all():
    o = q.compress_latents(m_np, lv_np, LAMBDAS_16)
    return {k: np.asarray(o[k][LAMBDAS_16[0]]) for k in KEYS}


def lazy_one():
    o = q.compress_latents(m_np, lv_np, LAMBDAS_16)
    return np.asarray(o["num_bits"][LAMBDAS_16[0]])


def timeit(f, n=25):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a, b = eager_r4(), lazy_all()
print("same numbers:", all(np.array_equal(a[k].reshape(16, -1)[0], b[k].reshape(-1)) for k in KEYS))
for rnd in range(4):
    print(f"round {rnd}: eager r4 {timeit(eager_r4):6.3f} ms   lazy, all read {timeit(lazy_all):6.3f} ms   lazy, num_bits alone {timeit(lazy_one):6.3f} ms")
