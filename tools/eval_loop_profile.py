#!/usr/bin/env python3
"""Developer tool: where the host time of one evaluation-loop image goes (cProfile over compress() + evaluation_reads())."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vbq_amd
from bench import LAMBDAS_16, N_BITS, make_inputs_with_table
from vbq_amd import utils as vutils

dev = torch.device("cuda")
rows, C = 36864, 256
mu_h, sg_h, tab_h = make_inputs_with_table(rows, C, 1000)
mu_bc, sg_bc = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)


class _Table:
    def inverse_cdf(self, xi):
        return np.ascontiguousarray(tab_h.T)


q = vbq_amd.ChannelwisePriorCDFQuantizer(C, N_BITS)
q.build_code_points(_Table())
lams = LAMBDAS_16
q.build_entropy_models_from_latents(mu_bc, sg_bc, lams, 1)
H, W = 32, 48
B = H * W
m_img = mu_bc[:B].reshape(1, H, W, C).contiguous()
lv_img = (2.0 * torch.log(sg_bc[:B])).reshape(1, H, W, C).contiguous()


class VAE:
    def encode(self, X):
        return m_img, lv_img

    def decode(self, Z):
        return 0.5 + 0.1 * Z[..., :3]


vae, X, pinned = VAE(), np.zeros((1, H, W, 3), np.float32), {}


def image():
    tmp = q.compress(X, vae, lams, clip=True)
    return vutils.evaluation_reads(tmp, lams, pinned)


for _ in range(50):
    image()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    image()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 500 * 1e3:.4f} ms per image")
t0 = time.perf_counter()
for _ in range(500):
    q.compress_latents(m_img, lv_img, lams, return_np=False)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 500 * 1e3:.4f} ms per compress_latents(return_np=False)")
t0 = time.perf_counter()
for _ in range(500):
    q.compress_latents(m_img, lv_img, lams)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 500 * 1e3:.4f} ms per compress_latents (lazy views)")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    image()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
