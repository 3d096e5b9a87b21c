#!/usr/bin/env python3
"""Developer benchmark of the image metrics (f4): one Kodak-sized image against M reconstructions, the three
colour modes of utils.evaluate_compression_quantizer.   python tools/metrics_bench.py [--M 32] [--cpu]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vbq_amd import metrics as M


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    H, W = 512, 768
    x = rng.integers(0, 256, (1, H, W, 3)).astype(np.uint8)
    xs = np.repeat(x, a.M, axis=0)
    ys = np.clip(xs + rng.normal(0, 8, xs.shape), 0, 255).astype(np.uint8)

    def run():
        for sl in (slice(0, 3), slice(0, 1), slice(1, 3)):
            p, q = np.ascontiguousarray(xs[..., sl]), np.ascontiguousarray(ys[..., sl])
            M.mse(p, q); M.ms_ssim(p, q)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"mse + ms_ssim, 3 colour modes, {a.M} reconstructions of one 512x768 image: {dt * 1e3:.1f} ms (host arrays in, PCIe included)")
    if a.cpu:
        from oracle import vbq_oracle as o
        t0 = time.perf_counter()
        o.image_mse(xs[:4], ys[:4]); o.ms_ssim(xs[:4], ys[:4])
        dt = time.perf_counter() - t0
        print(f"NumPy restatement on the host: {dt:.2f} s for 4 RGB pairs -> {dt / 4 * a.M * 2 * 1e3:.0f} ms for the same job (6 of 3 channel-planes x {a.M})")


if __name__ == "__main__":
    main()
