#!/usr/bin/env python3
"""Developer micro-benchmark: times single kernels through the C-ABI with HIP events.
    python tools/kbench.py [flat|cb|all] [--n 10000000] [--reps 20] [--lambdas 32]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import LAMBDAS, N_BITS, make_inputs_with_table as make_inputs
from vbq_amd import ops



def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lambdas", type=int, default=32)
    args = ap.parse_args()
    dev = torch.device("cuda")
    lam = LAMBDAS[:: max(1, 32 // args.lambdas)][:args.lambdas]
    L = len(lam)
    if args.what in ("flat", "all"):
        mu, sg, tab = make_inputs(args.n, 1, 0)
        mu, sg, tab = (torch.from_numpy(a).to(dev) for a in (mu.ravel(), sg.ravel(), tab))
        idx = torch.empty((L, args.n), dtype=torch.uint16, device=dev)
        med, best = timeit(lambda: ops.quantize(mu, sg, tab, lam, N=N_BITS, out_idx=idx), args.reps)
        E = args.n
        print(f"K1 flat  C=1   E={E:.3g} L={L}: {med:.3f} ms (min {best:.3f})  {E*L/med/1e6:.1f} G latents/s  "
              f"alg {(8+2*L)*E/med/1e6:.0f} GB/s")
        cnt = torch.zeros((L, 1, 2047), dtype=torch.int64, device=dev)
        med, best = timeit(lambda: ops.histogram(idx, 1, N=N_BITS, out=cnt), args.reps)
        print(f"K2 flat  C=1   E={E:.3g} L={L}: {med:.3f} ms (min {best:.3f})  alg {2*L*E/med/1e6:.0f} GB/s")
    if args.what in ("cb", "all"):
        rows, C = 36864, 256
        mu, sg, tab = make_inputs(rows, C, 0)
        mu_t, sg_t = (torch.from_numpy(np.ascontiguousarray(a.T)).to(dev) for a in (mu, sg))
        tab = torch.from_numpy(tab).to(dev)
        E = rows * C
        idx = torch.empty((L, C, rows), dtype=torch.uint16, device=dev)
        med, best = timeit(lambda: ops.quantize(mu_t, sg_t, tab, lam, N=N_BITS, layout="cb", out_idx=idx), args.reps)
        print(f"K1 flat  CB 256x36864 L={L}: {med:.3f} ms (min {best:.3f})  {E*L/med/1e6:.1f} G latents/s  "
              f"alg {(8+2*L)*E/med/1e6:.0f} GB/s")
        cnt = torch.zeros((L, C, 2047), dtype=torch.int64, device=dev)
        med, best = timeit(lambda: ops.histogram(idx, C, N=N_BITS, layout="cb", out=cnt), args.reps)
        print(f"K2 flat  CB 256x36864 L={L}: {med:.3f} ms (min {best:.3f})  alg {2*L*E/med/1e6:.0f} GB/s")
        mu_d, sg_d = torch.from_numpy(mu).to(dev), torch.from_numpy(sg).to(dev)
        idx2 = torch.empty((L, rows, C), dtype=torch.uint16, device=dev)
        med, best = timeit(lambda: ops.quantize(mu_d, sg_d, tab, lam, N=N_BITS, out_idx=idx2), args.reps)
        print(f"K1 tiled BC 36864x256 L={L}: {med:.3f} ms (min {best:.3f})  {E*L/med/1e6:.1f} G latents/s")
        med, best = timeit(lambda: ops.histogram(idx2, C, N=N_BITS, out=cnt), args.reps)
        print(f"K2 tiled BC 36864x256 L={L}: {med:.3f} ms (min {best:.3f})  alg {2*L*E/med/1e6:.0f} GB/s")
        assert torch.equal(idx2.permute(0, 2, 1).contiguous().view(torch.int16), idx.view(torch.int16))


if __name__ == "__main__":
    main()
