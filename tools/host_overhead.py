#!/usr/bin/env python3
"""Developer tool: host-side cost of one call (Python + ctypes + launch), measured on inputs so small that the kernels are free.

    python tools/host_overhead.py        (on the GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vbq_amd
from bench import LAMBDAS, N_BITS, make_inputs_with_table
from vbq_amd import ops


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


def main():
    dev = torch.device("cuda")
    rows, C = 64, 256
    mu_h, sg_h, tab_h = make_inputs_with_table(rows, C, 1000)
    mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev)
    sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev)
    mu_bc, sg_bc = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)
    tab = torch.from_numpy(tab_h).to(dev)
    idx1 = torch.empty((1, C, rows), dtype=torch.uint16, device=dev)
    idx32 = torch.empty((32, C, rows), dtype=torch.uint16, device=dev)
    lam1 = [LAMBDAS[12]]
    print(f"ops.quantize L=1, out given       {per_call(lambda: ops.quantize(mu, sg, tab, lam1, N=N_BITS, layout='cb', out_idx=idx1)):7.1f} us per call")
    print(f"ops.quantize L=1, out allocated   {per_call(lambda: ops.quantize(mu, sg, tab, lam1, N=N_BITS, layout='cb')):7.1f} us per call")
    print(f"ops.quantize L=32, out given      {per_call(lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout='cb', out_idx=idx32)):7.1f} us per call")
    print(f"vbq_amd.quantize facade L=1 (bc)  {per_call(lambda: vbq_amd.quantize(mu_bc, sg_bc, lam1, table=tab)):7.1f} us per call")
    print(f"vbq_amd.quantize facade L=32 (bc) {per_call(lambda: vbq_amd.quantize(mu_bc, sg_bc, LAMBDAS, table=tab)):7.1f} us per call")
    # NumPy tables (api._device_table): a writeable array is compared with what was uploaded on EVERY call (xxh3 when xxhash is
    # importable, np.array_equal against a kept copy otherwise -- the second block hides xxhash to show that route); a read-only
    # array that is the very object prepared before takes the identity path
    import vbq_amd.api as api
    tab_w = tab_h.copy()
    tab_ro = tab_h.copy()
    tab_ro.setflags(write=False)
    for hide in (False, True):
        real = api._xxh3
        if hide:
            api._xxh3 = lambda: None
        api._CHECKED.clear()
        tag = "no xxhash" if hide else ("xxhash" if real() is not None else "no xxhash")
        print(f"facade L=1, NumPy [256, 2047] table, writeable ({tag}) {per_call(lambda: vbq_amd.quantize(mu_bc, sg_bc, lam1, table=tab_w), 500):7.1f} us per call")
        print(f"facade L=1, NumPy [256, 2047] table, read-only ({tag}) {per_call(lambda: vbq_amd.quantize(mu_bc, sg_bc, lam1, table=tab_ro), 500):7.1f} us per call")
        api._xxh3 = real
    lc = torch.zeros((32, C, N_BITS + 1), dtype=torch.int64, device=dev)
    print(f"ops.level_counts L=32             {per_call(lambda: ops.level_counts(mu, sg, tab, LAMBDAS, N=N_BITS, layout='cb', out=lc)):7.1f} us per call")
    e = torch.empty(16, device=dev)
    print(f"torch.empty(4096 B)               {per_call(lambda: torch.empty(4096, dtype=torch.uint8, device=dev)):7.1f} us per call")
    print(f"torch add (reference point)       {per_call(lambda: e.add_(1.0)):7.1f} us per call")


if __name__ == "__main__":
    main()
