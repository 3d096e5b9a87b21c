#!/usr/bin/env python3
"""Developer tool: the resident grids of K1t / K1 beside a kernel that holds workgroup slots, as the
overlapped RCCL all-reduce of the distributed pipeline does.

    python tools/resident_vs_collective.py [--k 0,8,32,64] [--threads 512] [--workload kodak|c1]

For every k: a spin kernel (tools/spin.hip: k workgroups of 512 threads, 128 VGPRs, sleeping) is started on a second stream and
K1t / K1 / K2 of one build step are event-timed while it is resident.  Run once per launch policy:
    (default)                     resident grids sized to every slot of the chip
    VBQ_K1_DYNAMIC=1              K1 as short-lived workgroups (K1t stays resident)
    VBQ_RESERVED_WORKGROUPS=n     resident grids sized to (slots - n)
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import LAMBDAS, N_BITS, make_inputs_with_table as make_inputs
from vbq_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", default="0,8,32,64")
    ap.add_argument("--threads", type=int, default=512)
    ap.add_argument("--workload", default="kodak")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--counted", type=int, default=0, help="1: spin = counted s_sleep loop, 2: counted add loop (no s_memrealtime)")
    args = ap.parse_args()
    spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libspin.so"))
    spin.spin_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    spin.spin_launch_counted.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    dev = torch.device("cuda")
    rows, C = (36864, 256) if args.workload == "kodak" else (10_000_000, 1)
    mu_h, sg_h, tab_h = make_inputs(rows, C, 1000)
    mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev)
    sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev)
    tab = torch.from_numpy(tab_h).to(dev)
    L = len(LAMBDAS)
    rng = np.random.default_rng(5)
    ll = torch.from_numpy((np.arange(N_BITS + 1, dtype=np.float32)[None, None, :] +
                           np.abs(rng.normal(0, 1.0, (L, C, N_BITS + 1)))).astype(np.float32)).to(dev)
    idx = torch.empty((L, C, rows), dtype=torch.uint16, device=dev)
    lc = torch.zeros((L, C, N_BITS + 1), dtype=torch.int64, device=dev)
    cnt = torch.zeros((L, C, 2047), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    kernels = {
        "K1t": lambda: ops.level_counts(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out=lc),
        "K1": lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx, level_len=ll),
        "K2": lambda: ops.histogram_models(idx, C, cnt, N=N_BITS),
    }
    policy = ("dynamic K1" if os.environ.get("VBQ_K1_DYNAMIC") == "1" else "resident") + \
             (f", reserve {os.environ['VBQ_RESERVED_WORKGROUPS']}" if os.environ.get("VBQ_RESERVED_WORKGROUPS") else "")
    for fn in kernels.values():                                  # warm-up + clock ramp
        for _ in range(3):
            fn()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        kernels["K1"]()
        torch.cuda.synchronize()
    for k in [int(v) for v in args.k.split(",")]:
        res = {}
        for name, fn in kernels.items():
            torch.cuda.synchronize()
            if k > 0:                                            # resident for the ramp and the whole timed window
                if args.counted:
                    rc = spin.spin_launch_counted(ctypes.c_void_p(side.cuda_stream), k, args.counted, 60.0)
                else:
                    rc = spin.spin_launch(ctypes.c_void_p(side.cuda_stream), k, args.threads, 60.0)
                assert rc == 0, rc
            # the SAME untimed ramp with and without the neighbour: ~25 ms of back-to-back launches, no idle gap in front of
            # the timed ones (an idle gap alone costs 10-50 %: the first version of this tool slept 0.5 ms after starting the
            # spin kernel and blamed the neighbour for it)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.025:
                for _ in range(4):
                    fn()
                torch.cuda.current_stream().synchronize()        # this stream only: the neighbour keeps running
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
            for a, b in evs:
                a.record(); fn(); b.record()
            torch.cuda.current_stream().synchronize()
            still = not side.query() if k > 0 else None          # the neighbour must have outlived the timed launches
            torch.cuda.synchronize()
            assert still is not False, "the spin kernel ended before the timed launches did"
            t = sorted(a.elapsed_time(b) for a, b in evs)
            res[name] = (t[len(t) // 2], t[0], t[-1])
        print(f"{args.workload:6s} {policy:28s} k={k:3d} x {args.threads if not args.counted else 64} threads{' (counted ' + ('sleep' if args.counted == 1 else 'adds') + ')' if args.counted else ''}: " +
              "  ".join(f"{n} {m * 1e3:7.1f} us (min {lo * 1e3:.1f}, max {hi * 1e3:.1f})" for n, (m, lo, hi) in res.items()), flush=True)


if __name__ == "__main__":
    main()
