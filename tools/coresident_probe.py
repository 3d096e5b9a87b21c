#!/usr/bin/env python3
"""Developer probe: K1 at 2 / 3 / 4 resident workgroups per CU, alone and with ONE spinning wave on another stream."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import LAMBDAS, N_BITS, make_inputs_with_table as make_inputs
from vbq_amd import ops

spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libspin.so"))
spin.spin_launch_counted.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double]
dev = torch.device("cuda")
rows, C = 36864, 256
mu_h, sg_h, tab_h = make_inputs(rows, C, 1000)
mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev); sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev)
tab = torch.from_numpy(tab_h).to(dev)
L = len(LAMBDAS)
rng = np.random.default_rng(5)
ll = torch.from_numpy((np.arange(N_BITS + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1.0, (L, C, N_BITS + 1)))).astype(np.float32)).to(dev)
idx = torch.empty((L, C, rows), dtype=torch.uint16, device=dev)
side = torch.cuda.Stream()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx, level_len=ll); torch.cuda.synchronize()
for wg in (0, 4, 3, 2):
    for k in (0, 1):
        fn = lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx, level_len=ll, workgroups_per_cu=wg)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        if k:
            spin.spin_launch_counted(ctypes.c_void_p(side.cuda_stream), 1, 2, 12.0); time.sleep(0.0005)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in evs)
        print(f"K1 workgroups_per_cu={wg} spin waves={k}: {t[6]*1e3:7.1f} us (min {t[0]*1e3:.1f}, max {t[-1]*1e3:.1f})", flush=True)
