"""CPU baselines B1-B3 of BASELINE.md section 3 on this host (C oracle, OpenMP), for the record."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import LAMBDAS, N_BITS, make_inputs_with_table as make_inputs
from oracle import c_oracle as CO, vbq_oracle as O
th = CO.max_threads()
print("host threads:", th, "os.cpu_count():", os.cpu_count())
for rows, C in ((1536, 32), (36864, 32), (36864, 256), (10_000_000, 1)):
    mu, sg, tab = make_inputs(rows, C, 0)
    t0 = time.perf_counter(); idx = CO.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, threads=th); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); CO.quantize(mu[: max(1, rows // 16)], sg[: max(1, rows // 16)], tab, LAMBDAS, N=N_BITS, threads=1); d1 = time.perf_counter() - t1
    t2 = time.perf_counter(); CO.histogram(idx, C, N=N_BITS); d2 = time.perf_counter() - t2
    print(f"B1 [{rows} x {C}] x 32 lambdas: {rows*C*32/dt:.3g} latents/s on {th} threads, {max(1, rows//16)*C*32/d1:.3g} on 1 thread; B3 histogram {rows*C*32/d2:.3g} latents/s (1 thread)")
rng = np.random.default_rng(0)
n = 100_000
means = rng.normal(-0.08, 1.23, n).astype(np.float32); stds = np.exp(rng.normal(-2, 0.7, n)).astype(np.float32)
pts, lens = O.notebook_code_book(O.empirical_std(means), 10)
t0 = time.perf_counter(); CO.compress_coordinates(means, stds, 1.0, pts, lens, threads=th); dt = time.perf_counter() - t0
t1 = time.perf_counter(); CO.compress_coordinates(means[:10000], stds[:10000], 1.0, pts, lens, threads=1); d1 = time.perf_counter() - t1
t2 = time.perf_counter(); O.compress_coordinates(means, stds, 1.0, pts, lens); d2 = time.perf_counter() - t2
print(f"B2 notebook brute force (2047 points, f64), one beta, 1e5 elements: C {n/dt:.3g} latents/s on {th} threads, {1e4/d1:.3g} on 1 thread; NumPy as in the notebook {n/d2:.3g} latents/s")
