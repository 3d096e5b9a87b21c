#!/usr/bin/env python3
"""Developer benchmark of the analogy evaluator (f4): fused f32-MFMA rank GEMM vs NumPy on the host.
    python tools/ranks_bench.py [--V 100000 --K 100 --Q 19544] [--cpu]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vbq_amd import embeddings as E

MFMA_F32_PEAK = 157.3e12      # MI355X dense f32 matrix peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=100_000)
    ap.add_argument("--K", type=int, default=100)
    ap.add_argument("--Q", type=int, default=19_544)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cpu", action="store_true", help="also time the NumPy restatement of ipynb cell 14 on the host")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    emb = rng.normal(0, 1, (a.V, a.K)).astype(np.float32)
    an = rng.integers(0, a.V, (a.Q, 4)).astype(np.int32)
    e = torch.from_numpy(emb).cuda()
    for _ in range(2):
        E.prediction_ranks(e, an)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for s, t in ev:
        s.record(); E.prediction_ranks(e, an); t.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(t) for s, t in ev)[a.reps // 2]
    flop = 2.0 * a.Q * a.V * a.K
    print(f"prediction_ranks V={a.V} K={a.K} Q={a.Q}: {ms:.3f} ms per call  {flop / ms / 1e9:.1f} TFLOP/s "
          f"({flop / ms / 1e9 / (MFMA_F32_PEAK / 1e12):.2f} of the f32 MFMA peak, whole call incl. normalisation)")
    import json
    print(json.dumps({"metric": "analogy questions ranked/sec", "value": a.Q / ms * 1e3, "unit": "questions/s",
                      "config": {"workload": f"prediction_ranks V={a.V} K={a.K} Q={a.Q}"}, "ms_per_call": ms, "dtype": "f32",
                      "roofline": {"bound": "mfma", "achieved": flop / ms / 1e9, "peak": MFMA_F32_PEAK / 1e12,
                                   "unit": "TFLOP/s", "frac": flop / ms / 1e9 / (MFMA_F32_PEAK / 1e12), "traffic": None}}))
    if a.cpu:
        from oracle import vbq_oracle as o
        q = min(a.Q, 2000)
        t0 = time.perf_counter()
        o.prediction_ranks(emb, an[:q], dtype=np.float32)
        dt = time.perf_counter() - t0
        print(f"NumPy f32 on the host ({os.cpu_count()} hardware threads, BLAS default): {dt:.2f} s for {q} questions "
              f"-> {dt * a.Q / q * 1e3:.0f} ms per {a.Q}-question call")


if __name__ == "__main__":
    main()
