#!/usr/bin/env python3
"""Large randomized parity stress: HIP K1 (fast plane kernel) vs the C oracle, bit for bit.
    python tools/stress_parity.py [--n 4000000] [--rounds 6] [--levels | --small] [--seed K]
--levels: the counting kernels instead (K1t thresholds for raw lengths, K1h dense for corrected ones): bit-length
histograms of windows of 20 000 elements against np.bincount of the oracle's levels, with varied lambda sweeps
(geometric, random spacing, unsorted, short) -- one wrong element shows up as two wrong counts.
Covers raw and corrected (random) lengths, narrow / wide / duplicate tables, tiny and huge
sigma, mu on and between code points, and the f64-score mode."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scipy.stats import norm

from oracle import c_oracle as CO
from vbq_amd import ops

N = 10
XI = np.concatenate([(np.arange(2 ** n) + 0.5) / 2 ** n for n in range(N + 1)])
LAM = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 32)]


def case(rng, kind, n):
    scale = float(np.exp(rng.uniform(np.log(0.05), np.log(20.0))))
    tab = norm.ppf(XI, scale=scale)
    if kind == "dup":
        tab = np.round(tab / scale * 40) / 40 * scale
    if kind == "t":
        from scipy.stats import t as tdist
        tab = tdist.ppf(XI, df=3) * scale
    tab = tab.astype(np.float32)[None]
    mu = (scale * rng.standard_t(4, n)).astype(np.float32)
    sg = np.exp(rng.normal(-2, 1.5, n)).astype(np.float32) * np.float32(scale)
    srt = np.sort(tab[0])
    k = n // 20
    mu[:k] = srt[rng.integers(0, 2047, k)]                                   # exact hits
    j = rng.integers(0, 2046, k)
    mu[k:2 * k] = (0.5 * (srt[j].astype(np.float64) + srt[j + 1])).astype(np.float32)   # mid-points: L/R ties
    mu[2 * k:2 * k + 100] = np.float32(1e6) * scale
    sg[3 * k:3 * k + 1000] = np.float32(1e-6) * scale
    sg[3 * k + 1000:3 * k + 2000] = np.float32(1e4) * scale
    level_len = None
    if kind in ("corr", "dup"):
        ov = np.abs(rng.normal(0, 2.0, (len(LAM), 1, N + 1))).astype(np.float32)
        level_len = (np.arange(N + 1, dtype=np.float32)[None, None, :] + ov).astype(np.float32)
    return tab, mu, sg, level_len


def sweep(rng, r):
    """Lambda sweeps for the counting stress: the bench's, randomly spaced ones, unsorted, short."""
    if r % 4 == 0:
        return LAM
    L = int(rng.integers(1, 33))
    lam = np.exp(rng.uniform(np.log(2e-4), np.log(4e3), L))
    if r % 4 == 1:
        lam = np.sort(lam)
    if r % 4 == 3:
        lam = 2.0 ** rng.integers(-12, 12, L).astype(np.float64)       # powers of two, repeats likely (dense fallback)
    return [float(v) for v in lam]


def levels_main(args):
    from oracle import vbq_oracle as O
    dev = torch.device("cuda")
    rng = np.random.default_rng(777 + args.seed)
    lev = O.levels_of_sorted_ranks(N)
    W = 20_000
    total = bad = 0
    kinds = ["raw", "raw", "t", "dup", "corr", "raw"]
    for r in range(args.rounds):
        kind = kinds[r % len(kinds)]
        tab, mu, sg, ll = case(rng, kind, args.n)
        lam = sweep(rng, r) if kind != "corr" and kind != "dup" else LAM
        if kind == "dup":
            ll = None
        want_idx = CO.quantize(mu, sg, tab, lam, N=N, level_len=ll, threads=CO.max_threads())[:, :, 0]
        wl = lev[want_idx]                                                         # [L, n]
        nb = 0
        tabd = torch.from_numpy(tab).to(dev)
        lld = None if ll is None else torch.from_numpy(ll).to(dev)
        for s0 in range(0, args.n, W):
            m, s_ = torch.from_numpy(mu[s0:s0 + W]).to(dev), torch.from_numpy(sg[s0:s0 + W]).to(dev)
            got = ops.level_counts(m, s_, tabd, lam, N=N, level_len=lld).cpu().numpy()[:, 0, :]
            want = np.stack([np.bincount(wl[l, s0:s0 + W], minlength=N + 1) for l in range(len(lam))])
            nb += int(np.count_nonzero(got != want))
        total += wl.size
        bad += nb
        print(f"round {r} [{kind:4s}, {len(lam):2d} lambdas] {wl.size:.3g} latents in windows of {W}: wrong counts {nb}", flush=True)
    print(f"TOTAL {total:.4g} latents counted, {bad} mismatches")
    sys.exit(1 if bad else 0)


def small_main(args):
    """--small: calls with ONE or TWO lambdas (the pruned descent, K1p): every sweep value on its own and in random pairs,
    raw and corrected lengths (zero and huge overheads included: penalties that never prune / prune at once), all table
    kinds.  The oracle solves the whole 32-point sweep once per round; the GPU is called per lambda / pair."""
    dev = torch.device("cuda")
    rng = np.random.default_rng(4242 + args.seed)
    total = bad = 0
    kinds = ["raw", "corr", "dup", "t", "corr", "raw"]
    for r in range(args.rounds):
        kind = kinds[r % len(kinds)]
        tab, mu, sg, ll = case(rng, kind, args.n)
        if ll is not None and r % 2 == 1:                      # some levels free, some prohibitively long
            ll = ll.copy()
            ll[:, :, rng.integers(0, N + 1, 3)] = 0.0
            ll[:, :, rng.integers(0, N + 1, 2)] = 1.0e6
        want = CO.quantize(mu, sg, tab, LAM, N=N, level_len=ll, threads=CO.max_threads())[:, :, 0]
        m, s_, tabd = torch.from_numpy(mu).to(dev), torch.from_numpy(sg).to(dev), torch.from_numpy(tab).to(dev)
        lld = None if ll is None else torch.from_numpy(ll).to(dev)
        nb = 0
        for l in range(len(LAM)):
            got = ops.quantize(m, s_, tabd, [LAM[l]], N=N, level_len=None if lld is None else lld[l:l + 1].contiguous()).cpu().numpy()
            nb += int(np.count_nonzero(got[0] != want[l]))
            total += got.size
        for _ in range(8):
            a, b = (int(v) for v in rng.integers(0, len(LAM), 2))
            lens = None if lld is None else torch.stack([lld[a], lld[b]]).contiguous()
            got = ops.quantize(m, s_, tabd, [LAM[a], LAM[b]], N=N, level_len=lens).cpu().numpy()
            nb += int(np.count_nonzero(got[0] != want[a])) + int(np.count_nonzero(got[1] != want[b]))
            total += got.size
        bad += nb
        print(f"round {r} [{kind:4s}] 32 single-lambda calls + 8 pairs over {args.n:.3g} elements: mismatches {nb}", flush=True)
    print(f"TOTAL {total:.4g} latents compared, {bad} mismatches")
    sys.exit(1 if bad else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4_000_000)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--levels", action="store_true")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--seed", type=int, default=0, help="added to every generator's seed: a repeated soak sees new data")
    args = ap.parse_args()
    if args.levels:
        return levels_main(args)
    if args.small:
        return small_main(args)
    dev = torch.device("cuda")
    rng = np.random.default_rng(2024 + args.seed)
    total = bad = 0
    kinds = ["raw", "corr", "dup", "t", "raw", "corr"]
    for r in range(args.rounds):
        kind = kinds[r % len(kinds)]
        tab, mu, sg, ll = case(rng, kind, args.n)
        t0 = time.time()
        want = CO.quantize(mu, sg, tab, LAM, N=N, level_len=ll, threads=CO.max_threads())[:, :, 0]
        t1 = time.time()
        got = ops.quantize(torch.from_numpy(mu).to(dev), torch.from_numpy(sg).to(dev), torch.from_numpy(tab).to(dev), LAM,
                           N=N, level_len=None if ll is None else torch.from_numpy(ll).to(dev)).cpu().numpy()
        nb = int(np.count_nonzero(got != want))
        total += got.size
        bad += nb
        print(f"round {r} [{kind:4s}] {got.size:.3g} latents: mismatches {nb}   (oracle {t1 - t0:.1f} s)", flush=True)
        if kind == "raw":
            w64 = CO.quantize(mu[:500000], sg[:500000], tab, LAM, N=N, mode=1, threads=CO.max_threads())[:, :, 0]
            g64 = ops.quantize(torch.from_numpy(mu[:500000]).to(dev), torch.from_numpy(sg[:500000]).to(dev),
                               torch.from_numpy(tab).to(dev), LAM, N=N, mode="f64").cpu().numpy()
            nb = int(np.count_nonzero(g64 != w64))
            total += g64.size
            bad += nb
            print(f"        [f64 ] {g64.size:.3g} latents: mismatches {nb}", flush=True)
    print(f"TOTAL {total:.4g} latents compared, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
