#!/usr/bin/env python3
"""Randomized API fuzz of vbq_quantize_f32 / vbq_histogram_u16 / vbq_level_counts_f32 and the row-range forms against the C
oracle: random shapes, layouts, bit depths, lambda counts (above the 32-lambda chunk too), raw / corrected lengths, optional
outputs, lambdas outside the fast kernel's range, both arithmetic modes, random row cuts.
    python tools/fuzz_api.py [--cases 300] [--seed 0]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scipy.stats import norm

from oracle import c_oracle as CO
from vbq_amd import ops



def run(cases, seed, verbose=True):
    """Returns (failing cases, solves compared)."""
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda")
    bad = 0
    solves = 0
    for case in range(cases):
        bad_case, n = one_case(rng, dev)
        solves += n
        if bad_case:
            bad += 1
            if verbose:
                print(f"case {case}: {bad_case}")
    return bad, solves


def one_case(rng, dev):
    N = int(rng.integers(4, 13))
    T = 2 ** (N + 1) - 1
    C = int(rng.choice([1, 1, 2, 3, 5, 16, 17, 40]))
    rows = int(rng.choice([1, 2, 7, 64, 511, 512, 513, 1000, 3001]))
    L = int(rng.choice([1, 2, 5, 32, 33, 40]))
    layout = "bc" if C == 1 else ("cb" if N > 10 else str(rng.choice(["bc", "cb"])))      # N > 10: planes only
    mode = "f64" if rng.random() < 0.15 else "f32"
    xi = np.concatenate([(np.arange(2 ** n) + 0.5) / 2 ** n for n in range(N + 1)])
    scale = np.exp(rng.uniform(np.log(0.1), np.log(10), C))
    tab = norm.ppf(xi[None, :], scale=scale[:, None]).astype(np.float32)
    mu = (scale * rng.standard_t(4, (rows, C))).astype(np.float32)
    sg = (np.exp(rng.normal(-2, 1.2, (rows, C))) * scale).astype(np.float32)
    if rows > 4:
        srt = np.sort(tab[0])
        mu[0, 0] = srt[rng.integers(0, T)]
        mu[1, 0] = np.float32(0.5 * (float(srt[3]) + float(srt[4])))
    lam = np.exp(rng.uniform(np.log(1e-3), np.log(300), L))
    if rng.random() < 0.2:
        lam[rng.integers(0, L)] = float(rng.choice([0.0, 1e-14, 1e20]))        # outside the fast kernel's range
    ll = None
    if mode == "f32" and rng.random() < 0.6:
        ll = (np.arange(N + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1.5, (L, C, N + 1)))).astype(np.float32)
    wz, wb = bool(rng.random() < 0.4), bool(rng.random() < 0.4)
    want = CO.quantize(mu, sg, tab, list(lam), N=N, level_len=ll, mode=1 if mode == "f64" else 0, want_zhat=wz,
                       want_bits=wb, threads=8)
    want = want if isinstance(want, tuple) else (want,)
    m_in, s_in = (mu[:, 0], sg[:, 0]) if C == 1 else ((mu, sg) if layout == "bc" else (np.ascontiguousarray(mu.T), np.ascontiguousarray(sg.T)))
    got = ops.quantize(torch.from_numpy(m_in).to(dev), torch.from_numpy(s_in).to(dev), torch.from_numpy(tab).to(dev), list(lam), N=N,
                       level_len=None if ll is None else torch.from_numpy(ll).to(dev), layout=layout, mode=mode, want_zhat=wz,
                       want_bits=wb)
    got = got if isinstance(got, tuple) else (got,)

    def canon(t):
        x = t.cpu().numpy()
        if C == 1:
            return x.reshape(L, rows, 1)
        return x if layout == "bc" else x.transpose(0, 2, 1)
    miss = sum(int((canon(g) != w).sum()) for g, w in zip(got, want))
    hist = ops.histogram(got[0], C, N=N, layout=layout).cpu().numpy()
    hw = CO.histogram(want[0], C, N=N)
    miss += int((hist != hw).sum())
    # the same pass cut into random row ranges (vbq_quantize_rows_f32 / vbq_histogram_rows_u16)
    if mode == "f32" and rng.random() < 0.5:
        cuts = sorted({0, rows, *[int(v) for v in rng.integers(0, rows + 1, 2)]})
        idx2 = torch.zeros_like(got[0])
        cnt2 = torch.zeros(hist.shape, dtype=torch.int64, device=dev)
        for a, b in zip(cuts[:-1], cuts[1:]):
            ops.quantize(torch.from_numpy(m_in).to(dev), torch.from_numpy(s_in).to(dev), torch.from_numpy(tab).to(dev), list(lam), N=N,
                         level_len=None if ll is None else torch.from_numpy(ll).to(dev), layout=layout, out_idx=idx2, rows=(a, b),
                         workgroups_per_cu=int(rng.choice([0, 4])), reserved_workgroups=[None, 0, 64, 900][int(rng.integers(0, 4))])
            ops.histogram(idx2, C, N=N, layout=layout, out=cnt2, rows=(a, b))
        miss += int((canon(idx2) != want[0]).sum()) + int((cnt2.cpu().numpy() != hw).sum())
    # the counting kernels (K1t for raw lengths at N = 10, K1h otherwise): bit-length histogram of the same solve
    in_range = bool(np.all((lam >= 1.9e-12) & (lam <= 1.8e19)))
    if mode == "f32" and in_range and (C == 1 or layout == "cb" or rng.random() < 0.5):
        lay = layout if (C == 1 or layout == "cb") else "bc->cb"
        lc = ops.level_counts(torch.from_numpy(m_in).to(dev), torch.from_numpy(s_in).to(dev), torch.from_numpy(tab).to(dev), list(lam), N=N,
                              level_len=None if ll is None else torch.from_numpy(ll).to(dev), layout=lay,
                              reserved_workgroups=[None, 0, 64, 900][int(rng.integers(0, 4))]).cpu().numpy()
        k = np.arange(1, T + 1)
        lev = N - np.log2(k & -k).astype(np.int64)
        wl = np.zeros((L, C, N + 1), np.int64)
        for n in range(N + 1):
            wl[:, :, n] = hw[:, :, lev == n].sum(axis=2)
        miss += int((lc != wl).sum())
    desc = f"N={N} C={C} rows={rows} L={L} layout={layout} mode={mode} ll={ll is not None} zhat={wz} bits={wb}: {miss} mismatches"
    return (desc if miss else None), rows * C * L


def one_latents_case(rng, dev):
    """vbq_compress_latents_f32 (planes + solve + fused lookups in one call), vbq_gather_latents_u16 on its own and the facade's
    output forms against the oracle: random C (vector and scalar tiles), ragged rows, every spread kind, raw / corrected lengths."""
    import vbq_amd
    N = int(rng.choice([10, 10, 10, 8, 11, 5]))
    T = 2 ** (N + 1) - 1
    C = int(rng.choice([1, 2, 3, 4, 7, 8, 20, 64, 68, 70, 130]))
    rows = int(rng.choice([1, 3, 8, 31, 32, 33, 64, 200, 513, 1536, 3076, 4096]))     # the last two: the LDS-table lookups
    L = int(rng.choice([1, 2, 3, 5, 16, 20, 32]))
    if rows > 2000 and C > 20:
        C = int(rng.choice([4, 8, 20]))                              # keep the oracle's share of a case small
    xi = np.concatenate([(np.arange(2 ** n) + 0.5) / 2 ** n for n in range(N + 1)])
    scale = np.exp(rng.uniform(np.log(0.1), np.log(10), C))
    tab = norm.ppf(xi[None, :], scale=scale[:, None]).astype(np.float32)
    srt = np.sort(tab, axis=1)
    mu = (scale * rng.standard_t(4, (rows, C))).astype(np.float32)
    lv = rng.normal(-4, 2.0, (rows, C)).astype(np.float32) + 2 * np.log(scale).astype(np.float32)
    lam = [float(v) for v in np.exp(rng.uniform(np.log(1e-3), np.log(300), L))]
    kind = str(rng.choice(["sigma", "variance", "logvar"]))
    lv_d = torch.from_numpy(lv).to(dev)
    sig_d = torch.exp(lv_d) ** 0.5                                   # what every kind must amount to
    spread = {"sigma": sig_d, "variance": torch.exp(lv_d), "logvar": lv_d}[kind]
    if kind == "variance":
        sig_d = torch.sqrt(spread)
    sg = sig_d.cpu().numpy()
    ll = models = None
    if rng.random() < 0.6:
        ll = (np.arange(N + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1.5, (L, C, N + 1)))).astype(np.float32)
    if rng.random() < 0.7:
        models = rng.uniform(0.2, 16, (L, C, T)).astype(np.float32)
    wi, wz, wb = CO.quantize(mu, sg, tab, lam, N=N, level_len=ll, want_zhat=True, want_bits=True, threads=8)
    d = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    z, raw, nb = ops.compress_latents(d(mu), spread, d(tab), d(srt), lam, N=N, spread=kind, level_len=d(ll), models=d(models))
    miss = int((z.cpu().numpy() != wz).sum()) + int((raw.cpu().numpy() != (wb if ll is not None else wb.astype(np.int32))).sum())
    if models is not None:
        want_nb = models[np.arange(L)[:, None, None], np.arange(C)[None, None, :], wi.astype(np.int64)]
        miss += int((nb.cpu().numpy() != want_nb).sum())
    # the facade on the same latents: channel-last and planes out, values / lengths / no indices
    if N <= 10 or C == 1:
        form = int(rng.integers(0, 3))
        if form == 0:
            got = vbq_amd.quantize(d(mu), sig_d, lam, table=tab, N=N, lengths=ll, return_values=True, return_bits=True)
            miss += sum(int((g.cpu().numpy() != w).sum()) for g, w in zip(got, (wi, wz, wb)))
        elif form == 1:
            got = vbq_amd.quantize(d(mu), sig_d, lam, table=tab, N=N, lengths=ll, out_layout="planes", return_values=True)
            miss += sum(int((g.cpu().numpy().reshape(L, C, rows).transpose(0, 2, 1) != w).sum()) for g, w in zip(got, (wi, wz)))
        else:
            got = vbq_amd.quantize(d(mu), sig_d, lam, table=tab, N=N, lengths=ll, return_values=True, return_indices=False)
            miss += int((got.cpu().numpy() != wz).sum())
    desc = f"latents N={N} C={C} rows={rows} L={L} spread={kind} ll={ll is not None} models={models is not None}: {miss} mismatches"
    return (desc if miss else None), rows * C * L


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    bad, solves = run(a.cases, a.seed)
    print(f"{a.cases} cases, {solves:.3g} solves, {bad} failing cases")
    rng2 = np.random.default_rng(a.seed + 10_000)
    bad2 = solves2 = 0
    for case in range(a.cases):
        desc, n = one_latents_case(rng2, torch.device("cuda"))
        solves2 += n
        if desc:
            bad2 += 1
            print(f"case {case}: {desc}")
    print(f"{a.cases} per-image-call / facade cases, {solves2:.3g} solves, {bad2} failing cases")
    sys.exit(1 if bad or bad2 else 0)
