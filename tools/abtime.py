#!/usr/bin/env python3
"""Developer tool: kernel timings of the library selected by VBQ_HIP_LIBRARY on the bench's Kodak-24 / 1e7 inputs.

    VBQ_HIP_LIBRARY=tools/bin/libvbq_x.so python tools/abtime.py [--what k1t,k1e,k1,l1,k1nt,k2] [--check]

One line per measurement: median / min of 15 event-timed launches.  --check compares K1t / K1e / K1 / small-L results with the
C oracle on the first 2048 rows (a smoke test for an experiment, not the parity suite)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import LAMBDAS, N_BITS, make_inputs_with_table as make_inputs
from vbq_amd import ops



def timeit(fn, reps=15, ramp_s=0.15):
    """Median / minimum of `reps` event-timed calls, after an untimed ramp: the first ~50 ms of kernels after an idle gap run
    about 10 % below the clock the device then sustains (EXPERIMENTS.md), which is more than most A/B differences."""
    import time
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(5):
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
    ap.add_argument("--what", default="k1t,k1e,k1,l1,k1nt,k2")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    what = set(args.what.split(","))
    tag = os.path.basename(os.environ.get("VBQ_HIP_LIBRARY", "default"))
    dev = torch.device("cuda")
    rows, C = 36864, 256
    mu_h, sg_h, tab_h = make_inputs(rows, C, 1000)
    mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev)
    sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev)
    tab = torch.from_numpy(tab_h).to(dev)
    E = rows * C
    L = len(LAMBDAS)
    rng = np.random.default_rng(5)
    ll_h = (np.arange(N_BITS + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1.0, (L, C, N_BITS + 1)))).astype(np.float32)
    ll = torch.from_numpy(ll_h).to(dev)
    idx = torch.empty((L, C, rows), dtype=torch.uint16, device=dev)
    lc = torch.zeros((L, C, N_BITS + 1), dtype=torch.int64, device=dev)

    def out(name, med, best, extra=""):
        print(f"{tag:28s} {name:34s} {med*1e3:9.1f} us (min {best*1e3:8.1f}) {extra}", flush=True)

    if args.check:
        from oracle import c_oracle as CO
        from oracle import vbq_oracle as O
        n = 2048
        th = CO.max_threads()
        want_raw = CO.quantize(mu_h[:n], sg_h[:n], tab_h, LAMBDAS, N=N_BITS, threads=th)
        want_cor = CO.quantize(mu_h[:n], sg_h[:n], tab_h, LAMBDAS, N=N_BITS, level_len=ll_h, threads=th)
        muc, sgc = mu[:, :n].contiguous(), sg[:, :n].contiguous()
        got = ops.quantize(muc, sgc, tab, LAMBDAS, N=N_BITS, layout="cb").cpu().numpy().transpose(0, 2, 1)
        print(tag, "check K1e raw sweep:", bool(np.array_equal(got, want_raw)))
        got = ops.quantize(muc, sgc, tab, LAMBDAS, N=N_BITS, layout="cb", level_len=ll).cpu().numpy().transpose(0, 2, 1)
        print(tag, "check K1 corrected:", bool(np.array_equal(got, want_cor)))
        lv = O.levels_of_sorted_ranks(N_BITS)[want_raw]
        want_lc = np.stack([[np.bincount(lv[l, :, c], minlength=N_BITS + 1) for c in range(C)] for l in range(L)])
        got_lc = ops.level_counts(muc, sgc, tab, LAMBDAS, N=N_BITS, layout="cb").cpu().numpy()
        print(tag, "check K1t level counts:", bool(np.array_equal(got_lc, want_lc)))
        for l in (0, 12, 17, 31):
            for lens, w in ((None, want_raw), (ll[l:l + 1], want_cor)):
                got = ops.quantize(muc, sgc, tab, [LAMBDAS[l]], N=N_BITS, layout="cb", level_len=lens).cpu().numpy().transpose(0, 2, 1)
                print(tag, f"check L=1 lambda[{l}] {'raw' if lens is None else 'corrected'}:", bool(np.array_equal(got[0], w[l])))
        got = ops.quantize(muc, sgc, tab, LAMBDAS[8:12], N=N_BITS, layout="cb").cpu().numpy().transpose(0, 2, 1)
        print(tag, "check L=4 raw:", bool(np.array_equal(got, want_raw[8:12])))

    if "k1t" in what:
        med, best = timeit(lambda: ops.level_counts(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out=lc))
        out("K1t level_counts raw L=32", med, best, f"{8*E/med/1e6:.0f} GB/s-in")
        med, best = timeit(lambda: ops.level_counts(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out=lc, level_len=ll))
        out("K1h level_counts corrected L=32", med, best)
    if "k1e" in what:
        for Lx in (16, 32):
            lam = LAMBDAS[:: 32 // Lx]
            med, best = timeit(lambda: ops.quantize(mu, sg, tab, lam, N=N_BITS, layout="cb", out_idx=idx[:Lx]))
            out(f"K1e quantize raw L={Lx}", med, best, f"frac {(8+2*Lx)*E/med/1e6/8000:.3f}")
    if "k1" in what:
        for Lx in (32, 16, 8, 4, 2, 1):
            lam = LAMBDAS[:: 32 // Lx]
            lens = ll[:: 32 // Lx].contiguous()
            med, best = timeit(lambda: ops.quantize(mu, sg, tab, lam, N=N_BITS, layout="cb", out_idx=idx[:Lx], level_len=lens))
            out(f"K1 quantize corrected L={Lx}", med, best, f"frac {(8+2*Lx)*E/med/1e6/8000:.3f}")
    if "l1" in what:
        for l in (0, 8, 12, 16, 20, 31):
            lam = [LAMBDAS[l]]
            med, best = timeit(lambda: ops.quantize(mu, sg, tab, lam, N=N_BITS, layout="cb", out_idx=idx[:1]))
            out(f"L=1 raw lambda=2^{np.log2(LAMBDAS[l]):.2f}", med, best, f"{E/med/1e6:.1f} G latents/s")
        for Lx in (2, 4, 8):
            lam = LAMBDAS[8:8 + 2 * Lx:2]
            med, best = timeit(lambda: ops.quantize(mu, sg, tab, lam, N=N_BITS, layout="cb", out_idx=idx[:Lx]))
            out(f"L={Lx} raw (mid sweep)", med, best, f"{E*Lx/med/1e6:.1f} G latents/s")
    if "k2" in what:
        ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx, level_len=ll)
        cnt = torch.zeros((L, C, 2047), dtype=torch.int32, device=dev)
        med, best = timeit(lambda: ops.histogram_models(idx, C, cnt, N=N_BITS))
        out("K2 histogram_models L=32", med, best, f"frac {2*L*E/med/1e6/8000:.3f}")
        lut = torch.rand(rows + 4096, device=dev)                   # any table: the lookup's cost does not depend on its values
        mdl = torch.empty((L, C, 2047), dtype=torch.float32, device=dev)
        med, best = timeit(lambda: ops.histogram_models(idx, C, cnt, N=N_BITS, lut=lut, models=mdl))
        out("K2 + model lookup in the flush", med, best, f"frac {2*L*E/med/1e6/8000:.3f}")
    if "k1nt" in what:
        from vbq_amd import embeddings as Emb
        n = 10_000_000
        m_h, s_h, _ = make_inputs(n, 1, 1000)
        m, s = torch.from_numpy(m_h.reshape(n)).to(dev), torch.from_numpy(s_h.reshape(n)).to(dev)
        pts_h, _ = Emb.make_code_book(Emb.empirical_std(m), N_BITS)
        cb = torch.from_numpy(pts_h).to(dev)
        for Lx in (32, 1):
            betas = [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(1e5), 32))][:: 32 // Lx]
            ix = torch.empty((Lx, n), dtype=torch.uint16, device=dev)
            med, best = timeit(lambda: ops.quantize_notebook(m, s, cb, betas, N=N_BITS, want_values=False, out_idx=ix))
            out(f"K1nt notebook L={Lx} 1e7", med, best, f"frac {(8+2*Lx)*n/med/1e6/8000:.3f}")


if __name__ == "__main__":
    main()
