"""Parity stress of the notebook kernels (K1nt thresholds for sweeps of >= 6 betas, K1n per beta otherwise) vs the literal
2047-point brute force (C oracle of compress_coordinates, ipynb:429-443).
    python tools/stress_notebook.py [n] [rounds] [seed offset]
Rounds cycle through: a short sweep, the notebook's 50 betas, an unsorted sweep with repeats across a launch-chunk boundary
(dense kernel), a randomly spaced unsorted sweep of 6 to 64 betas, and an ADVERSARIAL sweep whose betas put the penalty weight of chosen elements exactly on their own
level-change thresholds (T_n = max_j min_i (err_i - err_j) / (j - i), computed here in float64) -- the case K1nt's guard bands
exist for.  With VBQ_FAST_DEBUG=2 (bands off) that round must produce mismatches; with 0 and 1 none."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import c_oracle as CO, vbq_oracle as O
from vbq_amd import ops
N = 10
rank_of_slot = O.level_major_to_rank(N)


def threshold_betas(means, stds, pts, lens, rng, count):
    """Betas that make fl32(fl32(2 beta) * fl32(sigma^2)) land on (or within an ulp of) a level-change threshold of one
    element each: the element's own err_n (float64, as the notebook computes them) give T_n; beta = T_n / (2 sigma^2)."""
    out = []
    for e in rng.permutation(means.size)[:4 * count]:
        err = (pts - np.float64(means[e])) ** 2
        du = np.array([err[lens == n].min() for n in range(N + 1)])
        var = np.float32(stds[e]) * np.float32(stds[e])
        T = [max(min((du[i] - du[j]) / (j - i) for i in range(n + 1)) for j in range(n + 1, N + 1)) for n in range(N)]
        T = [t for t in T if t > 0]
        if not T:
            continue
        t = T[int(rng.integers(0, len(T)))]
        b = np.float32(t / np.float64(var))                   # the f32 number whose product with var is (nearly) T
        beta = float(np.float64(b) / 2.0)
        if 1e-10 < beta < 1e15:
            out.append(beta)
        if len(out) >= count:
            break
    out = sorted(set(out))
    keep = [out[0]]
    for v in out[1:]:
        if v > keep[-1] * 1.05:                               # one bucket apart at least, so that K1nt takes the sweep
            keep.append(v)
    span = [v for v in keep if v < keep[0] * 2.0 ** 22]
    return span[:64]


def main():
    rng = np.random.default_rng(11 + (int(sys.argv[3]) if len(sys.argv) > 3 else 0))
    dev = torch.device("cuda")
    tot = bad = 0
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
        scale = float(np.exp(rng.uniform(np.log(0.05), np.log(20))))
        means = (scale * rng.standard_t(4, n)).astype(np.float32)
        stds = (np.exp(rng.normal(-2, 1.5, n)) * scale).astype(np.float32)
        pts, lens = O.notebook_code_book(np.float32(scale), N)
        srt = np.sort(pts)
        k = n // 10
        means[:k] = srt[rng.integers(0, 2047, k)].astype(np.float32)
        j = rng.integers(0, 2046, k)
        means[k:2 * k] = (0.5 * (srt[j] + srt[j + 1])).astype(np.float32)
        kind = rnd % 5
        if kind == 3:
            betas = threshold_betas(means, stds, pts, lens, rng, 48)
        elif kind == 4:                                    # random length, spacing and order
            betas = [float(b) for b in np.exp(rng.uniform(np.log(1e-4), np.log(1e7), int(rng.integers(6, 65))))]
        else:
            nbeta = [10, 50, 33][kind]
            betas = list(np.exp(np.linspace(np.log(0.01), np.log(1e5), nbeta)))
            if kind == 2:                                  # unsorted, with duplicates and a second launch chunk boundary
                betas = list(rng.permutation(betas + betas[:7] + betas[10:40]))
        if n % 2 == 0 and rnd % 2 == 1:                    # odd length: the unvectorised emission path
            means, stds = means[:-1], stds[:-1]
        idx, val = ops.quantize_notebook(torch.from_numpy(means).to(dev), torch.from_numpy(stds).to(dev), torch.from_numpy(pts).to(dev), betas, N=N)
        idx, val = idx.cpu().numpy(), val.cpu().numpy()
        md, sd, pd_ = torch.from_numpy(means).to(dev), torch.from_numpy(stds).to(dev), torch.from_numpy(pts).to(dev)
        for i, b in enumerate(betas):
            v, slot = CO.compress_coordinates(means, stds, b, pts, lens, threads=CO.max_threads())
            nb = int(np.count_nonzero(idx[i].astype(np.int64) != rank_of_slot[slot])) + int(np.count_nonzero(val[i] != v))
            tot += means.size; bad += nb
            if i % 4 == 0:
                # the same beta on its own and paired with the next one: the calls of test_beta (ipynb:466) take the pruned
                # descent (K1np), with and without the values
                pair = [b] if i % 8 == 0 else [b, betas[(i + 1) % len(betas)]]
                i1, v1 = ops.quantize_notebook(md, sd, pd_, pair, N=N, want_values=(i % 3 != 0))
                nb = int(np.count_nonzero(i1[0].cpu().numpy().astype(np.int64) != rank_of_slot[slot]))
                if v1 is not None:
                    nb += int(np.count_nonzero(v1[0].cpu().numpy() != v))
                tot += means.size; bad += nb
        print(f"round {rnd}: scale {scale:.3g}, {len(betas)} betas{' (on thresholds)' if kind == 3 else ''}, "
              f"{len(betas) * means.size:.3g} latents, mismatches so far {bad}", flush=True)
    print("TOTAL", tot, "mismatches", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
