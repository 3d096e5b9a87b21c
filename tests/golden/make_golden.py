#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE's own
NumPy code (read-only, from /root/reference) in the build container.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Only data (inputs + the reference's outputs) is written; no reference source travels.
The reference tree does not exist on the GPU box, so this script is never run there.

What is executed from the reference:
  * img-compression/utils.py, imported as a module.  It does `from numba import jit`
    at :211; numba is not installed here, so an identity decorator is registered
    under that name first (`@jit(nopython=True)` does not change semantics).
  * the notebook cells that define the word-embedding VBQ (cells 25, 26, 28, 29 of
    word-embeddings/compress-trained-word-embeddings.ipynb), exec'd from the JSON.
What cannot be executed (TensorFlow is absent): quantizer.py and learned_prior.py.
The candidate tensors fed to the reference's batch_quantize_indep_dims are therefore
built by oracle/vbq_oracle.py (our restatement of quantizer.py:25-80,156-183); G6
ties that construction back to the reference through its exhaustive
quantize_indep_dims over the full code book.

NumPy-version note: the reference ran under NumPy 1.17 where a
np.float64 scalar times an f32 array stays f32.  Under NumPy 2 that needs a Python
float / np.float32 scalar, which is what is passed below.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
from scipy.stats import norm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VBQ_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

_nb = types.ModuleType("numba")
_nb.jit = lambda *a, **k: (lambda f: f)
sys.modules.setdefault("numba", _nb)
sys.path.insert(0, os.path.join(REF, "img-compression"))
import utils as ref_utils  # noqa: E402  (the reference module)

from oracle import vbq_oracle as O  # noqa: E402

N = 10
LAMBDAS32 = (2.0 ** np.linspace(-8, 7.5, 32))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


# ---------------------------------------------------------------- G1 / G2 / G3
def g1_g2_g3():
    xi = np.concatenate([np.array(ref_utils.n_bit_binary_floats(n), dtype=np.float64) for n in range(N + 1)])
    save("g1_xi_grid.npz", xi=xi, N=np.int64(N))

    rng = np.random.default_rng(101)
    xs = np.concatenate([[0.4375, 0.004375, 0.04375, 0.0, 1.0, 0.5, 0.25, 0.75, 0.999, 1e-9],
                         rng.uniform(0, 1, 40)])
    ns = np.concatenate([[2, 2, 5, 3, 3, 1, 2, 2, 10, 10], rng.integers(0, 11, 40)]).astype(np.int64)
    lr = np.array([ref_utils.get_n_bit_interval(float(x), int(n)) for x, n in zip(xs, ns)], dtype=np.float64)
    save("g2_n_bit_interval.npz", x=xs, n=ns, left=lr[:, 0], right=lr[:, 1])

    x = np.concatenate([rng.uniform(0, 1, 300), [0.0, 1.0, 0.5, 2.0 ** -11, 1 - 2.0 ** -11, 0.25, 0.375]])
    L = np.empty((N + 1, x.shape[0]))
    R = np.empty_like(L)
    ref_utils.get_all_N_bit_intervals(x, N, L, R)
    save("g3_xi_intervals.npz", x=x, left=L, right=R, N=np.int64(N))


# ---------------------------------------------------------------- G4
def g4():
    rng = np.random.default_rng(104)
    K = 200
    mu = rng.normal(0, 1.0, K)
    sigma = np.exp(rng.normal(-2, 0.7, K))
    fun = ref_utils.curry_normal_logpdf(loc=mu, scale=sigma, ignore_const=False, backend=np)
    lambs = np.array([2.0 ** -8, 0.1, 1.0, 7.3, 128.0])
    zh, nb, xh, sc = [], [], [], []
    for lamb in lambs:
        r = ref_utils.encode_vectorized(fun, mu, float(lamb), norm.cdf, norm.ppf, max_bits_per_coord=N)
        zh.append(r["z_hat"]); nb.append(r["num_bits"]); xh.append(r["xi_hat"]); sc.append(r["score"])
    save("g4_encode_vectorized.npz", mu=mu, sigma=sigma, lambs=lambs, z_hat=np.array(zh),
         num_bits=np.array(nb), xi_hat=np.array(xh), score=np.array(sc), N=np.int64(N))


# ---------------------------------------------------------------- G5 / G6 / G8
def make_image_case(seed, B, C):
    rng = np.random.default_rng(seed)
    ch_mean = rng.normal(0, 0.3, C)
    ch_std = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(ch_mean, ch_std))
    mu = (ch_mean + ch_std * rng.normal(0, 1, (B, C))).astype(np.float32)
    sigma = np.clip(np.exp(rng.normal(-2, 0.7, (B, C))), 1e-4, 10).astype(np.float32)
    # edge rows: exact code-point hits, far outside the table, tiny / huge sigma, table min / max
    for c in range(C):
        t = orc.by_channel[c]
        mu[0, c] = t[rng.integers(0, t.shape[0])]
        mu[1, c] = np.float32(50.0)
        mu[2, c] = np.float32(-50.0)
        mu[3, c] = t[0]
        mu[4, c] = t[-1]
        mu[5, c] = t[1023]                       # the 0-bit point
        mu[6, c] = np.nextafter(t[1023], np.float32(np.inf))
        mu[7, c] = 0.5 * (t[1000] + t[1001])
    sigma[8] = np.float32(1e-4)
    sigma[9] = np.float32(10.0)
    return orc, mu, sigma, ch_mean, ch_std


def g5_g6_g8():
    B, C = 192, 4
    orc, mu, sigma, ch_mean, ch_std = make_image_case(105, B, C)
    left, right = O.get_all_N_bit_intervals(orc.grids, mu)
    P = O.assemble_candidates(left, right)
    Lraw = O.raw_code_lengths(N, B, C)
    fun = ref_utils.curry_normal_logpdf(loc=mu, scale=sigma, ignore_const=True, backend=object())  # the TF-branch closure
    # f32 op-by-op mode (what the TF path computes): lengths cast to f32, lambda as np.float32
    lam32 = [np.float32(l) for l in LAMBDAS32]
    Zd, Bd = ref_utils.batch_quantize_indep_dims((B, C), P, Lraw.astype(np.float32), fun, lambs=lam32,
                                                 backend=np, return_np=True)
    Z32 = np.stack([Zd[l] for l in lam32]); B32 = np.stack([Bd[l] for l in lam32])
    # as-written NumPy mode: int lengths, Python-float lambda -> f64 scores
    lam64 = [float(l) for l in LAMBDAS32]
    Zd, Bd = ref_utils.batch_quantize_indep_dims((B, C), P, Lraw, fun, lambs=lam64, backend=np, return_np=True)
    Z64 = np.stack([Zd[l] for l in lam64]); B64 = np.stack([Bd[l] for l in lam64])
    save("g5_batch_quantize.npz", mu=mu, sigma=sigma, ch_mean=ch_mean, ch_std=ch_std,
         all_code_points=orc.all_code_points, lambdas=LAMBDAS32, N=np.int64(N),
         zhat_f32=Z32, bits_f32=B32.astype(np.int32), zhat_f64=Z64, bits_f64=B64.astype(np.int32))

    # G8: corrected-length second pass.  Raw models come from the pass-1 bit histogram
    # (our restatement of quantizer.py:99-110); the solve is the reference's.
    raw_models = []
    for i in range(len(lam32)):
        counts = np.array([np.bincount(B32[i][:, c].astype(np.int64), minlength=N + 1) for c in range(C)])
        raw_models.append(O.neg_log2_freq(counts, 1))
    L4 = O.corrected_code_lengths(N, B, raw_models)
    Zd, Bd = ref_utils.batch_quantize_indep_dims((B, C), P, L4, fun, lambs=lam32, backend=np, return_np=True)
    # Only Z_hat is kept: the NumPy branch does `code_indices.choose(L)` on the whole 4-D
    # length stack (utils.py:404) instead of L[i], so its num_bits output is not the
    # per-lambda value the TF branch (utils.py:407-415) returns.
    Zc = np.stack([Zd[l] for l in lam32])
    save("g8_corrected_lengths.npz", raw_models=np.stack(raw_models), zhat=Zc)

    # G6: exhaustive search over all 2047 sorted code points, reference quantize_indep_dims, per row.
    rank_levels = O.levels_of_sorted_ranks(N)
    lens_sorted = np.repeat(rank_levels[None, :], C, axis=0).astype(np.float32)
    rows = np.arange(0, B)                   # every row (edge rows included) x every lambda: 24 576 exhaustive solves
    lam_sel = list(range(len(lam32)))
    zh = np.empty((len(lam_sel), rows.shape[0], C), np.float32)
    nb = np.empty((len(lam_sel), rows.shape[0], C), np.float32)
    for a, li in enumerate(lam_sel):
        for b, r in enumerate(rows):
            f_row = ref_utils.curry_normal_logpdf(loc=mu[r], scale=sigma[r], ignore_const=True, backend=object())
            z, n_ = ref_utils.quantize_indep_dims(mu[r], orc.by_channel, lens_sorted, f_row, lam32[li], backend=np)
            zh[a, b], nb[a, b] = z, n_
    save("g6_brute_force.npz", rows=rows, lam_idx=np.array(lam_sel), zhat=zh, bits=nb)


# ---------------------------------------------------------------- G7 (notebook)
def g7():
    nb_path = os.path.join(REF, "word-embeddings", "compress-trained-word-embeddings.ipynb")
    cells = json.load(open(nb_path))["cells"]
    src = {i: "".join(c["source"]) for i, c in enumerate(cells) if c["cell_type"] == "code"}
    rng = np.random.default_rng(107)
    V, D = 250, 12
    vecs_u = rng.normal(-0.0799, 1.2329, (V, D)).astype(np.float32)
    stds_u = np.clip(np.exp(rng.normal(-2, 0.7, (V, D))), 1e-4, 10).astype(np.float32)
    vecs_u[0, 0] = 40.0
    vecs_u[0, 1] = -40.0
    import collections
    import scipy.stats
    ns = dict(np=np, scipy=scipy, Counter=collections.Counter, vecs_u=vecs_u, stds_u=stds_u, print=lambda *a, **k: None)
    for cid, must in ((25, "empirical_std"), (26, "codepoints_and_lengths"), (28, "def compress_coordinates"),
                      (29, "def empirical_entropy")):
        assert must in src[cid], (cid, src[cid][:80])
        exec(src[cid], ns)
    betas = [0.01, 0.37, 1.0, 9.5, 312.0, 1e5]
    outs, ents = [], []
    for beta in betas:
        opt, _ = ns["compress_coordinates"](vecs_u, stds_u, float(beta))
        outs.append(opt.copy())
        ents.append(ns["empirical_entropy"](opt))
    save("g7_notebook.npz", means=vecs_u, stds=stds_u, empirical_std=np.asarray(ns["empirical_std"]),
         codepoints=ns["codepoints"], lengths=ns["lengths"], betas=np.array(betas),
         optima=np.stack(outs), entropy=np.array(ents))


# ---------------------------------------------------------------- G12 (duplicate code points, inputs beyond the table)
def g12():
    """The reference's exhaustive solver (utils.quantize_indep_dims over all 2047 sorted points) on what the restated
    TF glue is most likely to get wrong: (a) tables whose float32 cast REPEATS code points (a narrow prior far from
    zero: hundreds of equal neighbours, across and within bit levels -- the canonical-qidx case of quantizer.py:135);
    (b) inputs beyond both ends of the table by many different margins and just inside the outermost interval of every
    bit level (the edge padding of the per-level grids, quantizer.py:54-57,75-76).  Algorithm 1's 21 candidates must
    still contain the exhaustive optimum, with the same value and the same bit length."""
    rng = np.random.default_rng(112)
    C = 3
    ch_mean = np.array([100.0, 1000.0, 0.3])
    ch_std = np.array([1e-3, 1e-3, 1.7])
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(ch_mean, ch_std))
    srt = orc.by_channel                                                        # C x T sorted (== rank order)
    n_dup = [int(T_ - np.unique(srt[c]).size) for c, T_ in enumerate([srt.shape[1]] * C)]
    assert n_dup[0] > 50 and n_dup[1] > 1500 and n_dup[2] == 0, n_dup
    rows = []
    for c in range(C):
        t = srt[c]
        span = float(t[-1] - t[0]) if t[-1] > t[0] else 1.0
        col = list(ch_mean[c] + ch_std[c] * rng.normal(0, 1.5, 60))             # ordinary inputs
        col += list(t[rng.integers(0, t.size, 20)])                             # exact hits (repeated values included)
        for k in (1e-7, 1e-3, 0.01, 0.1, 0.5, 1, 3, 10, 100):                   # beyond both ends
            col += [t[-1] + k * span, t[0] - k * span]
        for n in range(1, N + 1):                                               # outermost interval of every level
            lv = orc.all_code_points[c, 2 ** n - 1: 2 ** (n + 1) - 1]
            col += [0.5 * (float(lv[-1]) + float(t[-1])), 0.5 * (float(lv[0]) + float(t[0])),
                    np.nextafter(lv[-1], np.float32(np.inf)), np.nextafter(lv[0], np.float32(-np.inf))]
        rows.append(np.array(col, dtype=np.float32))
    B = min(len(r) for r in rows)
    mu = np.stack([r[:B] for r in rows], axis=1)                                # B x C
    sigma = (ch_std[None, :] * np.exp(rng.normal(-2, 1.0, (B, C)))).astype(np.float32)
    lens_sorted = np.repeat(O.levels_of_sorted_ranks(N)[None, :], C, axis=0).astype(np.float32)
    lam32 = [np.float32(l) for l in LAMBDAS32]
    zh = np.empty((len(lam32), B, C), np.float32)
    nb = np.empty((len(lam32), B, C), np.float32)
    for a, lamb in enumerate(lam32):
        for r in range(B):
            f_row = ref_utils.curry_normal_logpdf(loc=mu[r], scale=sigma[r], ignore_const=True, backend=object())
            z, n_ = ref_utils.quantize_indep_dims(mu[r], srt, lens_sorted, f_row, lamb, backend=np)
            zh[a, r], nb[a, r] = z, n_
    save("g12_duplicates_edges.npz", ch_mean=ch_mean, ch_std=ch_std, all_code_points=orc.all_code_points, mu=mu,
         sigma=sigma, lambdas=LAMBDAS32, zhat=zh, bits=nb.astype(np.int32), n_duplicates=np.array(n_dup))


# ---------------------------------------------------------------- G13 (the notebook's chain, end to end)
def g13():
    """Cells 25, 26, 28, 29 of the notebook run as a CHAIN on one seeded embedding: the float32 moment (np.mean of
    21 000 squares: more than two of NumPy's 8192-element summation blocks plus a ragged one) -> the code book built from
    THAT number -> compress_coordinates -> empirical_entropy.  g7 feeds the stages separately; this fixes what the
    notebook actually computes when one stage's rounding feeds the next."""
    nb_path = os.path.join(REF, "word-embeddings", "compress-trained-word-embeddings.ipynb")
    cells = json.load(open(nb_path))["cells"]
    src = {i: "".join(c["source"]) for i, c in enumerate(cells) if c["cell_type"] == "code"}
    rng = np.random.default_rng(113)
    V, D = 3000, 7
    vecs_u = rng.normal(-0.0799, 1.2329, (V, D)).astype(np.float32)
    stds_u = np.clip(np.exp(rng.normal(-2, 0.7, (V, D))), 1e-4, 10).astype(np.float32)
    import collections
    import scipy.stats
    ns = dict(np=np, scipy=scipy, Counter=collections.Counter, vecs_u=vecs_u, stds_u=stds_u, print=lambda *a, **k: None)
    for cid, must in ((25, "empirical_std"), (26, "codepoints_and_lengths"), (28, "def compress_coordinates"),
                      (29, "def empirical_entropy")):
        assert must in src[cid], (cid, src[cid][:80])
        exec(src[cid], ns)
    assert np.asarray(ns["empirical_std"]).dtype == np.float32
    betas = [0.02, 0.6, 17.0, 2500.0]
    outs, ents = [], []
    for beta in betas:
        opt, _ = ns["compress_coordinates"](vecs_u, stds_u, float(beta))
        outs.append(opt.copy())
        ents.append(ns["empirical_entropy"](opt))
    save("g13_notebook_chain.npz", means=vecs_u, stds=stds_u, empirical_std=np.asarray(ns["empirical_std"]),
         codepoints=ns["codepoints"], lengths=ns["lengths"], betas=np.array(betas), optima=np.stack(outs),
         entropy=np.array(ents))


# ---------------------------------------------------------------- G9 (baseline quantizers)
def g9():
    """UniformQuantizer / KmeansQuantizer.quantize (quantizer.py:259-333).  quantizer.py imports tensorflow at
    module level, so the two NumPy-only classes are exec'd from their source lines (nothing is copied into
    the repo: only their outputs are saved)."""
    src = open(os.path.join(REF, "img-compression", "quantizer.py")).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith("class UniformQuantizer"))
    stop = next(i for i, l in enumerate(src) if l.startswith("class ChannelwiseSimpleQuantizer:"))
    ns = {"np": np}
    exec("\n".join(src[start:stop]), ns)
    rng = np.random.default_rng(109)
    x = rng.normal(0.2, 1.3, 5000).astype(np.float32)
    out = {"x": x}
    for K in (4, 16, 61):
        u = ns["UniformQuantizer"](K)
        u.fit(x, add_n_smoothing=1)      # int: the float default cannot be added in place to the int64 counts (:282)
        qz, I, nb = u.quantize(x)
        out[f"u{K}_min"], out[f"u{K}_delta"] = np.asarray(u.min), np.asarray(u.delta)
        out[f"u{K}_code_points"], out[f"u{K}_code_lengths"] = u.code_points, u.code_lengths
        out[f"u{K}_q"], out[f"u{K}_I"], out[f"u{K}_bits"] = qz, I, nb
    centers = np.sort(rng.normal(0, 1.5, 12)).astype(np.float64)[rng.permutation(12)]      # unsorted, as sklearn returns
    k = ns["KmeansQuantizer"](12)
    k.code_points = centers
    k.code_lengths = -np.log2(np.full(12, 1 / 12.0))
    qz, I, nb = k.quantize(x)
    out["k_centers"], out["k_q"], out["k_I"] = centers, qz, I
    save("g9_baselines.npz", **out)


# ---------------------------------------------------------------- G10 (analogy evaluator)
def g10():
    """prediction_ranks and quantize_coordinates (notebook cells 14, 36) exec'd from the JSON on a seeded
    synthetic embedding with planted analogies."""
    nb_path = os.path.join(REF, "word-embeddings", "compress-trained-word-embeddings.ipynb")
    cells = json.load(open(nb_path))["cells"]
    src = {i: "".join(c["source"]) for i, c in enumerate(cells) if c["cell_type"] == "code"}
    assert "def prediction_ranks" in src[14] and "def quantize_coordinates" in src[36]
    rng = np.random.default_rng(110)
    V, K, Q = 2000, 50, 500
    emb = rng.normal(0, 1, (V, K)).astype(np.float32)
    an = rng.integers(0, V, (Q, 4))
    # plant structure so that ranks are not all ~V/2: d = b - a + c + noise for the first 400 questions
    for i in range(300):
        a, b, c, d = an[i]
        emb[d] = emb[b] - emb[a] + emb[c] + rng.normal(0, 0.6 + 0.005 * i, K).astype(np.float32)
    ns = dict(np=np, analogies_id=an)
    exec(src[14], ns)
    exec(src[36], ns)
    ranks = ns["prediction_ranks"](emb)
    qz = {str(q): ns["quantize_coordinates"](emb, q) for q in (7, 1023)}
    assert all(v.dtype == np.float32 and np.array_equal(v, v.astype(np.int16)) for v in qz.values())
    ranks_q7 = ns["prediction_ranks"](qz["7"])
    save("g10_analogy.npz", emb=emb, analogies=an.astype(np.int32), ranks=ranks.astype(np.int64),
         ranks_q7=ranks_q7.astype(np.int64), **{f"quantized_{k}": v.astype(np.int16) for k, v in qz.items()})      # integer-valued f32, stored as int16


# ---------------------------------------------------------------- G11 (image metrics)
def g11():
    """img_comparison_metrics.mse / psnr / ms_ssim (NumPy + SciPy only: imported as is) on seeded uint8 batches:
    smooth images plus noise of increasing strength, three shapes incl. one smaller than the 11-tap window."""
    sys.path.insert(0, os.path.join(REF, "img-compression"))
    import img_comparison_metrics as M
    rng = np.random.default_rng(111)
    out = {}
    for name, (B, H, W, Cc) in {"a": (3, 72, 104, 3), "b": (2, 96, 80, 1), "c": (2, 40, 24, 2)}.items():
        yy, xx = np.mgrid[0:H, 0:W]
        base = 128 + 90 * np.sin(yy[None, :, :, None] / 7.0 + np.arange(B)[:, None, None, None]) * np.cos(
            xx[None, :, :, None] / 5.0 + np.arange(Cc)[None, None, None, :])
        x = np.clip(base + rng.normal(0, 6, (B, H, W, Cc)), 0, 255).astype(np.uint8)
        y = np.clip(x.astype(np.float64) + rng.normal(0, 1, (B, H, W, Cc)) * (3 + 9 * np.arange(B))[:, None, None, None], 0,
                    255).astype(np.uint8)
        out[f"{name}_x"], out[f"{name}_y"] = x, y
        out[f"{name}_mse"] = M.mse(x, y)
        out[f"{name}_psnr"] = M.psnr(x, y, max_val=255)
        out[f"{name}_msssim"] = M.ms_ssim(x, y, max_val=255)
    save("g11_image_metrics.npz", **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference tree not found; this script only runs in the build container"
    g1_g2_g3()
    g4()
    g5_g6_g8()
    g7()
    g12()
    g13()
    g9()
    g10()
    g11()
