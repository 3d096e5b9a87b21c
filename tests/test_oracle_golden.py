"""The NumPy oracle against the golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
from scipy.stats import norm

from oracle import c_oracle as CO
from oracle import vbq_oracle as O

N = 10


def test_g1_xi_grid(golden):
    g = golden("g1_xi_grid.npz")
    assert np.array_equal(O.dyadic_xi(int(g["N"])), g["xi"])
    # level-major slot -> sorted rank is a bijection onto 0..T-1 and sorts xi
    r = O.level_major_to_rank(N)
    assert np.array_equal(np.sort(r), np.arange(2 ** (N + 1) - 1))
    assert np.array_equal(g["xi"][np.argsort(r)], np.arange(1, 2 ** (N + 1)) / 2.0 ** (N + 1))
    assert np.array_equal(O.levels_of_sorted_ranks(N)[r], np.concatenate([[n] * 2 ** n for n in range(N + 1)]))


def test_g2_docstring_known_answers(golden):
    g = golden("g2_n_bit_interval.npz")
    for x, n, l, r in zip(g["x"], g["n"], g["left"], g["right"]):
        assert O.get_n_bit_interval(float(x), int(n)) == (l, r)
    assert O.get_n_bit_interval(0.4375, 2) == (0.375, 0.625)          # utils.py:31-32
    assert O.get_n_bit_interval(0.004375, 2) == (0.125, 0.125)        # utils.py:82
    assert O.get_n_bit_interval(0.04375, 5) == (0.015625, 0.046875)   # utils.py:83


def test_g3_xi_intervals(golden):
    g = golden("g3_xi_intervals.npz")
    L, R = O.xi_intervals(g["x"], int(g["N"]))
    assert np.array_equal(L, g["left"])
    assert np.array_equal(R, g["right"])


def test_g4_encode_vectorized(golden):
    g = golden("g4_encode_vectorized.npz")
    mu, sigma = g["mu"], g["sigma"]
    fun = lambda z: norm.logpdf(z, loc=mu, scale=sigma)
    for i, lamb in enumerate(g["lambs"]):
        r = O.encode_vectorized(fun, mu, float(lamb), norm.cdf, norm.ppf, int(g["N"]))
        assert np.array_equal(r["num_bits"], g["num_bits"][i])
        assert np.array_equal(r["xi_hat"], g["xi_hat"][i])
        assert np.array_equal(r["z_hat"], g["z_hat"][i])
        assert r["score"] == g["score"][i]


def _image_oracle(g):
    C = g["mu"].shape[1]
    orc = O.ChannelwiseOracle(C, int(g["N"]))
    orc.build_code_points(O.factored_gaussian_icdf(g["ch_mean"], g["ch_std"]))
    assert np.array_equal(orc.all_code_points, g["all_code_points"])
    return orc


def test_g5_batch_quantize_f32_and_f64(golden):
    g = golden("g5_batch_quantize.npz")
    orc = _image_oracle(g)
    lam = g["lambdas"]
    Z, bits = orc.compress_batch(g["mu"], g["sigma"], [np.float32(l) for l in lam], mode="f32")
    assert np.array_equal(Z, g["zhat_f32"])
    assert np.array_equal(bits, g["bits_f32"])
    Z, bits = orc.compress_batch(g["mu"], g["sigma"], [float(l) for l in lam], mode="f64")
    assert np.array_equal(Z, g["zhat_f64"])
    assert np.array_equal(bits, g["bits_f64"])
    # round-trip invariant of quantizer.py:136-137
    q = O.qidx_lookup(orc.by_channel, g["zhat_f32"][7])
    assert np.array_equal(np.take_along_axis(orc.by_channel, q, axis=1), g["zhat_f32"][7].T)


def test_g6_algorithm1_equals_exhaustive_search(golden):
    g5 = golden("g5_batch_quantize.npz")
    g6 = golden("g6_brute_force.npz")
    orc = _image_oracle(g5)
    rows = g6["rows"]
    lens_sorted = np.repeat(O.levels_of_sorted_ranks(N)[None], orc.C, axis=0).astype(np.float32)
    for a, li in enumerate(g6["lam_idx"]):
        lamb = np.float32(g5["lambdas"][li])
        # reference exhaustive result == our exhaustive restatement
        for b, r in enumerate(rows):
            z, nb, _ = O.brute_force_solve(g5["mu"][r], g5["sigma"][r], orc.by_channel, lens_sorted, lamb)
            assert np.array_equal(z, g6["zhat"][a, b])
            assert np.array_equal(nb, g6["bits"][a, b])
        # ... and == the 21-candidate result captured from batch_quantize_indep_dims
        assert np.array_equal(g6["zhat"][a], g5["zhat_f32"][li][rows])
        assert np.array_equal(g6["bits"][a].astype(np.int32), g5["bits_f32"][li][rows])


def test_g8_corrected_lengths(golden):
    g5 = golden("g5_batch_quantize.npz")
    g8 = golden("g8_corrected_lengths.npz")
    orc = _image_oracle(g5)
    lam = [np.float32(l) for l in g5["lambdas"]]
    (Z1, b1), (Z2, b2) = orc.build_entropy_models(g5["mu"], g5["sigma"], lam, add_n_smoothing=1)
    assert np.array_equal(Z1, g5["zhat_f32"])
    assert np.array_equal(np.stack([orc.raw_models[l] for l in lam]), g8["raw_models"])
    assert np.array_equal(Z2, g8["zhat"])
    # entropy models are proper code-length tables
    for l in lam:
        m = orc.entropy_models[l]
        assert m.shape == (orc.C, orc.T) and m.dtype == np.float32
        assert np.allclose(np.sum(2.0 ** (-m.astype(np.float64)), axis=1), 1.0, atol=1e-4)
    out = orc.compress_latents(g5["mu"], g5["sigma"], lam)
    assert set(out) == {"Z_hat", "raw_num_bits", "num_bits_cl", "num_bits"}
    assert np.array_equal(out["Z_hat"][lam[3]], Z2[3])


def test_g7_notebook(golden):
    g = golden("g7_notebook.npz")
    means, stds = g["means"], g["stds"]
    es = O.empirical_std(means)
    assert es.dtype == np.float32 and es == g["empirical_std"]
    pts, lens = O.notebook_code_book(es, 10)
    assert np.array_equal(pts, g["codepoints"]) and np.array_equal(lens, g["lengths"])
    for i, beta in enumerate(g["betas"]):
        out = O.compress_coordinates(means, stds, float(beta), pts, lens)
        assert out.dtype == np.float32
        assert np.array_equal(out, g["optima"][i])
        assert O.empirical_entropy(out) == g["entropy"][i]


def test_rank_formulation_matches_grid_search():
    """The merged-table rank arithmetic (what the C oracle fast path and the HIP
    kernel use) reproduces quantizer.py:65-80 at every level."""
    rng = np.random.default_rng(5)
    for C, scale in ((1, 1.2329), (3, 0.4)):
        orc = O.ChannelwiseOracle(C, N)
        orc.build_code_points(O.factored_gaussian_icdf(np.zeros(C), np.full(C, scale)))
        t = orc.by_channel
        z = np.concatenate([rng.normal(0, scale * 1.5, 20000), t[0][rng.integers(0, 2047, 3000)],
                            [50, -50, t[0][0], t[0][-1], np.nextafter(t[0][-1], np.float32(9)),
                             np.nextafter(t[0][0], np.float32(-9))]]).astype(np.float32)
        Z = np.repeat(z[:, None], C, axis=1)
        left, right = O.get_all_N_bit_intervals(orc.grids, Z)
        for c in range(C):
            assert np.all(np.diff(t[c]) > 0)
            kl, kr = O.interval_ranks(t[c], z, N)
            assert np.array_equal(t[c][kl - 1], left[c])
            assert np.array_equal(t[c][kr - 1], right[c])
            # level-major slot <-> rank
            assert np.array_equal(t[c][O.level_major_to_rank(N)], orc.all_code_points[c])


def test_bmshj_self_consistency():
    rng = np.random.default_rng(3)
    C = 5
    mats, bias, fac = O.BMSHJ2018Oracle.init_params(C, init_scale=1.0, rng=rng)
    fac = [f + rng.normal(0, 0.5, f.shape).astype(np.float32) for f in fac]
    mats = [m + rng.normal(0, 0.3, m.shape).astype(np.float32) for m in mats]
    p = O.BMSHJ2018Oracle(*O.BMSHJ2018Oracle.effective(mats, bias, fac))
    x = np.sort(rng.normal(0, 2, (400, C)).astype(np.float32), axis=0)
    cdf, pdf = p.cdf_pdf(x)
    assert np.all(np.diff(cdf, axis=0) >= 0) and np.all(pdf > 0)
    assert np.array_equal(cdf, p.cdf(x))
    h = 1e-2
    fd = (p.cdf(x + np.float32(h)).astype(np.float64) - p.cdf(x - np.float32(h))) / (2 * h)
    assert np.allclose(fd, pdf, rtol=2e-2, atol=2e-4)
    xi = np.repeat(O.dyadic_xi(6)[:, None], C, axis=1)
    z = p.inverse_cdf(xi)
    assert z.dtype == np.float32
    assert np.allclose(p.cdf(z), xi, atol=2e-6)


def test_cfg1_single_image_plumbing():
    """BASELINE.json configs[0]: one Kodak image's latents ([1536 x C], here C = 32), one lambda, on the
    CPU restatement only -- the reference's own CPU-runnable case, end to end through the two
    entropy-model passes and the compress_latents dict (quantizer.py:82-150, 190-240)."""
    rng = np.random.default_rng(42)
    B, C = 1536, 32
    scale = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(np.zeros(C), scale))
    mu = (scale * rng.normal(0, 1, (B, C))).astype(np.float32)
    sg = np.clip(np.exp(rng.normal(-2, 0.7, (B, C))), 1e-4, 10).astype(np.float32)
    lam = [np.float32(0.5)]
    orc.build_entropy_models(mu, sg, lam, add_n_smoothing=1)
    out = orc.compress_latents(mu, sg, lam)
    z, nb = out["Z_hat"][lam[0]], out["num_bits"][lam[0]]
    assert z.shape == (B, C) and nb.shape == (B, C) and nb.dtype == np.float32
    q = O.qidx_lookup(orc.by_channel, z)
    assert np.array_equal(np.take_along_axis(orc.by_channel, q, axis=1), z.T)          # quantizer.py:136-137
    bpl = float(nb.sum()) / (B * C)                                                    # bits per latent
    assert 0.5 < bpl < 8.0
    d = O.lagrangian(mu, sg, z, out["raw_num_bits"][lam[0]], 0.5)
    assert np.isfinite(d) and d > 0


def test_g9_baseline_quantizers(golden):
    """f3: oracle restatement of UniformQuantizer / KmeansQuantizer.quantize vs the reference classes."""
    from oracle import vbq_oracle as o
    g = golden("g9_baselines.npz")
    x = g["x"]
    for K in (4, 16, 61):
        f = o.uniform_fit(x, K, 1)
        assert f["min"] == g[f"u{K}_min"] and f["delta"] == g[f"u{K}_delta"]
        assert np.array_equal(f["code_points"], g[f"u{K}_code_points"])
        assert np.array_equal(f["code_lengths"], g[f"u{K}_code_lengths"])
        q, I, nb = o.uniform_quantize(x, f)
        assert q.dtype == g[f"u{K}_q"].dtype and np.array_equal(q, g[f"u{K}_q"])
        assert np.array_equal(I, g[f"u{K}_I"]) and np.array_equal(nb, g[f"u{K}_bits"])
    q, I = o.nearest_code(x, g["k_centers"])
    assert np.array_equal(q, g["k_q"]) and np.array_equal(I, g["k_I"])


def test_g10_prediction_ranks(golden):
    """f4: the f64 restatement and the C checker (fma chain in k order, what the GPU path computes) against the
    notebook's prediction_ranks run on NumPy/BLAS f32: equal except where another word's score is within rounding
    of the ground truth, and then by at most the number of such words."""
    from oracle import c_oracle as CO
    from oracle import vbq_oracle as o
    g = golden("g10_analogy.npz")
    emb, an = g["emb"], g["analogies"]
    for e, want in ((emb, g["ranks"]), (g["quantized_7"].astype(np.float32), g["ranks_q7"])):
        r64, near = o.prediction_ranks(e, an)
        assert np.all(np.abs(r64 - want) <= near)
        assert np.mean(r64 == want) > 0.95            # exact f64 breaks the f32 ties of integer-valued embeddings
        rc = CO.analogy_ranks(e, an, threads=4)
        assert np.all(np.abs(rc - want) <= near) and np.mean(rc == want) > 0.97
    assert np.array_equal(o.quantize_coordinates(emb, 7), g["quantized_7"])
    assert np.array_equal(o.quantize_coordinates(emb, 1023), g["quantized_1023"])
    assert o.quantize_coordinates(emb, 7).dtype == np.float32


def test_g11_image_metrics(golden):
    """f4: direct-sum restatement of mse / psnr / ms_ssim vs the reference module (fftconvolve): mse and psnr
    bit-exact (integer-valued sums), ms_ssim to 1e-9 relative."""
    from oracle import vbq_oracle as o
    g = golden("g11_image_metrics.npz")
    for k in "abc":
        x, y = g[f"{k}_x"], g[f"{k}_y"]
        assert np.array_equal(o.image_mse(x, y), g[f"{k}_mse"])
        assert np.array_equal(o.image_psnr(x, y), g[f"{k}_psnr"])
        np.testing.assert_allclose(o.ms_ssim(x, y), g[f"{k}_msssim"], rtol=1e-9, atol=0)


def test_g12_duplicates_and_edges(golden):
    """Reference exhaustive solves on tables with repeated float32 code points and on inputs beyond / at the rim of
    every level's grid: the restated Algorithm-1 glue (per-level searchsorted on padded grids, 21 candidates) and the
    C oracle's rank formulation must return the same value and the same bit length."""
    g = golden("g12_duplicates_edges.npz")
    C = g["mu"].shape[1]
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(g["ch_mean"], g["ch_std"]))
    assert np.array_equal(orc.all_code_points, g["all_code_points"])
    assert list(g["n_duplicates"]) == [T_ - np.unique(r).size for r, T_ in ((r, r.size) for r in orc.by_channel)]
    assert g["n_duplicates"][1] > 1500                                   # the table really is mostly repeats
    lam = [np.float32(l) for l in g["lambdas"]]
    Z, bits = orc.compress_batch(g["mu"], g["sigma"], lam, mode="f32")
    assert np.array_equal(Z, g["zhat"]) and np.array_equal(bits, g["bits"])
    idx = CO.quantize(g["mu"], g["sigma"], g["all_code_points"], list(g["lambdas"]), N=N)         # [L, B, C]
    lev = O.levels_of_sorted_ranks(N)
    for l in range(len(lam)):
        assert np.array_equal(np.take_along_axis(orc.by_channel, idx[l].T.astype(np.int64), axis=1).T, g["zhat"][l])
        assert np.array_equal(lev[idx[l]], g["bits"][l])
    # canonical qidx (quantizer.py:135): the first sorted position of the value, whatever rank the solver reports
    q = O.qidx_lookup(orc.by_channel, g["zhat"][5])
    assert np.array_equal(np.take_along_axis(orc.by_channel, q, axis=1), g["zhat"][5].T)
    assert np.all(q <= idx[5].T)


def test_g13_notebook_chain(golden):
    """The notebook's stages chained: NumPy float32 moment (restated in C, block by block) -> code book from that very
    number -> brute-force compress_coordinates -> entropy."""
    g = golden("g13_notebook_chain.npz")
    means, stds = g["means"], g["stds"]
    s = CO.numpy_sum_sq_f32(means)
    assert s == np.sum(means.ravel() ** 2)
    es = np.sqrt(np.float32(float(s) / means.size))
    assert es.dtype == np.float32 and es == g["empirical_std"] == O.empirical_std(means)
    pts, lens = O.notebook_code_book(es, 10)
    assert np.array_equal(pts, g["codepoints"]) and np.array_equal(lens, g["lengths"])
    for i, beta in enumerate(g["betas"]):
        val, _ = CO.compress_coordinates(means, stds, float(beta), pts, lens, threads=4)
        assert np.array_equal(val, g["optima"][i])
        assert O.empirical_entropy(val) == g["entropy"][i]
