"""The C oracle (oracle/vbq_oracle.c) against the golden vectors and the NumPy oracle."""
import numpy as np

from oracle import c_oracle as CO
from oracle import vbq_oracle as O

N = 10


def _sorted_tables(all_pts):
    r = O.level_major_to_rank(N)
    s = np.empty_like(all_pts)
    s[:, r] = all_pts
    return s


def test_c_quantize_matches_reference_golden(golden):
    g = golden("g5_batch_quantize.npz")
    tab = g["all_code_points"]
    srt = _sorted_tables(tab)
    assert np.all(np.diff(srt, axis=1) > 0)
    lev = O.levels_of_sorted_ranks(N)
    for mode, zk, bk in ((0, "zhat_f32", "bits_f32"), (1, "zhat_f64", "bits_f64")):
        idx, zh, bt = CO.quantize(g["mu"], g["sigma"], tab, g["lambdas"], N=N, mode=mode, want_zhat=True,
                                  want_bits=True)
        assert np.array_equal(zh, g[zk])
        assert np.array_equal(bt.astype(np.int32), g[bk])
        for l in (0, 13, 31):    # rank index <-> value / level
            assert np.array_equal(np.take_along_axis(srt, idx[l].T.astype(np.int64), axis=1).T, g[zk][l])
            assert np.array_equal(lev[idx[l]], g[bk][l])


def test_c_quantize_layout_and_threads(golden):
    g = golden("g5_batch_quantize.npz")
    a = CO.quantize(g["mu"], g["sigma"], g["all_code_points"], g["lambdas"][:5], N=N)
    b = CO.quantize(g["mu"].T.copy(), g["sigma"].T.copy(), g["all_code_points"], g["lambdas"][:5], N=N, layout=1,
                    threads=4)
    assert np.array_equal(a, np.transpose(b, (0, 2, 1)))


def test_c_quantize_corrected_lengths(golden):
    g5, g8 = golden("g5_batch_quantize.npz"), golden("g8_corrected_lengths.npz")
    level_len = np.stack([O.corrected_level_lengths(N, m).T for m in g8["raw_models"]])    # [L, C, N+1]
    idx, zh, bt = CO.quantize(g5["mu"], g5["sigma"], g5["all_code_points"], g5["lambdas"], N=N, level_len=level_len,
                              want_zhat=True, want_bits=True)
    assert np.array_equal(zh, g8["zhat"])
    lev = O.levels_of_sorted_ranks(N)[idx]                                                  # [L, B, C]
    want = np.take_along_axis(level_len[:, None], lev[..., None].astype(np.int64), axis=3)[..., 0]
    assert np.array_equal(bt, want)


def test_c_notebook_matches_reference_golden(golden):
    g = golden("g7_notebook.npz")
    for i, beta in enumerate(g["betas"]):
        val, slot = CO.compress_coordinates(g["means"], g["stds"], float(beta), g["codepoints"], g["lengths"])
        assert np.array_equal(val, g["optima"][i])
        assert np.array_equal(g["codepoints"][slot].astype(np.float32), g["optima"][i])


def test_c_histogram_and_moments():
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 2047, (3, 50, 4)).astype(np.uint16)
    cnt = CO.histogram(idx, 4, N=N)
    for l in range(3):
        for c in range(4):
            assert np.array_equal(cnt[l, c], np.bincount(idx[l, :, c], minlength=2047))
    x = rng.normal(0, 1, (1000, 3)).astype(np.float32)
    m = CO.moments(x, 3)
    assert np.allclose(m[:, 0], x.astype(np.float64).sum(0)) and np.allclose(m[:, 1], (x.astype(np.float64) ** 2).sum(0))


def test_c_matches_numpy_oracle_on_random_tables():
    rng = np.random.default_rng(11)
    C, B = 3, 400
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(rng.normal(0, 1, C), np.exp(rng.normal(0, 1, C))))
    mu = rng.normal(0, 2, (B, C)).astype(np.float32)
    sg = np.exp(rng.normal(-2, 1, (B, C))).astype(np.float32)
    lam = 2.0 ** np.linspace(-8, 7.5, 7)
    Z, bits = orc.compress_batch(mu, sg, [np.float32(l) for l in lam])
    idx, zh, bt = CO.quantize(mu, sg, orc.all_code_points, lam, N=N, want_zhat=True, want_bits=True)
    assert np.array_equal(zh, Z) and np.array_equal(bt.astype(np.int32), bits)


def test_numpy_pairwise_sum_is_what_numpy_does():
    """The notebook's moment (ipynb:374) is np.mean of a 1-D float32 array: NumPy's pairwise float32 summation.  The C
    restatement of that order must reproduce np.sum bit for bit on every size class (below 8, one block, ragged blocks,
    deep recursion) -- NumPy itself, the library the reference runs on, is the pin here."""
    rng = np.random.default_rng(8)
    sizes = [0, 1, 7, 8, 9, 127, 128, 129, 255, 256, 257, 1000, 1023, 1024, 1025, 3001 * 5, 250 * 12, 99991, 100000,
             1 << 20, (1 << 20) + 13, 10_000_000]
    for n in sizes:
        x = rng.normal(-0.0799, 1.2329, n).astype(np.float32)
        want = np.sum(x ** 2) if n else np.float32(0)
        assert want.dtype == np.float32
        assert CO.numpy_sum_sq_f32(x) == want, n
        if n:
            assert np.sqrt(np.float32(CO.numpy_sum_sq_f32(x) / np.float32(n))) == np.sqrt(np.mean(x.ravel() ** 2))
