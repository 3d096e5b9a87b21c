"""Parity of the HIP path (through the C-ABI, via vbq_amd.ops) with the oracle and the golden
vectors captured from the reference.  Integer / index outputs must be bit-identical."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import c_oracle as CO
from oracle import vbq_oracle as O

pytestmark = pytest.mark.gpu
N = 10
T = 2047


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from vbq_amd import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def sorted_tables(all_pts):
    r = O.level_major_to_rank(N)
    s = np.empty_like(all_pts)
    s[:, r] = all_pts
    return s


# ------------------------------------------------------------------ golden vectors (reference outputs)
@pytest.mark.parametrize("mode,zk,bk", [("f32", "zhat_f32", "bits_f32"), ("f64", "zhat_f64", "bits_f64")])
def test_quantize_golden_g5(ops, golden, mode, zk, bk):
    g = golden("g5_batch_quantize.npz")
    idx, zh, bt = ops.quantize(dev(g["mu"]), dev(g["sigma"]), dev(g["all_code_points"]), list(g["lambdas"]), N=N,
                               mode=mode, want_zhat=True, want_bits=True)
    assert np.array_equal(host(zh), g[zk])
    assert np.array_equal(host(bt).astype(np.int32), g[bk])
    srt = sorted_tables(g["all_code_points"])
    q = host(idx).astype(np.int64)
    for l in range(q.shape[0]):
        assert np.array_equal(np.take_along_axis(srt, q[l].T, axis=1).T, g[zk][l])      # quantizer.py:136-137
    # channel-major layout of the same data takes the FLAT kernel: same answers
    idx_cb = ops.quantize(dev(g["mu"].T), dev(g["sigma"].T), dev(g["all_code_points"]), list(g["lambdas"]), N=N,
                          mode=mode, layout="cb")
    assert np.array_equal(host(idx_cb).transpose(0, 2, 1), host(idx))


def test_quantize_golden_g8_corrected_lengths(ops, golden):
    g5, g8 = golden("g5_batch_quantize.npz"), golden("g8_corrected_lengths.npz")
    level_len = np.stack([O.corrected_level_lengths(N, m).T for m in g8["raw_models"]])
    for layout in ("bc", "cb"):
        mu, sg = (g5["mu"], g5["sigma"]) if layout == "bc" else (g5["mu"].T, g5["sigma"].T)
        idx, zh, bt = ops.quantize(dev(mu), dev(sg), dev(g5["all_code_points"]), list(g5["lambdas"]), N=N,
                                   level_len=dev(level_len), layout=layout, want_zhat=True, want_bits=True)
        zh, bt, idx = host(zh), host(bt), host(idx)
        if layout == "cb":
            zh, bt, idx = zh.transpose(0, 2, 1), bt.transpose(0, 2, 1), idx.transpose(0, 2, 1)
        assert np.array_equal(zh, g8["zhat"])
        lev = O.levels_of_sorted_ranks(N)[idx]
        want = np.take_along_axis(level_len[:, None], lev[..., None].astype(np.int64), axis=3)[..., 0]
        assert np.array_equal(bt, want)


def test_notebook_golden_g7(ops, golden):
    g = golden("g7_notebook.npz")
    idx, val = ops.quantize_notebook(dev(g["means"]), dev(g["stds"]), dev(g["codepoints"]), list(g["betas"]), N=N)
    assert np.array_equal(host(val), g["optima"])
    cnt = host(ops.histogram(idx.reshape(len(g["betas"]), -1), 1, N=N))
    for i in range(len(g["betas"])):
        assert O.entropy_from_counts(cnt[i, 0]) == pytest.approx(g["entropy"][i], rel=1e-12)


def test_argmax_candidates_golden_g5(ops, golden):
    g = golden("g5_batch_quantize.npz")
    orc = O.ChannelwiseOracle(g["mu"].shape[1], N)
    orc.build_code_points(O.factored_gaussian_icdf(g["ch_mean"], g["ch_std"]))
    left, right = O.get_all_N_bit_intervals(orc.grids, g["mu"])
    P = O.assemble_candidates(left, right)
    Lraw = O.raw_code_lengths(N, *g["mu"].shape).astype(np.float32)
    for mode, zk, bk in (("f32", "zhat_f32", "bits_f32"), ("f64", "zhat_f64", "bits_f64")):
        zh, bt = ops.argmax_candidates(dev(P), dev(Lraw), dev(g["mu"]), dev(g["sigma"]), list(g["lambdas"]), mode=mode)
        assert np.array_equal(host(zh), g[zk])
        assert np.array_equal(host(bt).astype(np.int32), g[bk])


# ------------------------------------------------------------------ seeded inputs vs the C oracle
def synth(rng, rows, C):
    scale = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(np.zeros(C), scale))
    mu = (scale * rng.normal(0, 1.0, (rows, C))).astype(np.float32)
    sg = np.clip(np.exp(rng.normal(-2, 0.7, (rows, C))), 1e-4, 10).astype(np.float32)
    return orc.all_code_points, mu, sg


LAM32 = list(2.0 ** np.linspace(-8, 7.5, 32))


@pytest.mark.parametrize("rows,C", [(1, 1), (3, 1), (1021, 1), (4096, 1), (100003, 1), (777, 2), (130, 16), (257, 17),
                                    (1536, 32), (300, 48)])
def test_quantize_vs_oracle_shapes(ops, rows, C):
    rng = np.random.default_rng(rows * 31 + C)
    tab, mu, sg = synth(rng, rows, C)
    lam = LAM32 if rows * C < 60000 else LAM32[::5]
    want = CO.quantize(mu, sg, tab, lam, N=N, threads=8)
    got = ops.quantize(dev(mu), dev(sg), dev(tab), lam, N=N)
    assert got.shape == (len(lam), rows, C)
    assert np.array_equal(host(got), want)
    if C > 1:
        got_cb = ops.quantize(dev(mu.T), dev(sg.T), dev(tab), lam, N=N, layout="cb")
        assert np.array_equal(host(got_cb).transpose(0, 2, 1), want)


def test_quantize_edge_inputs(ops):
    rng = np.random.default_rng(77)
    tab, mu, sg = synth(rng, 4096, 1)
    srt = np.sort(tab[0])
    mu = mu.ravel()
    mu[:2047] = srt                                   # exact code-point hits
    mu[2047:2047 + 1000] = np.nextafter(srt[:1000], np.float32(np.inf))
    mu[3047:3050] = [1e30, -1e30, 0.0]
    mu[3050:3060] = 0.5 * (srt[1000:1010] + srt[1001:1011])   # midpoints: exact L/R ties
    sg = sg.ravel()
    sg[3060:3070] = 1e-4
    sg[3070:3080] = 10.0
    sg[3080:3090] = np.float32(2.0) - np.float32(2.0 ** -23)   # all-ones mantissa
    want = CO.quantize(mu, sg, tab, LAM32, N=N)
    got = ops.quantize(dev(mu), dev(sg), dev(tab), LAM32, N=N)
    assert np.array_equal(host(got), want[:, :, 0])
    # empty input and more than one lambda chunk (> 32 lambdas)
    e = ops.quantize(dev(mu[:0]), dev(sg[:0]), dev(tab), LAM32, N=N)
    assert e.shape == (32, 0)
    lam70 = list(np.exp(np.linspace(np.log(0.01), np.log(1e5), 70)))
    assert np.array_equal(host(ops.quantize(dev(mu), dev(sg), dev(tab), lam70, N=N)),
                          CO.quantize(mu, sg, tab, lam70, N=N)[:, :, 0])


def test_quantize_duplicate_code_points(ops):
    """f32 tables with repeated values (hard part 4): indices still address equal values and
    the candidate semantics still match the per-level search of the reference."""
    rng = np.random.default_rng(5)
    xi = O.dyadic_xi(N)
    from scipy.stats import norm
    pts = np.round(norm.ppf(xi) * 64) / 64             # heavy duplication, still non-decreasing in xi
    tab = pts.astype(np.float32)[None]
    mu = rng.normal(0, 1.2, 5000).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, 5000)).astype(np.float32)
    want_i, want_z = CO.quantize(mu, sg, tab, LAM32, N=N, want_zhat=True)
    got_i, got_z = ops.quantize(dev(mu), dev(sg), dev(tab), LAM32, N=N, want_zhat=True)
    assert np.array_equal(host(got_z), want_z[:, :, 0])
    assert np.array_equal(host(got_i), want_i[:, :, 0])


@pytest.mark.parametrize("Nbits", [4, 5, 6, 7, 8, 9])
def test_quantize_other_bit_depths(ops, Nbits):
    rng = np.random.default_rng(Nbits)
    orc = O.ChannelwiseOracle(2, Nbits)
    orc.build_code_points(O.factored_gaussian_icdf(np.zeros(2), np.array([1.0, 0.5])))
    mu = rng.normal(0, 1, (500, 2)).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, (500, 2))).astype(np.float32)
    want = CO.quantize(mu, sg, orc.all_code_points, LAM32[::4], N=Nbits)
    got = ops.quantize(dev(mu), dev(sg), dev(orc.all_code_points), LAM32[::4], N=Nbits)
    assert np.array_equal(host(got), want)


def test_notebook_vs_oracle(ops):
    rng = np.random.default_rng(9)
    means = rng.normal(-0.08, 1.23, (999, 7)).astype(np.float32)
    stds = np.exp(rng.normal(-2, 0.7, (999, 7))).astype(np.float32)
    pts, lens = O.notebook_code_book(O.empirical_std(means), N)
    betas = [0.01, 0.5, 3.0, 100.0, 1e5]
    idx, val = ops.quantize_notebook(dev(means), dev(stds), dev(pts), betas, N=N)
    rank_of_slot = O.level_major_to_rank(N)
    for i, b in enumerate(betas):
        v, slot = CO.compress_coordinates(means, stds, b, pts, lens, threads=8)
        assert np.array_equal(host(val)[i], v)
        assert np.array_equal(host(idx)[i].astype(np.int64), rank_of_slot[slot])


HMG_SHAPES = [(5000, 1), (8192, 1), (1000, 3), (300, 16), (999, 40)]


@pytest.mark.parametrize("rows,C", HMG_SHAPES)
def test_histogram_vs_oracle(ops, rows, C):
    rng = np.random.default_rng(4)
    idx = rng.integers(0, T, (3, rows, C)).astype(np.uint16)
    idx[2] = 1023                                       # collapsed distribution (large lambda)
    want = CO.histogram(idx, C, N=N)
    assert np.array_equal(host(ops.histogram(dev(idx), C, N=N)), want)
    got_cb = ops.histogram(dev(idx.transpose(0, 2, 1)), C, N=N, layout="cb")
    assert np.array_equal(host(got_cb), want)
    # accumulation into an existing tensor
    acc = ops.histogram(dev(idx), C, N=N)
    ops.histogram(dev(idx), C, N=N, out=acc)
    assert np.array_equal(host(acc), 2 * want)
    # 32-bit counters (the all-reduce payload of the sharded pass)
    c32 = ops.histogram(dev(idx), C, N=N, dtype=torch.int32)
    assert c32.dtype == torch.int32 and np.array_equal(host(c32), want)


@pytest.mark.parametrize("rows,C", HMG_SHAPES)
def test_moments_vs_oracle(ops, rows, C):
    rng = np.random.default_rng(6)
    x = rng.normal(0.3, 2.0, (rows, C)).astype(np.float32)
    ref = CO.moments(x, C)
    assert np.allclose(host(ops.moments(dev(x))), ref, rtol=1e-12, atol=1e-9)
    assert np.allclose(host(ops.moments(dev(x.T), layout="cb")), ref, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("rows,C", HMG_SHAPES)
def test_gather_vs_numpy(ops, rows, C):
    rng = np.random.default_rng(8)
    idx = rng.integers(0, T, (3, rows, C)).astype(np.uint16)
    tab = rng.normal(0, 1, (3, C, T)).astype(np.float32)
    ii = idx.astype(np.int64)
    cc = np.arange(C)[None, :]
    g = host(ops.gather(dev(idx), dev(tab), C, N=N))
    for l in range(3):
        assert np.array_equal(g[l], tab[l][cc, ii[l]])
    g2 = host(ops.gather(dev(idx), dev(tab[0]), C, N=N))
    for l in range(3):
        assert np.array_equal(g2[l], tab[0][cc, ii[l]])
    g3 = host(ops.gather(dev(idx.transpose(0, 2, 1)), dev(tab), C, N=N, layout="cb"))
    assert np.array_equal(g3.transpose(0, 2, 1), g)
    # layout-changing form: channel-major indices in, channel-last values out (and back)
    g4 = host(ops.gather(dev(idx.transpose(0, 2, 1)), dev(tab), C, N=N, layout="cb", out_layout="bc"))
    assert g4.shape == g.shape and np.array_equal(g4, g)
    g5 = host(ops.gather(dev(idx), dev(tab[0]), C, N=N, layout="bc", out_layout="cb"))
    assert np.array_equal(g5.transpose(0, 2, 1), g2)


@pytest.mark.parametrize("rows,cols", [(1, 1), (64, 64), (100, 37), (36864, 32), (1000, 257)])
def test_transpose(ops, rows, cols):
    x = np.random.default_rng(rows).normal(size=(rows, cols)).astype(np.float32)
    assert np.array_equal(host(ops.transpose(dev(x))), x.T)


def test_bmshj_vs_oracle(ops):
    rng = np.random.default_rng(3)
    C = 6
    mats, bias, fac = O.BMSHJ2018Oracle.init_params(C, init_scale=1.0, rng=rng)
    fac = [f + rng.normal(0, 0.5, f.shape).astype(np.float32) for f in fac]
    mats = [m + rng.normal(0, 0.3, m.shape).astype(np.float32) for m in mats]
    eff = O.BMSHJ2018Oracle.effective(mats, bias, fac)
    p = O.BMSHJ2018Oracle(*eff)
    from vbq_amd.priors import pack_bmshj_params
    params = dev(pack_bmshj_params(*eff))
    x = rng.normal(0, 2, (300, C)).astype(np.float32)
    cdf, pdf, logpdf = ops.bmshj_cdf_pdf(params, dev(x), logpdf=True)
    rc, rp = p.cdf_pdf(x)
    assert np.allclose(host(cdf), rc, rtol=2e-6, atol=2e-7)
    assert np.allclose(host(pdf), rp, rtol=1e-4, atol=1e-7)
    assert np.allclose(host(logpdf), p.logpdf(x), rtol=1e-5, atol=1e-5)


def test_large_roundtrip_properties(ops):
    """BASELINE-size check through size-independent properties: table[idx] reproduces zhat,
    level(idx) reproduces bits, histogram totals equal the element count, lambda-monotone rate."""
    rng = np.random.default_rng(1)
    rows, C = 36864, 32
    tab, mu, sg = synth(rng, rows, C)
    d_tab = dev(tab)
    idx, zh, bt = ops.quantize(dev(mu), dev(sg), d_tab, LAM32, N=N, want_zhat=True, want_bits=True)
    srt = dev(sorted_tables(tab))
    assert torch.equal(ops.gather(idx, srt, C, N=N), zh)
    lev = torch.as_tensor(O.levels_of_sorted_ranks(N).astype(np.float32)).cuda()
    assert torch.equal(lev[idx.to(torch.int64)], bt)
    cnt = ops.histogram(idx, C, N=N)
    assert torch.all(cnt.sum(dim=2) == rows)
    rate = bt.sum(dim=(1, 2)).cpu().numpy()
    assert np.all(np.diff(rate) <= 0)
    # spot-check 2 lambdas against the oracle at full size
    want = CO.quantize(mu, sg, tab, [LAM32[3], LAM32[20]], N=N, threads=8)
    assert np.array_equal(host(idx)[[3, 20]], want)


def test_fast_kernel_tie_machinery():
    """The fast K1 decides rare ties by re-solving flagged lanes with the literal 21-candidate scan.
    (1) With every solve forced through that scan the result is still bit-identical to the oracle;
    (2) with the flags switched off mismatches appear on the same data -- i.e. the adversarial data
    does reach the tie cases and the flags are what keeps the fast path exact."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "stress_parity.py"), "--n", "1000000", "--rounds", "3"]

    def run(dbg):
        env = dict(os.environ, VBQ_FAST_DEBUG=str(dbg))
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        return r.returncode, r.stdout
    rc0, out0 = run(0)
    assert rc0 == 0, out0
    rc1, out1 = run(1)
    assert rc1 == 0, out1
    rc2, out2 = run(2)
    assert rc2 == 1 and "mismatches 0\n" not in out2.splitlines()[-1] + "\n", out2


@pytest.mark.parametrize("Nbits", [5, 7, 9])
def test_notebook_and_histogram_other_bit_depths(ops, Nbits):
    rng = np.random.default_rng(40 + Nbits)
    means = rng.normal(-0.08, 1.23, 3001).astype(np.float32)
    stds = np.exp(rng.normal(-2, 0.7, 3001)).astype(np.float32)
    pts, lens = O.notebook_code_book(O.empirical_std(means), Nbits)
    betas = [0.02, 1.0, 250.0]
    idx, val = ops.quantize_notebook(dev(means), dev(stds), dev(pts), betas, N=Nbits)
    rank_of_slot = O.level_major_to_rank(Nbits)
    for i, b in enumerate(betas):
        v, slot = CO.compress_coordinates(means, stds, b, pts, lens, threads=4)
        assert np.array_equal(host(val)[i], v)
        assert np.array_equal(host(idx)[i].astype(np.int64), rank_of_slot[slot])
    Tn = 2 ** (Nbits + 1) - 1
    want = CO.histogram(host(idx)[:, :, None], 1, N=Nbits)
    got = host(ops.histogram(idx, 1, N=Nbits))
    assert got.shape == (3, 1, Tn) and np.array_equal(got, want)


def test_api_fuzz_small(ops):
    """A short run of tools/fuzz_api.py: random shapes / layouts / bit depths / lambda counts / lengths / outputs."""
    from tools import fuzz_api
    bad, solves = fuzz_api.run(60, seed=123, verbose=True)
    assert bad == 0 and solves > 1e6


def test_latents_call_fuzz_small(ops):
    """A short run of the second half of tools/fuzz_api.py: the per-image call (vbq_compress_latents_f32: every spread kind,
    vector and scalar tiles of the planes / lookup kernels, raw and corrected lengths, with and without entropy models) and the
    facade's output forms against the oracle on random shapes."""
    from tools import fuzz_api
    rng = np.random.default_rng(321)
    solves = 0
    for case in range(60):
        desc, n = fuzz_api.one_latents_case(rng, torch.device("cuda"))
        assert desc is None, f"case {case}: {desc}"
        solves += n
    assert solves > 1e6


@pytest.mark.parametrize("Nbits", [11, 12])
def test_bit_depths_above_ten_on_planes(ops, Nbits):
    """N = 11, 12 (4095 / 8191 code points): K1 and K2 on channel-major planes and on one code book, raw and
    corrected lengths, optional outputs; the channel-last form is refused with a pointer to the transpose."""
    from vbq_amd._lib import VBQError
    rng = np.random.default_rng(200 + Nbits)
    C, rows = 3, 1500
    orc = O.ChannelwiseOracle(C, Nbits)
    orc.build_code_points(O.factored_gaussian_icdf(np.zeros(C), np.array([1.0, 0.4, 2.5])))
    mu = rng.normal(0, 1, (rows, C)).astype(np.float32)
    sg = np.exp(rng.normal(-3, 1.0, (rows, C))).astype(np.float32)
    mu[:40, 0] = np.sort(orc.all_code_points[0])[rng.integers(0, 2 ** (Nbits + 1) - 1, 40)]      # exact hits
    lam = LAM32[::3]
    ll = (np.arange(Nbits + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1, (len(lam), C, Nbits + 1)))).astype(np.float32)
    for level_len in (None, ll):
        want_i, want_z, want_b = CO.quantize(mu, sg, orc.all_code_points, lam, N=Nbits, level_len=level_len, want_zhat=True,
                                             want_bits=True, threads=8)
        got_i, got_z, got_b = ops.quantize(dev(np.ascontiguousarray(mu.T)), dev(np.ascontiguousarray(sg.T)), dev(orc.all_code_points),
                                           lam, N=Nbits, level_len=None if level_len is None else dev(level_len), layout="cb",
                                           want_zhat=True, want_bits=True)
        assert np.array_equal(host(got_i).transpose(0, 2, 1), want_i)
        assert np.array_equal(host(got_z).transpose(0, 2, 1), want_z) and np.array_equal(host(got_b).transpose(0, 2, 1), want_b)
        got1 = ops.quantize(dev(mu[:, 1].copy()), dev(sg[:, 1].copy()), dev(orc.all_code_points[1:2]), lam, N=Nbits,
                            level_len=None if level_len is None else dev(level_len[:, 1:2].copy()))
        assert np.array_equal(host(got1), want_i[:, :, 1])
    cnt = host(ops.histogram(got_i, C, N=Nbits, layout="cb"))
    assert np.array_equal(cnt, CO.histogram(want_i, C, N=Nbits))
    with pytest.raises(VBQError, match="planes"):
        ops.quantize(dev(mu), dev(sg), dev(orc.all_code_points), lam, N=Nbits)


@pytest.mark.parametrize("rows,C", [(1, 2), (513, 3), (1000, 16), (2047, 40), (4096, 256)])
def test_channel_last_in_planes_out(ops, rows, C):
    """VBQ_LAYOUT_BC_TO_CB: the fast kernel reads the latents channel-last and writes planes -- same indices, code
    points and bits as transposing first; requests the fast kernel cannot serve are refused."""
    from vbq_amd._lib import VBQError
    rng = np.random.default_rng(rows + C)
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(np.zeros(C), np.exp(rng.uniform(-1, 1, C))))
    mu = rng.normal(0, 1.3, (rows, C)).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.8, (rows, C))).astype(np.float32)
    lam = LAM32[::5]
    ll = (np.arange(N + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1, (len(lam), C, N + 1)))).astype(np.float32)
    want = CO.quantize(mu, sg, orc.all_code_points, lam, N=N, level_len=ll, want_zhat=True, want_bits=True, threads=8)
    got = ops.quantize(dev(mu), dev(sg), dev(orc.all_code_points), lam, N=N, level_len=dev(ll), layout="bc->cb", want_zhat=True,
                       want_bits=True)
    for g, w in zip(got, want):
        assert g.shape == (len(lam), C, rows) and np.array_equal(host(g).transpose(0, 2, 1), w)
    with pytest.raises(VBQError, match="fast f32 kernel"):
        ops.quantize(dev(mu), dev(sg), dev(orc.all_code_points), [0.0, 1.0, 2.0], N=N, layout="bc->cb")
    # one or two lambdas take the pruned descent (K1p), which only makes the reference's own comparisons: any lambda
    g0 = ops.quantize(dev(mu), dev(sg), dev(orc.all_code_points), [0.0, 1.0], N=N, layout="bc->cb")
    assert np.array_equal(host(g0).transpose(0, 2, 1), CO.quantize(mu, sg, orc.all_code_points, [0.0, 1.0], N=N, threads=8))
    with pytest.raises(VBQError, match="fast f32 kernel"):
        ops.quantize(dev(mu), dev(sg), dev(orc.all_code_points), lam, N=N, layout="bc->cb", mode="f64")


def test_caller_length_tables_outside_the_certificate_range(ops):
    """vbq.h: a caller-supplied length table may make lambda * len tiny (below 2^-39) or huge; the fast kernel's tie
    certificate does not cover that, so the workgroup that sees such a penalty takes its literal scan.  Same answers as the
    oracle, for the index kernel and the counting kernel, with the odd values in one channel only."""
    rng = np.random.default_rng(77)
    rows, C = 3000, 3
    tab, mu, sg = synth(rng, rows, C)
    lam = [2.0 ** -8, 0.05, 1.0, 37.0]
    ll = (np.arange(N + 1, dtype=np.float32)[None, None, :] + rng.uniform(0, 4, (len(lam), C, N + 1)).astype(np.float32)).astype(np.float32)
    ll[0, 1, 3] = np.float32(1e-30)                                  # lambda * len = 4e-33: far below 2^-39
    ll[2, 1, 7] = np.float32(3e30)                                   # lambda * len = 3e30: above 2^70
    ll[3, 1, 0] = np.float32(0.0)                                    # exactly zero is inside the range
    want = CO.quantize(mu, sg, tab, lam, N=N, level_len=ll, threads=8)
    got = ops.quantize(dev(mu.T), dev(sg.T), dev(tab), lam, N=N, level_len=dev(ll), layout="cb")
    assert np.array_equal(host(got).transpose(0, 2, 1), want)
    lc = ops.level_counts(dev(mu.T), dev(sg.T), dev(tab), lam, N=N, level_len=dev(ll), layout="cb")
    lev = O.levels_of_sorted_ranks(N)[want]
    wl = np.stack([[np.bincount(lev[l, :, c], minlength=N + 1) for c in range(C)] for l in range(len(lam))])
    assert np.array_equal(host(lc), wl)


def test_notebook_one_beta_negative_or_zero(ops):
    """One or two betas per call normally take the pruned descent (K1np), whose stop rule best <= w (n + 1) needs w >= 0: with
    a NEGATIVE beta deeper levels carry smaller penalties and can still win -- such calls must take the literal
    kernel and agree with the brute force; beta = 0 stays on the pruned kernel."""
    rng = np.random.default_rng(19)
    means = rng.normal(-0.08, 1.23, 20011).astype(np.float32)
    stds = np.exp(rng.normal(-2, 0.7, 20011)).astype(np.float32)
    pts, lens = O.notebook_code_book(O.empirical_std(means), N)
    rank_of_slot = O.level_major_to_rank(N)
    for betas in ([-0.5], [-3.0, 2.0], [0.0], [0.0, -1e-3], [1e-3]):
        idx, val = ops.quantize_notebook(dev(means), dev(stds), dev(pts), betas, N=N)
        for i, b in enumerate(betas):
            v, slot = CO.compress_coordinates(means, stds, b, pts, lens, threads=8)
            assert np.array_equal(host(val)[i], v), betas
            assert np.array_equal(host(idx)[i].astype(np.int64), rank_of_slot[slot]), betas


def test_reserved_workgroups_change_the_grid_not_the_results(ops):
    """The launch policy beside an overlapped collective, a per-call argument since ABI 5 (`reserved_workgroups`): K1 / K1t with
    slots left free -- a smaller resident grid for few channels, short-lived workgroups for many -- give the same indices and
    counts, and two threads running solves with DIFFERENT policies at the same time (own streams) each get their own grid
    and the right answers: there is no process-wide launch state to race on."""
    import ctypes as C
    import threading
    from scipy.stats import norm
    from vbq_amd import _lib
    h = _lib.lib()
    rng = np.random.default_rng(77)
    lam = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 32)]
    xi = np.concatenate([(np.arange(2 ** k) + 0.5) / 2 ** k for k in range(N + 1)])

    def grid(kernel, rows, n_ch, reserved):
        g = (C.c_int64 * 3)()
        assert h.vbq_solve_grid(kernel, rows, n_ch, 0, reserved, g) == 0
        return tuple(g)

    cases = []
    for rows, Cc in ((300_000, 1), (9000, 64), (2100, 256)):
        s_c = np.exp(rng.uniform(np.log(0.3), np.log(3.0), Cc))
        mu = (rng.standard_normal((Cc, rows)) * s_c[:, None]).astype(np.float32)
        sg = np.exp(rng.normal(-2, 0.7, (Cc, rows))).astype(np.float32)
        tab = norm.ppf(xi[None, :], scale=s_c[:, None]).astype(np.float32)
        ll = (np.arange(N + 1, dtype=np.float32) + np.abs(rng.normal(0, 1, (32, Cc, N + 1)))).astype(np.float32)
        got = {}
        for r in (0, 64, 100000):
            idx = ops.quantize(dev(mu), dev(sg), dev(tab), lam, N=N, level_len=dev(ll), layout="cb", reserved_workgroups=r)
            lc = ops.level_counts(dev(mu), dev(sg), dev(tab), lam, N=N, layout="cb", reserved_workgroups=r)
            got[r] = (host(idx), host(lc))
        want = CO.quantize(np.ascontiguousarray(mu.T), np.ascontiguousarray(sg.T), tab, lam, N=N, level_len=ll)
        assert np.array_equal(got[0][0].transpose(0, 2, 1), want)
        for r in (64, 100000):
            assert np.array_equal(got[r][0], got[0][0]) and np.array_equal(got[r][1], got[0][1]), (rows, Cc, r)
        big = 10 ** 7                                  # (the small test tensors do not fill the chip: their grids coincide)
        assert grid(0, big, Cc, 64) != grid(0, big, Cc, 0) or grid(1, big, Cc, 64) != grid(1, big, Cc, 0), (rows, Cc)
        cases.append((dev(mu), dev(sg), dev(tab), dev(ll), got[0]))
    with pytest.raises(Exception):
        ops.quantize(cases[0][0], cases[0][1], cases[0][2], lam, N=N, layout="cb", workgroups_per_cu=9)
    # two threads, two policies, at the same time
    torch.cuda.synchronize()
    errors = []

    def worker(reserved):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(6):
                    for mu_d, sg_d, tab_d, ll_d, (want_idx, want_lc) in cases:
                        idx = ops.quantize(mu_d, sg_d, tab_d, lam, N=N, level_len=ll_d, layout="cb", reserved_workgroups=reserved)
                        lc = ops.level_counts(mu_d, sg_d, tab_d, lam, N=N, layout="cb", reserved_workgroups=reserved)
                        st.synchronize()
                        if not (np.array_equal(host(idx), want_idx) and np.array_equal(host(lc), want_lc)):
                            errors.append(reserved)
        except Exception as e:                               # noqa: BLE001 -- reported below
            errors.append(repr(e))
    ts = [threading.Thread(target=worker, args=(r,)) for r in (0, 64, 700)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors