"""The pieces of the two-pass entropy-model build (quantizer.py:82-150) that round 2 added, against the oracle:
K1t / K1h (solve + bit-length histogram, no per-element output), the row-range forms of K1 / K2 that let the host overlap
them, and the table-driven code lengths.  Counts are integers: everything is compared with array_equal."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import c_oracle as CO
from oracle import vbq_oracle as O

pytestmark = pytest.mark.gpu
N = 10
T = 2047
LAM32 = list(2.0 ** np.linspace(-8, 7.5, 32))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from vbq_amd import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def synth(rng, rows, C):
    scale = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(np.zeros(C), scale))
    mu = (scale * rng.normal(0, 1.0, (rows, C))).astype(np.float32)
    sg = np.clip(np.exp(rng.normal(-2, 0.7, (rows, C))), 1e-4, 10).astype(np.float32)
    return orc.all_code_points, mu, sg


def oracle_level_counts(mu, sg, tab, lam, level_len=None):
    """[L, C, N+1] from the C oracle's indices: np.bincount(raw_num_bits[:, c], minlength=N+1) (quantizer.py:104)."""
    idx = CO.quantize(mu, sg, tab, lam, N=N, level_len=level_len, threads=8)          # [L, rows, C]
    lev = O.levels_of_sorted_ranks(N)[idx]
    L, rows, C = lev.shape
    return np.stack([[np.bincount(lev[l, :, c], minlength=N + 1) for c in range(C)] for l in range(L)]).astype(np.int64)


def random_level_len(rng, L, C):
    over = rng.uniform(0.0, 9.0, (L, C, N + 1)).astype(np.float32)
    return (np.arange(N + 1, dtype=np.float32)[None, None, :] + over).astype(np.float32)


@pytest.mark.parametrize("rows,C", [(1, 1), (5, 1), (1021, 1), (70001, 1), (777, 2), (257, 17), (1536, 32), (4104, 5)])
def test_level_counts_vs_oracle(ops, rows, C):
    rng = np.random.default_rng(rows * 7 + C)
    tab, mu, sg = synth(rng, rows, C)
    lam = LAM32 if rows * C < 60000 else LAM32[::4]
    for ll in (None, random_level_len(rng, len(lam), C)):
        want = oracle_level_counts(mu, sg, tab, lam, ll)
        lld = None if ll is None else dev(ll)
        if C == 1:
            got = ops.level_counts(dev(mu[:, 0]), dev(sg[:, 0]), dev(tab), lam, N=N, level_len=lld)
        else:
            got = ops.level_counts(dev(mu.T), dev(sg.T), dev(tab), lam, N=N, level_len=lld, layout="cb")
            got2 = ops.level_counts(dev(mu), dev(sg), dev(tab), lam, N=N, level_len=lld, layout="bc->cb")
            assert torch.equal(got, got2)
        assert got.dtype == torch.int64 and tuple(got.shape) == (len(lam), C, N + 1)
        assert np.array_equal(got.cpu().numpy(), want)
        assert int(got.sum().item()) == rows * C * len(lam)
    # accumulation into a caller's buffer
    out = ops.level_counts(dev(mu.T) if C > 1 else dev(mu[:, 0]), dev(sg.T) if C > 1 else dev(sg[:, 0]), dev(tab), lam, N=N,
                           layout="cb" if C > 1 else "bc", out=got.clone())
    assert np.array_equal(out.cpu().numpy(), want + oracle_level_counts(mu, sg, tab, lam, None))


def test_level_counts_ties_and_edges(ops):
    """Inputs that force the events the fast path cannot decide by itself (exact code-point hits, exact mid-points
    between neighbours of different levels, values outside the table, extreme sigmas): the level of the winner must
    still be the reference's, i.e. the literal scan takes over."""
    rng = np.random.default_rng(5)
    tab, _, _ = synth(rng, 4, 1)
    t = np.sort(tab[0])
    mids = (t[:-1] + t[1:]) * np.float32(0.5)
    mu = np.concatenate([t, mids, [t[0] - 50, t[-1] + 50, t[0], t[-1], 0.0], rng.normal(0, 1, 3000).astype(np.float32)]).astype(np.float32)
    sg = np.concatenate([np.full(t.size + mids.size + 5, 1.0), np.exp(rng.normal(-2, 2.5, 3000))]).astype(np.float32)
    sg[::7] = np.float32(0.99999994)                                # all-ones mantissa
    lam = [2.0 ** k for k in range(-10, 9)] + [1.0, 3.0]
    for ll in (None, random_level_len(rng, len(lam), 1), np.ones((len(lam), 1, N + 1), np.float32)):
        want = oracle_level_counts(mu[:, None], sg[:, None], tab, lam, ll)
        got = ops.level_counts(dev(mu), dev(sg), dev(tab), lam, N=N, level_len=None if ll is None else dev(ll))
        assert np.array_equal(got.cpu().numpy(), want)


def test_threshold_kernel_on_ties_and_odd_sweeps(ops):
    """K1t (raw lengths, no per-lambda loop) on what its guard bands exist for: thresholds that sit exactly on sweep
    points (lambda = (du_i - du_j) / (j - i) by construction), exact code-point hits and mid-points, inputs outside the
    table, extreme sigmas; sweeps that are unsorted, short, randomly spaced; and sweeps it must hand to the dense kernel
    (repeated values, two points in one bucket, more than 16 octaves)."""
    rng = np.random.default_rng(21)
    tab, _, _ = synth(rng, 4, 1)
    t = np.sort(tab[0])
    mids = (t[:-1] + t[1:]) * np.float32(0.5)
    mu = np.concatenate([t, mids, [t[0] - 50, t[-1] + 50, 0.0], rng.normal(0, 1, 6000).astype(np.float32)]).astype(np.float32)
    sg = np.concatenate([np.full(t.size + mids.size + 3, 1.0), np.exp(rng.normal(-2, 2.5, 6000))]).astype(np.float32)
    sg[::7] = np.float32(0.99999994)
    sg[5::11] = np.float32(2.0 ** -10)                               # du values that are exact multiples of powers of two
    sweeps = [LAM32, LAM32[::-1], [LAM32[i] for i in rng.permutation(32)[:9]], [0.37], [2.0 ** k for k in range(-10, 9)],
              [float(v) for v in np.sort(np.exp(rng.uniform(np.log(1e-3), np.log(1e3), 32)))],
              [1.0, 1.0, 2.0],                                       # repeated value      -> dense kernel
              [1.0, 1.0 + 2.0 ** -9, 4.0],                           # same bucket         -> dense kernel
              [1e-9, 1.0, 1e9]]                                      # > 16 octaves        -> dense kernel
    for lam in sweeps:
        want = oracle_level_counts(mu[:, None], sg[:, None], tab, lam, None)
        got = ops.level_counts(dev(mu), dev(sg), dev(tab), lam, N=N)
        assert np.array_equal(got.cpu().numpy(), want), lam
    # thresholds exactly on sweep points: pick lambda from the element's own cost differences
    idx = ops.quantize(dev(mu[-6000:]), dev(sg[-6000:]), dev(tab), [0.5], N=N, want_zhat=True)[1].cpu().numpy()[0]
    d = ((idx - mu[-6000:]) / sg[-6000:]) ** 2 * np.float32(0.5)
    lam = sorted({float(np.float32(v)) for v in d[(d > 1e-3) & (d < 1e3)][:400:13]})
    lam = [v for i, v in enumerate(lam) if i == 0 or v > lam[i - 1] * 1.02][:32]
    want = oracle_level_counts(mu[:, None], sg[:, None], tab, lam, None)
    assert np.array_equal(ops.level_counts(dev(mu), dev(sg), dev(tab), lam, N=N).cpu().numpy(), want)


@pytest.mark.timeout(900)
def test_threshold_kernel_guard_machinery():
    """tools/stress_parity.py --levels: (0) as built, every window's histogram equals the oracle's; (1) with every sweep
    point forced through the literal scan + correction it still does; (2) with the guard bands switched off wrong counts
    appear on the same data -- the adversarial inputs do reach the near-tie cases and the bands are what keeps K1t exact."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "stress_parity.py"), "--levels", "--n", "400000", "--rounds", "4"]

    def run(dbg):
        env = dict(os.environ, VBQ_FAST_DEBUG=str(dbg))
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
        return r.returncode, r.stdout + r.stderr[-2000:]
    rc0, out0 = run(0)
    assert rc0 == 0, out0
    rc1, out1 = run(1)
    assert rc1 == 0, out1
    rc2, out2 = run(2)
    assert rc2 == 1, out2


def test_level_counts_agrees_with_quantize_plus_histogram(ops):
    """Same counts as the two-kernel route (K1 indices -> K2 rank histogram -> sum over the ranks of a level)."""
    from vbq_amd import entropy
    rng = np.random.default_rng(11)
    tab, mu, sg = synth(rng, 36864 // 8, 24)
    mu_cb, sg_cb, tabd = dev(mu.T), dev(sg.T), dev(tab)
    idx = ops.quantize(mu_cb, sg_cb, tabd, LAM32, N=N, layout="cb")
    via_hist = entropy.level_counts_from_counts(ops.histogram(idx, 24, N=N, layout="cb"), N)
    assert torch.equal(ops.level_counts(mu_cb, sg_cb, tabd, LAM32, N=N, layout="cb"), via_hist)


def test_level_counts_rejects_what_it_does_not_serve(ops):
    from vbq_amd._lib import VBQError
    rng = np.random.default_rng(2)
    tab, mu, sg = synth(rng, 64, 3)
    with pytest.raises(VBQError, match="channel-last"):
        ops.level_counts(dev(mu), dev(sg), dev(tab), [1.0], N=N, layout="bc")
    with pytest.raises(VBQError, match="fast f32 kernel"):
        ops.level_counts(dev(mu.T), dev(sg.T), dev(tab), [1e-30], N=N, layout="cb")


@pytest.mark.parametrize("rows,C,cuts", [(4096, 6, (0, 1024, 1032, 4096)), (10007, 1, (0, 8, 5000, 10007)),
                                         (1000, 3, (0, 3, 501, 1000)), (2048, 1, (0, 0, 2048, 2048))])
def test_row_ranges_equal_whole(ops, rows, C, cuts):
    """vbq_quantize_rows_f32 / vbq_histogram_rows_u16: chunks (aligned or not, empty ones included) reproduce the
    one-launch result bit for bit, with and without the persistent 4-workgroups-per-CU grid."""
    rng = np.random.default_rng(rows + C)
    tab, mu, sg = synth(rng, rows, C)
    lam = LAM32[::3]
    ll = dev(random_level_len(rng, len(lam), C))
    m, s = (dev(mu.T), dev(sg.T)) if C > 1 else (dev(mu[:, 0]), dev(sg[:, 0]))
    layout = "cb" if C > 1 else "bc"
    whole, zw, bw = ops.quantize(m, s, dev(tab), lam, N=N, level_len=ll, layout=layout, want_zhat=True, want_bits=True)
    cw = ops.histogram(whole, C, N=N, layout=layout)
    for wg in (0, 4):
        idx = torch.full_like(whole, 0xffff)
        zh, bt = torch.full_like(zw, -1.0), torch.full_like(bw, -1.0)
        cnt = torch.zeros_like(cw)
        cnt32 = torch.zeros(cw.shape, dtype=torch.int32, device="cuda")
        for a, b in zip(cuts[:-1], cuts[1:]):
            ops.quantize(m, s, dev(tab), lam, N=N, level_len=ll, layout=layout, out_idx=idx, out_zhat=zh, out_bits=bt,
                         rows=(a, b), workgroups_per_cu=wg)
            ops.histogram(idx, C, N=N, layout=layout, out=cnt, rows=(a, b))
            ops.histogram(idx, C, N=N, layout=layout, out=cnt32, rows=(a, b))
        assert torch.equal(idx.view(torch.int16), whole.view(torch.int16))
        assert torch.equal(zh, zw) and torch.equal(bt, bw)
        assert torch.equal(cnt, cw) and torch.equal(cnt32.to(torch.int64), cw)
    from vbq_amd._lib import VBQError
    with pytest.raises(VBQError, match="row range"):
        ops.quantize(m, s, dev(tab), lam, N=N, level_len=ll, layout=layout, rows=(5, rows + 1))
    with pytest.raises(VBQError, match="row range"):
        ops.histogram(whole, C, N=N, layout=layout, rows=(7, 3))


def test_row_ranges_channel_last(ops):
    rng = np.random.default_rng(77)
    tab, mu, sg = synth(rng, 900, 20)
    lam = LAM32[::6]
    whole = ops.quantize(dev(mu), dev(sg), dev(tab), lam, N=N)                      # [L, rows, C], tiled kernel
    idx = torch.zeros_like(whole)
    cnt = torch.zeros((len(lam), 20, T), dtype=torch.int64, device="cuda")
    for a, b in ((0, 130), (130, 131), (131, 900)):
        ops.quantize(dev(mu), dev(sg), dev(tab), lam, N=N, out_idx=idx, rows=(a, b))
        ops.histogram(idx, 20, N=N, out=cnt, rows=(a, b))
    assert torch.equal(idx.view(torch.int16), whole.view(torch.int16))
    assert torch.equal(cnt, ops.histogram(whole, 20, N=N))


def test_code_lengths_from_counts_match_numpy(ops):
    """The table-driven form of quantizer.py:105-110 / 141-146: lut built on the host with the reference's NumPy
    float32 operations, looked up on the device, equals those operations applied to the counts directly."""
    from vbq_amd import entropy
    rng = np.random.default_rng(3)
    B, C, L, smooth = 36864, 7, 5, 1
    for K, period in ((N + 1, N + 1), (T, 0)):
        p = rng.dirichlet(np.full(K, 0.05), size=(L, C))
        counts = np.stack([[rng.multinomial(B, p[l, c]) for c in range(C)] for l in range(L)]).astype(np.int64)
        want_model = entropy.neg_log2_freq(counts, smooth)                              # [L, C, K] f32, NumPy
        lut = entropy.neg_log2_lut(B, K, smooth)
        assert lut is not None and lut.dtype == np.float32 and lut.shape == (B + 1,)
        for dt in (torch.int64, torch.int32):
            ln, model = ops.code_lengths_from_counts(dev(counts).to(dt), dev(lut), level_period=period, want_model=True)
            assert np.array_equal(model.cpu().numpy(), want_model)
            lv = np.arange(K, dtype=np.float32) if period else np.zeros(K, np.float32)
            assert np.array_equal(ln.cpu().numpy(), (lv + want_model).astype(np.float32))
    assert entropy.neg_log2_lut(1 << 24, T, 1) is None                                  # sums no longer exact in f32
    assert entropy.neg_log2_lut(1000, T, 0.5) is None                                   # non-integer smoothing


@pytest.mark.parametrize("rows,C,L,dt", [(3001, 70, 32, torch.int32), (512, 2048, 1, torch.int64), (1000, 5, 3, torch.int32),
                                         (8, 64, 32, torch.int64)])
def test_histogram_models_assigns_counts_and_lengths(ops, rows, C, L, dt):
    """vbq_histogram_models_u16, fused (>= 2048 rows of bins: one workgroup per row stores its bins and the looked-up lengths)
    and composed (smaller problems): counts are ASSIGNED (stale contents of the buffer must not survive), models = lut[counts]."""
    from vbq_amd import entropy
    rng = np.random.default_rng(rows + C)
    idx = torch.from_numpy(rng.integers(0, T, (L, C, rows)).astype(np.uint16)).cuda()
    idx[:, :, : rows // 2] = 1023                                      # a hot bin
    want = ops.histogram(idx, C, N=N, layout="cb")
    lut = entropy.neg_log2_lut(rows, T, 1)
    counts = torch.full((L, C, T), 12345, dtype=dt, device="cuda")
    models = torch.full((L, C, T), -1.0, dtype=torch.float32, device="cuda")
    ops.histogram_models(idx, C, counts, N=N, lut=torch.from_numpy(lut).cuda(), models=models)
    assert torch.equal(counts.to(torch.int64), want)
    assert np.array_equal(models.cpu().numpy(), entropy.neg_log2_freq(want, 1))
    counts2 = torch.full((L, C, T), 7, dtype=dt, device="cuda")
    ops.histogram_models(idx, C, counts2, N=N)                         # counts only
    assert torch.equal(counts2.to(torch.int64), want)


def test_integration_md_sequence_through_ctypes_only():
    """The C-ABI call sequence INTEGRATION.md shows for the whole two-pass build, issued through ctypes alone (no vbq_amd.ops,
    no pipeline object): the five calls must give the oracle's raw-length models and entropy models bit for bit."""
    import ctypes as C
    from vbq_amd import _lib
    h = _lib.lib()
    rng = np.random.default_rng(91)
    B, Cc, L = 2304, 72, 32                                        # 72 x 32 = 2304 rows of bins: the fused K2 flush applies
    tab_h, mu_h, sg_h = synth(rng, B, Cc)
    lam = LAM32
    dev_ = torch.device("cuda")
    mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev_)   # planes [C, B]
    sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev_)
    table = torch.from_numpy(tab_h).to(dev_)
    lut1 = -np.log2((np.arange(B + 1, dtype=np.float32) + 1) / np.float32(B + (N + 1)))
    lut2 = -np.log2((np.arange(B + 1, dtype=np.float32) + 1) / np.float32(B + T))
    d_lut1, d_lut2 = torch.from_numpy(lut1).to(dev_), torch.from_numpy(lut2).to(dev_)
    level_counts = torch.zeros((L, Cc, N + 1), dtype=torch.int64, device=dev_)
    level_len = torch.empty((L, Cc, N + 1), dtype=torch.float32, device=dev_)
    raw_models = torch.empty_like(level_len)
    idx = torch.empty((L, Cc, B), dtype=torch.uint16, device=dev_)
    counts = torch.empty((L, Cc, T), dtype=torch.int64, device=dev_)
    models = torch.empty((L, Cc, T), dtype=torch.float32, device=dev_)
    nws = h.vbq_quantize_workspace_bytes(Cc, L, N)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev_)
    lamc = (C.c_double * L)(*lam)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert h.vbq_level_counts_f32(p(mu), p(sg), B, Cc, 1, p(table), None, lamc, L, N, p(level_counts), p(ws), nws, 0, st) == 0
    assert h.vbq_code_lengths_from_counts(p(level_counts), 0, L * Cc * (N + 1), p(d_lut1), B + 1, N + 1, p(level_len), p(raw_models), st) == 0
    assert h.vbq_quantize_f32(p(mu), p(sg), B, Cc, 1, p(table), p(level_len), lamc, L, N, 0, p(idx), None, None, p(ws), nws, st) == 0
    assert h.vbq_histogram_models_u16(p(idx), B, Cc, L, N, p(counts), 0, p(d_lut2), B + 1, p(models), st) == 0
    torch.cuda.synchronize()
    orc = O.ChannelwiseOracle(Cc, N)
    orc.build_code_points(lambda xi_: tab_h.T)
    keys = [np.float32(l) for l in lam]
    orc.build_entropy_models(mu_h, sg_h, keys, add_n_smoothing=1)
    for i, k in enumerate(keys):
        assert np.array_equal(raw_models[i].cpu().numpy(), orc.raw_models[k])
        assert np.array_equal(models[i].cpu().numpy(), orc.entropy_models[k])
    assert int(counts.sum().item()) == L * Cc * B
    # ... and INTEGRATION.md's per-image call on top of the tables the build left on the device: one C call
    Bi = 768                                                       # one "image" of 768 positions, channel-last as it arrives
    means = torch.from_numpy(mu_h[:Bi]).to(dev_)
    var = torch.from_numpy(sg_h[:Bi]).to(dev_) ** 2                # 1 = VBQ_SPREAD_VARIANCE below
    sig_dev = torch.sqrt(var).cpu().numpy()                        # what the call derives (IEEE root of the variance handed in)
    table_sorted = torch.from_numpy(np.sort(tab_h, axis=1)).to(dev_)
    z = torch.empty((L, Bi, Cc), dtype=torch.float32, device=dev_)
    raw = torch.empty_like(z)
    nb = torch.empty_like(z)
    nws2 = h.vbq_compress_latents_workspace_bytes(Bi, Cc, L, N)
    ws2 = torch.empty(nws2, dtype=torch.uint8, device=dev_)
    assert h.vbq_compress_latents_f32(p(means), p(var), 1, Bi, Cc, p(table), p(table_sorted), p(level_len), p(models), lamc, L, N,
                                      p(z), p(raw), p(nb), p(ws2), nws2, st) == 0
    torch.cuda.synchronize()
    ref = orc.compress_latents(mu_h[:Bi], sig_dev, keys)
    for i, k in enumerate(keys):
        assert np.array_equal(z[i].cpu().numpy(), ref["Z_hat"][k]) and np.array_equal(raw[i].cpu().numpy(), ref["raw_num_bits"][k])
        assert np.array_equal(nb[i].cpu().numpy(), ref["num_bits"][k])
    assert h.vbq_compress_latents_f32(p(means), p(var), 1, Bi, Cc, p(table), p(table_sorted), p(level_len), p(models), lamc, L, N,
                                      p(z), p(raw), p(nb), p(ws2), nws2 - 1, st) != 0        # a short workspace is refused
    # ... and the whole build again as ONE C call on the channel-last latents (vbq_build_entropy_models_f32): the same tables
    mu_bc, sg_bc = torch.from_numpy(mu_h).to(dev_), torch.from_numpy(sg_h).to(dev_)
    lc2, ll2, rm2 = torch.full_like(level_counts, -1), torch.zeros_like(level_len), torch.zeros_like(raw_models)
    cnt2 = torch.full((L, Cc, T), -1, dtype=torch.int32, device=dev_)
    md2 = torch.zeros_like(models)
    nws3 = h.vbq_build_entropy_models_workspace_bytes(B, Cc, L, N)
    ws3 = torch.empty(nws3, dtype=torch.uint8, device=dev_)
    assert h.vbq_build_entropy_models_f32(p(mu_bc), p(sg_bc), 0, B, Cc, p(table), lamc, L, N, p(d_lut1), B + 1, p(d_lut2), B + 1,
                                          p(lc2), p(ll2), p(rm2), p(cnt2), 1, p(md2), p(ws3), nws3, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(lc2, level_counts) and torch.equal(ll2, level_len) and torch.equal(rm2, raw_models)
    assert torch.equal(cnt2.to(torch.int64), counts) and torch.equal(md2, models)
    assert h.vbq_build_entropy_models_f32(p(mu_bc), p(sg_bc), 0, B, Cc, p(table), lamc, L, N, p(d_lut1), B, p(d_lut2), B + 1,
                                          p(lc2), p(ll2), p(rm2), p(cnt2), 1, p(md2), p(ws3), nws3, 0, st) != 0   # a table shorter than B + 1


# ---------------------------------------------------------------------------------------------------------------------
# K1nt: the notebook's beta sweep (ipynb:429-443, cell 32) from per-element thresholds -- indices and values
# ---------------------------------------------------------------------------------------------------------------------
def _notebook_case(rng, n):
    scale = np.float32(np.exp(rng.uniform(np.log(0.2), np.log(5.0))))
    pts, lens = O.notebook_code_book(scale, N)
    srt = np.sort(pts)
    mids = (0.5 * (srt[:-1] + srt[1:])).astype(np.float32)
    means = np.concatenate([srt.astype(np.float32), mids, np.float32([srt[0] - 40 * scale, srt[-1] + 40 * scale, 0.0]),
                            (scale * rng.standard_t(4, n)).astype(np.float32)]).astype(np.float32)
    stds = (np.exp(rng.normal(-2, 1.5, means.size)) * scale).astype(np.float32)
    stds[::7] = np.float32(0.99999994) * scale
    stds[3::101] = np.float32(1e-6) * scale                  # err / sigma^2 beyond every threshold of the sweep
    stds[5::103] = np.float32(1e4) * scale
    return pts, lens, means, stds


def _check_notebook(ops, means, stds, pts, lens, betas, want_values=True):
    idx, val = ops.quantize_notebook(dev(means), dev(stds), dev(pts), betas, N=N, want_values=want_values)
    idx = idx.cpu().numpy().astype(np.int64)
    r2s = O.level_major_to_rank(N)
    for i, b in enumerate(betas):
        v, slot = CO.compress_coordinates(means, stds, b, pts, lens, threads=8)
        assert np.array_equal(idx[i], r2s[slot]), (len(betas), i, b)
        if want_values:
            assert np.array_equal(val[i].cpu().numpy(), v), (len(betas), i, b)


def test_notebook_threshold_kernel_sweeps(ops):
    """K1nt on exact code-point hits, mid-points, inputs outside the code book and extreme sigmas, for the sweeps it takes
    (6 to 64 betas, any order: the notebook's 50, 32, 64, reversed, a random subset) and the ones it hands to the per-beta
    kernel (fewer than 6 betas, repeated values, two betas in one bucket, more than 24 octaves, out-of-range betas); odd and
    even lengths (packed and unpacked stores); with and without the quantised values."""
    rng = np.random.default_rng(77)
    pts, lens, means, stds = _notebook_case(rng, 9000)
    nb50 = [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(1e5), 50))]        # ipynb cell 32
    sweeps = [nb50, nb50[::-1], [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(1e5), 32))],
              [float(b) for b in np.exp(np.linspace(np.log(0.003), np.log(3e3), 64))],
              [nb50[i] for i in rng.permutation(50)[:9]], nb50[:6], [2.0 ** k for k in range(-9, 12)],
              nb50[:5],                                              # too short           -> per-beta kernel
              [1.0, 1.0, 2.0, 3.0, 5.0, 9.0, 17.0],                  # repeated value      -> per-beta kernel
              [1.0, 1.0 + 2.0 ** -9, 2.0, 4.0, 8.0, 16.0],           # same bucket         -> per-beta kernel
              [1e-7, 1e-3, 1.0, 1e3, 1e6, 1e9, 1e12]]                # > 24 octaves        -> per-beta kernel
    for betas in sweeps:
        _check_notebook(ops, means, stds, pts, lens, betas)
    _check_notebook(ops, means[:-1], stds[:-1], pts, lens, nb50)                 # odd length
    _check_notebook(ops, means[1:], stds[1:], pts, lens, nb50[:32])              # odd length, base not 8-byte aligned
    _check_notebook(ops, means, stds, pts, lens, nb50, want_values=False)


@pytest.mark.timeout(900)
def test_notebook_threshold_kernel_guard_machinery():
    """tools/stress_notebook.py, whose last round puts betas exactly on level-change thresholds of chosen elements:
    (0) as built and (1) with every (element, beta) forced through the literal scan the indices and values equal the
    2047-point brute force; (2) with the guard bands switched off mismatches appear on the same data."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "stress_notebook.py"), "120000", "5"]

    def run(dbg):
        env = dict(os.environ, VBQ_FAST_DEBUG=str(dbg))
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
        return r.returncode, r.stdout + r.stderr[-2000:]
    rc0, out0 = run(0)
    assert rc0 == 0, out0
    rc1, out1 = run(1)
    assert rc1 == 0, out1
    rc2, out2 = run(2)
    assert rc2 == 1, out2


# ---------------------------------------------------------------------------------------------------------------------
# K1e: rank indices of a raw-length lambda sweep (16 to 32 lambdas) from K1t's thresholds
# ---------------------------------------------------------------------------------------------------------------------
def _threshold_lambdas(mu, sg, tab, rng, count):
    """Lambdas that sit on level-change thresholds of chosen elements: the element's own per-level distortions (the
    reference's four rounded f32 operations, utils.py:319-320) give T_n = max_j min_i (du_i - du_j) / (j - i)."""
    lev = O.levels_of_sorted_ranks(N)[np.argsort(np.argsort(tab, kind="stable"), kind="stable")]
    out = []
    for e in rng.permutation(mu.size)[:6 * count]:
        d = np.float32(-0.5) * ((tab - mu[e]) / sg[e]) ** 2
        du = np.array([np.float64(-d[lev == n].max()) for n in range(N + 1)])
        T = [max(min((du[i] - du[j]) / (j - i) for i in range(n + 1)) for j in range(n + 1, N + 1)) for n in range(N)]
        T = [t for t in T if 1e-3 < t < 50.0]
        if T:
            out.append(float(np.float32(T[int(rng.integers(0, len(T)))])))
    out = sorted(set(out))
    keep = [out[0]]
    for v in out[1:]:
        if v > keep[-1] * 1.03:
            keep.append(v)
    return keep[:32]


def test_raw_sweep_threshold_kernel(ops):
    """K1e on K1t's adversarial inputs (exact code-point hits, mid-points, outside the table, extreme sigmas, du values that
    are multiples of powers of two) for sweeps it takes (16-32 lambdas in any order, sorted / reversed / random / on the
    elements' own thresholds) and sweeps it hands to k_quant_fast; odd row counts; the channel-last input form."""
    rng = np.random.default_rng(31)
    tab, _, _ = synth(rng, 4, 1)
    t = np.sort(tab[0])
    mids = (t[:-1] + t[1:]) * np.float32(0.5)
    mu = np.concatenate([t, mids, [t[0] - 50, t[-1] + 50, 0.0], rng.normal(0, 1, 6000).astype(np.float32)]).astype(np.float32)
    sg = np.concatenate([np.full(t.size + mids.size + 3, 1.0), np.exp(rng.normal(-2, 2.5, 6000))]).astype(np.float32)
    sg[::7] = np.float32(0.99999994)
    sg[5::11] = np.float32(2.0 ** -10)
    on_thr = _threshold_lambdas(mu[-6000:], sg[-6000:], tab[0], rng, 60)
    assert len(on_thr) >= 16
    lam16 = [float(v) for v in 2.0 ** np.linspace(-8, 7, 16)]                 # post_process.py:115: the 16-point straight emission
    sweeps = [LAM32, LAM32[::-1], lam16, [lam16[i] for i in rng.permutation(16)],
              [LAM32[i] for i in rng.permutation(32)[:17]], [2.0 ** k for k in range(-10, 9)], on_thr,
              [float(v) for v in np.sort(np.exp(rng.uniform(np.log(1e-3), np.log(1e3), 32)))],
              LAM32[:15],                                            # too short        -> k_quant_fast
              LAM32[:20] + LAM32[:3],                                # repeated values  -> k_quant_fast
              [1e-9 * 4.0 ** k for k in range(20)]]                  # > 16 octaves     -> k_quant_fast
    for lam in sweeps:
        want = CO.quantize(mu[:, None], sg[:, None], tab, lam, N=N, threads=8)[:, :, 0]
        got = ops.quantize(dev(mu), dev(sg), dev(tab), lam, N=N).cpu().numpy()
        assert np.array_equal(got, want), lam
        got = ops.quantize(dev(mu[:-1]), dev(sg[:-1]), dev(tab), lam, N=N).cpu().numpy()         # odd length
        assert np.array_equal(got, want[:, :-1]), lam
    # several channels: planes and the channel-last input form
    tab3, mu3, sg3 = synth(rng, 3001, 5)
    want = CO.quantize(mu3, sg3, tab3, LAM32, N=N, threads=8)                                     # [L, rows, C]
    got = ops.quantize(dev(mu3.T), dev(sg3.T), dev(tab3), LAM32, N=N, layout="cb").cpu().numpy()
    assert np.array_equal(got.transpose(0, 2, 1), want)
    got = ops.quantize(dev(mu3), dev(sg3), dev(tab3), LAM32, N=N, layout="bc->cb").cpu().numpy()
    assert np.array_equal(got.transpose(0, 2, 1), want)


@pytest.mark.timeout(600)
def test_raw_sweep_threshold_kernel_guard_machinery():
    """K1e with every (element, lambda) forced through the literal scan (VBQ_FAST_DEBUG=1) still equals the oracle; with the
    guard bands and the near-tie marks switched off (=2) mismatches appear on lambdas placed on the elements' thresholds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_twopass as t\n"
            "from vbq_amd import ops\n"
            "rng = np.random.default_rng(5)\n"
            "tab, mu, sg = t.synth(rng, 200000, 1)\n"
            "mu, sg = mu[:, 0], sg[:, 0]\n"
            "lam = t._threshold_lambdas(mu[:4000], sg[:4000], tab[0], rng, 80)\n"
            "assert len(lam) >= 16, len(lam)\n"
            "want = t.CO.quantize(mu[:, None], sg[:, None], tab, lam, N=10, threads=8)[:, :, 0]\n"
            "got = ops.quantize(t.dev(mu), t.dev(sg), t.dev(tab), lam, N=10).cpu().numpy()\n"
            "print('MISMATCHES', int((got != want).sum()))\n") % (root, os.path.join(root, "tests"))

    def run(dbg):
        env = dict(os.environ, VBQ_FAST_DEBUG=str(dbg))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=500)
        assert r.returncode == 0, r.stdout + r.stderr[-3000:]
        return int(r.stdout.strip().split("MISMATCHES")[-1])
    assert run(0) == 0
    assert run(1) == 0
    assert run(2) > 0
