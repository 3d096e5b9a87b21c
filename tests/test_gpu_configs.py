"""BASELINE.json configs[2..4] at their FULL sizes on one GPU (the 8-GPU run itself is the driver's):

    configs[2]  word embeddings, one code book: 1e7 (the notebook's own 100 000 x 100, ipynb:169-172) and
                1.2e8 (4e5 x 300) elements, in the image pipeline's arithmetic (K1) and the notebook's (K1n)
    configs[3]  synthetic 1e8-element tensor, the whole alternation moment -> table -> pass 1 -> lengths -> pass 2
                -> models (quantizer.py:82-150 with the notebook's empirical prior, ipynb:373-390)
    configs[4]  one rank's 1.25e8-element shard of the 1e9-element tensor: K1 -> K2 with 32 lambdas, i.e. index
                offsets up to 4e9 elements (past 2^31 and 2^32), 32-bit counters as the all-reduce ships them

At these sizes the oracle cannot re-solve everything in seconds, so each test combines (a) the C oracle on windows
at the start, in the middle / across the 2^30-, 2^31- and 2^32-element marks of the output, and at the end, with (b)
size-independent properties on the whole tensor: table[idx] == Z_hat, level(idx) == bits, every histogram row sums to
the element count, the two routes to the bit-length histogram agree, rate non-increasing in lambda.
Inputs are generated on the device (seeded) so that no multi-GB host arrays are needed; only windows travel.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import c_oracle as CO
from oracle import vbq_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
N = 10
T = 2047
LAM32 = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 32)]
LEV = O.levels_of_sorted_ranks(N)
W = 100_000


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from vbq_amd import ops as _ops
    return _ops


def device_inputs(n, seed):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(seed)
    mu = torch.randn(n, device=dev, generator=g).mul_(1.2329).sub_(0.0799)
    sg = torch.randn(n, device=dev, generator=g).mul_(0.7).sub_(2.0).exp_().clamp_(1e-4, 10)
    return mu, sg


def gaussian_table(std):
    from scipy.stats import norm
    xi = np.concatenate([(np.arange(2 ** k) + 0.5) / 2 ** k for k in range(N + 1)])
    return norm.ppf(xi, scale=float(std)).astype(np.float32)[None, :]                   # [1, T] level-major


def windows(n, extra=()):
    starts = {0, n // 2 - W // 2, n - W}
    starts.update(int(min(max(0, s), n - W)) for s in extra)
    return sorted(starts)


def check_windows(idx, mu, sg, tab, lam, level_len, starts):
    """idx [L, n] against the C oracle on [start, start + W) for every lambda."""
    for s in starts:
        want = CO.quantize(mu[s:s + W].cpu().numpy()[:, None], sg[s:s + W].cpu().numpy()[:, None], tab, lam, N=N,
                           level_len=level_len, threads=16)[:, :, 0]
        got = idx[:, s:s + W].cpu().numpy()
        assert np.array_equal(got, want), f"window at {s}: {int((got != want).sum())} of {got.size} indices differ"


@pytest.mark.parametrize("n", [10_000_000, 120_000_000])
def test_config2_embeddings_k1(ops, n):
    """configs[2] in the image pipeline's f32 arithmetic: 32-lambda sweep over the whole embedding matrix."""
    mu, sg = device_inputs(n, seed=n % 1000 + 1)
    from vbq_amd import embeddings as Emb
    tab = gaussian_table(Emb.empirical_std(mu))
    tabd = torch.from_numpy(tab).cuda()
    idx = ops.quantize(mu, sg, tabd, LAM32, N=N)
    assert tuple(idx.shape) == (32, n)
    # lambda plane l starts at element l * n of the output: marks 2^30, 2^31, 2^32 fall inside planes 8, 17, 35 (n = 1.2e8)
    marks = [((1 << k) % n) - W // 2 for k in (30, 31, 32)] if n > 50_000_000 else []
    check_windows(idx, mu, sg, tab, LAM32, None, windows(n, marks))
    cnt = ops.histogram(idx, 1, N=N)
    assert torch.all(cnt.sum(dim=-1) == n)
    lc = ops.level_counts(mu, sg, tabd, LAM32, N=N)
    from vbq_amd import entropy
    assert torch.equal(lc, entropy.level_counts_from_counts(cnt, N))                      # two routes, full size
    bits = (lc[:, 0, :] * torch.arange(N + 1, device="cuda")).sum(dim=1).cpu().numpy()
    assert np.all(np.diff(bits) <= 0), "total raw rate must not grow with lambda"
    # Z_hat / bits outputs on a lambda subset, full size: table[idx] == Z_hat and level(idx) == bits
    sub = LAM32[3::9]
    i2, zh, bt = ops.quantize(mu, sg, tabd, sub, N=N, want_zhat=True, want_bits=True)
    assert torch.equal(i2.view(torch.int16), idx[3::9].view(torch.int16))
    srt = torch.from_numpy(np.sort(tab[0])).cuda()
    lev = torch.from_numpy(LEV.astype(np.float32)).cuda()
    for k in range(len(sub)):
        q = i2[k].to(torch.int64)
        assert torch.equal(srt[q], zh[k]) and torch.equal(lev[q], bt[k])


@pytest.mark.parametrize("n", [10_000_000, 120_000_000])
def test_config2_embeddings_notebook_arithmetic(ops, n):
    """configs[2] as the notebook computes it (compress_coordinates, ipynb:429-443): f64 brute force over all 2047 code
    points is the oracle; K1n must pick the same slot for every element of the windows, and the histogram behind
    empirical_entropy (ipynb:452-455) must count every element once."""
    from vbq_amd import embeddings as Emb
    mu, sg = device_inputs(n, seed=n % 1000 + 2)
    pts, lens = Emb.make_code_book(Emb.empirical_std(mu), N)
    betas = [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(1e5), 32))]
    idx, _ = ops.quantize_notebook(mu, sg, torch.from_numpy(pts).cuda(), betas, N=N, want_values=False)
    r2s = O.level_major_to_rank(N)
    w = 4000                                                                               # 2047-point brute force per element
    for s in (0, n // 2, ((1 << 31) % n), n - w):
        s = min(s, n - w)
        m, sd = mu[s:s + w].cpu().numpy(), sg[s:s + w].cpu().numpy()
        for k in (0, 7, 16, 25, 31):
            _, slot = CO.compress_coordinates(m, sd, betas[k], pts, lens, threads=16)
            assert np.array_equal(idx[k, s:s + w].cpu().numpy().astype(np.int64), r2s[slot])
    cnt = ops.histogram(idx, 1, N=N)
    assert torch.all(cnt.sum(dim=-1) == n)
    ent = [Emb.entropy_from_counts(cnt[k, 0]) for k in range(32)]
    assert all(np.isfinite(e) and 0 <= e <= n * 11 for e in ent) and ent[0] > ent[-1]
    sub = ops.histogram(idx[:, n - W:].contiguous(), 1, N=N).cpu().numpy()[:, 0]
    for k in (0, 16, 31):
        assert np.array_equal(sub[k], np.bincount(idx[k, n - W:].cpu().numpy().astype(np.int64), minlength=T))


def test_config3_synthetic_1e8_full_alternation(ops):
    """configs[3]: moment pass -> Gaussian table -> pass 1 -> length table -> pass 2 -> models on 1e8 elements, every
    stage against its reference arithmetic (the oracle on windows for the solves, NumPy float32 for the tables)."""
    from vbq_amd import embeddings as Emb, entropy
    from vbq_amd.pipeline import EntropyModelBuild
    n = 100_000_000
    mu, sg = device_inputs(n, seed=3)
    std = Emb.empirical_std(mu)                                                           # K3
    # the moment against an f64 reduction done another way (chunks on the host side of a torch f64 sum)
    ref = float(torch.sqrt((mu.double() ** 2).mean()).item())
    assert abs(float(std) - ref) <= 1e-6 * ref
    tab = gaussian_table(std)
    tabd = torch.from_numpy(tab).cuda()
    build = EntropyModelBuild(n, 1, LAM32, tabd, N=N, add_n_smoothing=1)
    assert build.lut1 is None and build.lut2 is None                                      # >= 2^24 samples per row: host tables
    m2, s2 = mu.reshape(1, n), sg.reshape(1, n)
    build.run(m2, s2)
    torch.cuda.synchronize()
    lc = build.level_counts.cpu().numpy()
    assert lc.sum() == 32 * n and np.all(lc.sum(axis=2) == n)
    # pass 1 = raw lengths: windows of the raw solve through K1, and the K1 -> K2 route to the same histogram
    idx1 = ops.quantize(mu, sg, tabd, LAM32, N=N)
    check_windows(idx1, mu, sg, tab, LAM32, None, windows(n))
    assert torch.equal(entropy.level_counts_from_counts(ops.histogram(idx1, 1, N=N), N), build.level_counts)
    del idx1
    # the length table: the reference's float32 arithmetic on those counts (quantizer.py:105-110, 171-175)
    raw = entropy.neg_log2_freq(lc, 1)
    ll = (np.arange(N + 1, dtype=np.float32) + raw).astype(np.float32)
    assert np.array_equal(build.level_len.cpu().numpy(), ll) and np.array_equal(build.raw_models.cpu().numpy(), raw)
    # pass 2: indices on windows against the oracle WITH that table; histogram totals; models
    marks = [((1 << k) % n) - W // 2 for k in (30, 31)]
    check_windows(build.idx[:, 0, :], mu, sg, tab, LAM32, ll, windows(n, marks))
    cnt = build.counts
    assert torch.all(cnt.sum(dim=-1) == n)
    assert torch.equal(ops.histogram(build.idx, 1, N=N, layout="cb"), cnt.to(torch.int64))
    sub = ops.histogram(build.idx[:, :, :W].contiguous(), 1, N=N, layout="cb").cpu().numpy()[:, 0]
    for k in (0, 13, 31):
        assert np.array_equal(sub[k], np.bincount(build.idx[k, 0, :W].cpu().numpy().astype(np.int64), minlength=T))
    models = entropy.neg_log2_freq(cnt, 1)                                                # :141-146 on the host (65 504 entries)
    assert models.shape == (32, 1, T) and np.all(np.isfinite(models)) and np.all(models > 0)
    # corrected lengths make shallow levels cheaper where they are popular: the average code length estimate must not
    # exceed the raw one by more than the overhead it adds, and rates fall with lambda
    est = [(cnt[k, 0].double().cpu().numpy() * models[k, 0]).sum() / n for k in range(32)]
    assert est[0] > est[-1] and all(np.isfinite(est))


def test_config4_shard_1p25e8_64bit_offsets(ops):
    """configs[4], one rank's shard: 1.25e8 elements x 32 lambdas = 4e9 indices in one output tensor (8 GB)."""
    n = 125_000_000
    mu, sg = device_inputs(n, seed=4)
    tab = gaussian_table(1.2355)
    tabd = torch.from_numpy(tab).cuda()
    rng = np.random.default_rng(4)
    ll = (np.arange(N + 1, dtype=np.float32)[None, None, :] + rng.uniform(0, 6, (32, 1, N + 1)).astype(np.float32)).astype(np.float32)
    idx = ops.quantize(mu, sg, tabd, LAM32, N=N, level_len=torch.from_numpy(ll).cuda())
    assert idx.numel() == 32 * n > 2 ** 31
    marks = [((1 << k) % n) - W // 2 for k in (30, 31, 32)]
    check_windows(idx, mu, sg, tab, LAM32, ll, windows(n, marks))
    c32 = torch.zeros((32, 1, T), dtype=torch.int32, device="cuda")
    ops.histogram(idx, 1, N=N, out=c32)
    c64 = ops.histogram(idx, 1, N=N)
    assert torch.equal(c32.to(torch.int64), c64) and torch.all(c64.sum(dim=-1) == n)
    # the same shard cut into row ranges (what a rank does when it overlaps or streams): identical indices and counts
    idx2 = torch.empty_like(idx)
    cuts = [0, 8 * 1_000_003, 62_500_000, n]
    c2 = torch.zeros_like(c64)
    for a, b in zip(cuts[:-1], cuts[1:]):
        ops.quantize(mu, sg, tabd, LAM32, N=N, level_len=torch.from_numpy(ll).cuda(), out_idx=idx2, rows=(a, b))
        ops.histogram(idx2, 1, N=N, out=c2, rows=(a, b))
    assert torch.equal(idx2.view(torch.int16), idx.view(torch.int16)) and torch.equal(c2, c64)
    lc = ops.level_counts(mu, sg, tabd, LAM32, N=N, level_len=torch.from_numpy(ll).cuda())
    from vbq_amd import entropy
    assert torch.equal(lc, entropy.level_counts_from_counts(c64, N))


def test_config1_kodak24_c256_full_size(ops):
    """configs[1] at the paper's width, the bench's headline tensor: 36864 x 256 latents (channel-last, as they arrive),
    per-channel code books, 32-lambda sweep with raw and with corrected lengths.  Oracle on three row windows of all
    channels and lambdas; on the whole tensor: table[idx] == Z_hat, level(idx) == bits, histogram rows sum to the row
    count, the two routes to the bit-length histogram agree, rate non-increasing in lambda, and the one-call facade on the
    channel-last tensors returns the plane kernels' indices transposed."""
    import vbq_amd
    from vbq_amd import entropy
    from scipy.stats import norm
    B, C = 36864, 256
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(11)
    s_c = np.exp(np.random.default_rng(11).uniform(np.log(0.3), np.log(3.0), C))
    mu = torch.randn((B, C), device=dev, generator=g) * torch.from_numpy(s_c.astype(np.float32)).to(dev)
    sg = torch.randn((B, C), device=dev, generator=g).mul_(0.7).sub_(2.0).exp_().clamp_(1e-4, 10)
    xi = np.concatenate([(np.arange(2 ** k) + 0.5) / 2 ** k for k in range(N + 1)])
    tab = norm.ppf(xi[None, :], scale=s_c[:, None]).astype(np.float32)                  # [C, T]
    tabd = torch.from_numpy(tab).to(dev)
    mu_cb, sg_cb = ops.transpose(mu), ops.transpose(sg)
    rng = np.random.default_rng(12)
    ll = (np.arange(N + 1, dtype=np.float32)[None, None, :] + rng.uniform(0, 4, (32, C, N + 1)).astype(np.float32)).astype(np.float32)
    lld = torch.from_numpy(ll).to(dev)
    wrows = 600
    for lens, lens_h in ((None, None), (lld, ll)):                                    # K1e (raw sweep) and K1 (corrected)
        idx = ops.quantize(mu_cb, sg_cb, tabd, LAM32, N=N, layout="cb", level_len=lens)          # [32, C, B]
        for s in (0, B // 2 - wrows // 2, B - wrows):
            want = CO.quantize(mu[s:s + wrows].cpu().numpy(), sg[s:s + wrows].cpu().numpy(), tab, LAM32, N=N, level_len=lens_h,
                               threads=16)                                               # [32, wrows, C]
            got = idx[:, :, s:s + wrows].permute(0, 2, 1).cpu().numpy()
            assert np.array_equal(got, want), f"rows {s}..: {int((got != want).sum())} of {got.size} indices differ"
        cnt = ops.histogram(idx, C, N=N, layout="cb")
        assert torch.all(cnt.sum(dim=-1) == B)
        lc = ops.level_counts(mu_cb, sg_cb, tabd, LAM32, N=N, layout="cb", level_len=lens)       # K1t / K1h
        assert torch.equal(lc, entropy.level_counts_from_counts(cnt, N))
        if lens is None:
            bits = (lc.sum(dim=1) * torch.arange(N + 1, device=dev)).sum(dim=1).cpu().numpy()
            assert np.all(np.diff(bits) <= 0), "total raw rate must not grow with lambda"
            idx_raw = idx
    sub = LAM32[5::13]
    i2, zh, bt = ops.quantize(mu_cb, sg_cb, tabd, sub, N=N, layout="cb", want_zhat=True, want_bits=True)
    assert torch.equal(i2.view(torch.int16), idx_raw[5::13].view(torch.int16))
    srt = torch.from_numpy(np.sort(tab, axis=1)).to(dev)
    lev = torch.from_numpy(LEV.astype(np.float32)).to(dev)
    for k in range(len(sub)):
        q = i2[k].to(torch.int64)                                                       # [C, B]
        assert torch.equal(torch.gather(srt, 1, q), zh[k]) and torch.equal(lev[q], bt[k])
    fac = vbq_amd.quantize(mu, sg, LAM32, table=tabd)                                 # [32, B, C]
    assert torch.equal(fac.view(torch.int16), idx_raw.permute(0, 2, 1).contiguous().view(torch.int16))


def test_level_counts_single_launch_above_5e8_rows(ops):
    """The counting kernels keep 16-bit partial counters in LDS and size their grids so that no counter can wrap
    (vbq_quantize_fast.hip: max_iters in both launchers).  One C = 1 launch over 5.2e8 rows -- four times the largest shard
    of BASELINE.json -- through K1t (raw lengths), K1h (corrected lengths) and the K1 -> K2 route must agree count for
    count; one wrapped half-word would show up as a difference of 65536."""
    from vbq_amd import entropy
    n = 520_000_000
    mu, sg = device_inputs(n, seed=9)
    tabd = torch.from_numpy(gaussian_table(1.2355)).cuda()
    lam = [0.004, 0.3, 40.0]                                                            # deep, middle and shallow levels busy
    lc_t = ops.level_counts(mu, sg, tabd, lam, N=N)                                     # K1t
    idx = ops.quantize(mu, sg, tabd, lam, N=N)                                          # K1 (3 lambdas: the dense kernel)
    via = entropy.level_counts_from_counts(ops.histogram(idx, 1, N=N), N)
    assert torch.all(lc_t.sum(dim=-1) == n) and torch.equal(lc_t, via)
    del idx
    raw_len = torch.arange(N + 1, dtype=torch.float32, device="cuda").expand(3, 1, -1).contiguous()
    lc_h = ops.level_counts(mu, sg, tabd, lam, N=N, level_len=raw_len)                  # K1h: same lengths through the dense counting kernel
    assert torch.equal(lc_h, via)


@pytest.mark.timeout(1200)
def test_raw_length_sweep_above_2e31_elements(ops):
    """A 16-lambda raw-length sweep (K1e's kind of call) over 2^31 + 2^20 elements: K1e addresses an index plane with a 32-bit
    byte offset, so planes of 2^31 elements or more must go to K1.  69 GB of indices + 17 GB of inputs: it fits.
    Oracle windows at the start, around element 2^31 (where a wrapped offset would land at the front of the plane) and at the
    end; the front window is checked AFTER the whole launch, so a wrapped store would have overwritten it."""
    free, _ = torch.cuda.mem_get_info()
    n = (1 << 31) + (1 << 20)
    if free < 100 * (1 << 30):
        pytest.skip("needs 100 GB of free device memory")
    lam16 = [float(v) for v in 2.0 ** np.linspace(-8, 7, 16)]
    mu, sg = device_inputs(n, seed=8)
    tab = gaussian_table(1.2355)
    idx = ops.quantize(mu, sg, torch.from_numpy(tab).cuda(), lam16, N=N)
    torch.cuda.synchronize()
    assert idx.shape == (16, n)
    check_windows(idx, mu, sg, tab, lam16, None, windows(n, [(1 << 31) - W // 2, (1 << 30) - W // 2, (1 << 20) - W // 2]))
    cnt = ops.histogram(idx, 1, N=N)
    assert torch.all(cnt.sum(dim=-1) == n)
