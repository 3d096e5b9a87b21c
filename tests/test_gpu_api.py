"""The reference-shaped Python surface (vbq_amd.ChannelwisePriorCDFQuantizer, vbq_amd.utils,
vbq_amd.embeddings, vbq_amd.priors, vbq_amd.quantize) on the GPU against the oracle and the
golden vectors.  These read like tests of the reference's own classes."""
import pickle

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import c_oracle as CO
from oracle import vbq_oracle as O

pytestmark = pytest.mark.gpu
N = 10
T = 2 ** (N + 1) - 1


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")


class FakeVAE:
    """encode() returns stored NHWC posterior means / log-variances, decode() is an affine map."""

    def __init__(self, means, logvars):
        self.means, self.logvars = means, logvars

    def encode(self, X):
        return self.means, self.logvars

    def decode(self, Z):
        Z = np.asarray(Z)
        return (0.1 * Z.mean(axis=-1, keepdims=True) + 0.5).repeat(3, axis=-1)


def _case(golden):
    g = golden("g5_batch_quantize.npz")
    from vbq_amd import ChannelwisePriorCDFQuantizer, priors
    C = g["mu"].shape[1]
    q = ChannelwisePriorCDFQuantizer(C, N)
    q.build_code_points(priors.FactoredGaussianPrior(g["ch_mean"], g["ch_std"]))
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(g["ch_mean"], g["ch_std"]))
    return g, q, orc


def test_compress_batch_channel_latents_golden(golden):
    g, q, _ = _case(golden)
    lambs = list(2.0 ** np.linspace(-8, 7.5, 32))          # np.float64 keys, as post_process.py:115
    Zd, Bd = q.compress_batch_channel_latents(g["mu"], g["sigma"], lambs)
    assert list(Zd) == lambs
    for i, lamb in enumerate(lambs):
        assert Zd[lamb].dtype == np.float32 and Bd[lamb].dtype == np.int32
        assert np.array_equal(Zd[lamb], g["zhat_f32"][i]) and np.array_equal(Bd[lamb], g["bits_f32"][i])
    left, right = q.get_all_N_bit_intervals(g["mu"])
    lo, ro = O.get_all_N_bit_intervals(q._search_grids, g["mu"])
    assert np.array_equal(left.cpu().numpy(), lo) and np.array_equal(right.cpu().numpy(), ro)


def test_build_entropy_models_and_compress(golden):
    g, q, orc = _case(golden)
    g8 = golden("g8_corrected_lengths.npz")
    lambs = list(2.0 ** np.linspace(-8, 7.5, 32))
    B, C = g["mu"].shape
    means = g["mu"].reshape(2, 8, B // 16, C)
    logvars = (2 * np.log(g["sigma"])).astype(np.float32).reshape(means.shape)
    vae = FakeVAE(means, logvars)
    X = np.zeros((2, 8, B // 16, 3), np.float32)           # FakeVAE.decode keeps the spatial size
    q.build_entropy_models(X, vae, lambs, add_n_smoothing=1)
    # the oracle runs the same two passes on the stds the quantizer derived (exp(logvar)**0.5 on the device)
    stds = (torch.exp(torch.from_numpy(logvars).cuda()) ** 0.5).cpu().numpy().reshape(B, C)
    lam32 = [np.float32(l) for l in lambs]
    (_, _), (Z2, _) = orc.build_entropy_models(g["mu"], stds, lam32, add_n_smoothing=1)
    for i, lamb in enumerate(lambs):
        assert np.array_equal(q.raw_code_length_entropy_models[lamb], orc.raw_models[lam32[i]])
        assert np.array_equal(q.entropy_models[lamb], orc.entropy_models[lam32[i]])
        assert q.entropy_models[lamb].dtype == np.float32
    if np.array_equal(stds, g["sigma"]):
        assert np.array_equal(np.stack([q.raw_code_length_entropy_models[l] for l in lambs]), g8["raw_models"])
    assert q.lambs == sorted(lambs)
    q = pickle.loads(pickle.dumps(q))                       # post_process.py:106-107 / 163-164
    out = q.compress(X, vae, lambs[::7], clip=True)
    ref = orc.compress_latents(g["mu"], stds, lam32[::7])
    assert set(out) == {"Z_hat", "raw_num_bits", "num_bits_cl", "num_bits", "X_hat"}
    for lamb, l32 in zip(lambs[::7], lam32[::7]):
        assert out["Z_hat"][lamb].shape == means.shape
        assert np.array_equal(out["Z_hat"][lamb].reshape(B, C), ref["Z_hat"][l32])
        assert np.array_equal(out["raw_num_bits"][lamb].reshape(B, C), ref["raw_num_bits"][l32])
        assert np.array_equal(out["num_bits_cl"][lamb], out["raw_num_bits"][lamb])
        assert np.array_equal(out["num_bits"][lamb].reshape(B, C), ref["num_bits"][l32])
        assert out["X_hat"][lamb].shape == X.shape and out["X_hat"][lamb].min() >= 0 and out["X_hat"][lamb].max() <= 1
    # what utils.evaluate_compression_quantizer reads (utils.py:547): total bits per lambda
    bits = [float(np.sum(out["num_bits"][l])) for l in lambs[::7]]
    assert all(np.isfinite(bits))
    # results are ordinary arrays (a kept one does not hold a page-locked block L times its size), and the staging block is
    # reused by the next call without touching what was handed out
    keep = out["Z_hat"][lambs[0]].copy()
    again = q.compress_latents(means, logvars, lambs[7:9])
    assert np.array_equal(out["Z_hat"][lambs[0]], keep) and out["Z_hat"][lambs[0]].base is not None
    assert out["Z_hat"][lambs[0]].base.flags.owndata
    assert np.array_equal(again["Z_hat"][lambs[7]], out["Z_hat"][lambs[7]])
    # device-resident outputs: same values, no device-to-host copy
    dev_out = q.compress_latents(means, logvars, lambs[::7], return_np=False)
    for lamb in lambs[::7]:
        for key in ("Z_hat", "raw_num_bits", "num_bits"):
            t = dev_out[key][lamb]
            assert isinstance(t, torch.Tensor) and t.is_cuda and tuple(t.shape) == means.shape
            assert np.array_equal(t.cpu().numpy(), out[key][lamb])


def test_utils_solvers_golden(golden):
    from vbq_amd import utils
    g5, g6 = golden("g5_batch_quantize.npz"), golden("g6_brute_force.npz")
    orc = O.ChannelwiseOracle(g5["mu"].shape[1], N)
    orc.build_code_points(O.factored_gaussian_icdf(g5["ch_mean"], g5["ch_std"]))
    left, right = O.get_all_N_bit_intervals(orc.grids, g5["mu"])
    P = O.assemble_candidates(left, right)
    Lraw = O.raw_code_lengths(N, *g5["mu"].shape)
    fun = utils.curry_normal_logpdf(loc=g5["mu"], scale=g5["sigma"], ignore_const=True)
    lam32 = [np.float32(l) for l in g5["lambdas"]]
    Zd, Bd = utils.batch_quantize_indep_dims(g5["mu"].shape, P, Lraw.astype(np.float32), fun, lam32)
    for i, l in enumerate(lam32):
        assert np.array_equal(Zd[l], g5["zhat_f32"][i]) and np.array_equal(Bd[l].astype(np.int32), g5["bits_f32"][i])
    lam64 = [float(l) for l in g5["lambdas"]]               # integer lengths: the as-written f64 scores
    Zd, Bd = utils.batch_quantize_indep_dims(g5["mu"].shape, P, Lraw, fun, lam64)
    for i, l in enumerate(lam64):
        assert np.array_equal(Zd[l], g5["zhat_f64"][i]) and np.array_equal(Bd[l], g5["bits_f64"][i])
    # exhaustive single-vector solver over the full sorted code book (utils.py:330-360)
    lens_sorted = np.repeat(O.levels_of_sorted_ranks(N)[None], orc.C, axis=0).astype(np.float32)
    for a, li in enumerate(g6["lam_idx"][:3]):
        for b, r in enumerate(g6["rows"][:6]):
            f_row = utils.curry_normal_logpdf(loc=g5["mu"][r], scale=g5["sigma"][r], ignore_const=True)
            z, nb = utils.quantize_indep_dims(g5["mu"][r], orc.by_channel, lens_sorted, f_row, lam32[li])
            assert np.array_equal(z, g6["zhat"][a, b]) and np.array_equal(nb, g6["bits"][a, b])


def test_embeddings_golden(golden):
    from vbq_amd import embeddings as E
    g = golden("g7_notebook.npz")
    es = E.empirical_std(g["means"])
    assert es.dtype == np.float32 and es == g["empirical_std"]            # NumPy's float32 summation order, on the GPU
    pts, lens = E.make_code_book(es, 10)
    assert np.array_equal(pts, g["codepoints"]) and np.array_equal(lens, g["lengths"])
    for i, beta in enumerate(g["betas"]):
        out = E.compress_coordinates(g["means"], g["stds"], float(beta), bitlengths=lens, codepoints=pts)
        assert out.dtype == np.float32 and out.shape == g["means"].shape
        # the notebook hands the result on (ipynb:466-470): entropy (and the ranks) are taken where it lives -- no host copy yet
        assert E.empirical_entropy(out) == pytest.approx(g["entropy"][i], rel=1e-12) and out.on_device
        assert np.array_equal(out, g["optima"][i]) and not out.on_device
        assert E.empirical_entropy(np.asarray(out)) == pytest.approx(g["entropy"][i], rel=1e-12)
        comp, bits = E.test_beta(g["means"], g["stds"], float(beta), pts)
        assert bits == pytest.approx(g["entropy"][i], rel=1e-12)
    idx, _ = E.compress_coordinates_sweep(g["means"], g["stds"], list(g["betas"]), pts, want_values=False)
    assert E.entropy_from_indices(idx) == pytest.approx(list(g["entropy"]), rel=1e-12)


def test_bmshj_prior_inverse_cdf_and_table():
    from vbq_amd import ChannelwisePriorCDFQuantizer, priors
    rng = np.random.default_rng(2)
    C = 4
    p = priors.BMSHJ2018Prior(C, init_scale=1.0, seed=3)
    p.set_weights([w + rng.normal(0, 0.3, w.shape).astype(np.float32) for w in p.get_weights()])
    ref = O.BMSHJ2018Oracle(*p.effective_parameters())
    x = rng.normal(0, 2, (5, 7, C)).astype(np.float32)
    c, d = p.cdf_pdf(x)
    rc, rd = ref.cdf_pdf(x)
    assert c.shape == x.shape and np.allclose(c, rc, rtol=2e-6, atol=2e-7) and np.allclose(d, rd, rtol=1e-4, atol=1e-7)
    assert np.allclose(p.logpdf(x), ref.logpdf(x), rtol=1e-4, atol=1e-5)
    xi = np.repeat(O.dyadic_xi(N)[:, None], C, axis=1)
    z = p.inverse_cdf(xi)
    zr = ref.inverse_cdf(xi)
    assert z.dtype == np.float32 and z.shape == xi.shape
    assert np.allclose(p.cdf(z), xi, atol=3e-6)                      # cdf(icdf(xi)) ~= xi
    assert np.allclose(z, zr, rtol=1e-4, atol=1e-4)                   # vs the NumPy restatement (tolerance: unpinned)
    assert abs(p.last_iterations - ref.last_iterations) <= 2
    q = ChannelwisePriorCDFQuantizer(C, N)
    q.build_code_points(p)                                            # post_process.py:103
    mu = rng.normal(0, 1, (300, C)).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, (300, C))).astype(np.float32)
    Zd, Bd = q.compress_batch_channel_latents(mu, sg, [0.01, 1.0])
    zi, zh, bt = CO.quantize(mu, sg, q.all_code_points, [0.01, 1.0], N=N, want_zhat=True, want_bits=True)
    assert np.array_equal(Zd[1.0], zh[1]) and np.array_equal(Bd[0.01], bt[0].astype(np.int32))


def test_quantize_facade():
    import vbq_amd
    rng = np.random.default_rng(8)
    mu = rng.normal(0, 1.2, 5000).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, 5000)).astype(np.float32)
    tab = vbq_amd.gaussian_table(1.2329, N)
    idx = vbq_amd.quantize(mu, sg, 0.37, table=tab)
    assert idx.dtype == np.uint16 and idx.shape == mu.shape
    assert np.array_equal(idx, CO.quantize(mu, sg, tab, [0.37], N=N)[0, :, 0])
    lam = [0.01, 1.0, 50.0]
    idx3, val3 = vbq_amd.quantize(torch.from_numpy(mu).cuda(), torch.from_numpy(sg).cuda(), lam, table=tab,
                                  return_values=True)
    wi, wz = CO.quantize(mu, sg, tab, lam, N=N, want_zhat=True)
    assert idx3.is_cuda and np.array_equal(idx3.cpu().numpy(), wi[:, :, 0]) and np.array_equal(val3.cpu().numpy(), wz[:, :, 0])
    from vbq_amd import priors
    idx_p = vbq_amd.quantize(mu, sg, 0.37, prior=priors.FactoredGaussianPrior(np.zeros(1), np.array([1.2329])))
    assert np.array_equal(idx_p, idx)
    # R-D Lagrangian gate of the north star: within 1e-5 relative of the oracle's (trivially, indices equal)
    lev = O.levels_of_sorted_ranks(N)
    srt = np.sort(tab[0])
    lag = O.lagrangian(mu, sg, srt[idx], lev[idx], 0.37)
    lag_ref = O.lagrangian(mu, sg, srt[wi[0, :, 0]] if False else srt[CO.quantize(mu, sg, tab, [0.37], N=N)[0, :, 0]],
                           lev[CO.quantize(mu, sg, tab, [0.37], N=N)[0, :, 0]], 0.37)
    assert abs(lag - lag_ref) <= 1e-5 * abs(lag_ref)


def test_quantize_facade_channel_last_takes_the_plane_kernels(monkeypatch):
    """vbq_amd.quantize(mu[B, C], sigma[B, C], lmbda, table=...) -- the literal north-star call on latents as they arrive
    (quantizer.py:90-91): transposes + plane kernels + one batched transpose back, results in the caller's layout, equal to
    the oracle; the table is validated once per object, not per call."""
    import vbq_amd
    from vbq_amd import api, ops
    rng = np.random.default_rng(21)
    B, C = 777, 24                                            # odd row count: the scalar transpose paths
    s_c = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    mu = (rng.standard_normal((B, C)) * s_c).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, (B, C))).astype(np.float32)
    tab = vbq_amd.gaussian_table(s_c, N)
    lam = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 32)]
    wi, wz, wb = CO.quantize(mu, sg, tab, lam, N=N, want_zhat=True, want_bits=True)
    calls = []
    real = ops.quantize
    monkeypatch.setattr(ops, "quantize", lambda *a, **k: (calls.append(k.get("layout")), real(*a, **k))[1])
    idx = vbq_amd.quantize(mu, sg, lam, table=tab)            # raw-length sweep of 32: K1e on planes
    assert calls == ["cb"]
    assert idx.shape == (32, B, C) and idx.dtype == np.uint16 and np.array_equal(idx, wi)
    mu_d, sg_d, tab_d = (torch.from_numpy(a).cuda() for a in (mu, sg, tab))
    i2, z2, b2 = vbq_amd.quantize(mu_d, sg_d, lam[:5], table=tab_d, return_values=True, return_bits=True)
    assert i2.is_cuda and tuple(i2.shape) == (5, B, C)
    assert np.array_equal(i2.cpu().numpy(), wi[:5]) and np.array_equal(z2.cpu().numpy(), wz[:5]) and np.array_equal(b2.cpu().numpy(), wb[:5])
    one = vbq_amd.quantize(mu_d, sg_d, lam[7], table=tab_d)
    assert tuple(one.shape) == (B, C) and np.array_equal(one.cpu().numpy(), wi[7])
    ll = (np.arange(N + 1, dtype=np.float32) + np.abs(rng.normal(0, 1, (3, C, N + 1)))).astype(np.float32)
    i3 = vbq_amd.quantize(mu_d, sg_d, lam[10:13], table=tab_d, lengths=ll)
    assert np.array_equal(i3.cpu().numpy(), CO.quantize(mu, sg, tab, lam[10:13], N=N, level_len=ll))
    icb = vbq_amd.quantize(mu_d.t().contiguous(), sg_d.t().contiguous(), lam[:3], table=tab_d, layout="cb")
    assert np.array_equal(icb.cpu().numpy().transpose(0, 2, 1), wi[:3])
    # the table check runs once per table object: the second call with the same tensor does not copy it to the host
    key_hits = len(api._CHECKED)
    copies = []
    real_cpu = torch.Tensor.cpu
    monkeypatch.setattr(torch.Tensor, "cpu", lambda t, *a, **k: (copies.append(tuple(t.shape)), real_cpu(t, *a, **k))[1])
    vbq_amd.quantize(mu_d, sg_d, lam[:2], table=tab_d)
    monkeypatch.setattr(torch.Tensor, "cpu", real_cpu)
    assert (C, 2 ** (N + 1) - 1) not in copies and len(api._CHECKED) == key_hits
    bad = tab.copy()
    bad[3, 5] = 1e9                                           # level 2's points no longer ascend
    with pytest.raises(ValueError):
        vbq_amd.quantize(mu, sg, 1.0, table=bad)
    assert vbq_amd.quantize(mu, sg, 1.0, table=bad, validate=False).shape == (B, C)   # the caller's responsibility then
    tab_d[3, 5] = 1e9                                         # an in-place edit of a checked tensor is checked again
    with pytest.raises(ValueError):
        vbq_amd.quantize(mu_d, sg_d, 1.0, table=tab_d)


def test_rd_sums_against_numpy():
    """vbq_rd_sums_u16: distortion and rate sums per lambda (f64) == NumPy f64 on the same indices, both layouts, with and
    without a rate table, lambda counts on both sides of the kernel's chunk of eight."""
    from vbq_amd import ops
    rng = np.random.default_rng(31)
    rows, C, T = 999, 5, 2 ** (N + 1) - 1
    mu = rng.normal(0, 1.2, (rows, C)).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, (rows, C))).astype(np.float32)
    srt = np.sort(rng.normal(0, 1.5, (C, T)).astype(np.float32), axis=1)
    ch = np.arange(C)[None, :]
    for L in (1, 8, 11, 32):
        idx = rng.integers(0, T, (L, rows, C)).astype(np.uint16)
        rate = np.abs(rng.normal(6, 3, (L, C, T))).astype(np.float32)
        z = np.stack([srt[ch, idx[l].astype(np.int64)] for l in range(L)]).astype(np.float64)
        want_d = ((z - mu.astype(np.float64)) ** 2 / (2.0 * sg.astype(np.float64) ** 2)).reshape(L, -1).sum(1)
        want_r = np.stack([rate[l][ch, idx[l].astype(np.int64)].astype(np.float64).sum() for l in range(L)])
        got = ops.rd_sums(torch.from_numpy(mu).cuda(), torch.from_numpy(sg).cuda(), torch.from_numpy(idx).cuda(),
                          torch.from_numpy(srt).cuda(), C, N=N, layout="bc", rate=torch.from_numpy(rate).cuda()).cpu().numpy()
        assert np.allclose(got[:, 0], want_d, rtol=1e-12) and np.allclose(got[:, 1], want_r, rtol=1e-12)
        got = ops.rd_sums(torch.from_numpy(np.ascontiguousarray(mu.T)).cuda(), torch.from_numpy(np.ascontiguousarray(sg.T)).cuda(),
                          torch.from_numpy(np.ascontiguousarray(idx.transpose(0, 2, 1))).cuda(), torch.from_numpy(srt).cuda(), C, N=N,
                          layout="cb", rate=torch.from_numpy(rate[0]).cuda()).cpu().numpy()
        want_r0 = np.stack([rate[0][ch, idx[l].astype(np.int64)].astype(np.float64).sum() for l in range(L)])
        assert np.allclose(got[:, 0], want_d, rtol=1e-12) and np.allclose(got[:, 1], want_r0, rtol=1e-12)
        got = ops.rd_sums(torch.from_numpy(mu).cuda(), torch.from_numpy(sg).cuda(), torch.from_numpy(idx).cuda(),
                          torch.from_numpy(srt).cuda(), C, N=N, layout="bc").cpu().numpy()
        assert np.allclose(got[:, 0], want_d, rtol=1e-12) and np.all(got[:, 1] == 0)


def test_transpose_planes_u16_and_f32():
    from vbq_amd import ops
    rng = np.random.default_rng(2)
    for shape in ((3, 64, 128), (2, 37, 53), (5, 8, 8), (1, 1, 7), (4, 256, 1536)):
        x = torch.from_numpy(rng.integers(0, 2047, shape).astype(np.uint16)).cuda()
        got = ops.transpose_planes(x)
        assert np.array_equal(got.cpu().numpy(), x.cpu().numpy().transpose(0, 2, 1))
        f = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
        assert torch.equal(ops.transpose_planes(f), f.permute(0, 2, 1).contiguous())
    with pytest.raises(ValueError):
        ops.transpose_planes(torch.zeros((4, 4), dtype=torch.float32, device="cuda"))


def _torch_nll(raw, x):
    """f64 torch restatement of loss = -mean(log(pdf + 1e-10)) with pdf = d cdf / dx by autograd
    (learned_prior.py:150-171, 405-408): the independent reference for the hand-written gradient."""
    mats, bias, fac = raw
    xx = x.clone().requires_grad_(True)                       # [n, C]
    h = xx.t()[:, None, :]                                    # [C, 1, n]
    for i in range(4):
        h = torch.matmul(torch.nn.functional.softplus(mats[i]), h) + bias[i]
        if i < 3:
            h = h + torch.tanh(fac[i]) * torch.tanh(h)
    cdf = torch.sigmoid(h)[:, 0, :].t()                       # [n, C]
    pdf, = torch.autograd.grad(cdf.sum(), xx, create_graph=True)
    return -(torch.log(pdf + 1e-10)).mean()


def test_bmshj_fit_gradient_and_convergence():
    from vbq_amd import ops, priors
    rng = np.random.default_rng(4)
    C, n = 3, 4000
    p = priors.BMSHJ2018Prior(C, init_scale=2.0, seed=5)
    p.set_weights([w + rng.normal(0, 0.2, w.shape).astype(np.float32) for w in p.get_weights()])
    data = np.stack([rng.normal(0.5, 0.7, n), rng.standard_t(4, n) * 0.5,
                     np.where(rng.random(n) < 0.5, rng.normal(-1, 0.3, n), rng.normal(1.2, 0.4, n))], axis=1).astype(np.float32)
    x_cb = ops.transpose(torch.from_numpy(data).cuda())
    loss, grads = p.loss_and_grads(x_cb)
    raw = ([torch.tensor(m, dtype=torch.float64, requires_grad=True) for m in p.matrices],
           [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in p.biases],
           [torch.tensor(f, dtype=torch.float64, requires_grad=True) for f in p.factors])
    ref = _torch_nll(raw, torch.tensor(data, dtype=torch.float64))
    ref.backward()
    assert loss == pytest.approx(float(ref.detach()), rel=2e-5)
    want = []
    for i in range(4):
        want += [raw[0][i].grad.numpy(), raw[1][i].grad.numpy()] + ([raw[2][i].grad.numpy()] if i < 3 else [])
    for g, w in zip(grads, want):
        assert g.shape == w.shape
        assert np.allclose(g, w, rtol=5e-3, atol=2e-5), np.abs(g - w).max()
    # the fit: NLL goes down and ends below a single Gaussian's on the bimodal channel
    p2 = priors.BMSHJ2018Prior(C, init_scale=1.0, seed=1)
    l0, _ = p2.loss_and_grads(x_cb)
    rec = p2.fit(data, lr=0.1, its=150, tol=1e-2, logging_freq=10)          # post_process.py:78 settings, fewer its
    assert len(rec) >= 15 and rec[-1]["loss"] < rec[0]["loss"] < l0
    assert p2.last_loss < l0 - 0.3
    lp = p2.logpdf(data)
    gauss_nll = 0.5 * np.log(2 * np.pi * data[:, 2].var()) + 0.5
    assert -lp[:, 2].mean() < gauss_nll
    xi = np.repeat(O.dyadic_xi(6)[:, None], C, axis=1)
    z = p2.inverse_cdf(xi)
    assert np.all(np.diff(z[np.argsort(O.dyadic_xi(6))], axis=0) > 0)         # a usable, monotone code book
    q = np.quantile(data, [0.25, 0.5, 0.75], axis=0)                          # fitted quartiles ~ empirical
    zq = p2.inverse_cdf(np.repeat(np.array([0.25, 0.5, 0.75])[:, None], C, axis=1))
    assert np.allclose(zq, q, atol=0.15)


def test_quantizer_bitstream_roundtrip(golden):
    """encode_batch / decode_batch: the decoded Z_hat equals what compress_batch_channel_latents returns, and
    the coded size is within 2 % of the reference-style estimate sum(num_bits) on the data the models were fit on."""
    g, q, _ = _case(golden)
    lambs = list(2.0 ** np.linspace(-8, 7.5, 6))
    mu = np.tile(g["mu"], (20, 1)) + np.random.default_rng(0).normal(0, 0.05, (20 * g["mu"].shape[0], 4)).astype(np.float32)
    sg = np.tile(g["sigma"], (20, 1))
    q.build_entropy_models_from_latents(mu, sg, lambs, add_n_smoothing=1)
    words, sizes, cdc = q.encode_batch(mu, sg, lambs, segment=512)
    Zd, _ = q.compress_batch_channel_latents(mu, sg, lambs)
    back = q.decode_batch(words, sizes, cdc, mu.shape[0], lambs)
    for lamb in lambs:
        assert np.array_equal(back[lamb], Zd[lamb])
    est = sum(float(np.sum(q.compress_latents(mu[None], 2 * np.log(sg)[None], [lamb])["num_bits"][lamb])) for lamb in lambs[:2])
    bits2 = cdc.compressed_bits(sizes[: 2 * q.num_channels])
    assert bits2 <= 1.03 * est + 40 * sizes[: 2 * q.num_channels].numel()


def test_quantizer_with_repeated_code_points():
    """Tables whose f32 cast repeats values (SURVEY 7.2 hard part 4): qidx is the FIRST sorted position of a
    repeated value (quantizer.py:135), so the histograms / entropy models merge the duplicates exactly as the
    reference's searchsorted does."""
    from scipy.stats import norm
    from vbq_amd import ChannelwisePriorCDFQuantizer

    class Coarse:
        def inverse_cdf(self, xi):
            return np.round(norm.ppf(xi) * np.array([24.0, 64.0])) / np.array([24.0, 64.0])
    C = 2
    q = ChannelwisePriorCDFQuantizer(C, N)
    q.build_code_points(Coarse())
    assert not q._strict
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(Coarse().inverse_cdf)
    rng = np.random.default_rng(21)
    mu = rng.normal(0, 1.1, (4000, C)).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, (4000, C))).astype(np.float32)
    lambs = [0.01, 0.3, 4.0]
    q.build_entropy_models_from_latents(mu, sg, lambs, add_n_smoothing=1)
    orc.build_entropy_models(mu, sg, [np.float32(l) for l in lambs], add_n_smoothing=1)
    for l in lambs:
        assert np.array_equal(q.raw_code_length_entropy_models[l], orc.raw_models[np.float32(l)])
        assert np.array_equal(q.entropy_models[l], orc.entropy_models[np.float32(l)])
    out = q.compress_latents(mu[None], (2 * np.log(sg))[None], lambs)
    stds = (torch.exp(torch.from_numpy(2 * np.log(sg)).cuda()) ** 0.5).cpu().numpy()
    if np.array_equal(stds, sg):
        ref = orc.compress_latents(mu, sg, [np.float32(l) for l in lambs])
        for l in lambs:
            assert np.array_equal(out["Z_hat"][l][0], ref["Z_hat"][np.float32(l)])
            assert np.array_equal(out["num_bits"][l][0], ref["num_bits"][np.float32(l)])


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 5), (64, 64), (100, 36), (260, 132), (36864 // 8, 256), (67, 128), (128, 67)])
def test_transpose_scalar_and_vector_paths(rows, cols):
    """vbq_transpose_f32: the 16-byte path (rows, cols multiples of 4) and the scalar path, ragged tiles included."""
    from vbq_amd import ops
    rng = np.random.default_rng(rows * 1000 + cols)
    x = rng.normal(size=(rows, cols)).astype(np.float32)
    got = ops.transpose(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.shape == (cols, rows) and np.array_equal(got, x.T)


def test_model_table_cache_follows_replaced_models(golden):
    """The device copies of the entropy-model tables are cached across compress_latents calls and must be rebuilt
    when a model array is replaced (assigning a new array, or a whole new dict)."""
    g, q, _ = _case(golden)
    rng = np.random.default_rng(12)
    C = q.num_channels
    means = rng.normal(0, 1, (1, 6, 5, C)).astype(np.float32)
    logvars = rng.normal(-4, 0.5, (1, 6, 5, C)).astype(np.float32)
    lambs = [0.1, 2.0]
    q.raw_code_length_entropy_models = None
    q.entropy_models = {l: rng.uniform(1, 12, (C, T)).astype(np.float32) for l in lambs}
    a = q.compress_latents(means, logvars, lambs)
    b = q.compress_latents(means, logvars, lambs)                     # served from the cache
    assert all(np.array_equal(a["num_bits"][l], b["num_bits"][l]) for l in lambs)
    q.entropy_models[lambs[1]] = q.entropy_models[lambs[1]] + np.float32(1.0)     # a new array under the same key
    c = q.compress_latents(means, logvars, lambs)
    assert np.array_equal(c["num_bits"][lambs[0]], a["num_bits"][lambs[0]])
    assert np.array_equal(c["num_bits"][lambs[1]], a["num_bits"][lambs[1]] + np.float32(1.0))
    q.entropy_models = {l: np.zeros((C, T), np.float32) for l in lambs}
    d = q.compress_latents(means, logvars, lambs)
    assert all(not d["num_bits"][l].any() for l in lambs)
    # an edit IN PLACE cannot be seen by an identity-keyed cache: cached arrays are read-only, so it fails loudly
    with pytest.raises(ValueError, match="read-only"):
        q.entropy_models[lambs[0]][0, 0] = 5.0


def test_entry_points_are_graph_capturable():
    """The C-ABI launches only stream-ordered work on the caller's stream (no allocation, no synchronisation), so a
    whole entropy-model pass -- layout change, K1, K2 -- can be captured into a HIP graph and replayed."""
    from vbq_amd import ops
    rng = np.random.default_rng(31)
    rows, C, L = 640, 8, 5
    lambs = [0.05, 0.3, 1.0, 7.0, 60.0]
    from scipy.stats import norm
    xi = np.concatenate([(np.arange(2 ** n) + 0.5) / 2 ** n for n in range(N + 1)])
    tab = torch.from_numpy(norm.ppf(xi[None, :], scale=np.linspace(0.5, 2.0, C)[:, None]).astype(np.float32)).cuda()
    mu_in = torch.from_numpy(rng.normal(0, 1, (rows, C)).astype(np.float32)).cuda()
    sg_in = torch.from_numpy(np.exp(rng.normal(-2, 0.6, (rows, C))).astype(np.float32)).cuda()
    mu, sg = torch.empty((C, rows), device="cuda"), torch.empty((C, rows), device="cuda")
    idx = torch.empty((L, C, rows), dtype=torch.uint16, device="cuda")
    cnt = torch.zeros((L, C, T), dtype=torch.int32, device="cuda")
    ws = torch.empty(ops._lib.lib().vbq_quantize_workspace_bytes(C, L, N), dtype=torch.uint8, device="cuda")

    def step():
        ops.transpose(mu_in, out=mu)
        ops.transpose(sg_in, out=sg)
        ops.quantize(mu, sg, tab, lambs, N=N, layout="cb", out_idx=idx, workspace=ws)
        cnt.zero_()
        ops.histogram(idx, C, N=N, layout="cb", out=cnt)

    step()
    torch.cuda.synchronize()
    want_idx, want_cnt = idx.clone(), cnt.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()                                           # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        step()
    # new inputs, replay: the graph must recompute from the (updated) input buffers
    mu_in.copy_(torch.from_numpy(rng.normal(0, 1, (rows, C)).astype(np.float32)).cuda())
    idx.zero_()
    g.replay()
    torch.cuda.synchronize()
    step_idx, step_cnt = idx.clone(), cnt.clone()
    step()
    torch.cuda.synchronize()
    assert torch.equal(step_idx.view(torch.int16), idx.view(torch.int16)) and torch.equal(step_cnt, cnt)
    assert not torch.equal(step_idx.view(torch.int16), want_idx.view(torch.int16))
    assert int(cnt.sum()) == L * C * rows


def test_quantizer_class_at_eleven_bits():
    """ChannelwisePriorCDFQuantizer(max_bits_per_coord=11) end to end against the oracle class: tables, two-pass
    entropy models, compress_latents dict."""
    from vbq_amd import ChannelwisePriorCDFQuantizer, priors
    Nb, C = 11, 3
    rng = np.random.default_rng(77)
    ch_std = np.array([0.7, 1.0, 2.0])
    q = ChannelwisePriorCDFQuantizer(C, Nb)
    q.build_code_points(priors.FactoredGaussianPrior(np.zeros(C), ch_std))
    orc = O.ChannelwiseOracle(C, Nb)
    orc.build_code_points(O.factored_gaussian_icdf(np.zeros(C), ch_std))
    assert np.array_equal(q.all_code_points, orc.all_code_points)
    mu = (ch_std * rng.normal(0, 1, (900, C))).astype(np.float32)
    sg = np.exp(rng.normal(-3, 0.8, (900, C))).astype(np.float32)
    lambs = [0.02, 0.5, 8.0]
    q.build_entropy_models_from_latents(mu, sg, lambs, add_n_smoothing=1)
    orc.build_entropy_models(mu, sg, lambs, add_n_smoothing=1)
    for l in lambs:
        assert np.array_equal(q.raw_code_length_entropy_models[l], orc.raw_models[l])
        assert np.array_equal(q.entropy_models[l], orc.entropy_models[l])
    means = mu[:600].reshape(1, 20, 30, C)
    logvars = (2 * np.log(sg[:600])).reshape(1, 20, 30, C).astype(np.float32)
    out = q.compress_latents(means, logvars, lambs)
    sg2 = np.exp(logvars.reshape(-1, C).astype(np.float32)) ** np.float32(0.5)
    want = orc.compress_latents(mu[:600], sg2, lambs)
    for l in lambs:
        assert np.array_equal(out["Z_hat"][l].reshape(-1, C), want["Z_hat"][l])
        assert np.array_equal(out["num_bits"][l].reshape(-1, C), want["num_bits"][l])


def test_g12_duplicates_and_edges_on_gpu(golden):
    """g12 (reference exhaustive solver on tables with repeated float32 code points and on inputs beyond / at the rim of
    every level's grid): K1's Z_hat / bit lengths, in both layouts, the counting kernel, and the quantizer class with
    its canonical qidx (quantizer.py:135,223)."""
    from vbq_amd import ChannelwisePriorCDFQuantizer, ops, priors
    g = golden("g12_duplicates_edges.npz")
    lam = [float(l) for l in g["lambdas"]]
    tab = torch.from_numpy(g["all_code_points"]).cuda()
    mu, sg = torch.from_numpy(g["mu"]).cuda(), torch.from_numpy(g["sigma"]).cuda()
    for layout, m, s in (("bc", mu, sg), ("cb", mu.t().contiguous(), sg.t().contiguous())):
        idx, zh, bt = ops.quantize(m, s, tab, lam, N=N, layout=layout, want_zhat=True, want_bits=True)
        zh, bt = zh.cpu().numpy(), bt.cpu().numpy()
        if layout == "cb":
            zh, bt = zh.transpose(0, 2, 1), bt.transpose(0, 2, 1)
        assert np.array_equal(zh, g["zhat"]) and np.array_equal(bt.astype(np.int32), g["bits"])
    lc = ops.level_counts(mu.t().contiguous(), sg.t().contiguous(), tab, lam, N=N, layout="cb").cpu().numpy()
    want = np.stack([[np.bincount(g["bits"][l][:, c], minlength=N + 1) for c in range(mu.shape[1])] for l in range(len(lam))])
    assert np.array_equal(lc, want)
    q = ChannelwisePriorCDFQuantizer(mu.shape[1], N)
    q.build_code_points(priors.FactoredGaussianPrior(g["ch_mean"], g["ch_std"]))
    assert np.array_equal(q.all_code_points, g["all_code_points"]) and not q._strict
    Zd, Bd = q.compress_batch_channel_latents(g["mu"], g["sigma"], lam)
    for i, l in enumerate(lam):
        assert np.array_equal(Zd[l], g["zhat"][i]) and np.array_equal(Bd[l], g["bits"][i])
    left, right = q.get_all_N_bit_intervals(g["mu"])                  # repeated points and the rims of every level's grid
    lo, ro = O.get_all_N_bit_intervals(q._search_grids, g["mu"])
    assert np.array_equal(left.cpu().numpy(), lo) and np.array_equal(right.cpu().numpy(), ro)


def test_g13_notebook_chain_on_gpu(golden):
    """The notebook's chain as it runs (ipynb:373-390, 429-455): moment on the GPU in NumPy's float32 order -> code book
    from THAT number -> K1n -> K2 entropy; every stage equal to what the notebook's cells produced."""
    from vbq_amd import embeddings as E
    g = golden("g13_notebook_chain.npz")
    es = E.empirical_std(g["means"])
    assert es.dtype == np.float32 and es == g["empirical_std"]
    assert abs(float(E.empirical_std(g["means"], exact=False)) - float(es)) <= 1e-6 * float(es)
    pts, lens = E.make_code_book(es, 10)
    assert np.array_equal(pts, g["codepoints"]) and np.array_equal(lens, g["lengths"])
    for i, beta in enumerate(g["betas"]):
        out = E.compress_coordinates(g["means"], g["stds"], float(beta), bitlengths=lens, codepoints=pts)
        assert out.dtype == np.float32 and np.array_equal(out, g["optima"][i])
        assert E.empirical_entropy(out) == pytest.approx(g["entropy"][i], rel=1e-12)


@pytest.mark.parametrize("n", [0, 1, 7, 8, 100, 129, 8191, 8192, 8193, 3 * 8192, 100_000, 1_000_003, 10_000_000])
def test_numpy_order_moment(n):
    """vbq_numpy_sum_sq_f32 against the C restatement of NumPy's summation order (pinned to np.sum on the CPU side)
    and, since NumPy is at hand, against np.sum itself."""
    from oracle import c_oracle as CO
    from vbq_amd import ops
    rng = np.random.default_rng(n)
    x = rng.normal(-0.0799, 1.2329, n).astype(np.float32)
    got = np.float32(ops.numpy_sum_sq(torch.from_numpy(x).cuda()).cpu().numpy()[0])
    assert got == CO.numpy_sum_sq_f32(x)
    if n:
        assert got == np.sum(x ** 2)


def test_xi_space_encoder_golden_g3_g4(golden):
    """The reference's first encoder (utils.py:215-304) through vbq_amd.utils: the interval search (g3: the numba kernel's
    outputs) and the whole encode_vectorized (g4: z_hat, num_bits, xi_hat, score for 5 lambdas with scipy's norm.cdf /
    norm.ppf / norm.logpdf as the caller's functions), bit for bit, plus the rims and exact grid points."""
    from scipy.stats import norm
    from vbq_amd import utils
    g3 = golden("g3_xi_intervals.npz")
    n = int(g3["N"])
    left = np.empty((n + 1, g3["x"].size))
    right = np.empty_like(left)
    utils.get_all_N_bit_intervals(g3["x"], n, left, right)
    assert np.array_equal(left, g3["left"]) and np.array_equal(right, g3["right"])
    x = np.array([0.0, 1.0, 0.5, 0.25, 0.75, 2.0 ** -11, 1 - 2.0 ** -11, 0.4375, 1 / 3, np.nextafter(0.5, 1), np.nextafter(0.5, 0)])
    l2, r2 = np.empty((17, x.size)), np.empty((17, x.size))
    utils.get_all_N_bit_intervals(x, 16, l2, r2)
    lo, ro = O.xi_intervals(x, 16)
    assert np.array_equal(l2, lo) and np.array_equal(r2, ro)
    assert (l2[2, 7], r2[2, 7]) == (0.375, 0.625)                      # the reference docstring's known answer (utils.py:31-32)
    g4 = golden("g4_encode_vectorized.npz")
    mu, sigma = g4["mu"], g4["sigma"]
    fun = lambda z: norm.logpdf(z, loc=mu, scale=sigma)                # what curry_normal_logpdf(backend=np) returns (utils.py:314-316)
    for i, lamb in enumerate(g4["lambs"]):
        r = utils.encode_vectorized(fun, mu, float(lamb), norm.cdf, norm.ppf, max_bits_per_coord=int(g4["N"]))
        assert np.array_equal(r["z_hat"], g4["z_hat"][i]) and np.array_equal(r["xi_hat"], g4["xi_hat"][i])
        assert np.array_equal(r["num_bits"], g4["num_bits"][i]) and r["num_bits"].dtype == np.int64
        assert r["score"] == g4["score"][i]
        ref = O.encode_vectorized(fun, mu, float(lamb), norm.cdf, norm.ppf, int(g4["N"]))
        assert all(np.array_equal(r[k], ref[k]) for k in ("z_hat", "xi_hat", "num_bits")) and r["score"] == ref["score"]


def test_input_validation_is_available_and_off_by_default():
    from vbq_amd import ChannelwisePriorCDFQuantizer, ops, priors
    mu = torch.randn(5000, device="cuda")
    sg = torch.rand(5000, device="cuda") + 0.1
    ops.check_inputs(mu, sg)
    for bad_mu, bad_sg, what in ((float("nan"), 1.0, "1 non-finite means"), (float("inf"), 1.0, "1 non-finite means"),
                                 (0.0, 0.0, "1 standard deviations"), (0.0, -1.0, "1 standard deviations"),
                                 (0.0, float("nan"), "1 standard deviations")):
        m, s = mu.clone(), sg.clone()
        m[4321], s[4321] = bad_mu, bad_sg
        with pytest.raises(ValueError, match=what):
            ops.check_inputs(m, s)
    q = ChannelwisePriorCDFQuantizer(2, N, validate_inputs=True)
    q.build_code_points(priors.FactoredGaussianPrior(np.zeros(2), np.ones(2)))
    means = np.zeros((10, 2), np.float32)
    stds = np.ones((10, 2), np.float32)
    q.compress_batch_channel_latents(means, stds, [1.0])
    stds[3, 1] = 0.0
    with pytest.raises(ValueError, match="standard deviations"):
        q.compress_batch_channel_latents(means, stds, [1.0])
    q.validate_inputs = False                                          # the reference's behaviour: no check, no exception
    q.compress_batch_channel_latents(means, np.ones((10, 2), np.float32), [1.0])


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the per-image call as one C call, device-resident model tables, the facade's planes-out form

@pytest.mark.parametrize("rows,cols", [(64, 64), (777, 24), (130, 3), (1, 1), (4096, 256)])
def test_prep_planes_is_transpose_and_exact_sqrt(rows, cols):
    """vbq_prep_planes_f32: both layout changes in one launch; with spread='variance' the sigma plane is the IEEE square
    root of exp(logvar), with spread='logvar' sqrt(exp(.)) of the log-variances themselves -- the very numbers torch's
    `exp(logvars) ** 0.5` gives (quantizer.py:197,202)."""
    from vbq_amd import ops
    rng = np.random.default_rng(rows * 1000 + cols)
    mu = torch.from_numpy(rng.normal(0, 2, (rows, cols)).astype(np.float32)).cuda()
    lv = rng.normal(-4, 3, (rows, cols)).astype(np.float32)
    lv.reshape(-1)[:5] = [-200.0, 88.0, -87.5, 0.0, -103.0][: min(5, lv.size)]     # 0, near-overflow, denormal variances
    var = torch.exp(torch.from_numpy(lv).cuda())
    m_p, s_p = ops.prep_planes(mu, var, spread="variance")
    m_l, s_l = ops.prep_planes(mu, torch.from_numpy(lv).cuda(), spread="logvar")
    assert torch.equal(m_l, m_p) and torch.equal(s_l, s_p)
    with pytest.raises(ValueError):
        ops.prep_planes(mu, var, spread="stddev")
    assert torch.equal(m_p, mu.t().contiguous())
    assert torch.equal(s_p, (var ** 0.5).t().contiguous()) and torch.equal(s_p, torch.sqrt(var).t().contiguous())
    m2, s2 = ops.prep_planes(mu, var)
    assert torch.equal(m2, m_p) and torch.equal(s2, var.t().contiguous())


@pytest.mark.timeout(600)
def test_logvar_to_sigma_is_torchs_for_every_float32():
    """compress_latents hands the encoder's log-variances to the planes kernel, which takes sigma = sqrtf(expf(.)) itself
    (quantizer.py:197,202 `tf.exp(posterior_logvars) ** 0.5`).  That is only a drop-in for the torch ops it replaced if it is
    the SAME function: checked here for every one of the 2^32 float32 bit patterns (NaNs compare as NaNs)."""
    from vbq_amd import ops
    dev_ = torch.device("cuda")
    n = 1 << 26
    for chunk in range(64):
        bits = (torch.arange(chunk * n, (chunk + 1) * n, dtype=torch.int64, device=dev_) - ((1 << 32) if chunk >= 32 else 0)).to(torch.int32)
        x = bits.view(torch.float32).reshape(n // 256, 256)
        want = torch.exp(x) ** 0.5
        got = ops.prep_planes(x, x, spread="logvar")[1].t()
        differ = (got.view(torch.int32) != want.view(torch.int32)) & ~(torch.isnan(got) & torch.isnan(want))
        assert not bool(differ.any()), f"chunk {chunk}: {int(differ.sum())} values differ from torch.exp(x) ** 0.5"
        del bits, x, want, got, differ


@pytest.mark.parametrize("L,C,B,with_len", [(3, 70, 77, False), (2, 1, 1000, True), (5, 64, 32, True), (1, 130, 33, False),
                                             (3, 128, 136, False), (2, 68, 200, True), (16, 256, 1536, True),
                                             # large enough for the LDS-table form (k_lookup_lds): whole and ragged channel groups,
                                             # row counts that are not a multiple of its 256-row blocks
                                             (16, 64, 4096, True), (3, 20, 16388, False), (2, 36, 3076, True)])
def test_gather_latents_one_pass_against_numpy(L, C, B, with_len):
    """vbq_gather_latents_u16: Z_hat, raw_num_bits, num_bits and the indices themselves, channel-last, from index planes in
    one pass -- against NumPy fancy indexing; ragged tiles, foreign indices >= T clamped like vbq_gather_f32."""
    from vbq_amd import ops
    rng = np.random.default_rng(L * 100 + C)
    idx = rng.integers(0, T, (L, C, B)).astype(np.uint16)
    idx[0, 0, :3] = [0, T - 1, 65535 if B > 2 else T - 1][:3] if B >= 3 else idx[0, 0, :3]
    srt = np.sort(rng.normal(0, 1, (C, T)).astype(np.float32), axis=1)
    ll = rng.uniform(0, 20, (L, C, N + 1)).astype(np.float32) if with_len else None
    models = rng.uniform(0.5, 15, (L, C, T)).astype(np.float32)
    z, raw, nb, qi = ops.gather_latents(torch.from_numpy(idx).cuda(), N=N, table_sorted=torch.from_numpy(srt).cuda(),
                                        level_len=None if ll is None else torch.from_numpy(ll).cuda(),
                                        models=torch.from_numpy(models).cuda(), want_num_bits=True, want_idx=True)
    q = np.minimum(idx.astype(np.int64), T - 1)
    lev = O.levels_of_sorted_ranks(N)[q]
    ch = np.arange(C)[None, :, None]
    ls = np.arange(L)[:, None, None]
    assert np.array_equal(z.cpu().numpy(), srt[ch, q].transpose(0, 2, 1))
    want_raw = lev.astype(np.int32) if ll is None else ll[ls, ch, lev]
    assert raw.dtype == (torch.int32 if ll is None else torch.float32)
    assert np.array_equal(raw.cpu().numpy(), want_raw.transpose(0, 2, 1))
    assert np.array_equal(nb.cpu().numpy(), models[ls, ch, q].transpose(0, 2, 1))
    assert np.array_equal(qi.cpu().numpy(), q.astype(np.uint16).transpose(0, 2, 1))
    only_z = ops.gather_latents(torch.from_numpy(idx).cuda(), N=N, table_sorted=torch.from_numpy(srt).cuda(), want_raw_bits=False)
    assert only_z[1] is None and only_z[2] is None and only_z[3] is None and torch.equal(only_z[0], z)


def test_compress_latents_one_call_equals_oracle(golden):
    """vbq_compress_latents_f32 (planes + solve + fused lookups in one C call) against the oracle's compress_latents: raw
    lengths (int32 bits) and corrected lengths with entropy models, 1 / 3 / 16 / 32 lambdas (K1p, K1, K1e routes)."""
    from vbq_amd import ops
    g, q, orc = _case(golden)
    mu, sg = g["mu"], g["sigma"]
    B, C = mu.shape
    tab = torch.from_numpy(q.all_code_points).cuda()
    srt = torch.from_numpy(q.code_points_by_channel).cuda()
    var = torch.from_numpy(sg).cuda() ** 2                        # not bit-exactly sigma^2 -> sqrt: use sigma directly below
    lam_all = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 32)]
    for lam in (lam_all[5:6], lam_all[3:6], lam_all[::2], lam_all):
        wi, wz, wb = CO.quantize(mu, sg, q.all_code_points, lam, N=N, want_zhat=True, want_bits=True)
        z, raw, nb = ops.compress_latents(torch.from_numpy(mu).cuda(), torch.from_numpy(sg).cuda(), tab, srt, lam, N=N)
        assert nb is None and raw.dtype == torch.int32
        assert np.array_equal(z.cpu().numpy(), wz) and np.array_equal(raw.cpu().numpy(), wb.astype(np.int32))
    rng = np.random.default_rng(4)
    lam = lam_all[::4]
    ll = (np.arange(N + 1, dtype=np.float32) + np.abs(rng.normal(0, 1, (len(lam), C, N + 1)))).astype(np.float32)
    models = rng.uniform(0.5, 15, (len(lam), C, T)).astype(np.float32)
    wi, wz, wb = CO.quantize(mu, sg, q.all_code_points, lam, N=N, level_len=ll, want_zhat=True, want_bits=True)
    z, raw, nb = ops.compress_latents(torch.from_numpy(mu).cuda(), torch.from_numpy(sg).cuda(), tab, srt, lam, N=N,
                                      level_len=torch.from_numpy(ll).cuda(), models=torch.from_numpy(models).cuda())
    assert np.array_equal(z.cpu().numpy(), wz) and np.array_equal(raw.cpu().numpy(), wb)
    want_nb = models[np.arange(len(lam))[:, None, None], np.arange(C)[None, None, :], wi.astype(np.int64)]
    assert np.array_equal(nb.cpu().numpy(), want_nb)
    del var


def test_models_stay_on_the_device_until_read(golden):
    """build_entropy_models leaves entropy_models / raw_code_length_entropy_models / the histograms on the device behind
    dict-like views; compress_latents(return_np=False) uses them there (no host copy made); the first host read gives the
    oracle's tables bit for bit and int32 counts."""
    from vbq_amd.quantizer import DeviceModels
    g, q, orc = _case(golden)
    lambs = list(2.0 ** np.linspace(-8, 7.5, 32))
    lam32 = [np.float32(l) for l in lambs]
    B, C = g["mu"].shape
    q.build_entropy_models_from_latents(g["mu"], g["sigma"], lambs, 1)
    for d in (q.entropy_models, q.raw_code_length_entropy_models, q._code_counts):
        assert isinstance(d, DeviceModels) and d.on_device and list(d) == lambs
    assert q.lambs == sorted(lambs) and bool(q.raw_code_length_entropy_models)
    means = g["mu"].reshape(1, 8, B // 8, C)
    logvars = (2 * np.log(g["sigma"])).astype(np.float32).reshape(means.shape)
    dev_out = q.compress_latents(means, logvars, lambs[::5], return_np=False)
    sub = q.compress_latents(means, logvars, [lambs[9], lambs[2]], return_np=False)          # a re-ordered subset
    assert q.entropy_models.on_device and q.raw_code_length_entropy_models.on_device and q._code_counts.on_device
    stds = (torch.exp(torch.from_numpy(logvars).cuda()) ** 0.5).cpu().numpy().reshape(B, C)
    orc.build_entropy_models(g["mu"], g["sigma"], lam32, add_n_smoothing=1)
    ref = orc.compress_latents(g["mu"], stds, lam32)
    for lamb in lambs[::5] + [lambs[9], lambs[2]]:
        out = sub if lamb in (lambs[9], lambs[2]) and lamb not in lambs[::5] else dev_out
        l32 = np.float32(lamb)
        for key in ("Z_hat", "raw_num_bits", "num_bits"):
            assert np.array_equal(out[key][lamb].cpu().numpy().reshape(B, C), ref[key][l32]), (key, lamb)
    # first host read: one copy, the oracle's numbers
    for i, lamb in enumerate(lambs):
        assert np.array_equal(q.entropy_models[lamb], orc.entropy_models[lam32[i]])
        assert np.array_equal(q.raw_code_length_entropy_models[lamb], orc.raw_models[lam32[i]])
    assert not q.entropy_models.on_device and q._code_counts.on_device
    cnt = np.stack([q._code_counts[l] for l in lambs])
    assert cnt.dtype == np.int32 and int(cnt.sum()) == B * C * len(lambs)
    with pytest.raises(KeyError):
        q.compress_latents(means, logvars, [123.0])
    # a second build gives NEW tables; the dicts of the first one keep theirs
    old = q.entropy_models
    q.build_entropy_models_from_latents(g["mu"][: B // 2], g["sigma"][: B // 2], lambs[:4], 1)
    assert list(q.entropy_models) == lambs[:4] and q.entropy_models.on_device
    assert np.array_equal(old[lambs[0]], orc.entropy_models[lam32[0]])
    assert not np.array_equal(q.entropy_models[lambs[0]], old[lambs[0]])


def test_quantize_facade_planes_out_and_in_place_table_edit():
    """out_layout='planes' hands back what the kernels write ([L, C, rows]) whatever the input layout; return_indices=False
    with return_values gives Z_hat alone; a NumPy table edited IN PLACE is a new table (content-keyed cache)."""
    import vbq_amd
    rng = np.random.default_rng(33)
    B, C = 515, 20
    s_c = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    mu = (rng.standard_normal((B, C)) * s_c).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, (B, C))).astype(np.float32)
    tab = vbq_amd.gaussian_table(s_c, N)
    lam = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 32)]
    wi, wz, wb = CO.quantize(mu, sg, tab, lam, N=N, want_zhat=True, want_bits=True)
    mu_d, sg_d = torch.from_numpy(mu).cuda(), torch.from_numpy(sg).cuda()
    ip = vbq_amd.quantize(mu_d, sg_d, lam, table=tab, out_layout="planes")
    assert tuple(ip.shape) == (32, C, B) and np.array_equal(ip.cpu().numpy().transpose(0, 2, 1), wi)
    ip2, zp2, bp2 = vbq_amd.quantize(mu_d, sg_d, lam[:3], table=tab, out_layout="planes", return_values=True, return_bits=True)
    assert np.array_equal(ip2.cpu().numpy().transpose(0, 2, 1), wi[:3]) and np.array_equal(zp2.cpu().numpy().transpose(0, 2, 1), wz[:3])
    assert np.array_equal(bp2.cpu().numpy().transpose(0, 2, 1), wb[:3])
    z_only = vbq_amd.quantize(mu_d, sg_d, lam, table=tab, return_values=True, return_indices=False)
    assert tuple(z_only.shape) == (32, B, C) and np.array_equal(z_only.cpu().numpy(), wz)
    i3, z3, b3 = vbq_amd.quantize(mu, sg, lam[4:9], table=tab, return_values=True, return_bits=True)
    assert np.array_equal(i3, wi[4:9]) and np.array_equal(z3, wz[4:9]) and np.array_equal(b3, wb[4:9]) and b3.dtype == np.float32
    ll = (np.arange(N + 1, dtype=np.float32) + np.abs(rng.normal(0, 1, (3, C, N + 1)))).astype(np.float32)
    wi4, wz4, wb4 = CO.quantize(mu, sg, tab, lam[10:13], N=N, level_len=ll, want_zhat=True, want_bits=True)
    i4, z4, b4 = vbq_amd.quantize(mu_d, sg_d, lam[10:13], table=tab, lengths=ll, return_values=True, return_bits=True)
    assert np.array_equal(i4.cpu().numpy(), wi4) and np.array_equal(z4.cpu().numpy(), wz4) and np.array_equal(b4.cpu().numpy(), wb4)
    icb, zcb = vbq_amd.quantize(mu_d.t().contiguous(), sg_d.t().contiguous(), lam[:3], table=tab, layout="cb", return_values=True)
    assert np.array_equal(icb.cpu().numpy().transpose(0, 2, 1), wi[:3]) and np.array_equal(zcb.cpu().numpy().transpose(0, 2, 1), wz[:3])
    one, zone = vbq_amd.quantize(mu[:, 0], sg[:, 0], lam[6], table=tab[:1], return_values=True)
    assert np.array_equal(one, wi[6][:, 0]) and np.array_equal(zone, wz[6][:, 0])
    # the SAME ndarray edited in place: fresh values (another valid table) or a ValueError (a broken one), never a stale copy
    tab2 = tab.copy()
    first = vbq_amd.quantize(mu, sg, 1.0, table=tab2)
    tab2[:] = vbq_amd.gaussian_table(1.7 * s_c, N)
    second = vbq_amd.quantize(mu, sg, 1.0, table=tab2)
    assert np.array_equal(second, CO.quantize(mu, sg, tab2, [1.0], N=N)[0]) and not np.array_equal(first, second)
    tab2[3, 5] = 1e9
    with pytest.raises(ValueError):
        vbq_amd.quantize(mu, sg, 1.0, table=tab2)
    # ... the same guarantee without xxhash (np.array_equal against the kept copy; never a cryptographic hash of 2 MB per call)
    from vbq_amd import api
    import hashlib
    real_xxh3, real_blake = api._xxh3, hashlib.blake2b
    try:
        api._xxh3 = lambda: None
        hashlib.blake2b = None                                 # any use would raise TypeError
        api._CHECKED.clear()
        tab3 = tab.copy()
        a = vbq_amd.quantize(mu, sg, 1.0, table=tab3)
        assert np.array_equal(vbq_amd.quantize(mu, sg, 1.0, table=tab3), a) and len(api._CHECKED) == 1
        tab3[:] = vbq_amd.gaussian_table(0.6 * s_c, N)
        b = vbq_amd.quantize(mu, sg, 1.0, table=tab3)
        assert np.array_equal(b, CO.quantize(mu, sg, tab3, [1.0], N=N)[0]) and not np.array_equal(a, b)
    finally:
        api._xxh3, hashlib.blake2b = real_xxh3, real_blake
    # a READ-ONLY array that is the very object prepared before cannot have changed: no pass over its content on later calls ...
    api._CHECKED.clear()
    tab4 = tab.copy()
    tab4.setflags(write=False)
    c0 = vbq_amd.quantize(mu, sg, lam[3], table=tab4)
    real_same = api._NumpyEntry.same_content
    try:
        api._NumpyEntry.same_content = lambda self, arr: (_ for _ in ()).throw(AssertionError("content check on an immutable table"))
        assert np.array_equal(vbq_amd.quantize(mu, sg, lam[3], table=tab4, validate=False), c0)
        assert np.array_equal(vbq_amd.quantize(mu, sg, lam[3], table=tab4), wi[3])
    finally:
        api._NumpyEntry.same_content = real_same
    # ... but once it is writeable again it is compared again (and a read-only VIEW of a writeable array never took that path)
    tab4.setflags(write=True)
    tab4[:] = vbq_amd.gaussian_table(1.3 * s_c, N)
    assert np.array_equal(vbq_amd.quantize(mu, sg, lam[3], table=tab4), CO.quantize(mu, sg, tab4, [lam[3]], N=N)[0])
    base = tab.copy()
    view = base[:]
    view.setflags(write=False)
    v0 = vbq_amd.quantize(mu, sg, lam[3], table=view)
    base[:] = vbq_amd.gaussian_table(2.1 * s_c, N)
    v1 = vbq_amd.quantize(mu, sg, lam[3], table=view)
    assert np.array_equal(v0, wi[3]) and np.array_equal(v1, CO.quantize(mu, sg, base, [lam[3]], N=N)[0]) and not np.array_equal(v0, v1)
    # validate=False is cached as well, and a later validate=True still checks
    api._CHECKED.clear()
    bad = tab.copy()
    bad[3, 5] = 1e9
    vbq_amd.quantize(mu, sg, 1.0, table=bad, validate=False)
    assert len(api._CHECKED) == 1
    vbq_amd.quantize(mu, sg, 1.0, table=bad, validate=False)
    assert len(api._CHECKED) == 1
    with pytest.raises(ValueError):
        vbq_amd.quantize(mu, sg, 1.0, table=bad)


@pytest.mark.timeout(600)
def test_channel_last_above_the_old_transpose_limit():
    """5e6 rows of channel-last latents (the transposes used to put rows on grid.y: 4.19e6 was the limit) through the facade
    and through vbq_transpose_f32 / vbq_transpose_planes."""
    import vbq_amd
    from vbq_amd import ops
    rows, C = 5_000_000, 2
    rng = np.random.default_rng(1)
    mu = torch.from_numpy(rng.normal(0, 1.2, (rows, C)).astype(np.float32)).cuda()
    sg = torch.from_numpy(np.exp(rng.normal(-2, 0.7, (rows, C))).astype(np.float32)).cuda()
    assert torch.equal(ops.transpose(mu), mu.t().contiguous())
    tab = vbq_amd.gaussian_table([1.2, 1.3], N)
    idx = vbq_amd.quantize(mu, sg, [0.5, 4.0], table=tab)
    assert tuple(idx.shape) == (2, rows, C)
    for s in (0, 4_194_304 - 500, rows - 1000):
        want = CO.quantize(mu[s:s + 1000].cpu().numpy(), sg[s:s + 1000].cpu().numpy(), tab, [0.5, 4.0], N=N)
        assert np.array_equal(idx[:, s:s + 1000].cpu().numpy(), want)
    planes = vbq_amd.quantize(mu, sg, [0.5, 4.0], table=tab, out_layout="planes")
    assert torch.equal(ops.transpose_planes(planes), idx)


def test_compress_latents_large_batch_through_the_class():
    """A batch large enough for the LDS-table lookups (k_lookup_lds: Z_hat and num_bits; raw_num_bits from the tile kernel) through
    ChannelwisePriorCDFQuantizer: build the entropy models on 4 096 x 8 latents, compress the same batch as one [4, 32, 32, 8]
    tensor, compare all three outputs with the oracle's compress_latents."""
    from vbq_amd import ChannelwisePriorCDFQuantizer, priors
    rng = np.random.default_rng(77)
    C, B = 8, 4096
    ch_std = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    q = ChannelwisePriorCDFQuantizer(C, N)
    q.build_code_points(priors.FactoredGaussianPrior(np.zeros(C), ch_std))
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(np.zeros(C), ch_std))
    mu = (rng.standard_normal((B, C)) * ch_std).astype(np.float32)
    logvars = rng.normal(-4, 1.0, (B, C)).astype(np.float32)
    stds = (torch.exp(torch.from_numpy(logvars).cuda()) ** 0.5).cpu().numpy()
    lambs = list(2.0 ** np.linspace(-8, 7, 16))                     # post_process.py:115
    lam32 = [np.float32(l) for l in lambs]
    q.build_entropy_models_from_latents(mu, stds, lambs, 1)
    orc.build_entropy_models(mu, stds, lam32, add_n_smoothing=1)
    out = q.compress_latents(mu.reshape(4, 32, 32, C), logvars.reshape(4, 32, 32, C), lambs, return_np=False)
    assert q.entropy_models.on_device
    ref = orc.compress_latents(mu, stds, lam32)
    for lamb, l32 in zip(lambs, lam32):
        for key in ("Z_hat", "raw_num_bits", "num_bits"):
            got = out[key][lamb]
            assert tuple(got.shape) == (4, 32, 32, C)
            assert np.array_equal(got.cpu().numpy().reshape(B, C), ref[key][l32]), (key, lamb)
    assert np.array_equal(q.entropy_models[lambs[3]], orc.entropy_models[lam32[3]])


class TorchFakeVAE:
    """A VAE that lives on the device: encode() returns CUDA tensors, decode() insists on getting one."""

    def __init__(self, means, logvars):
        self.means, self.logvars = torch.from_numpy(means).cuda(), torch.from_numpy(logvars).cuda()
        self.decoded_from = None

    def encode(self, X):
        return self.means, self.logvars

    def decode(self, Z):
        assert isinstance(Z, torch.Tensor) and Z.is_cuda, "compress() must hand the decoder the device tensor"
        self.decoded_from = Z
        return (0.1 * Z.mean(dim=-1, keepdim=True) + 0.5).repeat_interleave(3, dim=-1)


def test_per_image_results_stay_on_the_device_until_read(golden):
    """compress_latents / compress (quantizer.py:190-256) with return_np=True hand out ndarray-like lazy views: nothing crosses
    PCIe until a value is read on the host; compress() gives a torch VAE's decoder the device Z_hat (no D2H + H2D round trip);
    the first host read of a quantity copies that quantity (and only starts the DMA of its siblings); values, dict layout and
    pickling are those of the eager NumPy form; the evaluation loop's sums (utils.py:547-552) are taken on the device in NumPy's
    float32 order."""
    from vbq_amd import utils
    from vbq_amd.lazy import LazyArray
    g, q, orc = _case(golden)
    lambs = list(2.0 ** np.linspace(-8, 7, 16))
    lam32 = [np.float32(l) for l in lambs]
    B, C = g["mu"].shape
    q.build_entropy_models_from_latents(g["mu"], g["sigma"], lambs, 1)
    means = g["mu"].reshape(1, 8, B // 8, C)
    logvars = (2 * np.log(g["sigma"])).astype(np.float32).reshape(means.shape)
    stds = (torch.exp(torch.from_numpy(logvars).cuda()) ** 0.5).cpu().numpy().reshape(B, C)
    orc.build_entropy_models(g["mu"], g["sigma"], lam32, add_n_smoothing=1)
    ref = orc.compress_latents(g["mu"], stds, lam32)
    vae = TorchFakeVAE(means, logvars)
    X = np.zeros((1, 8, B // 8, 3), np.float32)
    stager = q._stager()
    out = q.compress(X, vae, lambs, clip=True)
    assert set(out) == {"Z_hat", "raw_num_bits", "num_bits_cl", "num_bits", "X_hat"} and list(out["Z_hat"]) == lambs
    assert stager.transfers == 0 and q.entropy_models.on_device                     # nothing has crossed PCIe
    z0 = out["Z_hat"][lambs[0]]
    assert isinstance(z0, LazyArray) and z0.on_device and z0.shape == means.shape and z0.dtype == np.float32
    assert vae.decoded_from.shape == (16,) + means.shape[1:] and vae.decoded_from.data_ptr() == z0.tensor.data_ptr()
    assert out["X_hat"][lambs[3]].shape == X.shape and out["X_hat"][lambs[3]].on_device
    # zero-copy hand-over to device libraries (no transfer): the CUDA array interface and DLPack of the tensor behind the view
    zt = torch.as_tensor(z0, device="cuda")
    zd = torch.from_dlpack(out["num_bits"][lambs[1]])
    assert zt.data_ptr() == z0.tensor.data_ptr() and tuple(zt.shape) == z0.shape and zd.data_ptr() == out["num_bits"][lambs[1]].tensor.data_ptr()
    assert stager.transfers == 0 and z0.on_device
    # the evaluation loop's reads: sums on the device == np.sum of the host arrays, bit for bit
    sums = utils._sums_per_setting(out["num_bits"], lambs)
    sums_cl = utils._sums_per_setting(out["num_bits_cl"], lambs)
    u8 = utils._reconstructions_u8(out["X_hat"], lambs)
    all3 = utils.evaluation_reads(out, lambs, {})                                    # the three of them behind ONE synchronisation
    assert np.array_equal(all3[0], sums) and np.array_equal(all3[1], sums_cl) and np.array_equal(all3[2], u8)
    assert stager.transfers == 0 and sums.dtype == np.float32 and u8.dtype == np.uint8 and u8.shape == (16,) + X.shape[1:]
    # first host read of ONE quantity: that one (its 16 lambdas at once), nothing else
    nb = np.asarray(out["num_bits"][lambs[5]])
    assert stager.transfers == 1 and not out["num_bits"][lambs[0]].on_device and out["Z_hat"][lambs[0]].on_device
    assert out["X_hat"][lambs[0]].on_device
    for i, (lamb, l32) in enumerate(zip(lambs, lam32)):
        assert np.array_equal(out["num_bits"][lamb].reshape(B, C), ref["num_bits"][l32])
        assert np.array_equal(out["Z_hat"][lamb].reshape(B, C), ref["Z_hat"][l32])
        assert np.array_equal(out["raw_num_bits"][lamb].reshape(B, C), ref["raw_num_bits"][l32])
        assert out["num_bits_cl"][lamb] is out["raw_num_bits"][lamb]
        assert sums[i] == np.sum(np.asarray(out["num_bits"][lamb])[0]) and sums_cl[i] == np.sum(np.asarray(out["num_bits_cl"][lamb])[0])
        xh = np.asarray(out["X_hat"][lamb])
        assert xh.min() >= 0 and xh.max() <= 1 and np.array_equal(u8[i], np.clip(np.round(xh[0] * 255), 0, 255).astype(np.uint8))
        want_x = np.clip((0.1 * ref["Z_hat"][l32].reshape(means.shape).mean(axis=-1, keepdims=True) + 0.5).repeat(3, axis=-1), 0, 1)
        np.testing.assert_allclose(xh, want_x, rtol=1e-6)
    assert stager.transfers == 4 and np.array_equal(nb, np.asarray(out["num_bits"][lambs[5]]))      # no second copy of anything
    # a kept result survives later calls (the staging blocks are reused); pickles as plain arrays
    keep = np.asarray(out["Z_hat"][lambs[2]]).copy()
    out2 = q.compress_latents(means, logvars, lambs[:3])
    assert np.array_equal(out2["Z_hat"][lambs[2]], keep) and np.array_equal(out["Z_hat"][lambs[2]], keep)
    p = pickle.loads(pickle.dumps(out2))
    assert type(p["num_bits"][lambs[1]]) is np.ndarray and np.array_equal(p["num_bits"][lambs[1]], out["num_bits"][lambs[1]])
    # a loop that reads the same quantities call after call gets their transfers started together (what the last call's results
    # were read for is prefetched on the first read of the next one's): out2 was read for all three (pickled)
    n0 = stager.transfers
    out3 = q.compress_latents(means, logvars, lambs[4:6])
    _ = np.asarray(out3["Z_hat"][lambs[4]])                                             # starts the DMA of out3's num_bits too
    assert stager.transfers == n0 + 3 and out3["num_bits"][lambs[4]].on_device and out3["raw_num_bits"][lambs[4]].on_device
    # a sibling whose staging block was overwritten by a later call in the meantime is simply transferred again
    out4 = q.compress_latents(means, logvars, lambs[8:9])
    _ = np.asarray(out4["num_bits"][lambs[8]])                                          # ... whose block out4 now takes
    assert np.array_equal(out3["num_bits"][lambs[5]], out["num_bits"][lambs[5]])
    assert np.array_equal(out3["raw_num_bits"][lambs[4]], out["raw_num_bits"][lambs[4]])
    # NumPy VAE: the reference's own form still works (Z_hat goes through the host), same numbers
    out_np = q.compress(X, FakeVAE(means, logvars), lambs[:2], clip=True)
    assert isinstance(out_np["X_hat"][lambs[0]], np.ndarray)
    np.testing.assert_allclose(out_np["X_hat"][lambs[1]], np.asarray(out["X_hat"][lambs[1]]), rtol=1e-6)


def test_lazy_results_in_the_reference_style_loop(golden):
    """The evaluation loop as the reference writes it -- `tmp = q.compress(...); read(tmp)`, image after image, the old results
    dropped by the rebinding (utils.py:542-554) -- on the lazy views:
      * the host array of a dropped result is handed back and REFILLED by the next call (HostStager.spare; round 5 kept a Future
        alive in `busy` that held the array, so its reference count never showed it free);
      * `busy` holds nothing once a result has been read;
      * 'X_hat' belongs to its call's group: a loop that reads 'X_hat' and 'num_bits' gets num_bits' transfer started on the
        first read of the next call's X_hat (side by side), not one after the other;
      * a write through the view reaches the device tensor."""
    import gc
    import vbq_amd.lazy as LZ
    g, q, orc = _case(golden)
    lambs = list(2.0 ** np.linspace(-8, 7, 16))
    B, C = g["mu"].shape
    q.build_entropy_models_from_latents(g["mu"], g["sigma"], lambs, 1)
    means = g["mu"].reshape(1, 8, B // 8, C)
    logvars = (2 * np.log(g["sigma"])).astype(np.float32).reshape(means.shape)
    old = LZ._THREADED_FROM
    LZ._THREADED_FROM = 1 << 10                                   # the golden case is small: take the pool path anyway
    try:
        stager = q._stager()
        first = None
        ids = []
        for it in range(4):
            tmp = q.compress_latents(means, logvars, lambs)       # rebinding drops the previous call's results
            gc.collect()
            nb = np.asarray(tmp["num_bits"][lambs[3]])
            ids.append(nb.base.ctypes.data if nb.base is not None else nb.ctypes.data)
            assert "num_bits" not in stager.busy                   # the copy-out is over, nothing pins the array
            if first is None:
                first = nb.copy()
            assert np.array_equal(nb, first)
            del nb
        assert len(set(ids[1:])) <= 2 and ids[2] in (ids[0], ids[1]), ids    # the arrays of dropped results are refilled
        # X_hat is part of its call: reading X_hat and num_bits call after call prefetches the sibling
        vae = TorchFakeVAE(means, logvars)
        X = np.zeros((1, 8, B // 8, 3), np.float32)
        out = q.compress(X, vae, lambs)
        zs, xs = out["Z_hat"][lambs[0]]._stack, out["X_hat"][lambs[0]]._stack
        assert xs.group == zs.group and any(r() is xs for r in zs.siblings)
        _ = np.asarray(out["X_hat"][lambs[0]]); _ = np.asarray(out["num_bits"][lambs[0]])
        out = q.compress(X, vae, lambs)
        n0 = stager.transfers
        _ = np.asarray(out["X_hat"][lambs[1]])                     # first read of the next call: num_bits' DMA starts with it
        assert stager.transfers == n0 + 2 and out["num_bits"][lambs[0]]._stack._future is not None
        # writes reach the device
        z = out["Z_hat"][lambs[2]]
        z[0, 0, 0, :4] = 123.0
        assert torch.all(z.tensor[0, 0, 0, :4] == 123.0) and np.all(np.asarray(z)[0, 0, 0, :4] == 123.0)
        with pytest.raises(ValueError):
            np.asarray(z)[0, 0, 0, 0] = 1.0
    finally:
        LZ._THREADED_FROM = old


class InputDependentVAE:
    """A torch VAE whose latents depend on the image: encode(X) works on a NumPy image (the reference's call) and on a device
    tensor (the captured form); decode insists on a device tensor."""

    def __init__(self, means, logvars):
        self.means, self.logvars = torch.from_numpy(means).cuda(), torch.from_numpy(logvars).cuda()
        self.np_calls = self.dev_calls = 0

    def encode(self, X):
        if isinstance(X, torch.Tensor):
            self.dev_calls += 1
            shift = X.mean()
        else:
            self.np_calls += 1
            shift = torch.from_numpy(np.asarray(X, np.float32)).cuda().mean()
        return self.means + shift, self.logvars

    def decode(self, Z):
        assert isinstance(Z, torch.Tensor) and Z.is_cuda
        return (0.1 * Z.mean(dim=-1, keepdim=True) + 0.5).repeat_interleave(3, dim=-1)


class HostOnlyEncoderVAE(InputDependentVAE):
    def encode(self, X):
        if isinstance(X, torch.Tensor):
            raise TypeError("this encoder wants the NumPy image")
        return super().encode(X)


class SyncingDecoderVAE(InputDependentVAE):
    def decode(self, Z):
        float(Z.sum().item())                                      # a synchronisation: legal eagerly, fatal inside a capture
        return super().decode(Z)


def test_compress_replay_equals_compress_image_after_image(golden):
    """quantizer.compress_replay (vbq_amd.replay): compress + the evaluation loop's reads (utils.py:542-556) captured once per
    image shape into a HIP graph and replayed per image.  Every image: the same dict layout, the same values and the same
    (sums, sums_cl, uint8 X_hat) as `compress` + `utils.evaluation_reads` on that image; the encoder inside the graph where it
    takes a device tensor, outside where it does not, no graph at all for a NumPy VAE; a rebuilt model invalidates the capture."""
    from vbq_amd import utils
    g, q, orc = _case(golden)
    lambs = list(2.0 ** np.linspace(-8, 7, 16))
    B, C = g["mu"].shape
    q.build_entropy_models_from_latents(g["mu"], g["sigma"], lambs, 1)
    means = g["mu"].reshape(1, 8, B // 8, C)
    logvars = (2 * np.log(g["sigma"])).astype(np.float32).reshape(means.shape)
    rng = np.random.default_rng(3)
    images = [rng.uniform(0, 1, (1, 8, B // 8, 3)).astype(np.float32) for _ in range(4)]
    for cls, want_mode in ((InputDependentVAE, "full"), (HostOnlyEncoderVAE, "latents"), (SyncingDecoderVAE, "eager"), (FakeVAE, "eager")):
        vae = cls(means, logvars)
        for i, X in enumerate(images + images[:1]):
            out, (sums, sums_cl, u8) = q.compress_replay(X, vae, lambs, clip=True)
            ref = q.compress(X, vae, lambs, clip=True)
            r_sums, r_cl, r_u8 = utils.evaluation_reads(ref, lambs, {})
            assert set(out) == set(ref) and list(out["Z_hat"]) == lambs
            assert np.array_equal(sums, r_sums) and np.array_equal(sums_cl, r_cl) and np.array_equal(u8, r_u8)
            for k in ("Z_hat", "raw_num_bits", "num_bits", "num_bits_cl", "X_hat"):
                for lamb in (lambs[0], lambs[7], lambs[-1]):
                    assert np.array_equal(np.asarray(out[k][lamb]), np.asarray(ref[k][lamb])), (cls.__name__, i, k)
        rp = next(r for r in q._dev_cache["_replays"].values() if r.vae is vae)
        assert rp.mode == want_mode and (rp.replays == len(images) + 1) == (want_mode != "eager")
        if want_mode == "full":
            assert vae.dev_calls >= 1
    # a second shape gets its own graph; the first one keeps working
    vae = InputDependentVAE(means, logvars)
    X2 = np.zeros((1, 4, B // 4, 3), np.float32)
    vae2 = InputDependentVAE(means.reshape(1, 4, B // 4, C), logvars.reshape(1, 4, B // 4, C))
    a, _ = q.compress_replay(images[0], vae, lambs)
    b, _ = q.compress_replay(X2, vae2, lambs)
    assert np.asarray(a["Z_hat"][lambs[0]]).shape == means.shape and np.asarray(b["Z_hat"][lambs[0]]).shape == (1, 4, B // 4, C)
    # new models (a rebuild): the captured graph read the old tables -- it must be captured again, and give the new numbers
    q.build_entropy_models_from_latents(g["mu"][::-1].copy(), g["sigma"], lambs, 1)
    out, (sums, _, _) = q.compress_replay(images[1], vae, lambs)
    ref = q.compress(images[1], vae, lambs)
    assert np.array_equal(sums, utils.evaluation_reads(ref, lambs, {})[0])
    assert np.array_equal(np.asarray(out["num_bits"][lambs[5]]), np.asarray(ref["num_bits"][lambs[5]]))


@pytest.mark.parametrize("rows,n", [(1, 0), (3, 1), (2, 7), (16, 8191), (16, 8192), (5, 8193), (16, 12288), (3, 100_003), (16, 393_216), (2, 3_000_001)])
def test_numpy_row_sums(rows, n):
    """vbq_numpy_row_sums_f32 == np.sum(x[r]) bit for bit (float32, NumPy's blocks of 8192 / pairwise order), rows that start on
    16 bytes or not."""
    from vbq_amd import ops
    rng = np.random.default_rng(rows * 1000 + n)
    x = (rng.gamma(2.0, 3.0, (rows, n)) + rng.normal(0, 1e-3, (rows, n))).astype(np.float32)
    got = ops.numpy_row_sums(torch.from_numpy(x).cuda()).cpu().numpy()
    want = np.array([np.sum(x[r]) for r in range(rows)], dtype=np.float32)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    if n >= 16:                                                     # a 3-D "latent" view of the same rows sums the same way
        x3 = x[:, : (n // 8) * 8].copy().reshape(rows, 2, -1, 4)
        got3 = ops.numpy_row_sums(torch.from_numpy(x3).cuda()).cpu().numpy()
        assert np.array_equal(got3, np.array([np.sum(x3[r]) for r in range(rows)], dtype=np.float32))
