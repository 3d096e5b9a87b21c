"""f4: downstream evaluators of the word-embedding notebook on the GPU (fused MFMA rank GEMM)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")


def test_prediction_ranks_golden_and_checker(golden):
    _need_gpu()
    from oracle import c_oracle as CO
    from oracle import vbq_oracle as o
    from vbq_amd import embeddings as E
    g = golden("g10_analogy.npz")
    emb, an = g["emb"], g["analogies"]
    for e, want in ((emb, g["ranks"]), (g["quantized_7"].astype(np.float32), g["ranks_q7"])):
        got = E.prediction_ranks(e, an)
        assert got.dtype == np.int64 and got.shape == want.shape
        # bit for bit the documented arithmetic (fma chain over ascending k) ...
        assert np.array_equal(got, CO.analogy_ranks(e, an, threads=4))
        # ... and the notebook's NumPy/BLAS result up to near-ties with the ground-truth score
        _, near = o.prediction_ranks(e, an)
        assert np.all(np.abs(got - want) <= near) and np.mean(got == want) > 0.97
    q7 = E.quantize_coordinates(emb, 7)
    assert q7.dtype == np.float32 and np.array_equal(q7, g["quantized_7"])
    assert np.array_equal(E.quantize_coordinates(emb, 1023), g["quantized_1023"])


@pytest.mark.parametrize("V,K,Q", [(1, 3, 2), (127, 16, 1), (129, 17, 130), (1000, 300, 257), (5000, 33, 64)])
def test_prediction_ranks_shapes(V, K, Q):
    _need_gpu()
    from oracle import c_oracle as CO
    from vbq_amd import embeddings as E
    rng = np.random.default_rng(V * 7 + K)
    emb = rng.normal(0, 1, (V, K)).astype(np.float32)
    emb[rng.integers(0, V)] = 0.0                                 # a zero vector: 0 / 1e-8
    an = rng.integers(0, V, (Q, 4)).astype(np.int32)
    if V > 4:
        emb[min(3, V - 1)] = emb[min(2, V - 1)]                   # duplicate words: exact score ties
    got = E.prediction_ranks(emb, an)
    assert np.array_equal(got, CO.analogy_ranks(emb, an, threads=4))
    assert got.min() >= 0 and got.max() <= V - 1
    t = E.prediction_ranks(torch.from_numpy(emb).cuda(), an)      # tensor in -> tensor out
    assert isinstance(t, torch.Tensor) and np.array_equal(t.cpu().numpy(), got)
    assert E.prediction_ranks(emb, np.zeros((0, 4), np.int32)).shape == (0,)
    with pytest.raises(IndexError):
        E.prediction_ranks(emb, np.array([[0, 0, 0, V]], np.int32))


def test_full_size_properties_and_test_beta():
    """Notebook-sized evaluation (19.5k questions x 100k words x 100 dims) through properties: the rank of a
    question whose target is exactly its prediction is 0; ranks are invariant to a positive rescaling of the
    embedding by powers of two; test_beta returns the notebook's 4-tuple."""
    _need_gpu()
    from vbq_amd import embeddings as E
    rng = np.random.default_rng(5)
    V, K, Q = 100_000, 100, 19_544
    emb = rng.normal(0, 1, (V, K)).astype(np.float32)
    an = rng.integers(0, V, (Q, 4)).astype(np.int32)
    r = E.prediction_ranks(emb, an)
    assert r.shape == (Q,) and r.min() >= 0 and r.max() < V
    r2 = E.prediction_ranks(emb * np.float32(4.0), an)
    assert np.mean(r2 == r) > 0.999                               # 1e-8 in the denominator is the only difference
    mrr, acc, h10 = E.analogy_metrics(r)
    assert 0 < mrr < 1 and 0 <= acc <= h10 <= 1
    stds = np.full_like(emb[:2000], 0.05)
    cp, _ = E.make_code_book(E.empirical_std(emb[:2000]))
    out = E.test_beta(emb[:2000], stds, 1.0, cp, analogies_id=an[:200] % 2000)
    assert len(out) == 4 and out[3] > 0


def test_image_metrics_golden(golden):
    """mse / psnr bit-exact, ms_ssim to 1e-9 against the reference module's outputs (g11) and the oracle."""
    _need_gpu()
    from oracle import vbq_oracle as o
    from vbq_amd import metrics as M
    g = golden("g11_image_metrics.npz")
    for k in "abc":
        x, y = g[f"{k}_x"], g[f"{k}_y"]
        assert np.array_equal(M.mse(x, y), g[f"{k}_mse"])
        assert np.array_equal(M.psnr(x, y, max_val=255), g[f"{k}_psnr"])
        got = M.ms_ssim(x, y, max_val=255)
        np.testing.assert_allclose(got, g[f"{k}_msssim"], rtol=1e-9, atol=0)
        np.testing.assert_allclose(got, o.ms_ssim(x, y), rtol=1e-12, atol=0)
    assert np.array_equal(M.ms_ssim(g["a_x"], g["a_x"]), np.ones(3))            # identical images
    with pytest.raises(RuntimeError):
        M.mse(g["a_x"], g["b_x"])
    with pytest.raises(RuntimeError):
        M.ms_ssim(g["a_x"][0], g["a_x"][0])


def test_image_metrics_kodak_size():
    """One Kodak-sized image against 8 reconstructions (the M axis of utils.evaluate_compression_quantizer):
    scales and the 2x2 decimation against the oracle at full size, monotone in the noise level."""
    _need_gpu()
    from oracle import vbq_oracle as o
    from vbq_amd import metrics as M
    rng = np.random.default_rng(3)
    H, W = 512, 768
    yy, xx = np.mgrid[0:H, 0:W]
    x = np.clip(128 + 80 * np.sin(yy / 17.0)[..., None] * np.cos(xx / 11.0)[..., None] + rng.normal(0, 10, (H, W, 3)), 0, 255)
    x = x.astype(np.uint8)
    xs = np.repeat(x[None], 8, axis=0)
    ys = np.clip(xs + rng.normal(0, 1, xs.shape) * (1 + 4 * np.arange(8))[:, None, None, None], 0, 255).astype(np.uint8)
    got = M.ms_ssim(xs, ys)
    assert np.all(np.diff(got) < 0) and got[0] > 0.99
    np.testing.assert_allclose(got[[0, 7]], o.ms_ssim(xs[[0, 7]], ys[[0, 7]]), rtol=1e-12)
    assert np.array_equal(M.mse(xs, ys), o.image_mse(xs, ys))
    assert np.allclose(M.convert_to_db(got), -10 * np.log10(1 - got))


def test_evaluate_compression_quantizer(tmp_path):
    """utils.evaluate_compression_quantizer end to end on two PNG files with a stand-in VAE: result keys, shapes,
    bit accounting equal to the compress() dict, metrics equal to the oracle on the same reconstructions."""
    _need_gpu()
    from PIL import Image
    from oracle import vbq_oracle as o
    from vbq_amd import ChannelwisePriorCDFQuantizer, priors, utils
    rng = np.random.default_rng(8)
    C, H, W = 4, 48, 64

    class VAE:
        def encode(self, X):
            r = np.random.default_rng(int(X.sum() * 1000) % 2 ** 31)
            return (r.normal(0, 1, (1, H // 4, W // 4, C)).astype(np.float32),
                    r.normal(-4, 0.5, (1, H // 4, W // 4, C)).astype(np.float32))

        def decode(self, Z):
            Z = np.asarray(Z)
            up = np.repeat(np.repeat(Z[..., :3], 4, axis=1), 4, axis=2)
            return 0.5 + 0.2 * up

    files = []
    for i in range(2):
        arr = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        p = tmp_path / f"img{i}.png"
        Image.fromarray(arr).save(p)
        files.append(str(p))
    q = ChannelwisePriorCDFQuantizer(C, 10)
    q.build_code_points(priors.FactoredGaussianPrior(np.zeros(C), np.ones(C)))
    lambs = [0.01, 1.0, 30.0]
    vae = VAE()
    q.build_entropy_models(np.asarray(Image.open(files[0]).convert("RGB"))[None] / 255., vae, lambs, add_n_smoothing=1)
    res = utils.evaluate_compression_quantizer(q, vae, files, lambs, return_reconstructions=True)
    for key in ("B", "BPP", "BPPCL", "BPL", "MSE (RGB)", "PSNR (Luma)", "MS-SSIM (Chroma)", "MS-SSIM (RGB) (dB)"):
        assert res[key].shape == (2, 3), key
    assert len(res["reconstructions"]) == 2 and res["reconstructions"][0].shape == (3, H, W, 3)
    assert np.all(np.diff(res["B"], axis=1) <= 0)                       # rate falls as lambda grows
    x = np.asarray(Image.open(files[1]).convert("RGB"))
    xs = np.repeat(x[None], 3, axis=0)
    assert np.array_equal(res["MSE (RGB)"][1], o.image_mse(xs, res["reconstructions"][1]))
    np.testing.assert_allclose(res["MS-SSIM (RGB)"][1], o.ms_ssim(xs, res["reconstructions"][1]), rtol=1e-12)
    out = q.compress(x[None] / 255., vae, lambs)
    assert res["B"][1, 0] == np.sum(out["num_bits"][lambs[0]])
    for m, lamb in enumerate(lambs):                                     # the device sums ARE the reference's np.sum (utils.py:547-552)
        nb = np.asarray(out["num_bits"][lamb])[0]
        assert res["B"][1, m] == np.sum(nb) and res["BPL"][1, m] == np.sum(nb) / nb.size
        assert res["BPP"][1, m] == np.sum(nb) / (H * W) and res["BPPCL"][1, m] == np.sum(np.asarray(out["num_bits_cl"][lamb])[0]) / (H * W)

    # the same loop with a VAE that lives on the device: Z_hat and X_hat never visit the host as float32, same results
    import torch

    class TorchVAE(VAE):
        def encode(self, X):
            m, lv = VAE.encode(self, np.asarray(X))
            return torch.from_numpy(m).cuda(), torch.from_numpy(lv).cuda()

        def decode(self, Z):
            assert Z.is_cuda
            return 0.5 + 0.2 * Z[..., :3].repeat_interleave(4, dim=1).repeat_interleave(4, dim=2)
    res_t = utils.evaluate_compression_quantizer(q, TorchVAE(), files, lambs, return_reconstructions=True)
    for key in ("B", "BPP", "BPPCL", "BPL", "MSE (RGB)", "PSNR (Luma)", "MS-SSIM (Chroma)"):
        assert np.array_equal(res_t[key], res[key]), key
    assert all(np.array_equal(a, b) for a, b in zip(res_t["reconstructions"], res["reconstructions"]))
    with pytest.raises(Exception):
        utils.evaluate_compression_quantizer(q, vae, files, lambs, use_tf=True)


def test_beta_sweep_equals_single_betas():
    """E.test_betas (all betas in one K1n / K2 launch) == the notebook's per-beta test_beta, row by row."""
    _need_gpu()
    from vbq_amd import embeddings as E
    rng = np.random.default_rng(21)
    V, K = 1500, 40
    means = rng.normal(0, 1, (V, K)).astype(np.float32)
    stds = np.exp(rng.normal(-2, 0.5, (V, K))).astype(np.float32)
    an = rng.integers(0, V, (300, 4)).astype(np.int32)
    cp, _ = E.make_code_book(E.empirical_std(means))
    betas = [0.01, 1.0, 50.0, 1e4]
    sweep = E.test_betas(means, stds, betas, cp, an)
    assert sweep.shape == (4, 4)
    for i, b in enumerate(betas):
        assert np.array_equal(sweep[i], np.array(E.test_beta(means, stds, b, cp, analogies_id=an), dtype=np.float64))
    assert np.all(np.diff(sweep[:, 3]) < 0)                       # the rate falls as beta grows


def test_evaluate_compression_jpg(tmp_path):
    """utils.evaluate_compression_jpg: PIL encodes, the GPU scores; keys and values against the oracle."""
    _need_gpu()
    from io import BytesIO

    from PIL import Image
    from oracle import vbq_oracle as o
    from vbq_amd import utils
    yy, xx = np.mgrid[0:96, 0:128]
    arr = np.clip(128 + 100 * np.sin(yy / 9.0)[..., None] * np.cos(xx[..., None] / 7.0 + np.arange(3)), 0, 255).astype(np.uint8)
    p = tmp_path / "img.png"
    Image.fromarray(arr).save(p)
    quality = (5, 40, 90)
    res = utils.evaluate_compression_jpg(str(p), quality=quality, return_reconstructions=True)
    assert res["BPP"].shape == (3,) and np.all(np.diff(res["BPP"]) > 0)
    assert res["reconstructions"].shape == (3, 96, 128, 3)
    xs = np.repeat(arr[None], 3, axis=0)
    assert np.array_equal(res["MSE (RGB)"], o.image_mse(xs, res["reconstructions"]))
    np.testing.assert_allclose(res["MS-SSIM (RGB)"], o.ms_ssim(xs, res["reconstructions"]), rtol=1e-12)
    assert np.all(np.diff(res["PSNR (Luma)"]) > 0) and "MS-SSIM (Chroma) (dB)" in res
    buf = BytesIO()
    Image.open(p).convert("RGB").save(buf, "jpeg", quality=40)
    assert res["BPP"][1] == buf.tell() * 8 / (96 * 128)


def test_unit_to_u8_is_numpys_round_clip_cast():
    """vbq_unit_to_u8_f32 == np.clip(np.round(x * 255), 0, 255).astype(np.uint8) (utils.py:555) on values inside and outside [0, 1],
    exact halves (round half to even), tiny and huge magnitudes."""
    _need_gpu()
    import torch
    from vbq_amd import ops
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-0.2, 1.2, 200_001).astype(np.float32),
                        ((np.arange(-3, 260, dtype=np.float32) + np.float32(0.5)) / np.float32(255)),
                        (np.arange(0, 256, dtype=np.float32) / np.float32(255)),
                        np.float32([0.0, -0.0, 1.0, 1e-30, -1e-30, 1e30, -1e30, 0.0019607844, 0.99803925])])
    want = np.clip(np.round(x * 255), 0, 255).astype(np.uint8)
    got = ops.unit_to_u8(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    assert ops.unit_to_u8(torch.zeros((0, 3), device="cuda")).shape == (0, 3)
