"""f3: the comparison quantizers against golden vectors produced by the reference's own classes
(tests/golden/make_golden.py g9)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_uniform_and_kmeans_quantizers_golden(golden):
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from vbq_amd.baselines import ChannelwiseSimpleQuantizer, KmeansQuantizer, UniformQuantizer
    g = golden("g9_baselines.npz")
    x = g["x"]
    for K in (4, 16, 61):
        u = UniformQuantizer(K)
        u.fit(x, add_n_smoothing=1)
        assert u.min == g[f"u{K}_min"] and u.delta == g[f"u{K}_delta"]
        assert np.array_equal(u.code_points, g[f"u{K}_code_points"])
        assert np.array_equal(u.code_lengths, g[f"u{K}_code_lengths"])
        q, I, nb = u.quantize(x)
        assert q.dtype == np.float32 and np.array_equal(q, g[f"u{K}_q"])
        assert np.array_equal(I, g[f"u{K}_I"]) and np.array_equal(nb, g[f"u{K}_bits"])
    k = KmeansQuantizer(12)
    k.code_points = g["k_centers"]
    k.code_lengths = -np.log2(np.full(12, 1 / 12.0))
    q, I, nb = k.quantize(x)
    assert np.array_equal(q, g["k_q"]) and np.array_equal(I, g["k_I"])
    # channel-wise wrapper plumbing
    cq = ChannelwiseSimpleQuantizer(UniformQuantizer, 3, 8)
    lat = np.stack([x[:900], 2 * x[900:1800], x[1800:2700] - 1], axis=1).reshape(1, 30, 30, 3)
    cq.fit_latents(lat, 1)
    out = cq.compress_latents(lat)
    assert out["Z_hat"].shape == lat.shape and out["num_bits"].shape == lat.shape
    u0 = UniformQuantizer(8)
    u0.fit(lat.reshape(-1, 3)[:, 1], 1)
    assert np.array_equal(out["Z_hat"].reshape(-1, 3)[:, 1], u0.quantize(lat.reshape(-1, 3)[:, 1])[0])
