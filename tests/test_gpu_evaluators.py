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
