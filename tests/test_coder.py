"""f2: the rANS coder.  CPU part: frequency quantisation and the C checker's round trip.
GPU part: bitstreams bit-identical to the checker, decode(encode(idx)) == idx, rate vs estimate."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO

N = 10
T = 2047


def _streams(rng, S, n, spread):
    """Index streams shaped like K1 output: mass concentrated on few ranks for small `spread`."""
    idx = np.empty((S, n), np.uint16)
    for s in range(S):
        centre = rng.integers(200, 1800)
        v = np.rint(rng.normal(centre, spread[s % len(spread)], n)).astype(np.int64)
        idx[s] = np.clip(v, 0, T - 1)
    return idx


def test_quantize_frequencies_and_c_roundtrip():
    from vbq_amd.coder import ideal_bits, quantize_frequencies
    rng = np.random.default_rng(0)
    idx = _streams(rng, 6, 5000, [0.3, 3.0, 40.0, 400.0])
    counts = np.stack([np.bincount(r, minlength=T) for r in idx])
    freq = quantize_frequencies(counts, add_n_smoothing=1)
    assert freq.dtype == np.uint16 and freq.shape == (6, T)
    assert np.all(freq >= 1) and np.all(freq.astype(np.int64).sum(axis=1) == 1 << 15)
    assert np.array_equal(freq, quantize_frequencies(torch.from_numpy(counts)))         # deterministic
    for seg in (64, 1000, 5000, 7000):
        words, sizes = CO.rans_encode(idx, freq, seg)
        assert np.array_equal(CO.rans_decode(words, sizes, freq, idx.shape[1], seg), idx)
        bits = int(sizes.astype(np.int64).sum()) * 16
        est = ideal_bits(counts, freq)
        nseg = sizes.size
        assert est <= bits <= est + 48 * nseg                      # <= 32 bits of state + rounding per segment


@pytest.mark.gpu
def test_rans_gpu_matches_checker_and_roundtrips():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from vbq_amd.coder import RansCodec, ideal_bits, quantize_frequencies
    rng = np.random.default_rng(1)
    for S, n, seg in ((5, 3000, 1024), (3, 36864, 1024), (2, 1000, 7), (64, 2048, 256)):
        idx = _streams(rng, S, n, [0.2, 2.0, 25.0, 300.0])
        counts = np.stack([np.bincount(r, minlength=T) for r in idx])
        freq = quantize_frequencies(counts)
        codec = RansCodec(freq, N=N, segment=seg)
        d_idx = torch.from_numpy(idx).cuda()
        words, sizes = codec.encode(d_idx)
        w_ref, s_ref = CO.rans_encode(idx, freq, seg)
        assert np.array_equal(sizes.cpu().numpy(), s_ref)
        keep = np.arange(seg + 2)[None, None, :] < s_ref[..., None].astype(np.int64)
        assert np.array_equal(words.cpu().numpy()[keep], w_ref[keep])          # bit-identical streams
        back = codec.decode(words, sizes, n)
        assert torch.equal(back.view(torch.int16), d_idx.view(torch.int16))
        assert np.array_equal(CO.rans_decode(words.cpu().numpy(), sizes.cpu().numpy(), freq, n, seg), idx)
        bits = codec.compressed_bits(sizes)
        assert len(codec.pack(words, sizes)) * 8 == bits
        est = ideal_bits(counts, freq)
        assert est <= bits <= est + 48 * sizes.numel()


@pytest.mark.gpu
def test_rans_on_real_quantizer_output():
    """End to end on K1 output: coded size within 2 % (+ per-segment constant) of the reference's estimate."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from scipy.stats import norm
    from vbq_amd import ops
    from vbq_amd.coder import RansCodec, quantize_frequencies
    rng = np.random.default_rng(2)
    C, B = 8, 20000
    lambdas = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 8)]
    scale = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    xi = np.concatenate([(np.arange(2 ** k) + 0.5) / 2 ** k for k in range(N + 1)])
    tab = norm.ppf(xi[None], scale=scale[:, None]).astype(np.float32)
    mu = (scale[:, None] * rng.standard_normal((C, B))).astype(np.float32)
    sg = np.exp(-2 + 0.7 * rng.standard_normal((C, B))).astype(np.float32)
    idx = ops.quantize(torch.from_numpy(mu).cuda(), torch.from_numpy(sg).cuda(), torch.from_numpy(tab).cuda(), lambdas,
                       N=N, layout="cb")                                           # [L, C, B]
    counts = ops.histogram(idx, C, N=N, layout="cb")                                # [L, C, T]
    freq = quantize_frequencies(counts, add_n_smoothing=1)
    codec = RansCodec(freq.reshape(-1, T), N=N, segment=1024)
    words, sizes = codec.encode(idx)
    assert torch.equal(codec.decode(words, sizes, B).view(torch.int16), idx.reshape(-1, B).view(torch.int16))
    c = counts.cpu().numpy().astype(np.float64)
    sm = c + 1.0
    est = float(np.sum(c * -np.log2(sm / sm.sum(axis=2, keepdims=True))))         # quantizer.py:141-144 estimate
    bits = codec.compressed_bits(sizes)
    assert bits <= 1.02 * est + 40 * sizes.numel()
    assert bits >= 0.98 * est
