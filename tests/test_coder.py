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


@pytest.mark.gpu
def test_rans_decode_rejects_damaged_streams():
    """The decoder treats words / sizes as untrusted: out-of-range segment sizes, truncated segments, flipped
    words and foreign frequency tables raise VBQError (no out-of-bounds read, no silently wrong symbols), and a
    histogram / gather of arbitrary u16 indices stays inside its tables."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from vbq_amd import _lib, ops
    from vbq_amd.coder import RansCodec, quantize_frequencies
    rng = np.random.default_rng(9)
    S, n, seg = 4, 5000, 512
    idx = _streams(rng, S, n, [0.5, 3.0, 30.0, 200.0])
    counts = np.stack([np.bincount(r, minlength=T) for r in idx])
    codec = RansCodec(quantize_frequencies(counts), N=N, segment=seg)
    words, sizes = codec.encode(torch.from_numpy(idx).cuda())
    assert torch.equal(codec.decode(words, sizes, n).cpu(), torch.from_numpy(idx))
    for bad_size in (0, 1, seg + 3, 0xffffffff):
        s2 = sizes.clone()
        s2.view(torch.int32)[1, 2] = np.int64(bad_size).astype(np.int32) if bad_size > 0x7fffffff else bad_size
        with pytest.raises(_lib.VBQError, match="segment size"):
            codec.decode(words, s2, n)
    s2 = sizes.clone()                                                    # a truncated segment: fewer words than it needs
    s2.view(torch.int32)[0, 0] = 2
    with pytest.raises(_lib.VBQError, match="ran out of words|final state"):
        codec.decode(words, s2, n)
    w2 = words.clone()                                                    # flipped payload bits
    w2.view(torch.int16)[2, 1, :8] ^= 0x5a5a
    with pytest.raises(_lib.VBQError):
        codec.decode(w2, sizes, n)
    other = RansCodec(quantize_frequencies(counts[::-1].copy()), N=N, segment=seg)    # decoded with another model
    with pytest.raises(_lib.VBQError):
        other.decode(words, sizes, n)
    with pytest.raises(ValueError):
        codec.decode(words[:, :-1], sizes, n)
    # arbitrary u16 "indices" (not produced by K1): memory-safe, and vbq_index_max_u16 tells the caller
    junk = torch.from_numpy(rng.integers(0, 65536, (2, 3, 4096)).astype(np.uint16)).cuda()
    assert ops.index_max(junk) == int(junk.cpu().numpy().max())
    cnt = ops.histogram(junk, 3, N=N, layout="cb")
    assert 0 < int(cnt.sum().item()) <= junk.numel()                 # memory-safe; the counts of such input mean nothing
    tab = torch.arange(3 * T, dtype=torch.float32, device="cuda").reshape(3, T)
    got = ops.gather(junk, tab, 3, N=N, layout="cb").cpu().numpy()
    ref = (np.arange(3)[None, :, None] * T + np.minimum(junk.cpu().numpy().astype(np.int64), T - 1)).astype(np.float32)
    assert np.array_equal(got, ref)
