"""Comparison quantizers with the reference's API (img-compression/quantizer.py:259-459):
UniformQuantizer, KmeansQuantizer, ChannelwiseSimpleQuantizer and its multi-level wrapper.
The elementwise passes run on the GPU (vbq_baselines.hip); k-means itself is sklearn on the host,
exactly as in the reference (:309-311).  NumPy in, NumPy out."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from ._lib import VBQError, check


def _dev_f32(a):
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: vbq_amd.baselines has no CPU implementation")
    dt = a.dtype if isinstance(a, (torch.Tensor, np.ndarray)) else np.asarray(a).dtype
    if dt in (torch.float64, np.float64) and not isinstance(a, (list, tuple)):
        # the reference would bin float64 samples in float64 (floor((x - min) / delta), vq.vq); the kernels are f32
        raise ValueError("float64 samples are not supported: the comparison quantizers run in float32, the dtype of the "
                         "latents they are applied to (cast explicitly if f32 binning is what you want)")
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
    return t.to(torch.device("cuda", torch.cuda.current_device()), torch.float32).contiguous().reshape(-1)


def _code_lengths(counts, n_samples, add_n_smoothing):
    """quantizer.py:281-288 / 314-321 on the bincount."""
    counts = np.asarray(counts)
    if np.any(counts == 0):
        counts = counts + add_n_smoothing
    return -np.log2(counts / n_samples)


class UniformQuantizer:
    def __init__(self, quantization_levels, int_type=np.int32):
        self.quantization_levels = quantization_levels
        self.int_type = int_type

    def _run(self, samples, want_counts):
        x = _dev_f32(samples)
        N = self.quantization_levels
        I = torch.empty_like(x)
        q = torch.empty_like(x)
        counts = torch.zeros(N, dtype=torch.int64, device=x.device) if want_counts else None
        offset = np.float32(self.min + self.delta / 2)
        check(_lib.lib().vbq_uniform_quantize_f32(ops._ptr(x), x.numel(), C.c_float(float(self.min)),
                                                  C.c_float(float(self.delta)), C.c_float(float(offset)), N, ops._ptr(I),
                                                  ops._ptr(q), ops._ptr(counts), ops._stream(x)), "vbq_uniform_quantize_f32")
        return I, q, counts

    def fit(self, samples, add_n_smoothing=1.):
        s = np.asarray(samples)
        mn, mx = np.min(s), np.max(s)
        N = self.quantization_levels
        delta = (mx - mn) / N
        offset = mn + delta / 2
        self.min, self.max, self.delta = mn, mx, delta
        self.code_points = offset + delta * np.arange(N)
        _, _, counts = self._run(s, True)
        self.code_lengths = _code_lengths(counts.cpu().numpy(), len(s), add_n_smoothing)

    def quantize(self, samples):
        shape = np.shape(samples)
        I, q, _ = self._run(samples, False)
        I = I.cpu().numpy().reshape(shape)
        num_bits = np.take(self.code_lengths, I.astype(self.int_type))
        return q.cpu().numpy().reshape(shape), I, num_bits


class KmeansQuantizer:
    def __init__(self, quantization_levels, int_type=np.int32):
        self.quantization_levels = quantization_levels
        self.int_type = int_type

    def _vq(self, samples, want_counts):
        x = _dev_f32(samples)
        codes = torch.from_numpy(np.ascontiguousarray(self.code_points, dtype=np.float64)).to(x.device)
        I = torch.empty(x.numel(), dtype=torch.int32, device=x.device)
        q = torch.empty(x.numel(), dtype=torch.float64, device=x.device)
        counts = torch.zeros(codes.numel(), dtype=torch.int64, device=x.device) if want_counts else None
        check(_lib.lib().vbq_nearest_code_f64(ops._ptr(x), x.numel(), ops._ptr(codes), codes.numel(), ops._ptr(I), ops._ptr(q),
                                              ops._ptr(counts), ops._stream(x)), "vbq_nearest_code_f64")
        return I, q, counts

    def fit(self, samples, add_n_smoothing=1.):
        from sklearn.cluster import KMeans
        N = self.quantization_levels
        km = KMeans(n_clusters=N)
        s = np.asarray(samples).reshape((-1, 1))
        km.fit(s)
        self.code_points = km.cluster_centers_.ravel()                     # not guaranteed sorted (:312)
        counts = np.bincount(km.labels_.astype(self.int_type), minlength=N)
        self.code_lengths = _code_lengths(counts, len(s), add_n_smoothing)

    def quantize(self, samples):
        shape = np.shape(samples)
        I, q, _ = self._vq(samples, False)
        I = I.cpu().numpy().reshape(shape)
        num_bits = np.take(self.code_lengths, I.astype(self.int_type))
        return q.cpu().numpy().reshape(shape), I, num_bits


class ChannelwiseSimpleQuantizer:
    """quantizer.py:336-404."""

    def __init__(self, scalar_quantizer_type, num_channels, quantization_levels):
        self.quantization_levels = quantization_levels
        self.num_channels = num_channels
        self._quantizers = [scalar_quantizer_type(quantization_levels) for _ in range(num_channels)]

    def fit_latents(self, posterior_means, add_n_smoothing):
        C_ = np.shape(posterior_means)[-1]
        assert C_ == self.num_channels
        means = np.reshape(np.asarray(posterior_means), (-1, C_))
        for c, q in enumerate(self._quantizers):
            q.fit(means[:, c], add_n_smoothing)
        self.code_points = np.array([q.code_points for q in self._quantizers])
        self.code_lengths = np.array([q.code_lengths for q in self._quantizers])

    def fit(self, X, vae, add_n_smoothing):
        posterior_means, _ = vae.encode(X)
        self.fit_latents(posterior_means, add_n_smoothing)

    def compress_latents(self, posterior_means):
        pm = np.asarray(posterior_means)
        C_ = pm.shape[-1]
        assert C_ == self.num_channels
        means = np.reshape(pm, (-1, C_))
        quantized, num_bits = [], []
        for c, q in enumerate(self._quantizers):
            qz, _, nb = q.quantize(means[:, c])
            quantized.append(qz)
            num_bits.append(nb)
        Z_hat = np.reshape(np.array(quantized).transpose(), pm.shape)
        num_bits = np.reshape(np.array(num_bits).transpose(), pm.shape)
        return dict(Z_hat=Z_hat, num_bits=num_bits)

    def compress(self, X, vae, clip=True):
        posterior_means, _ = vae.encode(X)
        output = self.compress_latents(posterior_means)
        X_hat = np.asarray(vae.decode(output["Z_hat"]))
        output["X_hat"] = np.clip(X_hat, 0, 1) if clip else X_hat
        return output


class ChannelwiseSimpleQuantizerWrapper:
    """quantizer.py:407-459: one ChannelwiseSimpleQuantizer per number of levels, behind the compress()
    signature utils.evaluate_compression_quantizer expects."""

    def __init__(self, scalar_quantizer_type, num_channels, quantization_levels):
        self.quantization_levels = quantization_levels
        self.num_channels = num_channels
        self._quantizers = [ChannelwiseSimpleQuantizer(scalar_quantizer_type, num_channels, l) for l in quantization_levels]

    def fit(self, X, vae, add_n_smoothing):
        posterior_means, _ = vae.encode(X)
        for q in self._quantizers:
            q.fit_latents(posterior_means, add_n_smoothing)

    def compress(self, X, vae, quantization_levels, clip=True):
        assert quantization_levels == self.quantization_levels
        posterior_means, _ = vae.encode(X)
        pm = np.asarray(posterior_means)
        output = {"Z_hat": {}, "num_bits": {}}
        for l, q in zip(self.quantization_levels, self._quantizers):
            tmp = q.compress_latents(pm)
            for field in output:
                output[field][l] = tmp[field]
        Z = np.stack([output["Z_hat"][l] for l in self.quantization_levels])
        X_hat = np.reshape(np.asarray(vae.decode(np.reshape(Z, (-1,) + pm.shape[1:]))),
                           (len(self.quantization_levels),) + tuple(np.shape(X)))
        if clip:
            X_hat = np.clip(X_hat, 0, 1)
        output["X_hat"] = {l: X_hat[i] for i, l in enumerate(self.quantization_levels)}
        return output
