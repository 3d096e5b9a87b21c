"""Image comparison metrics with the call surface of img-compression/img_comparison_metrics.py
(mse :6-16, psnr :19-33, ms_ssim :160-220) on the GPU (vbq_metrics.hip), in the reference's float64
arithmetic, plus convert_to_db (utils.py:497-499).  Batches [B, H, W, C] of integer images (uint8)
or float arrays; NumPy in (or torch tensors already on the device), NumPy out."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from ._lib import VBQError, check

float_type = "float64"


def _device():
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: vbq_amd.metrics has no CPU implementation")
    return torch.device("cuda", torch.cuda.current_device())


def _on_device(x):
    return isinstance(x, torch.Tensor) and x.is_cuda


def _check_pair(img1, img2):
    """NumPy arrays -- or, for callers that already hold the images on the device, torch tensors there (no PCIe)."""
    a, b = (x if _on_device(x) else np.asarray(x) for x in (img1, img2))
    if tuple(a.shape) != tuple(b.shape):
        raise RuntimeError("Input images must have the same shape (%s vs. %s)." % (tuple(a.shape), tuple(b.shape)))
    if a.ndim != 4:
        raise RuntimeError("Input images must have four dimensions, not %d" % a.ndim)
    return a, b


def _is_u8(a):
    return a.dtype == (torch.uint8 if _on_device(a) else np.uint8)


def _as_f64_device(a):
    """uint8 batches are widened on the device (vbq_u8_to_f64); anything else is cast on the host first."""
    if _on_device(a):
        if a.dtype != torch.uint8:
            return a.to(torch.float64).contiguous()
        u = a.contiguous()
    else:
        dev = _device()
        if a.dtype != np.uint8:
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        u = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = torch.empty(u.shape, dtype=torch.float64, device=u.device)
    check(_lib.lib().vbq_u8_to_f64(ops._ptr(u), u.numel(), ops._ptr(out), ops._stream(u)), "vbq_u8_to_f64")
    return out


def mse(img1, img2):
    """img_comparison_metrics.py:6-16: mean squared difference over (H, W, C), float64 [B]."""
    a, b = _check_pair(img1, img2)
    if _is_u8(a) and _is_u8(b):
        ta, tb = (x.contiguous() if _on_device(x) else torch.from_numpy(np.ascontiguousarray(x)).to(_device()) for x in (a, b))
        out = torch.empty(a.shape[0], dtype=torch.int64, device=ta.device)
        n = int(np.prod(tuple(a.shape[1:])))
        check(_lib.lib().vbq_image_sqerr_u8(ops._ptr(ta), ops._ptr(tb), a.shape[0], n, ops._ptr(out), ops._stream(ta)),
              "vbq_image_sqerr_u8")
        return out.cpu().numpy().astype(np.float64) / n           # the integer sum is exact, as NumPy's f64 sum is
    a64, b64 = _as_f64_device(a), _as_f64_device(b)
    return torch.mean((a64 - b64) ** 2, dim=(1, 2, 3)).cpu().numpy()


def psnr(img1, img2, max_val=255, mse=None):
    """img_comparison_metrics.py:19-33."""
    if mse is None:
        mse = globals()["mse"](img1, img2)
    return 20 * np.log10(max_val) - 10 * np.log10(mse)


def _gauss_1d(size, sigma):
    """Separable factor of _FSpecialGauss (:70-81): g_ij = e_i e_j / (sum e)^2."""
    radius = size // 2
    offset = 0.5 if size % 2 == 0 else 0.0
    x = np.arange(size, dtype=np.float64) - radius + offset
    e = np.exp(-(x ** 2) / (2.0 * sigma ** 2))
    return e / e.sum()


def _ssim_for_multiscale(im1, im2, max_val, filter_size, filter_sigma, k1, k2):
    """:84-157 on device tensors [B, H, W, C] f64 -> (ssim [B], cs [B]) device tensors."""
    B, H, W, Cc = im1.shape
    size = min(filter_size, H, W)
    sigma = size * filter_sigma / filter_size if filter_size else 0
    if not filter_size:
        raise VBQError("filter_size=0 (no blur) is not supported on the GPU path")
    win = torch.from_numpy(_gauss_1d(size, sigma)).to(im1.device)
    h = _lib.lib()
    nbytes = h.vbq_ssim_scale_workspace_bytes(B, H, W, Cc, size)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=im1.device)
    ssim = torch.empty(B, dtype=torch.float64, device=im1.device)
    cs = torch.empty(B, dtype=torch.float64, device=im1.device)
    check(h.vbq_ssim_scale_f64(ops._ptr(im1), ops._ptr(im2), B, H, W, Cc, ops._ptr(win), size, C.c_double((k1 * max_val) ** 2),
                               C.c_double((k2 * max_val) ** 2), ops._ptr(ssim), ops._ptr(cs), ops._ptr(ws), C.c_size_t(nbytes),
                               ops._stream(im1)), "vbq_ssim_scale_f64")
    return ssim, cs


def _downsample(im):
    B, H, W, Cc = im.shape
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, Cc), dtype=torch.float64, device=im.device)
    check(_lib.lib().vbq_downsample2_f64(ops._ptr(im), B, H, W, Cc, ops._ptr(out), ops._stream(im)), "vbq_downsample2_f64")
    return out


def ms_ssim(img1, img2, max_val=255, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, weights=None):
    """img_comparison_metrics.py:160-220: MS-SSIM per image of the batch, float64 [B]."""
    a, b = _check_pair(img1, img2)
    weights = np.array(weights if weights else [0.0448, 0.2856, 0.3001, 0.2363, 0.1333])
    levels = weights.size
    im1, im2 = _as_f64_device(a), _as_f64_device(b)
    per_scale = []                                           # (ssim, cs) of every scale stay on the device: ONE copy at the end
    for i in range(levels):
        per_scale.extend(_ssim_for_multiscale(im1, im2, max_val, filter_size, filter_sigma, k1, k2))
        if i + 1 < levels:
            im1, im2 = _downsample(im1), _downsample(im2)
    both = torch.stack(per_scale).cpu().numpy().reshape(levels, 2, a.shape[0])
    mssim, mcs = both[:, 0], both[:, 1]
    return np.prod(mcs[0:levels - 1] ** weights[0:levels - 1, np.newaxis], axis=0) * (mssim[levels - 1] ** weights[levels - 1])


def convert_to_db(d):
    """utils.py:497-499."""
    return -10 * np.log10(1 - d)
