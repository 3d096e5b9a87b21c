"""Drop-in counterparts of the solver functions of img-compression/utils.py.

Same names and argument meaning as the reference (utils.py:23-24, 307-327, 330-360, 363-423);
`backend` is accepted for source compatibility and ignored -- the arithmetic always runs in the
HIP kernel K1c (vbq_argmax_candidates_f32).  The distortion `fun` must come from
`curry_normal_logpdf` below (the only distortion the reference ever passes, quantizer.py:185):
the kernel hard-wires  -0.5*((z-loc)/scale)**2  in separately rounded f32 operations.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import VBQError


def n_bit_binary_floats(n):
    """utils.py:23-24."""
    return [i * 2 ** (-n) + 2 ** (-n - 1) for i in range(2 ** n)]


class NormalNegHalfSqErr:
    """Callable returned by curry_normal_logpdf(ignore_const=True): f(z) = -0.5*((z-loc)/scale)**2
    (utils.py:319-320).  Carries loc/scale so that the solvers can hand them to the kernel."""

    def __init__(self, loc, scale, const=None):
        self.loc, self.scale, self.const = loc, scale, const

    def __call__(self, z):
        v = -0.5 * ((z - self.loc) / self.scale) ** 2
        return v if self.const is None else v + self.const


def curry_normal_logpdf(loc, scale, ignore_const=False, backend=np):
    """utils.py:307-327.  With ignore_const=False the additive constant does not depend on the
    candidate, so the argmax (and therefore every solver result) is the same."""
    if ignore_const:
        return NormalNegHalfSqErr(loc, scale)
    s = scale.detach().cpu().numpy() if isinstance(scale, torch.Tensor) else np.asarray(scale)
    return NormalNegHalfSqErr(loc, scale, const=-np.log(s) - 0.5 * np.log(2 * np.pi))


def _dev_f32(a, device):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
    return t.to(device, torch.float32).contiguous()


def _device():
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: vbq_amd.utils has no CPU implementation")
    return torch.device("cuda", torch.cuda.current_device())


def _mode(code_lengths, lambs):
    """The NumPy backend of the reference computes f64 scores when the lengths are integers
    (utils.py:388 leaves L uncast); the TF backend casts them to f32.  Mirror that by dtype."""
    dt = code_lengths.dtype
    is_int = (dt in (torch.int32, torch.int64, torch.int16)) if isinstance(code_lengths, torch.Tensor) \
        else np.issubdtype(np.asarray(code_lengths).dtype, np.integer)
    return "f64" if is_int else "f32"


def batch_quantize_indep_dims(Z_shape, code_points, code_lengths, fun, lambs, backend=np, return_np=True, mode=None):
    """utils.py:363-423.  code_points: K x M, or M x B x K; code_lengths likewise, or L x M x B x K."""
    if not isinstance(fun, NormalNegHalfSqErr):
        raise TypeError("fun must be created by vbq_amd.utils.curry_normal_logpdf (Gaussian distortion)")
    dev = _device()
    B, K = Z_shape
    mode = mode or _mode(code_lengths, lambs)
    P = _dev_f32(code_points, dev)
    Lt = _dev_f32(code_lengths, dev)
    if P.dim() == 2:                                                                # utils.py:384-386
        assert P.shape[0] == K
        P = P.t()[:, None, :].expand(-1, B, -1).contiguous()
        Lt = Lt.t()[:, None, :].expand(-1, B, -1).contiguous()
    mu = _dev_f32(fun.loc, dev).expand(B, K).contiguous()
    sg = _dev_f32(fun.scale, dev).expand(B, K).contiguous()
    lambs = list(lambs)
    zh, bt = ops.argmax_candidates(P, Lt, mu, sg, [float(l) for l in lambs], mode=mode)
    int_len = (not torch.is_tensor(code_lengths) and np.issubdtype(np.asarray(code_lengths).dtype, np.integer)) or \
              (torch.is_tensor(code_lengths) and not code_lengths.dtype.is_floating_point)
    Z_hat_dict, num_bits_dict = {}, {}
    for i, lamb in enumerate(lambs):
        z = zh[i]
        b = bt[i].to(torch.int32) if int_len else bt[i]
        Z_hat_dict[lamb] = z.cpu().numpy() if return_np else z
        num_bits_dict[lamb] = b.cpu().numpy() if return_np else b
    return Z_hat_dict, num_bits_dict


def quantize_indep_dims(z, code_points, code_lengths, fun, lamb, backend=np, mode=None):
    """utils.py:330-360: one K-vector against K x M code points (every point is scored)."""
    K = len(z)
    Zd, Bd = batch_quantize_indep_dims((1, K), code_points, code_lengths, fun, [lamb], return_np=True, mode=mode)
    return Zd[lamb][0], Bd[lamb][0]


def get_all_N_bit_intervals(x, N, left_endpoints, right_endpoints):
    """utils.py:215-260 (the numba kernel), same in-place signature: fills the two (N+1) x K float64 arrays with the
    n-bit grid points around every x[k].  Runs on the GPU (vbq_xi_intervals_f64)."""
    from . import _lib, ops
    dev = _device()
    xd = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64))).to(dev)
    K = xd.numel()
    ld = torch.empty((N + 1, K), dtype=torch.float64, device=dev)
    rd = torch.empty((N + 1, K), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().vbq_xi_intervals_f64(ops._ptr(xd), K, int(N), ops._ptr(ld), ops._ptr(rd), ops._stream(xd)),
               "vbq_xi_intervals_f64")
    left_endpoints[...] = ld.cpu().numpy()
    right_endpoints[...] = rd.cpu().numpy()


def encode_vectorized(fun, z, lamb, squash, unsquash, max_bits_per_coord=16):
    """utils.py:263-304, the xi-space encoder.  `fun`, `squash`, `unsquash` are the caller's NumPy functions and are
    evaluated where the caller defined them; the interval search before them and the selection after them run on the GPU
    in the reference's float64.  Returns the reference's dict (z_hat, score, num_bits, xi_hat)."""
    from . import _lib, ops
    dev = _device()
    z = np.asarray(z)
    K, N = len(z), int(max_bits_per_coord)
    xi = np.asarray(squash(z), dtype=np.float64)
    xd = torch.from_numpy(np.ascontiguousarray(xi)).to(dev)
    ends = torch.empty((2, N + 1, K), dtype=torch.float64, device=dev)          # np.stack([left, right]) (:286)
    h = _lib.lib()
    _lib.check(h.vbq_xi_intervals_f64(ops._ptr(xd), K, N, ops._ptr(ends[0]), ops._ptr(ends[1]), ops._stream(xd)),
               "vbq_xi_intervals_f64")
    endpoints = ends.cpu().numpy()
    unsquashed = np.ascontiguousarray(np.asarray(unsquash(endpoints), dtype=np.float64))      # 2 x (N+1) x K (:287)
    F = np.ascontiguousarray(np.asarray(fun(unsquashed), dtype=np.float64))                    # (:288)
    Fd, ud = torch.from_numpy(F).to(dev), torch.from_numpy(unsquashed).to(dev)
    z_hat = torch.empty(K, dtype=torch.float64, device=dev)
    xi_hat = torch.empty(K, dtype=torch.float64, device=dev)
    f_z = torch.empty(K, dtype=torch.float64, device=dev)
    nb = torch.empty(K, dtype=torch.int64, device=dev)
    _lib.check(h.vbq_xi_select_f64(ops._ptr(Fd), ops._ptr(ends), ops._ptr(ud), K, N, float(lamb), ops._ptr(z_hat), ops._ptr(nb),
                                   ops._ptr(xi_hat), ops._ptr(f_z), ops._stream(Fd)), "vbq_xi_select_f64")
    return dict(z_hat=z_hat.cpu().numpy(), score=np.sum(f_z.cpu().numpy()), num_bits=nb.cpu().numpy(), xi_hat=xi_hat.cpu().numpy())


def convert_to_db(d):
    """utils.py:497-499 (BMSHJ ICLR 2018, p. 8)."""
    return -10 * np.log10(1 - d)


def _quality_by_mode(x, x_yc, x_hats, x_hats_yc):
    """MSE / PSNR / MS-SSIM of M reconstructions against one image in the three colour modes of utils.py:574-597:
    returns {'<metric> (<mode>)': float64 [M]}.  All M pairs of a mode go to the GPU as one batch."""
    from . import metrics as img_comparison_metrics
    M = len(x_hats)
    out = {}
    for mode in ("RGB", "Luma", "Chroma"):
        if mode == "RGB":
            x_comp, x_hats_comp = x, x_hats
        elif mode == "Luma":
            x_comp, x_hats_comp = x_yc[..., 0:1], x_hats_yc[..., 0:1]
        else:
            x_comp, x_hats_comp = x_yc[..., 1:], x_hats_yc[..., 1:]
        xs_comp = np.repeat(np.ascontiguousarray(x_comp)[None, ...], repeats=M, axis=0)
        x_hats_comp = np.ascontiguousarray(x_hats_comp)
        out["MSE (%s)" % mode] = img_comparison_metrics.mse(xs_comp, x_hats_comp)
        out["PSNR (%s)" % mode] = img_comparison_metrics.psnr(xs_comp, x_hats_comp, max_val=255)
        out["MS-SSIM (%s)" % mode] = img_comparison_metrics.ms_ssim(xs_comp, x_hats_comp, max_val=255)
    return out


def evaluate_compression_jpg(img_file, quality=(1, 10), return_reconstructions=False, use_tf=False):
    """utils.py:636-742: the JPEG baseline curve of one image -- PIL encodes at every quality setting (host, as in
    the reference), the reconstructions are scored on the GPU.  Same result dict."""
    from io import BytesIO

    from PIL import Image
    if use_tf:
        raise VBQError("use_tf=True needs TensorFlow's image ops; this build evaluates mse / psnr / ms_ssim natively")
    orig = Image.open(img_file)
    img = orig.convert("RGB")
    img_hats = []
    file_sizes = np.empty(len(quality))
    for i, qual in enumerate(quality):
        buf = BytesIO()
        img.save(buf, "jpeg", quality=qual)
        file_sizes[i] = buf.tell()
        img_hats.append(Image.open(buf).convert("RGB"))
    num_pixels = orig.size[0] * orig.size[1]
    results = {"BPP": file_sizes * 8 / num_pixels}
    x, x_yc = np.asarray(img), np.asarray(img.convert("YCbCr"))
    x_hats = np.array([np.asarray(h) for h in img_hats])
    x_hats_yc = np.array([np.asarray(h.convert("YCbCr")) for h in img_hats])
    if return_reconstructions:
        results["reconstructions"] = x_hats
    results.update(_quality_by_mode(x, x_yc, x_hats, x_hats_yc))
    results.update({"%s (dB)" % k: convert_to_db(v) for k, v in results.items() if "MS-SSIM" in k})
    return results


def _device_row_sums(d, settings):
    """Device tensor f32 [M] of np.sum(np.asarray(d[lamb])[0]) per setting in NumPy's float32 order (vbq_numpy_row_sums_f32) when
    the arrays are device-resident float32 views of one call with a batch of one image; else None."""
    from .lazy import common_stack
    t = common_stack([d[lamb] for lamb in settings])
    if t is not None and t.dtype == torch.float32 and t.dim() >= 2 and t.shape[1] == 1 and t.is_contiguous():
        return ops.numpy_row_sums(t)
    return None


def _sums_per_setting(d, settings):
    """[np.sum(np.asarray(d[lamb])[0]) for lamb in settings] (utils.py:547-552) as float32 scalars: on the device, in NumPy's
    own summation order, while the arrays are still there; on the host otherwise."""
    t = _device_row_sums(d, settings)
    if t is not None:
        return t.cpu().numpy()
    return np.array([np.sum(np.asarray(d[lamb])[0]) for lamb in settings])


def _device_u8(d, settings):
    from .lazy import common_stack
    t = common_stack([d[lamb] for lamb in settings])
    if t is not None and t.dtype == torch.float32:
        return ops.unit_to_u8(t[:, 0])
    return None


def _reconstructions_u8(d, settings):
    """np.clip(np.round(X_hat * 255), 0, 255).astype(np.uint8) of the first image of every setting (utils.py:554-556) ->
    [M, H, W, 3] uint8 on the host; taken on the device when the reconstructions still live there (a quarter of the bytes
    cross PCIe)."""
    t = _device_u8(d, settings)
    if t is not None:
        return t.cpu().numpy()
    return np.stack([np.clip(np.round(np.asarray(d[lamb])[0] * 255), 0, 255).astype(np.uint8) for lamb in settings])


def evaluation_device_reads(tmp, settings):
    """The three quantities the evaluation loop reads (see evaluation_reads) as small DEVICE tensors -- float32 [M] sums of
    'num_bits', of 'num_bits_cl', uint8 [M, H, W, 3] reconstructions -- or None when the results are not device-resident stacks of
    one call.  Nothing here synchronises (vbq_amd.replay captures it into the per-shape HIP graph)."""
    num_bits_cl = tmp.get("num_bits_cl", tmp["num_bits"])
    if not num_bits_cl:                                          # no raw-length models: the loop reads num_bits alone (utils.py:551)
        num_bits_cl = tmp["num_bits"]
    dev = (_device_row_sums(tmp["num_bits"], settings), _device_row_sums(num_bits_cl, settings), _device_u8(tmp["X_hat"], settings))
    return None if any(t is None for t in dev) else dev


def evaluation_reads(tmp, settings, staging=None):
    """Everything the evaluation loop reads from one `quantizer.compress(...)` result (utils.py:547-556) -> (sums of 'num_bits',
    sums of 'num_bits_cl', uint8 reconstructions), float32 [M] / float32 [M] / uint8 [M, H, W, 3] on the host.  With
    device-resident results: three small device tensors, ONE synchronisation (asynchronous copies into pinned memory kept in
    `staging`, a dict the caller holds between images); no latent-shaped array crosses PCIe."""
    num_bits_cl = tmp.get("num_bits_cl", tmp["num_bits"])
    dev = evaluation_device_reads(tmp, settings)
    if dev is None:
        return _sums_per_setting(tmp["num_bits"], settings), _sums_per_setting(num_bits_cl, settings), _reconstructions_u8(tmp["X_hat"], settings)
    staging = {} if staging is None else staging
    host = []
    for i, t in enumerate(dev):
        h = staging.get(i)
        if h is None or h.numel() < t.numel() or h.dtype != t.dtype:
            h = staging[i] = torch.empty(max(t.numel(), 1), dtype=t.dtype, pin_memory=True)
        hv = h[:t.numel()].view(t.shape)
        hv.copy_(t, non_blocking=True)
        host.append(hv)
    torch.cuda.current_stream(dev[0].device).synchronize()
    return tuple(np.array(h.numpy()) for h in host)


def evaluate_compression_quantizer(quantizer, vae, test_img_files, settings, model_input_float_type="float32",
                                   return_reconstructions=False, use_tf=False):
    """utils.py:502-634: compress every image with `quantizer.compress(X, vae, settings, clip=True)` and report bits
    and image quality per (image, setting).  Same result dict (keys 'B', 'BPP', 'BPPCL', 'BPL', '<metric> (<mode>)'
    for MSE / PSNR / MS-SSIM x RGB / Luma / Chroma, 'MS-SSIM (<mode>) (dB)', 'reconstructions').  Image I/O and
    colour conversion stay with PIL on the host, as in the reference; the metrics of all M reconstructions of an
    image are evaluated in one batch on the GPU (vbq_amd.metrics).  `use_tf=True` (TensorFlow's ssim ops) is not
    available."""
    from PIL import Image
    if use_tf:
        raise VBQError("use_tf=True needs TensorFlow's image ops; this build evaluates mse / psnr / ms_ssim natively")
    N, M = len(test_img_files), len(settings)
    results = {"reconstructions": []}
    for key in ("B", "BPP", "BPPCL", "BPL"):
        results[key] = np.empty([N, M])
    modes = ("RGB", "Luma", "Chroma")
    results.update({"%s (%s)" % (metric, mode): np.empty([N, M]) for mode in modes for metric in ("MSE", "PSNR", "MS-SSIM")})
    staging = {}
    for n, f in enumerate(test_img_files):
        orig = Image.open(f)
        img = orig.convert("RGB")
        num_pixels = orig.size[0] * orig.size[1]
        x = np.asarray(img)
        X = (x / 255.)[None, ...].astype(model_input_float_type)
        # one HIP graph replay per image of a shape seen before (vbq_amd.replay), where the quantizer offers it: the very calls
        # below, captured once -- `tmp` is then valid until the next image, which is all this loop needs
        replay = getattr(quantizer, "compress_replay", None)
        tmp, reads = replay(X, vae, settings, clip=True) if replay is not None else (quantizer.compress(X, vae, settings, clip=True), None)
        # utils.py:547-556: nbits = np.sum(num_bits), np.sum(num_bits_cl) per setting, X_hat as uint8.  While the per-lambda arrays
        # are still on the device (vbq_amd.lazy) the sums are taken there, in NumPy's own float32 order (vbq_numpy_row_sums_f32),
        # and 2 M floats + the uint8 images come to the host instead of latent-shaped float arrays; otherwise np.sum as in the
        # reference.
        sums, sums_cl, x_hat_u8 = reads if reads is not None else evaluation_reads(tmp, settings, staging)
        img_hats, x_hats = [], []
        for m, lamb in enumerate(settings):
            nbits = sums[m]
            results["B"][n, m] = nbits
            results["BPP"][n, m] = nbits / num_pixels
            results["BPL"][n, m] = nbits / (tmp["num_bits"][lamb].size // len(tmp["num_bits"][lamb]))
            results["BPPCL"][n, m] = sums_cl[m] / num_pixels
            x_hat = x_hat_u8[m]
            img_hats.append(Image.fromarray(x_hat))
            x_hats.append(x_hat)
        x_yc = np.asarray(img.convert("YCbCr"))
        x_hats = np.asarray(x_hats)
        x_hats_yc = np.array([np.asarray(h.convert("YCbCr")) for h in img_hats])
        if return_reconstructions:
            results["reconstructions"].append(x_hats)
        for key, val in _quality_by_mode(x, x_yc, x_hats, x_hats_yc).items():
            results[key][n] = val
    results.update({"%s (dB)" % k: convert_to_db(v) for k, v in results.items() if "MS-SSIM" in k})
    return results
