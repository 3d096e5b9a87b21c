"""Torch-tensor front end of the C-ABI (include/vbq.h).  PyTorch supplies device memory and
the current HIP stream; all arithmetic happens in libvbq_hip.so.  Every function raises
(VBQError / ValueError) instead of falling back to anything slower."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import LAYOUT_BC, LAYOUT_BC_TO_CB, LAYOUT_CB, MODE_F32, MODE_F64_SCORE, VBQError, check

_LAYOUTS = {"bc": LAYOUT_BC, "cb": LAYOUT_CB, LAYOUT_BC: LAYOUT_BC, LAYOUT_CB: LAYOUT_CB}
_MODES = {"f32": MODE_F32, "f64": MODE_F64_SCORE, MODE_F32: MODE_F32, MODE_F64_SCORE: MODE_F64_SCORE}


def table_size(N: int) -> int:
    return 2 ** (N + 1) - 1


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def raw_stream(device: torch.device) -> int:
    """Handle of the current HIP stream of `device` (an integer; 0 = the default stream) without building a torch.cuda.Stream
    object: torch.cuda.current_stream costs 4 us of Python per call, and every op asks once."""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def _stream(t: torch.Tensor):
    return C.c_void_p(raw_stream(t.device))


def _dev(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a torch tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise VBQError(f"{name}: tensor is on {t.device}; the VBQ kernels only run on a ROCm device "
                       "(there is no CPU implementation in this package)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _doubles(vals: Sequence[float]):
    arr = (C.c_double * len(vals))(*[float(v) for v in vals])
    return arr


def _rows_channels(shape, layout):
    if len(shape) == 1:
        return shape[0], 1
    if len(shape) != 2:
        raise ValueError(f"expected a 1-D or 2-D tensor, got shape {tuple(shape)}")
    return (shape[0], shape[1]) if layout == LAYOUT_BC else (shape[1], shape[0])


def quantize(mu: torch.Tensor, sigma: torch.Tensor, table_lm: torch.Tensor, lambdas: Sequence[float], *,
             N: int = 10, level_len: Optional[torch.Tensor] = None, layout="bc", mode="f32",
             want_zhat: bool = False, want_bits: bool = False, out_idx: Optional[torch.Tensor] = None,
             out_zhat: Optional[torch.Tensor] = None, out_bits: Optional[torch.Tensor] = None,
             workspace: Optional[torch.Tensor] = None, rows: Optional[Sequence[int]] = None,
             workgroups_per_cu: int = 0, reserved_workgroups: Optional[int] = None):
    """K1 (vbq_quantize_f32).  mu, sigma: f32 [rows, C] (layout 'bc') / [C, rows] ('cb') / [n] (C = 1).
    table_lm: f32 [C, T] level-major.  level_len: optional f32 [L, C, N+1].
    Returns idx u16 [L, *mu.shape] and, when asked, zhat / bits f32 of the same shape.
    layout 'bc->cb': inputs channel-last [rows, C], outputs channel-major planes [L, C, rows] (no input transposes;
    f32 mode and lambdas in the fast kernel's range only, VBQError otherwise).
    rows=(r0, r1): only that row range is solved (vbq_quantize_rows_f32; the output tensors are still full-size) --
    the host cuts a pass into chunks to overlap K2 with K1; workgroups_per_cu, reserved_workgroups (slots this call's
    resident grid leaves to a kernel of another stream; None: the library's default): see include/vbq.h."""
    to_planes = layout in ("bc->cb", LAYOUT_BC_TO_CB)
    layout = LAYOUT_BC if to_planes else _LAYOUTS[layout]
    mode = _MODES[mode]
    rows_range = rows
    mu = _dev(mu, torch.float32, "mu")
    sigma = _dev(sigma, torch.float32, "sigma")
    if mu.shape != sigma.shape:
        raise ValueError(f"mu {tuple(mu.shape)} and sigma {tuple(sigma.shape)} differ in shape")
    rows, Cc = _rows_channels(mu.shape, layout)
    T = table_size(N)
    table_lm = _dev(table_lm, torch.float32, "table_lm")
    if table_lm.numel() != Cc * T:
        raise ValueError(f"table_lm has {table_lm.numel()} entries, expected C*T = {Cc}*{T}")
    L = len(lambdas)
    if L < 1:
        raise ValueError("need at least one lambda")
    if level_len is not None:
        level_len = _dev(level_len, torch.float32, "level_len")
        if tuple(level_len.shape) != (L, Cc, N + 1):
            raise ValueError(f"level_len shape {tuple(level_len.shape)} != {(L, Cc, N + 1)}")
    h = _lib.lib()
    if to_planes and mu.dim() == 2 and Cc > 1:
        layout, oshape = LAYOUT_BC_TO_CB, (L, Cc, rows)
    else:
        oshape = (L,) + tuple(mu.shape)

    def _out(given, dtype, want, name):
        if given is not None:
            if tuple(given.shape) != oshape or given.dtype != dtype or not given.is_cuda or not given.is_contiguous():
                raise ValueError(f"{name}: expected a contiguous {dtype} device tensor of shape {oshape}")
            return given
        return torch.empty(oshape, dtype=dtype, device=mu.device) if want else None

    idx = _out(out_idx, torch.uint16, True, "out_idx")
    zhat = _out(out_zhat, torch.float32, want_zhat, "out_zhat")
    bits = _out(out_bits, torch.float32, want_bits, "out_bits")
    want_zhat, want_bits = zhat is not None, bits is not None
    wsb = h.vbq_quantize_workspace_bytes(Cc, L, N)
    ws = workspace if workspace is not None else torch.empty(wsb, dtype=torch.uint8, device=mu.device)
    if ws.numel() * ws.element_size() < wsb or not ws.is_cuda:
        raise ValueError(f"workspace must be a device tensor of at least {wsb} bytes")
    if mu.numel() == 0:
        out = (idx,) + ((zhat,) if want_zhat else ()) + ((bits,) if want_bits else ())
        return out if len(out) > 1 else idx
    if rows_range is None and not workgroups_per_cu and reserved_workgroups is None:
        check(h.vbq_quantize_f32(_ptr(mu), _ptr(sigma), rows, Cc, layout, _ptr(table_lm), _ptr(level_len),
                                 _doubles(lambdas), L, N, mode, _ptr(idx), _ptr(zhat), _ptr(bits), _ptr(ws), wsb,
                                 _stream(mu)), "vbq_quantize_f32")
    else:
        r0, r1 = (0, rows) if rows_range is None else (int(rows_range[0]), int(rows_range[1]))
        check(h.vbq_quantize_rows_f32(_ptr(mu), _ptr(sigma), rows, Cc, layout, _ptr(table_lm), _ptr(level_len),
                                      _doubles(lambdas), L, N, mode, _ptr(idx), _ptr(zhat), _ptr(bits), _ptr(ws), wsb,
                                      r0, r1, int(workgroups_per_cu), -1 if reserved_workgroups is None else int(reserved_workgroups),
                                      _stream(mu)), "vbq_quantize_rows_f32")
    out = (idx,)
    if want_zhat:
        out += (zhat,)
    if want_bits:
        out += (bits,)
    return out if len(out) > 1 else idx


def check_inputs(mu: torch.Tensor, sigma: torch.Tensor):
    """vbq_check_inputs_f32: raise ValueError when mu holds NaN / infinity or sigma holds NaN / infinity / values <= 0 --
    the inputs vbq_quantize_f32 does not define an answer for (synchronises: one 8-byte read)."""
    mu = _dev(mu, torch.float32, "mu")
    sigma = _dev(sigma, torch.float32, "sigma")
    if mu.shape != sigma.shape:
        raise ValueError(f"mu {tuple(mu.shape)} and sigma {tuple(sigma.shape)} differ in shape")
    bad = torch.zeros(2, dtype=torch.uint32, device=mu.device)
    check(_lib.lib().vbq_check_inputs_f32(_ptr(mu), _ptr(sigma), mu.numel(), _ptr(bad), _stream(mu)), "vbq_check_inputs_f32")
    b = bad.cpu().numpy()
    if b[0] or b[1]:
        raise ValueError(f"invalid latents: {int(b[0])} non-finite means, {int(b[1])} standard deviations that are not finite and positive")


def level_counts(mu: torch.Tensor, sigma: torch.Tensor, table_lm: torch.Tensor, lambdas: Sequence[float], *,
                 N: int = 10, level_len: Optional[torch.Tensor] = None, layout="bc", out: Optional[torch.Tensor] = None,
                 workspace: Optional[torch.Tensor] = None, reserved_workgroups: Optional[int] = None):
    """K1t / K1h (vbq_level_counts_f32; thresholds for raw lengths at N = 10, dense otherwise): the solve of `quantize` followed by the per-(lambda, channel) histogram of the
    winners' bit levels, in one kernel with no per-element output.  Returns int64 [L, C, N+1] (added into `out`)."""
    to_planes = layout in ("bc->cb", LAYOUT_BC_TO_CB)
    layout = LAYOUT_BC if to_planes else _LAYOUTS[layout]
    mu = _dev(mu, torch.float32, "mu")
    sigma = _dev(sigma, torch.float32, "sigma")
    if mu.shape != sigma.shape:
        raise ValueError(f"mu {tuple(mu.shape)} and sigma {tuple(sigma.shape)} differ in shape")
    rows, Cc = _rows_channels(mu.shape, layout)
    T = table_size(N)
    table_lm = _dev(table_lm, torch.float32, "table_lm")
    if table_lm.numel() != Cc * T:
        raise ValueError(f"table_lm has {table_lm.numel()} entries, expected C*T = {Cc}*{T}")
    L = len(lambdas)
    if L < 1:
        raise ValueError("need at least one lambda")
    if level_len is not None:
        level_len = _dev(level_len, torch.float32, "level_len")
        if tuple(level_len.shape) != (L, Cc, N + 1):
            raise ValueError(f"level_len shape {tuple(level_len.shape)} != {(L, Cc, N + 1)}")
    if to_planes and mu.dim() == 2 and Cc > 1:
        layout = LAYOUT_BC_TO_CB
    if out is None:
        out = torch.zeros((L, Cc, N + 1), dtype=torch.int64, device=mu.device)
    elif tuple(out.shape) != (L, Cc, N + 1) or out.dtype != torch.int64 or not out.is_cuda or not out.is_contiguous():
        raise ValueError(f"out: expected a contiguous int64 device tensor of shape {(L, Cc, N + 1)}")
    h = _lib.lib()
    wsb = h.vbq_quantize_workspace_bytes(Cc, L, N)
    ws = workspace if workspace is not None else torch.empty(wsb, dtype=torch.uint8, device=mu.device)
    if ws.numel() * ws.element_size() < wsb or not ws.is_cuda:
        raise ValueError(f"workspace must be a device tensor of at least {wsb} bytes")
    if mu.numel():
        check(h.vbq_level_counts_f32(_ptr(mu), _ptr(sigma), rows, Cc, layout, _ptr(table_lm), _ptr(level_len),
                                     _doubles(lambdas), L, N, _ptr(out), _ptr(ws), wsb,
                                     -1 if reserved_workgroups is None else int(reserved_workgroups), _stream(mu)),
              "vbq_level_counts_f32")
    return out


def code_lengths_from_counts(counts: torch.Tensor, lut: torch.Tensor, *, level_period: int = 0, want_len: bool = True,
                             want_model: bool = False):
    """vbq_code_lengths_from_counts: f32 tensors shaped like `counts` -- (level +) lut[count] and / or lut[count]."""
    if counts.dtype not in (torch.int64, torch.int32):
        raise ValueError("counts must be int64 or int32")
    counts = _dev(counts, counts.dtype, "counts")
    lut = _dev(lut, torch.float32, "lut")
    out_len = torch.empty(counts.shape, dtype=torch.float32, device=counts.device) if want_len else None
    out_model = torch.empty(counts.shape, dtype=torch.float32, device=counts.device) if want_model else None
    check(_lib.lib().vbq_code_lengths_from_counts(_ptr(counts), int(counts.dtype == torch.int32), counts.numel(), _ptr(lut),
                                                  lut.numel(), int(level_period), _ptr(out_len), _ptr(out_model),
                                                  _stream(counts)), "vbq_code_lengths_from_counts")
    if want_len and want_model:
        return out_len, out_model
    return out_len if want_len else out_model


def quantize_notebook(means: torch.Tensor, stds: torch.Tensor, codebook_lm: torch.Tensor, betas: Sequence[float], *,
                      N: int = 10, want_values: bool = True, out_idx: Optional[torch.Tensor] = None):
    """K1n (vbq_quantize_notebook_f64).  Returns (idx u16 [n_beta, *shape], values f32 or None)."""
    means = _dev(means, torch.float32, "means")
    stds = _dev(stds, torch.float32, "stds")
    if means.shape != stds.shape:
        raise ValueError("means and stds differ in shape")
    codebook_lm = _dev(codebook_lm, torch.float64, "codebook_lm")
    if codebook_lm.numel() != table_size(N):
        raise ValueError(f"codebook has {codebook_lm.numel()} entries, expected {table_size(N)}")
    nb = len(betas)
    n = means.numel()
    if out_idx is not None:
        if tuple(out_idx.shape) != (nb,) + tuple(means.shape) or out_idx.dtype != torch.uint16 or not out_idx.is_contiguous():
            raise ValueError(f"out_idx: expected a contiguous uint16 tensor of shape {(nb,) + tuple(means.shape)}")
        idx = out_idx
    else:
        idx = torch.empty((nb,) + tuple(means.shape), dtype=torch.uint16, device=means.device)
    val = torch.empty((nb,) + tuple(means.shape), dtype=torch.float32, device=means.device) if want_values else None
    check(_lib.lib().vbq_quantize_notebook_f64(_ptr(means), _ptr(stds), n, _ptr(codebook_lm), _doubles(betas), nb, N,
                                               _ptr(idx), _ptr(val), _stream(means)), "vbq_quantize_notebook_f64")
    return idx, val


def histogram(idx: torch.Tensor, n_ch: int, *, N: int = 10, layout="bc", out: Optional[torch.Tensor] = None,
              dtype=torch.int64, rows: Optional[Sequence[int]] = None):
    """K2 (vbq_histogram_u16 / _i32).  idx: u16 [L, rows, C] / [L, C, rows] / [L, n].  Returns counts
    [L, C, T] (added into `out` when given; `out.dtype` int64 or int32 selects the entry point)."""
    layout = _LAYOUTS[layout]
    rows_range = rows
    idx = _dev(idx, torch.uint16, "idx")
    L = idx.shape[0]
    E = idx[0].numel()
    if E % n_ch:
        raise ValueError(f"{E} indices per lambda is not a multiple of n_ch={n_ch}")
    rows = E // n_ch
    T = table_size(N)
    if out is None:
        out = torch.zeros((L, n_ch, T), dtype=dtype, device=idx.device)
    else:
        if out.dtype not in (torch.int64, torch.int32):
            raise ValueError("out must be int64 or int32")
        if not out.is_contiguous():
            raise ValueError("out must be contiguous (counts are accumulated in place)")
        out = _dev(out, out.dtype, "out")
        if tuple(out.shape) != (L, n_ch, T):
            raise ValueError(f"out shape {tuple(out.shape)} != {(L, n_ch, T)}")
    h = _lib.lib()
    if rows_range is not None:
        check(h.vbq_histogram_rows_u16(_ptr(idx), rows, n_ch, layout, L, N, _ptr(out), int(out.dtype == torch.int32),
                                       int(rows_range[0]), int(rows_range[1]), _stream(idx)), "vbq_histogram_rows_u16")
        return out
    fn, name = (h.vbq_histogram_u16, "vbq_histogram_u16") if out.dtype == torch.int64 else \
               (h.vbq_histogram_u16_i32, "vbq_histogram_u16_i32")
    check(fn(_ptr(idx), rows, n_ch, layout, L, N, _ptr(out), _stream(idx)), name)
    return out


def histogram_models(idx: torch.Tensor, n_ch: int, counts: torch.Tensor, *, N: int = 10, lut: Optional[torch.Tensor] = None,
                     models: Optional[torch.Tensor] = None):
    """vbq_histogram_models_u16: counts[L, C, T] := histogram of the planes idx [L, C, rows] (assigned, not added) and, with
    `lut` / `models` (f32 [L, C, T]), models := lut[counts] in the same pass.  Returns (counts, models)."""
    idx = _dev(idx, torch.uint16, "idx")
    if idx.dim() != 3 or idx.shape[1] != n_ch:
        raise ValueError(f"idx must be planes [L, {n_ch}, rows], got {tuple(idx.shape)}")
    L, _, rows = idx.shape
    T = table_size(N)
    if counts.dtype not in (torch.int64, torch.int32) or tuple(counts.shape) != (L, n_ch, T) or not counts.is_cuda or not counts.is_contiguous():
        raise ValueError(f"counts: expected a contiguous int32 / int64 device tensor of shape {(L, n_ch, T)}")
    if (lut is None) != (models is None):
        raise ValueError("lut and models go together")
    if models is not None:
        lut = _dev(lut, torch.float32, "lut")
        if models.dtype != torch.float32 or tuple(models.shape) != (L, n_ch, T) or not models.is_cuda or not models.is_contiguous():
            raise ValueError(f"models: expected a contiguous f32 device tensor of shape {(L, n_ch, T)}")
    check(_lib.lib().vbq_histogram_models_u16(_ptr(idx), rows, n_ch, L, N, _ptr(counts), int(counts.dtype == torch.int32), _ptr(lut),
                                              lut.numel() if lut is not None else 0, _ptr(models), _stream(idx)),
          "vbq_histogram_models_u16")
    return counts, models


def index_max(idx: torch.Tensor) -> int:
    """vbq_index_max_u16: the largest index of a u16 array (synchronises).  For indices of foreign origin: K2 and
    gather are memory-safe for anything, but only indices < T are meaningful."""
    idx = _dev(idx, torch.uint16, "idx")
    m = torch.zeros(1, dtype=torch.uint32, device=idx.device)
    check(_lib.lib().vbq_index_max_u16(_ptr(idx), idx.numel(), _ptr(m), _stream(idx)), "vbq_index_max_u16")
    return int(m.cpu().item())


def moments(x: torch.Tensor, *, layout="bc", out: Optional[torch.Tensor] = None):
    """K3 (vbq_moments_f32).  Returns f64 [C, 2] = (sum x, sum x^2) per channel."""
    layout = _LAYOUTS[layout]
    x = _dev(x, torch.float32, "x")
    rows, Cc = _rows_channels(x.shape, layout)
    if out is None:
        out = torch.zeros((Cc, 2), dtype=torch.float64, device=x.device)
    check(_lib.lib().vbq_moments_f32(_ptr(x), rows, Cc, layout, _ptr(out), _stream(x)), "vbq_moments_f32")
    return out


def numpy_sum_sq(x: torch.Tensor) -> torch.Tensor:
    """vbq_numpy_sum_sq_f32: np.sum(x.ravel()**2) of a float32 tensor in NumPy's own summation order -> f32 [1] (device)."""
    x = _dev(x, torch.float32, "x").reshape(-1)
    if x.data_ptr() % 16:                       # a view into a larger tensor: the block kernel loads 16 bytes per lane
        x = x.clone()
    out = torch.zeros(1, dtype=torch.float32, device=x.device)
    h = _lib.lib()
    wsb = h.vbq_numpy_sum_sq_workspace_bytes(x.numel())
    ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=x.device)
    check(h.vbq_numpy_sum_sq_f32(_ptr(x), x.numel(), _ptr(out), _ptr(ws), wsb, _stream(x)), "vbq_numpy_sum_sq_f32")
    return out


def unit_to_u8(x: torch.Tensor) -> torch.Tensor:
    """vbq_unit_to_u8_f32: np.clip(np.round(x * 255), 0, 255).astype(np.uint8) of a float32 device tensor (utils.py:555)."""
    x = _dev(x, torch.float32, "x")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(_lib.lib().vbq_unit_to_u8_f32(_ptr(x), x.numel(), _ptr(out), _stream(x)), "vbq_unit_to_u8_f32")
    return out


def numpy_row_sums(x: torch.Tensor, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """vbq_numpy_row_sums_f32: np.sum(x[r]) for every r of a contiguous float32 device tensor [rows, ...], in NumPy's own float32
    summation order (bit for bit) -> f32 [rows] on the device.  The reductions of the evaluation loop (utils.py:547-552) without
    bringing the per-lambda arrays to the host."""
    x = _dev(x, torch.float32, "x")
    rows = x.shape[0]
    n = x[0].numel() if rows else 0
    out = torch.empty(rows, dtype=torch.float32, device=x.device)
    h = _lib.lib()
    wsb = h.vbq_numpy_row_sums_workspace_bytes(rows, n)
    ws = workspace if workspace is not None and workspace.numel() * workspace.element_size() >= wsb else \
        torch.empty(max(wsb, 4), dtype=torch.uint8, device=x.device)
    check(h.vbq_numpy_row_sums_f32(_ptr(x), rows, n, _ptr(out), _ptr(ws), wsb, _stream(x)), "vbq_numpy_row_sums_f32")
    return out


def gather(idx: torch.Tensor, tab: torch.Tensor, n_ch: int, *, N: int = 10, layout="bc", out_layout=None):
    """vbq_gather_f32: out[l][e] = tab[(l,) c(e), idx[l][e]].  tab: f32 [C, T] or [L, C, T] indexed by RANK.
    With out_layout != layout the result comes back transposed (e.g. idx [L, C, B] -> out [L, B, C])."""
    layout = _LAYOUTS[layout]
    out_layout = layout if out_layout is None else _LAYOUTS[out_layout]
    idx = _dev(idx, torch.uint16, "idx")
    tab = _dev(tab, torch.float32, "tab")
    L = idx.shape[0]
    E = idx[0].numel()
    rows = E // n_ch
    T = table_size(N)
    per_lambda = tab.dim() == 3
    if tuple(tab.shape) != ((L, n_ch, T) if per_lambda else (n_ch, T)):
        raise ValueError(f"tab shape {tuple(tab.shape)} does not match (L={L}, C={n_ch}, T={T})")
    oshape = tuple(idx.shape)
    if out_layout != layout and idx.dim() == 3 and n_ch == 1:
        oshape = (L, idx.shape[2], idx.shape[1])          # same memory, other view
    if out_layout != layout and n_ch > 1:
        if idx.dim() != 3:
            raise ValueError("a layout change needs idx of shape [L, rows, C] or [L, C, rows]")
        oshape = (L, idx.shape[2], idx.shape[1])
    out = torch.empty(oshape, dtype=torch.float32, device=idx.device)
    check(_lib.lib().vbq_gather_f32(_ptr(idx), rows, n_ch, layout, L, N, _ptr(tab), int(per_lambda), _ptr(out),
                                    out_layout, _stream(idx)), "vbq_gather_f32")
    return out


_SPREADS = {"sigma": 0, "variance": 1, "logvar": 2}


def _spread_kind(spread: str) -> int:
    try:
        return _SPREADS[spread]
    except KeyError:
        raise ValueError(f"spread must be one of {sorted(_SPREADS)}, got {spread!r}") from None


def prep_planes(means_bc: torch.Tensor, spread_bc: torch.Tensor, *, spread: str = "sigma",
                out_mu: Optional[torch.Tensor] = None, out_sigma: Optional[torch.Tensor] = None):
    """vbq_prep_planes_f32: channel-last [rows, C] means and spreads -> channel-major planes [C, rows] in ONE launch.
    spread: what `spread_bc` holds -- 'sigma', 'variance' (sigma = sqrt(.)) or 'logvar' (sigma = sqrt(exp(.)), the
    `tf.exp(posterior_logvars) ** 0.5` of quantizer.py:197,202 inside the launch)."""
    means_bc = _dev(means_bc, torch.float32, "means")
    spread_bc = _dev(spread_bc, torch.float32, "spread")
    if means_bc.dim() != 2 or means_bc.shape != spread_bc.shape:
        raise ValueError(f"expected two [rows, C] tensors, got {tuple(means_bc.shape)} / {tuple(spread_bc.shape)}")
    r, c = means_bc.shape
    outs = []
    for o, name in ((out_mu, "out_mu"), (out_sigma, "out_sigma")):
        if o is None:
            o = torch.empty((c, r), dtype=torch.float32, device=means_bc.device)
        elif tuple(o.shape) != (c, r) or o.dtype != torch.float32 or not o.is_contiguous() or not o.is_cuda:
            raise ValueError(f"{name} must be a contiguous f32 device tensor of shape {(c, r)}")
        outs.append(o)
    check(_lib.lib().vbq_prep_planes_f32(_ptr(means_bc), _ptr(spread_bc), _spread_kind(spread), r, c, _ptr(outs[0]),
                                         _ptr(outs[1]), _stream(means_bc)), "vbq_prep_planes_f32")
    return outs[0], outs[1]


def gather_latents(idx_planes: torch.Tensor, *, N: int = 10, table_sorted: Optional[torch.Tensor] = None,
                   level_len: Optional[torch.Tensor] = None, models: Optional[torch.Tensor] = None, want_zhat: bool = True,
                   want_raw_bits: bool = True, want_num_bits: bool = False, want_idx: bool = False):
    """vbq_gather_latents_u16: ONE pass over rank indices in planes [L, C, B] -> channel-last [L, B, C] results
    (Z_hat f32, raw_num_bits int32 / f32 with level_len, num_bits f32, idx u16; None for the ones not asked for)."""
    idx_planes = _dev(idx_planes, torch.uint16, "idx_planes")
    if idx_planes.dim() != 3:
        raise ValueError(f"idx_planes must be [L, C, B], got {tuple(idx_planes.shape)}")
    L, Cc, B = idx_planes.shape
    T = table_size(N)
    if want_zhat:
        table_sorted = _dev(table_sorted, torch.float32, "table_sorted")
        if table_sorted.numel() != Cc * T:
            raise ValueError(f"table_sorted has {table_sorted.numel()} entries, expected {Cc}*{T}")
    if level_len is not None:
        level_len = _dev(level_len, torch.float32, "level_len")
        if tuple(level_len.shape) != (L, Cc, N + 1):
            raise ValueError(f"level_len shape {tuple(level_len.shape)} != {(L, Cc, N + 1)}")
    if want_num_bits:
        models = _dev(models, torch.float32, "models")
        if tuple(models.shape) != (L, Cc, T):
            raise ValueError(f"models shape {tuple(models.shape)} != {(L, Cc, T)}")
    dev = idx_planes.device
    z = torch.empty((L, B, Cc), dtype=torch.float32, device=dev) if want_zhat else None
    raw = torch.empty((L, B, Cc), dtype=torch.int32 if level_len is None else torch.float32, device=dev) if want_raw_bits else None
    nb = torch.empty((L, B, Cc), dtype=torch.float32, device=dev) if want_num_bits else None
    qi = torch.empty((L, B, Cc), dtype=torch.uint16, device=dev) if want_idx else None
    check(_lib.lib().vbq_gather_latents_u16(_ptr(idx_planes), B, Cc, L, N, _ptr(table_sorted) if want_zhat else None, _ptr(level_len),
                                            _ptr(models) if want_num_bits else None, _ptr(z), _ptr(raw), _ptr(nb), _ptr(qi),
                                            _stream(idx_planes)), "vbq_gather_latents_u16")
    return z, raw, nb, qi


def compress_latents(means_bc: torch.Tensor, spread_bc: torch.Tensor, table_lm: torch.Tensor, table_sorted: torch.Tensor,
                     lambdas: Sequence[float], *, N: int = 10, spread: str = "sigma",
                     level_len: Optional[torch.Tensor] = None, models: Optional[torch.Tensor] = None,
                     workspace: Optional[torch.Tensor] = None):
    """vbq_compress_latents_f32: the per-image call of quantizer.py:190-240 in one C call (planes, solve, fused lookups).
    means / spreads channel-last [B, C] (spread: 'sigma' | 'variance' | 'logvar', see prep_planes); returns
    (Z_hat f32, raw_num_bits int32 | f32, num_bits f32 | None), [L, B, C]."""
    means_bc = _dev(means_bc, torch.float32, "means")
    spread_bc = _dev(spread_bc, torch.float32, "spread")
    if means_bc.dim() != 2 or means_bc.shape != spread_bc.shape:
        raise ValueError(f"expected two [rows, C] tensors, got {tuple(means_bc.shape)} / {tuple(spread_bc.shape)}")
    B, Cc = means_bc.shape
    L = len(lambdas)
    T = table_size(N)
    if L < 1:
        raise ValueError("need at least one lambda")
    table_lm = _dev(table_lm, torch.float32, "table_lm")
    table_sorted = _dev(table_sorted, torch.float32, "table_sorted")
    if table_lm.numel() != Cc * T or table_sorted.numel() != Cc * T:
        raise ValueError(f"tables must hold C*T = {Cc}*{T} entries")
    if level_len is not None:
        level_len = _dev(level_len, torch.float32, "level_len")
        if tuple(level_len.shape) != (L, Cc, N + 1):
            raise ValueError(f"level_len shape {tuple(level_len.shape)} != {(L, Cc, N + 1)}")
    if models is not None:
        models = _dev(models, torch.float32, "models")
        if tuple(models.shape) != (L, Cc, T):
            raise ValueError(f"models shape {tuple(models.shape)} != {(L, Cc, T)}")
    dev = means_bc.device
    h = _lib.lib()
    wsb = h.vbq_compress_latents_workspace_bytes(B, Cc, L, N)
    ws = workspace if workspace is not None else torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    if ws.numel() * ws.element_size() < wsb or not ws.is_cuda:
        raise ValueError(f"workspace must be a device tensor of at least {wsb} bytes")
    z = torch.empty((L, B, Cc), dtype=torch.float32, device=dev)
    raw = torch.empty((L, B, Cc), dtype=torch.int32 if level_len is None else torch.float32, device=dev)
    nb = torch.empty((L, B, Cc), dtype=torch.float32, device=dev) if models is not None else None
    if B:
        check(h.vbq_compress_latents_f32(_ptr(means_bc), _ptr(spread_bc), _spread_kind(spread), B, Cc, _ptr(table_lm),
                                         _ptr(table_sorted), _ptr(level_len), _ptr(models), _doubles(lambdas), L, N, _ptr(z),
                                         _ptr(raw), _ptr(nb), _ptr(ws), ws.numel() * ws.element_size(), _stream(means_bc)),
              "vbq_compress_latents_f32")
    return z, raw, nb


def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None):
    """vbq_transpose_f32: [rows, cols] f32 -> [cols, rows] (channel-last <-> channel-major planes)."""
    x = _dev(x, torch.float32, "x")
    if x.dim() != 2:
        raise ValueError("transpose expects a 2-D tensor")
    r, c = x.shape
    if out is None:
        out = torch.empty((c, r), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (c, r) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous f32 tensor of shape {(c, r)}")
    check(_lib.lib().vbq_transpose_f32(_ptr(x), r, c, _ptr(out), _stream(x)), "vbq_transpose_f32")
    return out


def rd_sums(mu: torch.Tensor, sigma: torch.Tensor, idx: torch.Tensor, tab_sorted: torch.Tensor, n_ch: int, *, N: int = 10,
            layout="bc", rate: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """vbq_rd_sums_u16: f64 [L, 2] = per lambda (sum of (z - mu)^2 / (2 sigma^2), sum of rate[idx]) over all elements, z looked
    up in the SORTED table by rank index.  rate: f32 [C, T] or [L, C, T] (entropy models), or None."""
    layout = _LAYOUTS[layout]
    mu = _dev(mu, torch.float32, "mu")
    sigma = _dev(sigma, torch.float32, "sigma")
    idx = _dev(idx, torch.uint16, "idx")
    tab_sorted = _dev(tab_sorted, torch.float32, "tab_sorted")
    L = idx.shape[0]
    E = mu.numel()
    T = table_size(N)
    if idx[0].numel() != E or sigma.numel() != E or E % n_ch:
        raise ValueError("mu, sigma and idx[l] must hold the same number of elements, a multiple of n_ch")
    if tab_sorted.numel() != n_ch * T:
        raise ValueError(f"tab_sorted has {tab_sorted.numel()} entries, expected {n_ch}*{T}")
    per_lambda = 0
    if rate is not None:
        rate = _dev(rate, torch.float32, "rate")
        per_lambda = int(rate.dim() == 3)
        if tuple(rate.shape) != ((L, n_ch, T) if per_lambda else (n_ch, T)):
            raise ValueError(f"rate shape {tuple(rate.shape)} does not match (L={L}, C={n_ch}, T={T})")
    if out is None:
        out = torch.zeros((L, 2), dtype=torch.float64, device=mu.device)
    check(_lib.lib().vbq_rd_sums_u16(_ptr(mu), _ptr(sigma), _ptr(idx), E // n_ch, n_ch, layout, L, N, _ptr(tab_sorted), _ptr(rate),
                                     per_lambda, _ptr(out), _stream(mu)), "vbq_rd_sums_u16")
    return out


def transpose_planes(x: torch.Tensor, out: Optional[torch.Tensor] = None):
    """vbq_transpose_planes: [batch, rows, cols] -> [batch, cols, rows] for uint16 / float32 / int32 stacks (e.g. plane
    indices [L, C, B] -> channel-last [L, B, C])."""
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise VBQError("transpose_planes: expected a tensor on a ROCm device")
    if x.dim() != 3 or x.element_size() not in (2, 4):
        raise ValueError("transpose_planes expects a 3-D tensor of 2- or 4-byte elements")
    x = x.contiguous()
    b, r, c = x.shape
    if out is None:
        out = torch.empty((b, c, r), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (b, c, r) or out.dtype != x.dtype or not out.is_contiguous() or not out.is_cuda:
        raise ValueError(f"out must be a contiguous {x.dtype} device tensor of shape {(b, c, r)}")
    check(_lib.lib().vbq_transpose_planes(_ptr(x), b, r, c, x.element_size(), _ptr(out), _stream(x)), "vbq_transpose_planes")
    return out


def argmax_candidates(P: torch.Tensor, lens: torch.Tensor, mu: torch.Tensor, sigma: torch.Tensor,
                      lambdas: Sequence[float], *, mode="f32", want_j=False):
    """K1c (vbq_argmax_candidates_f32).  P: f32 [M, *shape]; lens: f32 [M, *shape] or [L, M, *shape]."""
    mode = _MODES[mode]
    P = _dev(P, torch.float32, "P")
    lens = _dev(lens, torch.float32, "lens")
    mu = _dev(mu, torch.float32, "mu")
    sigma = _dev(sigma, torch.float32, "sigma")
    M = P.shape[0]
    L = len(lambdas)
    if tuple(P.shape[1:]) != tuple(mu.shape) or mu.shape != sigma.shape:
        raise ValueError("P must be [M, *mu.shape] and sigma must match mu")
    per_lambda = lens.dim() == P.dim() + 1
    if tuple(lens.shape) != (((L,) if per_lambda else ()) + tuple(P.shape)):
        raise ValueError(f"lens shape {tuple(lens.shape)} incompatible with P {tuple(P.shape)} and L={L}")
    n = mu.numel()
    zhat = torch.empty((L,) + tuple(mu.shape), dtype=torch.float32, device=mu.device)
    bits = torch.empty_like(zhat)
    j = torch.empty((L,) + tuple(mu.shape), dtype=torch.int32, device=mu.device) if want_j else None
    check(_lib.lib().vbq_argmax_candidates_f32(_ptr(P), _ptr(lens), int(per_lambda), _ptr(mu), _ptr(sigma), n,
                                               _doubles(lambdas), L, M, mode, _ptr(j), _ptr(zhat), _ptr(bits),
                                               _stream(mu)), "vbq_argmax_candidates_f32")
    return (zhat, bits, j) if want_j else (zhat, bits)


def bmshj_cdf_pdf(params: torch.Tensor, x: torch.Tensor, *, cdf=True, pdf=True, logpdf=False):
    """K4 (vbq_bmshj_cdf_pdf_f32).  params: f32 [C, 43] effective parameters; x: f32 [..., C]."""
    params = _dev(params, torch.float32, "params")
    x = _dev(x, torch.float32, "x")
    Cc = params.shape[0]
    if params.shape[1] != _lib.BMSHJ_PARAMS_PER_CHANNEL or x.shape[-1] != Cc:
        raise ValueError(f"params must be [C, 43] and x [..., C]; got {tuple(params.shape)}, {tuple(x.shape)}")
    rows = x.numel() // Cc
    outs = [torch.empty_like(x) if f else None for f in (cdf, pdf, logpdf)]
    check(_lib.lib().vbq_bmshj_cdf_pdf_f32(_ptr(params), _ptr(x), rows, Cc, _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                           _stream(x)), "vbq_bmshj_cdf_pdf_f32")
    return tuple(outs)


def bmshj_icdf_step(params, xi, left, right, mid, flags):
    """K4 (vbq_bmshj_icdf_step_f32): one in-place bisection update; flags u32[2] must be preset to (0, 0x7f800000)."""
    Cc = params.shape[0]
    rows = xi.numel() // Cc
    check(_lib.lib().vbq_bmshj_icdf_step_f32(_ptr(params), _ptr(xi), rows, Cc, _ptr(left), _ptr(right), _ptr(mid),
                                             _ptr(flags), _stream(xi)), "vbq_bmshj_icdf_step_f32")


def bmshj_icdf_chain(params, xi, left, right, mid, flags, n_steps: int, tol: float, first: bool = True):
    """K4 (vbq_bmshj_icdf_chain_f32): n_steps bisection updates enqueued at once, the reference's stopping rule applied on the
    device between them; flags u32 [n_steps + 1, 2] is written (read it after a synchronisation)."""
    Cc = params.shape[0]
    rows = xi.numel() // Cc
    check(_lib.lib().vbq_bmshj_icdf_chain_f32(_ptr(params), _ptr(xi), rows, Cc, _ptr(left), _ptr(right), _ptr(mid), _ptr(flags),
                                              int(n_steps), float(tol), int(bool(first)), _stream(xi)), "vbq_bmshj_icdf_chain_f32")


def bmshj_nll_grad(params: torch.Tensor, x_cb: torch.Tensor, out: Optional[torch.Tensor] = None):
    """K4 (vbq_bmshj_nll_grad_f32).  params f32 [C, 43] effective; x_cb f32 [C, n] planes.
    Returns f64 [C, 44]: d(sum -log(pdf+1e-10))/d(params) and, in column 43, the sum itself."""
    params = _dev(params, torch.float32, "params")
    x_cb = _dev(x_cb, torch.float32, "x_cb")
    Cc, n = x_cb.shape
    if tuple(params.shape) != (Cc, _lib.BMSHJ_PARAMS_PER_CHANNEL):
        raise ValueError(f"params must be [{Cc}, 43], got {tuple(params.shape)}")
    if out is None:
        out = torch.zeros((Cc, 44), dtype=torch.float64, device=x_cb.device)
    check(_lib.lib().vbq_bmshj_nll_grad_f32(_ptr(params), _ptr(x_cb), n, Cc, _ptr(out), _stream(x_cb)),
          "vbq_bmshj_nll_grad_f32")
    return out
