"""ctypes binding of libvbq_hip.so (include/vbq.h).  No CPU fallback exists: if the
library is missing or fails to load, every op raises."""
from __future__ import annotations

import ctypes as C
import os


_LIB = None

c_f32p = C.c_void_p      # device pointers travel as integers
c_u16p = C.c_void_p

OK = 0
ABI_VERSION = 5
LAYOUT_BC, LAYOUT_CB, LAYOUT_BC_TO_CB = 0, 1, 2
MODE_F32, MODE_F64_SCORE = 0, 1
BMSHJ_PARAMS_PER_CHANNEL = 43
COMM_ID_BYTES = 128

# name -> (restype, argtypes); mirrors include/vbq.h one to one (tests/test_abi.py checks it)
SIGNATURES = {
    "vbq_abi_version": (C.c_int, []),
    "vbq_last_error": (C.c_char_p, []),
    "vbq_solve_grid": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "vbq_device_count": (C.c_int, []),
    "vbq_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "vbq_quantize_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "vbq_check_inputs_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vbq_quantize_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vbq_n_bit_intervals_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "vbq_quantize_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "vbq_level_counts_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_int32, C.c_void_p]),
    "vbq_code_lengths_from_counts": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_host_neg_log2_freq_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_float, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "vbq_host_stage_run": (None, [C.c_void_p]),
    "vbq_histogram_rows_u16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                         C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "vbq_xi_intervals_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_xi_select_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_quantize_notebook_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_double),
                                            C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_histogram_u16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p]),
    "vbq_histogram_u16_i32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p]),
    "vbq_histogram_models_u16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vbq_index_max_u16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vbq_moments_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "vbq_numpy_sum_sq_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "vbq_numpy_sum_sq_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vbq_numpy_row_sums_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "vbq_numpy_row_sums_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vbq_gather_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "vbq_rd_sums_u16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "vbq_transpose_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "vbq_transpose_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "vbq_prep_planes_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_gather_latents_u16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_compress_latents_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "vbq_compress_latents_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vbq_build_entropy_models_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "vbq_build_entropy_models_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                                               C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "vbq_argmax_candidates_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_bmshj_cdf_pdf_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "vbq_bmshj_icdf_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_bmshj_icdf_chain_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "vbq_uniform_quantize_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_nearest_code_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "vbq_analogy_ranks_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int64]),
    "vbq_analogy_ranks_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p]),
    "vbq_image_sqerr_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "vbq_u8_to_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vbq_unit_to_u8_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vbq_ssim_scale_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "vbq_ssim_scale_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                     C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vbq_downsample2_f64": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "vbq_pack_counts_3x21": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "vbq_unpack_counts_3x21": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vbq_comm_unique_id": (C.c_int, [C.c_void_p]),
    "vbq_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32]),
    "vbq_allreduce_hist": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "vbq_comm_destroy": (C.c_int, [C.c_void_p]),
    "vbq_rans_encode_u16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "vbq_rans_decode_u16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbq_bmshj_nll_grad_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
}


class HostStageDesc(C.Structure):
    """include/vbq.h: vbq_host_stage."""
    _fields_ = [("h_counts", C.c_void_p), ("counts_are_i32", C.c_int32), ("add_level", C.c_int32), ("n_rows", C.c_int64),
                ("K", C.c_int64), ("add_n_smoothing", C.c_float), ("status", C.c_int32), ("log2_loop", C.c_void_p),
                ("log2_data", C.c_void_p), ("h_out_model", C.c_void_p), ("h_out_len", C.c_void_p), ("runs", C.c_int64)]


class VBQError(RuntimeError):
    pass


def library_path() -> str:
    from . import build as _build
    return os.environ.get("VBQ_HIP_LIBRARY", _build.LIB)


def lib():
    """Load (once) and return the ctypes handle.  Raises VBQError when the HIP extension has
    not been built -- there is deliberately no slower path to fall back to."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise VBQError(f"HIP extension not built: {path} is missing. Run `python -m vbq_amd.build` "
                       "(needs hipcc; cross-compiles gfx950 without a GPU).")
    try:
        h = C.CDLL(path)
    except OSError as e:
        raise VBQError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise VBQError(f"{path} does not export {name}; rebuild with `python -m vbq_amd.build --force`") from e
        fn.restype = res
        fn.argtypes = args
    if h.vbq_abi_version() != ABI_VERSION:
        raise VBQError(f"{path}: ABI version {h.vbq_abi_version()} != {ABI_VERSION}; rebuild with `python -m vbq_amd.build --force`")
    _LIB = h
    return h


def check(status: int, what: str):
    if status != OK:
        msg = lib().vbq_last_error().decode("utf-8", "replace")
        raise VBQError(f"{what} failed ({status}): {msg}")
