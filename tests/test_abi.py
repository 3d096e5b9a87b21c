"""The C-ABI library loads on a CPU-only host and exports exactly what include/vbq.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vbq.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vbq_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    from vbq_amd import build
    return build.build_hip()


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("vbq_quantize_f32", "vbq_quantize_notebook_f64", "vbq_histogram_u16", "vbq_histogram_u16_i32", "vbq_moments_f32",
                 "vbq_transpose_f32", "vbq_transpose_planes", "vbq_rd_sums_u16",
                 "vbq_gather_f32", "vbq_argmax_candidates_f32", "vbq_bmshj_cdf_pdf_f32", "vbq_bmshj_icdf_step_f32", "vbq_bmshj_icdf_chain_f32", "vbq_bmshj_nll_grad_f32", "vbq_rans_encode_u16", "vbq_rans_decode_u16", "vbq_uniform_quantize_f32", "vbq_nearest_code_f64",
           "vbq_analogy_ranks_workspace_bytes", "vbq_analogy_ranks_f32", "vbq_image_sqerr_u8", "vbq_u8_to_f64",
           "vbq_pack_counts_3x21", "vbq_unpack_counts_3x21", "vbq_ssim_scale_workspace_bytes", "vbq_ssim_scale_f64", "vbq_downsample2_f64",
                 "vbq_last_error", "vbq_abi_version"):
        assert must in names


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (vbq_[a-z0-9_]+)", out))
    assert set(declared_functions()) <= exported, sorted(set(declared_functions()) - exported)
    # and nothing but the C-ABI leaks out with C linkage
    assert all(n in declared_functions() for n in exported), sorted(exported - set(declared_functions()))


def test_ctypes_binding_matches_header(built_lib):
    from vbq_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    h = _lib.lib()
    assert h.vbq_abi_version() == 5
    assert isinstance(h.vbq_device_count(), int)
    # argument validation happens before any device work: callable without a GPU
    assert h.vbq_quantize_workspace_bytes(256, 32, 10) >= 256 * 32 * 11 * 4
    r = h.vbq_quantize_f32(None, None, 5, 1, 0, None, None, None, 1, 10, 0, None, None, None, None, 0, None)
    assert r == -1 and b"null pointer" in h.vbq_last_error()
    r = h.vbq_quantize_f32(None, None, 0, 1, 7, None, None, None, 1, 10, 0, None, None, None, None, 0, None)
    assert r == -1 and b"layout" in h.vbq_last_error()
    r = h.vbq_histogram_u16(None, -1, 1, 0, 1, 10, None, None)
    assert r == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vbq_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setenv("VBQ_HIP_LIBRARY", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.VBQError, match="not built"):
        _lib.lib()
    monkeypatch.delenv("VBQ_HIP_LIBRARY")
    monkeypatch.setattr(_lib, "_LIB", None)
    assert _lib.lib() is not None


def test_argument_validation_of_the_newer_entry_points(built_lib):
    """Every entry point validates sizes and pointers before touching the device: callable on a CPU-only host."""
    import ctypes as C
    from vbq_amd import _lib
    h = _lib.lib()
    assert h.vbq_analogy_ranks_workspace_bytes(100_000, 100, 19_544) > 4 * 100_000 * 100
    assert h.vbq_analogy_ranks_workspace_bytes(0, 100, 10) == 0
    assert h.vbq_analogy_ranks_f32(None, 0, 100, None, 5, None, None, 0, None) == -1 and b"bad shape" in h.vbq_last_error()
    assert h.vbq_analogy_ranks_f32(None, 10, 4, None, 5, None, None, 0, None) == -1 and b"null pointer" in h.vbq_last_error()
    assert h.vbq_analogy_ranks_f32(None, 10, 4, None, 0, None, None, 0, None) == 0            # no questions: nothing to do
    assert h.vbq_uniform_quantize_f32(None, 5, C.c_float(0), C.c_float(0), C.c_float(0), 4, None, None, None, None) == -1
    assert h.vbq_nearest_code_f64(None, 5, None, 0, None, None, None, None) == -1
    assert h.vbq_ssim_scale_workspace_bytes(2, 64, 64, 3, 11) == 2 * 8 * 2 * 3 * 4 * 4
    assert h.vbq_ssim_scale_workspace_bytes(2, 8, 64, 3, 11) == 0                              # window larger than the image
    assert h.vbq_ssim_scale_f64(None, None, 1, 8, 8, 1, None, 12, C.c_double(1), C.c_double(1), None, None, None, 0, None) == -1
    assert b"window" in h.vbq_last_error()
    assert h.vbq_image_sqerr_u8(None, None, -1, 4, None, None) == -1
    assert h.vbq_pack_counts_3x21(None, -1, None, 1, None, None) == -1 and h.vbq_pack_counts_3x21(None, 0, None, 1, None, None) == 0
    assert h.vbq_pack_counts_3x21(None, 3, None, 0, None, None) == -1 and b"n_ranks" in h.vbq_last_error()
    assert h.vbq_allreduce_hist(None, None, -1, 0, None) == -1 and h.vbq_allreduce_hist(None, None, 0, 0, None) == 0
    assert h.vbq_allreduce_hist(None, None, 5, 1, None) == -1 and b"null pointer" in h.vbq_last_error()
    assert h.vbq_comm_init(None, 2, None, 0) == -1 and h.vbq_comm_unique_id(None) == -1 and h.vbq_comm_destroy(None) == 0
    assert h.vbq_level_counts_f32(None, None, 4, 1, 0, None, None, None, 1, 10, None, None, 0, 0, None) == -1
    assert h.vbq_quantize_rows_f32(None, None, 4, 1, 0, None, None, None, 1, 10, 0, None, None, None, None, 0, 3, 2, 0, 0, None) == -1
    assert b"row range" in h.vbq_last_error()
    assert h.vbq_histogram_rows_u16(None, 4, 1, 0, 1, 10, None, 0, 0, 5, None) == -1 and b"row range" in h.vbq_last_error()
    assert h.vbq_code_lengths_from_counts(None, 0, 3, None, 0, 0, None, None, None) == -1
    assert h.vbq_index_max_u16(None, 0, None, None) == 0 and h.vbq_index_max_u16(None, 4, None, None) == -1
    assert h.vbq_downsample2_f64(None, 1, 0, 4, 1, None, None) == -1
    assert h.vbq_transpose_f32(None, 4, 4, None, None) == -1


def test_launch_policy_is_a_per_call_argument_not_process_state(built_lib):
    """"re-entrant (no globals besides the error string)".  The reserved-slot policy of the resident solve grids is an
    argument of vbq_quantize_rows_f32 / vbq_level_counts_f32 / vbq_build_entropy_models_f32 (ABI 5; the process-wide setter of
    ABI 4 is gone) and vbq_solve_grid answers what grid a call takes: a pure function of its arguments -- many threads asking
    with different policies at the same time each get their own answer."""
    import ctypes as C
    import threading
    from vbq_amd import _lib
    h = _lib.lib()
    assert "vbq_set_reserved_workgroups" not in declared_functions() and not hasattr(h, "vbq_set_reserved_workgroups")

    def grid(kernel, rows, n_ch, wg=0, reserved=0):
        g = (C.c_int64 * 3)()
        assert h.vbq_solve_grid(kernel, rows, n_ch, wg, reserved, g) == 0, h.vbq_last_error()
        return tuple(g)

    K1, K1T = 0, 1
    cus = 256 if h.vbq_device_count() == 0 else None        # without a device the library assumes an MI355X (256 CUs)
    if cus:
        # Kodak-24 (36 864 rows x 256 channels): every slot -> 4 resident workgroups per channel; 64 slots reserved -> K1 cannot
        # shrink by whole rounds of 256 without giving up a quarter of the chip and takes short-lived workgroups, K1t drops to 3
        assert grid(K1, 36864, 256, 0, 0) == (4, 256, 1) and grid(K1T, 36864, 256, 0, 0) == (4, 256, 1)
        assert grid(K1, 36864, 256, 0, 64)[2] == 0 and grid(K1T, 36864, 256, 0, 64) == (3, 256, 1)
        # one code book: the resident grid shrinks by the reserved slots, stays resident
        a, b = grid(K1, 10 ** 8, 1, 0, 0), grid(K1, 10 ** 8, 1, 0, 64)
        assert a[2] == b[2] == 1 and a[0] <= 1024 and b[0] <= 960 and b[0] < a[0]
        # WITHOUT a reservation K1t never gives up a workgroup per CU, whatever the channel count
        assert grid(K1T, 36864, 300, 0, 0) == (3, 300, 1) and grid(K1T, 36864, 300, 0, 64) == (2, 300, 1)
        # an explicit workgroups_per_cu wins over the reservation heuristics (the chunked form beside K2)
        assert grid(K1, 36864, 256, 3, 0) == (3, 256, 1)
    assert h.vbq_solve_grid(7, 10, 1, 0, 0, (C.c_int64 * 3)()) == -1 and b"unknown kernel" in h.vbq_last_error()
    assert h.vbq_solve_grid(0, 10, 1, 0, 0, None) == -1
    want = {r: (grid(K1, 36864, 64, 0, r), grid(K1T, 10 ** 7, 1, 0, r)) for r in (0, 8, 64, 200, 700)}
    bad = []

    def worker(r):
        for _ in range(300):
            if (grid(K1, 36864, 64, 0, r), grid(K1T, 10 ** 7, 1, 0, r)) != want[r]:
                bad.append(r)
    ts = [threading.Thread(target=worker, args=(r,)) for r in want for _ in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad
    assert len({want[r] for r in want}) >= 3                   # the policies really differ in the grids they give
