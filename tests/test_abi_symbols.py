"""The C-ABI library loads on a GPU-less host and exports every symbol include/toda.h declares."""
import os
import re

from toda_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "toda.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(toda_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 20
    lib = L.load()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/toda.h but not exported"
        assert s in L.SIGNATURES, f"{s} has no ctypes prototype in toda_amd/lib.py"
    assert sorted(L.SIGNATURES) == syms
    assert lib.toda_abi_version() == 3


def test_host_side_size_queries_need_no_gpu():
    lib = L.load()
    shape = L.host_i32([41, 1504, 1504])
    nbytes = lib.toda_gridindex_bytes(2, L.hptr(shape))
    cells = (2 * 41 * 1504 * 1504 + 31) // 32
    assert nbytes >= cells * 8
    assert lib.toda_spconv_packed_weight_floats(27, 16, 32) == 27 * 1 * 2 * 256          # fp32 fragment format only
    # a pair the split matrix path covers: the larger of the fp32 (4 B) and the three-plane bf16 (6 B per weight) formats
    assert lib.toda_spconv_packed_weight_floats(27, 64, 64) == 27 * 64 * 64 * 6 // 4
    assert lib.toda_spconv_split_supported(64, 64) == 1 and lib.toda_spconv_split_supported(16, 16) == 0
    assert lib.toda_matrix_path() in (0, 1) and lib.toda_set_matrix_path(7) != 0
    assert lib.toda_spconv_packed_weight_floats(27, 5, 16) == 27 * 256
    assert lib.toda_voxelize_workspace_bytes(180000, 150000) > 180000 * 4 * 5


def test_bad_arguments_fail_loudly_without_touching_the_device():
    lib = L.load()
    rc = lib.toda_spconv_gather_gemm(None, 0, 200, None, None, 0, 27, 16, None, None, None)
    assert rc == -1 and b"channels" in lib.toda_last_error()


def test_more_argument_checks_and_size_queries():
    """Every entry point validates before it launches: error code + message, nothing dereferenced."""
    import ctypes
    lib = L.load()
    assert lib.toda_spconv_wgrad(None, 10, None, None, 10, 27, 64, 200, None, None, 0, None) == -1 and b"channels" in lib.toda_last_error()
    need = lib.toda_spconv_wgrad_workspace_bytes(100000, 27, 64, 64)
    assert need >= 27 * 64 * 64 * 4
    assert lib.toda_spconv_wgrad(None, 10, None, None, 100000, 27, 64, 64, None, None, need - 1, None) != 0 and b"workspace" in lib.toda_last_error()
    assert lib.toda_nms_workspace_bytes(1000) >= 1000 * 16 * 8
    assert lib.toda_nms_rotated(None, 1000, 0.5, None, None, None, 8, None) != 0 and b"workspace" in lib.toda_last_error()
    assert lib.toda_rows_select_workspace_bytes(100000) >= 100000 * 4
    assert lib.toda_rows_select_append(None, 1000, None, 4, None, 1, 0, None, 10, None, None, 16, None) != 0 and b"workspace" in lib.toda_last_error()
    assert lib.toda_rows_moments(None, 10, 24, None, None) == -1 and b"channels" in lib.toda_last_error()           # 24 does not divide 256
    assert lib.toda_rows_reduce_doubles(389533, 64) >= 2 * 64 * 2
    edges = (ctypes.c_double * 40)()
    ep = ctypes.cast(edges, ctypes.c_void_p)
    assert lib.toda_points_polar_cell(None, 10, None, 4, 0.0, ep, 33, ep, 2, 0.0, 1.0, None, None) == -1 and b"bins" in lib.toda_last_error()
    assert lib.toda_points_in_boxes(None, 10, None, 4, None, 3, 7, 5, None, None) == -1 and b"mode" in lib.toda_last_error()
    assert lib.toda_points_in_boxes(None, 10, None, 2, None, 3, 7, 0, None, None) == -1                              # needs x, y, z
    assert lib.toda_timing_begin(0) == -1
    # zero-sized problems succeed without touching any pointer
    assert lib.toda_spconv_gather_gemm_ordered(None, 0, 64, None, None, 0, 27, 64, None, None, None, None) == 0
    assert lib.toda_points_sector(None, 0, None, 4, 0.0, 1.0, None, None) == 0
    assert lib.toda_boxes_overlap_bev(None, 0, None, 5, None, None) == 0


def test_host_tensors_are_refused():
    import pytest
    import torch
    from toda_amd import ops

    with pytest.raises(RuntimeError):
        ops.mean_vfe(torch.zeros(2, 3, 4), torch.ones(2))


def test_round3_entry_points_validate_before_they_launch():
    """The entry points added in round 3 (compacting gather-GEMM, slice BatchNorm2d, optimizer step):
    support queries, workspace sizes, argument errors - no GPU needed."""
    lib = L.load()
    assert lib.toda_spconv_gather_gemm_compact_supported(16, 16, 27) == 1 and lib.toda_spconv_gather_gemm_compact_supported(5, 16, 27) == 1
    assert lib.toda_spconv_gather_gemm_compact_supported(64, 16, 27) == 0 and lib.toda_spconv_gather_gemm_compact_supported(16, 16, 3) == 0
    assert lib.toda_spconv_gather_gemm_compact_supported(16, 18, 27) == 0                                   # produced channels: a multiple of 4
    assert lib.toda_spconv_gather_gemm_compact(None, 10, 64, None, 16, 64, 0, 0, None, 10, 27, 16, None, None, None) == -1 and b"K = 27" in lib.toda_last_error()
    assert lib.toda_spconv_gather_gemm_compact(None, 10, 16, None, 32, 16, 1, 0, None, 10, 27, 16, None, None, None) == -1 and b"does not map" in lib.toda_last_error()
    assert lib.toda_spconv_gather_gemm_compact(None, 10, 16, None, 16, 16, 0, 0, None, 0, 27, 16, None, None, None) == 0       # no output rows: nothing to do
    one = L.host_f32([0.0])
    p = L.hptr(one)
    assert lib.toda_bn2d_fwd_into(p, 2, 64, 100, p, p, None, None, 0.01, 1e-3, 1, p, 96, 40, p, None, 0, None) == -1 and b"outside" in lib.toda_last_error()
    assert lib.toda_bn2d_bwd_from(p, p, 96, 40, 2, 64, 100, p, p, p, 1, p, p, p, None, 0, None) == -1 and b"outside" in lib.toda_last_error()
    assert lib.toda_clip_adam_chunk() == 8192
    assert lib.toda_clip_adam_step(None, None, None, None, None, None, None, 4, None, None, 10.0, 1e-3, 0.9, 0.99, 1e-8, 0.01, 1, None) == -1 and b"null" in lib.toda_last_error()
    assert lib.toda_clip_adam_step(p, p, p, p, p, p, p, 4, p, p, 10.0, 1e-3, 0.9, 0.99, 1e-8, 0.01, 0, None) == -1 and b"step" in lib.toda_last_error()
    assert lib.toda_clip_adam_step(p, p, p, p, p, p, p, 0, p, p, 10.0, 1e-3, 0.9, 0.99, 1e-8, 0.01, 1, None) == 0             # no tensors: nothing to do
