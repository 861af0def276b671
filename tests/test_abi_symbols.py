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
    assert lib.toda_abi_version() == 1


def test_host_side_size_queries_need_no_gpu():
    lib = L.load()
    shape = L.host_i32([41, 1504, 1504])
    nbytes = lib.toda_gridindex_bytes(2, L.hptr(shape))
    cells = (2 * 41 * 1504 * 1504 + 31) // 32
    assert nbytes >= cells * 8
    assert lib.toda_spconv_packed_weight_floats(27, 64, 64) == 27 * 4 * 4 * 256
    assert lib.toda_spconv_packed_weight_floats(27, 5, 16) == 27 * 256
    assert lib.toda_voxelize_workspace_bytes(180000, 150000) > 180000 * 4 * 5


def test_bad_arguments_fail_loudly_without_touching_the_device():
    lib = L.load()
    rc = lib.toda_spconv_gather_gemm(None, 0, 200, None, None, 0, 27, 16, None, None, None)
    assert rc == -1 and b"channels" in lib.toda_last_error()


def test_host_tensors_are_refused():
    import pytest
    import torch
    from toda_amd import ops

    with pytest.raises(RuntimeError):
        ops.mean_vfe(torch.zeros(2, 3, 4), torch.ones(2))
