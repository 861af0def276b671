"""ctypes binding of libtoda_hip.so (the C ABI of include/toda.h).

There is no CPU fallback: if the shared library is missing, cannot be loaded, or an entry point
reports an error, a RuntimeError is raised.  Tensors are marshalled as raw device pointers and the
current HIP stream; the library never allocates device memory.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TODA_HIP_LIB") or os.path.join(_HERE, "libtoda_hip.so")   # env override: A/B of kernel builds

_vp, _i, _sz, _dbl = C.c_void_p, C.c_int, C.c_size_t, C.c_double

# name -> (restype, argtypes); must list every symbol include/toda.h declares
SIGNATURES = {
    "toda_last_error": (C.c_char_p, []),
    "toda_abi_version": (_i, []),
    "toda_device_fault": (_i, []),
    "toda_voxelize_workspace_bytes": (_sz, [_i, _i]),
    "toda_voxelize_hard": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "toda_mean_vfe_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "toda_mean_vfe_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "toda_gridindex_bytes": (_sz, [_i, _vp]),
    "toda_gridindex_from_coords": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "toda_gridindex_from_conv": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "toda_rulebook_subm": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "toda_rulebook_conv": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "toda_voxelize_batch_workspace_bytes": (_sz, [_i, _i]),
    "toda_voxelize_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "toda_gridindex_from_coords_unordered": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "toda_gridindex_clear": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp]),
    "toda_gridindex_from_bitmap": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "toda_spconv_packed_weight_floats": (_sz, [_i, _i, _i]),
    "toda_matrix_path": (_i, []),
    "toda_set_matrix_path": (_i, [_i]),
    "toda_spconv_split_supported": (_i, [_i, _i]),
    "toda_spconv_pack_weight": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "toda_spconv_pack_weights": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "toda_spconv_gather_gemm": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "toda_spconv_gather_gemm_ordered": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "toda_rulebook_class_order": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "toda_spconv_gather_gemm_classed": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "toda_spconv_gather_gemm_stats_supported": (_i, [_i, _i]),
    "toda_spconv_gather_gemm_stats_doubles": (_sz, [_i, _i]),
    "toda_spconv_gather_gemm_stats": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "toda_spconv_gather_gemm_stats_partials": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "toda_spconv_gather_gemm_compact_supported": (_i, [_i, _i, _i]),
    "toda_spconv_gather_gemm_compact": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "toda_spconv_gather_gemm_compact_stats_doubles": (_sz, [_i, _i]),
    "toda_spconv_gather_gemm_compact_stats": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "toda_spconv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "toda_spconv_wgrad": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "toda_sparse_to_dense_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "toda_sparse_to_dense_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "toda_pillar_scatter_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "toda_pillar_scatter_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "toda_rows_reduce_doubles": (_sz, [_i, _i]),
    "toda_rows_moments": (_i, [_vp, _i, _i, _vp, _vp]),
    "toda_rows_affine_act": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "toda_bn_finalize": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _i, _vp, _vp, _vp, _vp, _vp]),
    "toda_bn_finalize_partials": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "toda_rows_bn_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "toda_rows_bn_bwd_res": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "toda_rows_bn_bwd_colsum_doubles": (_sz, [_i, _i]),
    "toda_rows_bn_bwd_res_colsum": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "toda_conv3x3s2_supported": (_i, [_i, _i, _i, _i, _i]),
    "toda_conv3x3s2_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "toda_conv3x3s2_dgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "toda_conv3x3s2_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "toda_conv3x3s2_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "toda_deconv_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "toda_deconv_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "toda_deconv_dgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "toda_deconv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "toda_deconv_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "toda_bn2d_supported": (_i, [_i, _i, _i]),
    "toda_bn2d_sync_bytes": (_sz, []),
    "toda_bn2d_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _i, _vp, _vp, _vp, C.c_uint, _vp]),
    "toda_bn2d_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, C.c_uint, _vp]),
    "toda_bn2d_fwd_into": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _i, _vp, _i, _i, _vp, _vp, C.c_uint, _vp]),
    "toda_bn2d_bwd_from": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, C.c_uint, _vp]),
    "toda_boxes_iou_bev": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "toda_boxes_overlap_bev": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "toda_nms_workspace_bytes": (_sz, [_i]),
    "toda_nms_rotated": (_i, [_vp, _i, C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "toda_clip_adam_chunk": (_i, []),
    "toda_clip_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _i, _vp]),
    "toda_points_in_boxes": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "toda_points_sector": (_i, [_vp, _i, _vp, _i, _dbl, _dbl, _vp, _vp]),
    "toda_points_rect": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "toda_points_polar_cell": (_i, [_vp, _i, _vp, _i, C.c_float, _vp, _i, _vp, _i, C.c_float, C.c_float, _vp, _vp]),
    "toda_points_polar_select": (_i, [_vp, _i, _vp, _i, _dbl, _dbl, _i, _i, _dbl, _vp, _vp, _vp]),
    "toda_points_pitch_range_workspace_bytes": (_sz, []),
    "toda_points_pitch_range": (_i, [_vp, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "toda_points_pitch_band": (_i, [_vp, _i, _vp, _i, C.c_float, C.c_float, C.c_float, _vp, _i, _vp, _vp]),
    "toda_rows_select_workspace_bytes": (_sz, [_i]),
    "toda_rows_select_append": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "toda_points_rotate_z": (_i, [_vp, _i, _vp, _i, _dbl, _dbl, _vp, _vp]),
    "toda_points_world_transform": (_i, [_vp, _i, _vp, _i, _i, _i, _i, C.c_float, C.c_float, _i, C.c_float, _vp, _vp]),
    "toda_conv3x3_supported": (_i, [_i, _i, _i, _i, _i]),
    "toda_conv3x3_weight_floats": (_sz, [_i, _i]),
    "toda_conv3x3_transform_weight": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "toda_conv3x3_workspace_bytes": (_sz, []),
    "toda_conv3x3_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "toda_conv3x3_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "toda_conv3x3_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "toda_conv3x3_narrow_supported": (_i, [_i, _i, _i, _i, _i]),
    "toda_conv3x3_narrow_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_longlong, _vp, _vp]),
    "toda_conv3x3_narrow_dgrad": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, C.c_longlong, _vp, _vp]),
    "toda_conv3x3_narrow_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "toda_conv3x3_narrow_wgrad": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, C.c_longlong, _vp, _vp, _sz, _vp]),
    "toda_center_loss_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "toda_center_loss_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "toda_center_loss_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, C.c_float, C.c_float, _vp, _vp, _sz, _vp]),
    "toda_timing_begin": (_i, [_i]),
    "toda_timing_end": (_i, [_vp, _i, _vp]),
    "toda_center_assign": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _dbl, _i, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load libtoda_hip.so (once) and attach prototypes.  Raises RuntimeError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  toda_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().toda_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def ptr(t):
    """Device pointer of a CUDA tensor (None -> NULL).  Refuses host tensors loudly."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("toda_amd ops need tensors on the GPU (there is no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError("toda_amd ops need contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Raw handle of torch's current HIP stream on the current device.  (torch.cuda.current_stream() builds a Stream object
    and resolves the device through three Python layers: ~9 us per call, ~200 calls per training step.)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def host_ptrs(tensors):
    """Host array of device pointers (None -> NULL) for the entry points that take several tensors per call."""
    return (C.c_void_p * len(tensors))(*[ptr(t) for t in tensors])


def host_addrs(addrs):
    """Host array of raw device addresses (slices of one tensor)."""
    return (C.c_void_p * len(addrs))(*[int(a) for a in addrs])


def host_i32(vals):
    arr = (C.c_int32 * len(vals))(*[int(v) for v in vals])
    return arr


def host_f32(vals):
    arr = (C.c_float * len(vals))(*[float(v) for v in vals])
    return arr


def host_f64(vals):
    arr = (C.c_double * len(vals))(*[float(v) for v in vals])
    return arr


def hptr(arr):
    return C.cast(arr, C.c_void_p)
