"""numpy front-end of the CPU oracle (oracle/toda_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (toda_amd) never imports it.
Each function mirrors one entry point of include/toda.h on host arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtoda_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "toda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def set_threads(n):
    """Pin the oracle's OpenMP thread count; returns the count in effect."""
    return int(lib().oracle_set_threads(int(n)))


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def grid_size(pc_range, voxel_size):
    """data_processor.py:117-118: np.round((hi - lo) / voxel).astype(int64), xyz order."""
    r = np.asarray(pc_range, dtype=np.float64)
    return np.round((r[3:6] - r[0:3]) / np.asarray(voxel_size, dtype=np.float64)).astype(np.int64)


def voxelize_hard(points, pc_range, voxel_size, max_pts, max_voxels):
    pts = _f32(points)
    n, c = pts.shape
    rng = _f32(pc_range)
    vs = _f32(voxel_size)
    grid = _i32(grid_size(pc_range, voxel_size))
    voxels = np.empty((max_voxels, max_pts, c), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    num = np.empty((max_voxels,), np.int32)
    m = C.c_int32(0)
    rc = lib().oracle_voxelize_hard(_p(pts), n, c, _p(rng), _p(vs), _p(grid), int(max_pts),
                                    int(max_voxels), _p(voxels), _p(coords), _p(num), C.byref(m))
    assert rc == 0
    m = m.value
    return voxels[:m].copy(), coords[:m].copy(), num[:m].copy()


def mean_vfe_fwd(voxels, num_pts):
    v = _f32(voxels)
    m, p, c = v.shape
    out = np.empty((m, c), np.float32)
    lib().oracle_mean_vfe_fwd(_p(v), _p(_f32(num_pts)), m, p, c, _p(out))
    return out


def mean_vfe_bwd(gout, num_pts, p):
    g = _f32(gout)
    m, c = g.shape
    gv = np.empty((m, p, c), np.float32)
    lib().oracle_mean_vfe_bwd(_p(g), _p(_f32(num_pts)), m, p, c, _p(gv))
    return gv


def _k3(v):
    v = [v] * 3 if np.isscalar(v) else list(v)
    return _i32(v)


def conv_out_shape(shape_in, ksize, stride, pad):
    ks, st, pd = _k3(ksize), _k3(stride), _k3(pad)
    return [int((int(s) + 2 * int(p) - int(k)) // int(t) + 1) for s, k, t, p in zip(shape_in, ks, st, pd)]


def rulebook_subm(idx, batch, shape, ksize=3, dilation=1):
    idx = _i32(idx)
    n = idx.shape[0]
    ks, dl, sh = _k3(ksize), _k3(dilation), _i32(shape)
    K = int(ks.prod())
    nbr = np.empty((K, n), np.int32)
    cnt = np.zeros((K,), np.int32)
    rc = lib().oracle_rulebook_subm(_p(idx), n, int(batch), _p(sh), _p(ks), _p(dl), _p(nbr), _p(cnt))
    assert rc == 0
    return nbr, cnt


def rulebook_conv(idx_in, batch, shape_in, ksize, stride, pad):
    idx_in = _i32(idx_in)
    n_in = idx_in.shape[0]
    ks, st, pd, shi = _k3(ksize), _k3(stride), _k3(pad), _i32(shape_in)
    sho = _i32(conv_out_shape(shape_in, ks, st, pd))
    K = int(ks.prod())
    cap = max(1, n_in * K)
    idx_out = np.empty((cap, 4), np.int32)
    n_out = lib().oracle_conv_out_indices(_p(idx_in), n_in, int(batch), _p(shi), _p(ks), _p(st), _p(pd),
                                          _p(sho), _p(idx_out), cap)
    assert n_out >= 0
    idx_out = idx_out[:n_out].copy()
    o2i = np.empty((K, n_out), np.int32)
    i2o = np.empty((K, n_in), np.int32)
    cnt = np.zeros((K,), np.int32)
    rc = lib().oracle_rulebook_conv(_p(idx_in), n_in, int(batch), _p(shi), _p(ks), _p(st), _p(pd), _p(sho),
                                    _p(idx_out), n_out, _p(o2i), _p(i2o), _p(cnt))
    assert rc == 0
    return idx_out, [int(v) for v in sho], o2i, i2o, cnt


def spconv_fwd(feat, weight, nbr, bias=None):
    """weight [Cout, kz, ky, kx, Cin]; nbr [K, n_out]."""
    x = _f32(feat)
    w = _f32(weight)
    cout, cin = w.shape[0], w.shape[-1]
    K, n_out = nbr.shape
    w = w.reshape(cout, K, cin)
    out = np.empty((n_out, cout), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().oracle_spconv_fwd(_p(x), cin, _p(w), _p(_i32(nbr)), n_out, K, cout, _p(b), _p(out))
    return out


def spconv_dgrad(dout, weight, nbr_i2o, flip_k):
    g = _f32(dout)
    w = _f32(weight)
    cout, cin = w.shape[0], w.shape[-1]
    K, n_in = nbr_i2o.shape
    w = w.reshape(cout, K, cin)
    din = np.empty((n_in, cin), np.float32)
    lib().oracle_spconv_dgrad(_p(g), cout, _p(w), _p(_i32(nbr_i2o)), n_in, K, cin, int(bool(flip_k)), _p(din))
    return din


def spconv_wgrad(feat, dout, nbr, wshape):
    x, g = _f32(feat), _f32(dout)
    cout, cin = wshape[0], wshape[-1]
    K, n_out = nbr.shape
    dw = np.empty((cout, K, cin), np.float32)
    lib().oracle_spconv_wgrad(_p(x), _p(g), _p(_i32(nbr)), n_out, K, cin, cout, _p(dw))
    return dw.reshape(wshape)


def sparse_to_dense_fwd(feat, idx, batch, shape):
    x = _f32(feat)
    n, c = x.shape
    dense = np.empty((batch, c) + tuple(int(s) for s in shape), np.float32)
    lib().oracle_sparse_to_dense_fwd(_p(x), _p(_i32(idx)), n, c, int(batch), _p(_i32(shape)), _p(dense))
    return dense


def sparse_to_dense_bwd(gdense, idx, shape):
    g = _f32(gdense)
    batch, c = g.shape[:2]
    idx = _i32(idx)
    n = idx.shape[0]
    gf = np.empty((n, c), np.float32)
    lib().oracle_sparse_to_dense_bwd(_p(g), _p(idx), n, c, int(batch), _p(_i32(shape)), _p(gf))
    return gf


def pillar_scatter_fwd(feat, idx, batch, ny, nx):
    x = _f32(feat)
    n, c = x.shape
    canvas = np.empty((batch, c, ny, nx), np.float32)
    lib().oracle_pillar_scatter_fwd(_p(x), _p(_i32(idx)), n, c, int(batch), int(ny), int(nx), _p(canvas))
    return canvas


def rows_moments(x):
    x = _f32(x)
    n, c = x.shape
    s = np.empty((2 * c,), np.float64)
    lib().oracle_rows_moments(_p(x), n, c, _p(s))
    return s


def rows_affine_act(x, scale, shift, residual=None, relu=True):
    x = _f32(x)
    n, c = x.shape
    y = np.empty_like(x)
    r = _f32(residual) if residual is not None else None
    lib().oracle_rows_affine_act(_p(x), _p(_f32(scale)), _p(_f32(shift)), _p(r), n, c, int(bool(relu)), _p(y))
    return y


def center_assign(gt_boxes, num_classes, fm_w, fm_h, pc_range, voxel_size, fm_stride, max_objs=500,
                  overlap=0.1, min_radius=2):
    gt = _f32(gt_boxes)
    batch, n_gt, code = gt.shape
    hm = np.empty((batch, num_classes, fm_h, fm_w), np.float32)
    rb = np.empty((batch, max_objs, code), np.float32)
    inds = np.empty((batch, max_objs), np.int64)
    mask = np.empty((batch, max_objs), np.int64)
    lib().oracle_center_assign.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib().oracle_center_assign(_p(gt), batch, n_gt, code, int(num_classes), int(fm_w), int(fm_h),
                               _p(_f32(pc_range)), _p(_f32(voxel_size)), int(fm_stride), int(max_objs),
                               float(overlap), int(min_radius), _p(hm), _p(rb), _p(inds), _p(mask))
    return hm, rb, inds, mask


def boxes_iou_bev(boxes_a, boxes_b):
    a, b = _f32(boxes_a)[:, :7].copy(), _f32(boxes_b)[:, :7].copy()
    iou = np.empty((len(a), len(b)), np.float32)
    lib().oracle_boxes_iou_bev(_p(a), len(a), _p(b), len(b), _p(iou))
    return iou


def nms_rotated(boxes_sorted, thresh):
    """Greedy NMS over boxes already sorted by descending score -> kept indices (int64)."""
    b = _f32(boxes_sorted)[:, :7].copy()
    keep = np.empty((len(b),), np.int64)
    lib().oracle_nms_rotated.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    n = lib().oracle_nms_rotated(_p(b), len(b), float(thresh), _p(keep))
    return keep[:n].copy()


def points_in_boxes(points, boxes, mode=0):
    """[k, n] int32 membership matrix (mode 0: roiaware points_in_boxes_cpu, mode 1: get_points_in_box)."""
    pts = _f32(points)
    bx = _f32(boxes)
    if bx.ndim != 2 or bx.shape[0] == 0:
        return np.zeros((0, pts.shape[0]), np.int32)
    out = np.empty((bx.shape[0], pts.shape[0]), np.int32)
    lib().oracle_points_in_boxes(_p(pts), pts.shape[0], pts.shape[1], _p(bx), bx.shape[0], bx.shape[1], int(mode), _p(out))
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    a, b = _f32(boxes_a)[:, :7].copy(), _f32(boxes_b)[:, :7].copy()
    out = np.zeros((len(a), len(b)), np.float32)
    if len(a) and len(b):
        lib().oracle_boxes_overlap_bev(_p(a), len(a), _p(b), len(b), _p(out))
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """3-D IoU = BEV overlap area x height overlap / union volume (reference iou3d_nms_utils.py:52-82), fp32."""
    a, b = _f32(boxes_a)[:, :7], _f32(boxes_b)[:, :7]
    ov = boxes_overlap_bev(a, b)
    a_hi, a_lo = (a[:, 2] + a[:, 5] / 2)[:, None], (a[:, 2] - a[:, 5] / 2)[:, None]
    b_hi, b_lo = (b[:, 2] + b[:, 5] / 2)[None, :], (b[:, 2] - b[:, 5] / 2)[None, :]
    oh = np.clip(np.minimum(a_hi, b_hi) - np.maximum(a_lo, b_lo), 0, None).astype(np.float32)
    o3 = ov * oh
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return o3 / np.clip(va + vb - o3, 1e-6, None)
