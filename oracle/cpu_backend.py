"""CPU stand-ins for toda_amd.ops built on the oracle (TEST INFRASTRUCTURE ONLY).

`with oracle_backend():` swaps the HIP-backed operators of toda_amd.ops for oracle-backed ones with
the same signatures, so the unmodified model code (toda_amd.pcdet) runs on the CPU.  Used by
  * tests: end-to-end parity of the GPU path against the CPU restatement, and CPU-only plumbing
    tests in the GPU-less build container;
  * bench.py: the `cpu_baseline` leg (kind "port").
The product never imports this module; outside the context manager toda_amd.ops is untouched and
still refuses host tensors.
"""
import contextlib

import numpy as np
import torch

from . import oracle as O


def _np(t):
    return t.detach().cpu().numpy()


class _Rulebook:
    def __init__(self, kind, ksize, n_in, n_out, nbr_fwd, nbr_bwd, flip_bwd, pair_cnt):
        self.kind, self.ksize = kind, ksize
        self.k_vol = int(np.prod(ksize))
        self.n_in, self.n_out = n_in, n_out
        self.nbr_fwd, self.nbr_bwd, self.flip_bwd = nbr_fwd, nbr_bwd, flip_bwd
        self.pair_cnt = torch.from_numpy(pair_cnt)

    def num_pairs(self):
        return int(self.pair_cnt.sum())


def _triple(v):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


def voxelize(points, pc_range, voxel_size, max_pts, max_voxels):
    v, c, n = O.voxelize_hard(_np(points), pc_range, voxel_size, max_pts, max_voxels)
    return torch.from_numpy(v), torch.from_numpy(c), torch.from_numpy(n)


def voxelize_batch(points_list, pc_range, voxel_size, max_pts, max_voxels):
    vox, coords, nums = [], [], []
    for b, p in enumerate(points_list):
        v, c, n = voxelize(p, pc_range, voxel_size, max_pts, max_voxels)
        vox.append(v)
        nums.append(n)
        coords.append(torch.cat([torch.full((len(c), 1), b, dtype=torch.int32), c], 1))
    return torch.cat(vox), torch.cat(coords), torch.cat(nums)


class _MeanVFE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, voxels, num_points):
        num = _np(num_points).astype(np.float32)
        ctx.num, ctx.p = num, voxels.shape[1]
        return torch.from_numpy(O.mean_vfe_fwd(_np(voxels), num))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(O.mean_vfe_bwd(_np(g), ctx.num, ctx.p)), None


def mean_vfe(voxels, num_points):
    return _MeanVFE.apply(voxels, num_points)


def build_subm_rulebook(indices, batch, shape, ksize=3, dilation=1, grid_index=None):
    ks = _triple(ksize)
    nbr, cnt = O.rulebook_subm(_np(indices), batch, shape, ks, _triple(dilation))
    return _Rulebook("subm", ks, nbr.shape[1], nbr.shape[1], nbr, nbr, True, cnt), None


def build_conv_rulebook(indices, batch, shape, ksize, stride, padding):
    ks = _triple(ksize)
    idx_out, sho, o2i, i2o, cnt = O.rulebook_conv(_np(indices), batch, shape, ks, _triple(stride), _triple(padding))
    rb = _Rulebook("conv", ks, i2o.shape[1], o2i.shape[1], o2i, i2o, False, cnt)
    return torch.from_numpy(idx_out), sho, rb, None


def pack_weight(weight, transpose, flip_k):
    return None


class _SparseConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, bias, rb, wp):
        ctx.save_for_backward(features, weight)
        ctx.rb, ctx.has_bias = rb, bias is not None
        out = O.spconv_fwd(_np(features), _np(weight), rb.nbr_fwd, _np(bias) if bias is not None else None)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        features, weight = ctx.saved_tensors
        rb = ctx.rb
        gnp = np.ascontiguousarray(_np(g))
        gf = gw = gb = None
        if ctx.needs_input_grad[0]:
            gf = torch.from_numpy(O.spconv_dgrad(gnp, _np(weight), rb.nbr_bwd, rb.flip_bwd))
        if ctx.needs_input_grad[1]:
            gw = torch.from_numpy(O.spconv_wgrad(_np(features), gnp, rb.nbr_fwd, tuple(weight.shape)))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gf, gw, gb, None, None


def sparse_conv(features, weight, bias, rulebook, packed_weight=None):
    return _SparseConv.apply(features, weight, bias, rulebook, packed_weight)


class _ToDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, batch, shape):
        ctx.idx, ctx.shape = _np(indices), [int(s) for s in shape]
        return torch.from_numpy(O.sparse_to_dense_fwd(_np(features), ctx.idx, batch, ctx.shape))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(O.sparse_to_dense_bwd(np.ascontiguousarray(_np(g)), ctx.idx, ctx.shape)), None, None, None


def sparse_to_dense(features, indices, batch, shape):
    return _ToDense.apply(features, indices, int(batch), shape)


def center_assign(gt_boxes, num_classes, fm_w, fm_h, pc_range, voxel_size, fm_stride, max_objs=500, overlap=0.1,
                  min_radius=2):
    hm, rb, inds, mask = O.center_assign(_np(gt_boxes), num_classes, fm_w, fm_h, pc_range, voxel_size, fm_stride,
                                         max_objs, overlap, min_radius)
    return torch.from_numpy(hm), torch.from_numpy(rb), torch.from_numpy(inds), torch.from_numpy(mask)


_PATCHED = ["voxelize", "voxelize_batch", "mean_vfe", "build_subm_rulebook", "build_conv_rulebook", "pack_weight",
            "sparse_conv", "sparse_to_dense", "center_assign"]


@contextlib.contextmanager
def oracle_backend():
    """Temporarily route toda_amd.ops through the CPU oracle (tests / cpu_baseline only)."""
    from toda_amd import ops

    saved = {name: getattr(ops, name) for name in _PATCHED}
    cuda_saved = torch.Tensor.cuda
    try:
        for name in _PATCHED:
            setattr(ops, name, globals()[name])
        torch.Tensor.cuda = lambda self, *a, **k: self  # load_data_to_gpu -> stay on the host
        yield
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)
        torch.Tensor.cuda = cuda_saved
