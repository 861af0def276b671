"""Box helpers on the anchor-head path (reference pcdet/utils/box_utils.py:255-298)."""
import numpy as np
import torch

from . import common_utils


def boxes_iou_normal(boxes_a, boxes_b):
    """Axis-aligned IoU of [N,4] x [M,4] boxes (x1, y1, x2, y2)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 4
    lo = torch.max(boxes_a[:, None, 0:2], boxes_b[None, :, 0:2])
    hi = torch.min(boxes_a[:, None, 2:4], boxes_b[None, :, 2:4])
    wh = torch.clamp_min(hi - lo, min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (boxes_a[:, 2] - boxes_a[:, 0]) * (boxes_a[:, 3] - boxes_a[:, 1])
    area_b = (boxes_b[:, 2] - boxes_b[:, 0]) * (boxes_b[:, 3] - boxes_b[:, 1])
    return inter / torch.clamp_min(area_a[:, None] + area_b[None, :] - inter, min=1e-6)


def boxes3d_lidar_to_aligned_bev_boxes(boxes3d):
    """Snap headings to the nearest axis: boxes whose |heading mod pi| >= pi/4 swap dx and dy."""
    rot = common_utils.limit_period(boxes3d[:, 6], offset=0.5, period=np.pi).abs()
    dims = torch.where(rot[:, None] < np.pi / 4, boxes3d[:, [3, 4]], boxes3d[:, [4, 3]])
    return torch.cat((boxes3d[:, 0:2] - dims / 2, boxes3d[:, 0:2] + dims / 2), dim=1)


def boxes3d_nearest_bev_iou(boxes_a, boxes_b):
    return boxes_iou_normal(boxes3d_lidar_to_aligned_bev_boxes(boxes_a), boxes3d_lidar_to_aligned_bev_boxes(boxes_b))


_CORNER_SIGNS = ((1, 1, -1), (1, -1, -1), (-1, -1, -1), (-1, 1, -1), (1, 1, 1), (1, -1, 1), (-1, -1, 1), (-1, 1, 1))


def boxes_to_corners_3d(boxes3d):
    """[N, >=7] (x y z dx dy dz heading) -> [N, 8, 3]; corners 0-3 bottom (+x+y, +x-y, -x-y, -x+y), 4-7 top, same
    order as the reference (box_utils.py:28-54).  fp32 torch arithmetic, numpy in -> numpy out."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    half = boxes3d.new_tensor(_CORNER_SIGNS) / 2
    local = boxes3d[:, None, 3:6].repeat(1, 8, 1) * half[None, :, :]
    corners = common_utils.rotate_points_along_z(local.view(-1, 8, 3), boxes3d[:, 6]).view(-1, 8, 3)
    corners += boxes3d[:, None, 0:3]
    return corners.numpy() if is_numpy else corners


def mask_boxes_outside_range_numpy(boxes, limit_range, min_num_corners=1):
    """True for boxes with at least `min_num_corners` of their 8 corners inside [min xyz, max xyz], ends inclusive
    (reference box_utils.py:57-72; all three coordinates are tested)."""
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=bool)
    corners = boxes_to_corners_3d(boxes[:, 0:7])
    lim = np.asarray(limit_range)
    inside = ((corners >= lim[0:3]) & (corners <= lim[3:6])).all(axis=2)
    return inside.sum(axis=1) >= min_num_corners


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """dx, dy, dz grown by extra_width (reference box_utils.py:145-158); returns a torch tensor like the reference."""
    boxes3d, _ = common_utils.check_numpy_to_torch(boxes3d)
    large = boxes3d.clone()
    large[:, 3:6] += boxes3d.new_tensor(extra_width)[None, :]
    return large


def remove_points_in_boxes3d(points, boxes3d):
    """points [N, 3+C] without those inside any of boxes3d [M, 7] (reference box_utils.py:75-89).  CUDA clouds are filtered
    by the in-box kernel + a stable compaction and stay on the device; numpy / CPU clouds take the same route and come back."""
    from toda_amd import ops

    is_numpy = isinstance(points, np.ndarray)
    was_cuda = torch.is_tensor(points) and points.is_cuda
    pts = torch.as_tensor(points, dtype=torch.float32).cuda().contiguous()
    bx = torch.as_tensor(np.asarray(boxes3d.cpu() if torch.is_tensor(boxes3d) else boxes3d), dtype=torch.float32)[:, :7].cuda().contiguous()
    if bx.shape[0] and pts.shape[0]:
        flags = ops.points_in_boxes(pts, bx, mode=0)
        pts = ops.RowBuffer(pts.shape[0], pts.shape[1], pts.device).append(pts, flags, 1, invert=True).finish()
    if was_cuda:
        return pts
    return pts.cpu().numpy() if is_numpy else pts.cpu()
