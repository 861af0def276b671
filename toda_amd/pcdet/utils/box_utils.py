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
