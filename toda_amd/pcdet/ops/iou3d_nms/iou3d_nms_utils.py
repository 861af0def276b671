"""Reference pcdet/ops/iou3d_nms/iou3d_nms_utils.py over the C ABI (toda_boxes_iou_bev, toda_boxes_overlap_bev,
toda_nms_rotated).  Boxes are (x, y, z, dx, dy, dz, heading)."""
import numpy as np
import torch

from toda_amd import ops

from ...models.model_utils.model_nms_utils import nms_gpu  # noqa: F401  (reference :84-101)


def boxes_iou_bev(boxes_a, boxes_b):
    """[N, 7] x [M, 7] CUDA tensors -> rotated BEV IoU [N, M] (reference :31-45)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return ops.boxes_iou_bev(boxes_a, boxes_b)


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """The reference's CPU entry point (:12-28): numpy or CPU tensors in, same kind out; computed on the device."""
    is_numpy = isinstance(boxes_a, np.ndarray)
    a = torch.as_tensor(boxes_a, dtype=torch.float32)
    b = torch.as_tensor(boxes_b, dtype=torch.float32)
    assert a.shape[1] == 7 and b.shape[1] == 7
    out = ops.boxes_iou_bev(a.cuda(), b.cuda()).cpu() if a.shape[0] and b.shape[0] else a.new_zeros((a.shape[0], b.shape[0]))
    return out.numpy() if is_numpy else out


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """3-D IoU [N, M]: BEV intersection area x height overlap over the union volume (reference :52-82)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_top, a_bot = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1), (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_top, b_bot = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1), (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    area = ops.boxes_overlap_bev(boxes_a, boxes_b)
    height = torch.clamp(torch.min(a_top, b_top) - torch.max(a_bot, b_bot), min=0)
    inter = area * height
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return inter / torch.clamp(vol_a + vol_b - inter, min=1e-6)
