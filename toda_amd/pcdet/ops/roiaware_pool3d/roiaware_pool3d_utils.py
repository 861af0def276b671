"""Point-in-box queries of the reference's pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:9-42 over toda_points_in_boxes
(the RoI-aware pooling itself belongs to the two-stage detectors and is out of scope)."""
import numpy as np
import torch

from toda_amd import ops


def points_in_boxes_gpu(points, boxes):
    """points [B, N, 3], boxes [B, M, 7] (CUDA) -> [B, N] int32: index of the first box holding each point, -1 for none."""
    assert boxes.shape[0] == points.shape[0] and boxes.shape[2] == 7 and points.shape[2] == 3
    out = [ops.points_in_boxes(points[b].contiguous().float(), boxes[b].contiguous().float(), mode=2) for b in range(points.shape[0])]
    return torch.stack(out, 0)


def points_in_boxes_cpu(points, boxes):
    """points [N, 3], boxes [M, 7] (numpy or CPU tensors) -> [M, N] int32 membership matrix (reference :9-25); computed on
    the device, one flag pass per box."""
    is_numpy = isinstance(points, np.ndarray)
    p = torch.as_tensor(points, dtype=torch.float32).cuda().contiguous()
    b = torch.as_tensor(boxes, dtype=torch.float32).cuda().contiguous()
    assert b.shape[1] == 7 and p.shape[1] == 3
    rows = [ops.points_in_boxes(p, b[i:i + 1], mode=0) for i in range(b.shape[0])]
    out = torch.stack(rows, 0).cpu() if rows else torch.zeros((0, p.shape[0]), dtype=torch.int32)
    return out.numpy() if is_numpy else out
