"""Global augmentations (reference pcdet/datasets/augmentor/augmentor_utils.py:8-81), same names, same random draws.
numpy clouds are transformed on the host with the reference's expressions; CUDA clouds by one fused kernel
(toda_points_world_transform).  The boxes (a few dozen rows) are always handled on the host."""
import numpy as np
import torch

from ...utils import common_utils


def _on_device(points):
    return torch.is_tensor(points) and points.is_cuda


def random_flip_along_x(gt_boxes, points, return_flip=False):
    enable = np.random.choice([False, True], replace=False, p=[0.5, 0.5])
    if enable:
        gt_boxes[:, 1] = -gt_boxes[:, 1]
        gt_boxes[:, 6] = -gt_boxes[:, 6]
        if _on_device(points):
            from .... import ops
            points = ops.points_world_transform(points.contiguous(), flip_x=True)
        else:
            points[:, 1] = -points[:, 1]
        if gt_boxes.shape[1] > 8:
            gt_boxes[:, 8] = -gt_boxes[:, 8]
    return (gt_boxes, points, enable) if return_flip else (gt_boxes, points)


def random_flip_along_y(gt_boxes, points, return_flip=False):
    enable = np.random.choice([False, True], replace=False, p=[0.5, 0.5])
    if enable:
        gt_boxes[:, 0] = -gt_boxes[:, 0]
        gt_boxes[:, 6] = -(gt_boxes[:, 6] + np.pi)
        if _on_device(points):
            from .... import ops
            points = ops.points_world_transform(points.contiguous(), flip_y=True)
        else:
            points[:, 0] = -points[:, 0]
        if gt_boxes.shape[1] > 8:
            gt_boxes[:, 7] = -gt_boxes[:, 7]
    return (gt_boxes, points, enable) if return_flip else (gt_boxes, points)


def global_rotation(gt_boxes, points, rot_range, return_rot=False):
    noise_rotation = np.random.uniform(rot_range[0], rot_range[1])
    angle = np.array([noise_rotation])
    if _on_device(points):
        from .... import ops
        a = torch.from_numpy(angle).float()                       # the reference rotates with fp32 cos / sin of the fp32 angle
        points = ops.points_world_transform(points.contiguous(), rot=(float(torch.cos(a)), float(torch.sin(a))))
    else:
        points = common_utils.rotate_points_along_z(points[np.newaxis, :, :], angle)[0]
    gt_boxes[:, 0:3] = common_utils.rotate_points_along_z(gt_boxes[np.newaxis, :, 0:3], angle)[0]
    gt_boxes[:, 6] += noise_rotation
    if gt_boxes.shape[1] > 8:
        vel = np.hstack((gt_boxes[:, 7:9], np.zeros((gt_boxes.shape[0], 1))))[np.newaxis, :, :]
        gt_boxes[:, 7:9] = common_utils.rotate_points_along_z(vel, angle)[0][:, 0:2]
    return (gt_boxes, points, noise_rotation) if return_rot else (gt_boxes, points)


def global_scaling(gt_boxes, points, scale_range, return_scale=False):
    if scale_range[1] - scale_range[0] < 1e-3:
        return (gt_boxes, points, 1.0) if return_scale else (gt_boxes, points)
    noise_scale = np.random.uniform(scale_range[0], scale_range[1])
    if _on_device(points):
        from .... import ops
        points = ops.points_world_transform(points.contiguous(), scale=np.float32(noise_scale))
    else:
        points[:, :3] *= noise_scale
    gt_boxes[:, :6] *= noise_scale
    return (gt_boxes, points, noise_scale) if return_scale else (gt_boxes, points)


def get_points_in_box(points, gt_box):
    """Points inside one box, margin 0.1 m in x / y, none in z, borders included (reference augmentor_utils.py:474-491).
    Host numpy; the device form of the same test is ops.points_in_boxes(mode=1)."""
    import math
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    cx, cy, cz, dx, dy, dz, rz = (gt_box[i] for i in range(7))
    sx, sy, sz = x - cx, y - cy, z - cz
    cosa, sina = math.cos(-rz), math.sin(-rz)
    lx = sx * cosa + sy * (-sina)
    ly = sx * sina + sy * cosa
    mask = np.logical_and(abs(sz) <= dz / 2.0, np.logical_and(abs(lx) <= dx / 2.0 + 1e-1, abs(ly) <= dy / 2.0 + 1e-1))
    return points[mask], mask
