"""DataProcessor (reference pcdet/datasets/processor/data_processor.py:63-211): the per-sample
queue range-mask -> shuffle -> voxelise, selected by NAME from the DATA_PROCESSOR list.

transform_points_to_voxels has two modes (config key VOXELIZE_ON, default 'gpu_batch'):
  'gpu_batch'  defer to the training process: the model function voxelises the whole batch on the
               MI355X (toda_amd.pcdet.models.voxelize_on_gpu) - no [M,P,C] tensor crosses PCIe;
  'sample'     voxelise here through the spconv-style generator (toda_amd.spconv.utils), which
               also runs on the GPU; use with num_workers=0."""
from functools import partial

import numpy as np
import torch

from ...utils import box_utils, common_utils


def mask_boxes_outside_range(boxes, limit_range, min_num_corners=1):
    return box_utils.mask_boxes_outside_range_numpy(boxes, limit_range, min_num_corners=min_num_corners)


class DataProcessor:
    def __init__(self, processor_configs, point_cloud_range, training, num_point_features):
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.training = training
        self.num_point_features = num_point_features
        self.mode = "train" if training else "test"
        self.grid_size = self.voxel_size = None
        self.voxel_cfg = None
        self.voxel_generator = None
        self.data_processor_queue = [getattr(self, c.NAME)(config=c) for c in processor_configs]

    def mask_points_and_boxes_outside_range(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.mask_points_and_boxes_outside_range, config=config)
        pts = data_dict.get("points")
        if pts is not None and torch.is_tensor(pts) and pts.is_cuda:
            # GPU input pipeline: closed x/y range test + stable compaction on the device (csrc/points.hip)
            from .... import ops
            r = self.point_cloud_range
            keep = ops.points_rect(pts.contiguous(), r[0:2], r[3:5], closed=True)
            data_dict["points"] = ops.RowBuffer(pts.shape[0], pts.shape[1], pts.device).append(pts.contiguous(), keep, 1).finish()
        elif pts is not None:
            keep = common_utils.mask_points_by_range(pts, self.point_cloud_range)
            data_dict["points"] = pts[keep]
        if data_dict.get("gt_boxes") is not None and config.REMOVE_OUTSIDE_BOXES and self.training:
            keep = mask_boxes_outside_range(data_dict["gt_boxes"], self.point_cloud_range,
                                            min_num_corners=config.get("min_num_corners", 1))
            data_dict["gt_boxes"] = data_dict["gt_boxes"][keep]
        return data_dict

    def shuffle_points(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.shuffle_points, config=config)
        if config.SHUFFLE_ENABLED[self.mode]:
            pts = data_dict["points"]
            if config.get("SHUFFLE_ON_DEVICE", False) and torch.is_tensor(pts) and pts.is_cuda:
                # opt-in: draw the permutation on the device (torch's generator instead of numpy's stream; saves the host
                # permutation + its 8 B/point upload, ~0.5 ms per 180k-point cloud)
                data_dict["points"] = pts.index_select(0, torch.randperm(pts.shape[0], device=pts.device))
                return data_dict
            rng = data_dict.get("_rng")
            n = data_dict["points"].shape[0]
            order = rng.permutation(n) if rng is not None else np.random.permutation(n)
            pts = data_dict["points"]
            if torch.is_tensor(pts):      # the permutation is drawn on the host (same numpy stream as the reference)
                data_dict["points"] = pts.index_select(0, torch.from_numpy(order).to(pts.device))
            else:
                data_dict["points"] = pts[order]
        return data_dict

    def _bind_voxel_geometry(self, config):
        extent = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]).astype(np.float64)
        self.grid_size = np.round(extent / np.array(config.VOXEL_SIZE, dtype=np.float64)).astype(np.int64)
        self.voxel_size = list(config.VOXEL_SIZE)
        self.voxel_cfg = {
            "point_cloud_range": [float(v) for v in self.point_cloud_range],
            "voxel_size": [float(v) for v in config.VOXEL_SIZE],
            "max_points_per_voxel": int(config.MAX_POINTS_PER_VOXEL),
            "max_num_voxels": int(config.MAX_NUMBER_OF_VOXELS[self.mode]),
        }

    def transform_points_to_voxels_placeholder(self, data_dict=None, config=None):
        if data_dict is None:
            self._bind_voxel_geometry(config)
            return partial(self.transform_points_to_voxels_placeholder, config=config)
        return data_dict

    def transform_points_to_voxels(self, data_dict=None, config=None):
        if data_dict is None:
            self._bind_voxel_geometry(config)
            return partial(self.transform_points_to_voxels, config=config)
        if config.get("VOXELIZE_ON", "gpu_batch") == "gpu_batch":
            return data_dict
        if self.voxel_generator is None:  # built lazily, as the reference does (pickling)
            from .... import spconv
            self.voxel_generator = spconv.utils.VoxelGenerator(
                voxel_size=self.voxel_cfg["voxel_size"], point_cloud_range=self.voxel_cfg["point_cloud_range"],
                max_num_points=self.voxel_cfg["max_points_per_voxel"], max_voxels=self.voxel_cfg["max_num_voxels"])
        pts = data_dict["points"]
        voxels, coords, num = self.voxel_generator.generate(pts if torch.is_tensor(pts) else np.ascontiguousarray(pts, np.float32))
        if not data_dict["use_lead_xyz"]:
            voxels = voxels[..., 3:]
        data_dict.update(voxels=voxels, voxel_coords=coords, voxel_num_points=num)
        return data_dict

    def forward(self, data_dict):
        for step in self.data_processor_queue:
            data_dict = step(data_dict=data_dict)
        return data_dict
