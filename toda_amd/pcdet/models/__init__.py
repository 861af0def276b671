"""build_network / load_data_to_gpu / model_fn_decorator (reference pcdet/models/__init__.py:16-69).

One addition for the MI355X path: when a batch carries raw `points` but no `voxels`, voxelisation
runs on the GPU here (ops.voxelize_batch) instead of in DataLoader workers on the CPU."""
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    """Every ndarray -> fp32 CUDA tensor, integers included (reference :23-34)."""
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray):
            continue
        if key in ("frame_id", "metadata", "calib"):
            continue
        if key == "image_shape":
            batch_dict[key] = torch.from_numpy(val).int().cuda()
        else:
            batch_dict[key] = torch.from_numpy(val).float().cuda(non_blocking=True)


def voxelize_on_gpu(batch_dict, voxel_cfg):
    """points [sum N, 1 + C] (batch index in column 0) -> voxels / voxel_coords / voxel_num_points.
    voxel_cfg: dict(point_cloud_range, voxel_size, max_points_per_voxel, max_num_voxels)."""
    from ... import ops

    pts = batch_dict["points"]
    bs = int(batch_dict["batch_size"])
    counts = batch_dict.get("points_per_sample")
    if counts is None:
        counts = torch.bincount(pts[:, 0].long(), minlength=bs).tolist()
    clouds, start = [], 0
    for n in counts:
        clouds.append(pts[start:start + int(n), 1:].contiguous())
        start += int(n)
    vox, coords, num = ops.voxelize_batch(clouds, voxel_cfg["point_cloud_range"], voxel_cfg["voxel_size"],
                                          voxel_cfg["max_points_per_voxel"], voxel_cfg["max_num_voxels"])
    batch_dict["voxels"], batch_dict["voxel_coords"], batch_dict["voxel_num_points"] = vox, coords, num
    return batch_dict


ModelReturn = namedtuple("ModelReturn", ["loss", "tb_dict", "disp_dict"])


def model_fn_decorator():
    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        if "voxels" not in batch_dict and "points" in batch_dict:
            net = model.module if hasattr(model, "module") else model
            voxelize_on_gpu(batch_dict, net.dataset.voxel_cfg)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict["loss"].mean()
        (model.module if hasattr(model, "module") and not hasattr(model, "update_global_step") else model).update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func
