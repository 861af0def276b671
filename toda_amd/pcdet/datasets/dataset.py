"""DatasetTemplate (reference pcdet/datasets/dataset.py:13-233): the attributes build_network reads
(class_names, point_feature_encoder, grid_size, point_cloud_range, voxel_size,
depth_downsample_factor), prepare_data and collate_batch."""
from collections import defaultdict

import numpy as np
import torch
import torch.utils.data as torch_data

from .processor.data_processor import DataProcessor
from .processor.point_feature_encoder import PointFeatureEncoder


class DatasetTemplate(torch_data.Dataset):
    def __init__(self, dataset_cfg=None, class_names=None, training=True, root_path=None, logger=None):
        super().__init__()
        self.dataset_cfg = dataset_cfg
        self.training = training
        self.class_names = list(class_names) if class_names is not None else None
        self.logger = logger
        self.root_path = root_path
        if dataset_cfg is None or class_names is None:
            return
        if self.root_path is None and dataset_cfg.get("DATA_PATH", None):      # reference dataset.py:22
            from pathlib import Path
            self.root_path = Path(dataset_cfg.DATA_PATH)
        self.point_cloud_range = np.array(dataset_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        self.point_feature_encoder = PointFeatureEncoder(dataset_cfg.POINT_FEATURE_ENCODING,
                                                         point_cloud_range=self.point_cloud_range)
        self.data_augmentor = None
        if training and dataset_cfg.get("DATA_AUGMENTOR", None) is not None:
            from .augmentor.data_augmentor import DataAugmentor
            self.data_augmentor = DataAugmentor(self.root_path, dataset_cfg.DATA_AUGMENTOR, self.class_names, logger=logger)
        self.data_processor = DataProcessor(dataset_cfg.DATA_PROCESSOR, point_cloud_range=self.point_cloud_range,
                                            training=training,
                                            num_point_features=self.point_feature_encoder.num_point_features)
        self.grid_size = self.data_processor.grid_size
        self.voxel_size = self.data_processor.voxel_size
        self.voxel_cfg = self.data_processor.voxel_cfg
        self.total_epochs = 0
        self._merge_all_iters_to_one_epoch = False
        self.depth_downsample_factor = None

    @property
    def mode(self):
        return "train" if self.training else "test"

    def merge_all_iters_to_one_epoch(self, merge=True, epochs=None):
        self._merge_all_iters_to_one_epoch = merge
        if merge:
            self.total_epochs = epochs

    def prepare_data(self, data_dict):
        """gt filtering by class -> class-id column -> feature encoding -> processor queue."""
        if self.training and self.data_augmentor is not None and data_dict.get("gt_boxes") is not None:
            mask = np.array([n in self.class_names for n in data_dict["gt_names"]], dtype=bool)      # reference :127-135
            data_dict = self.data_augmentor.forward({**data_dict, "gt_boxes_mask": mask})
        if data_dict.get("gt_boxes") is not None:
            names = data_dict["gt_names"]
            keep = np.array([n in self.class_names for n in names], dtype=bool)
            boxes = data_dict["gt_boxes"][keep]
            ids = np.array([self.class_names.index(n) + 1 for n in names[keep]], dtype=np.float32).reshape(-1, 1)
            data_dict["gt_boxes"] = np.concatenate([boxes, ids], axis=1).astype(np.float32)
            data_dict["gt_names"] = names[keep]
        if data_dict.get("points") is not None:
            data_dict = self.point_feature_encoder.forward(data_dict)
        data_dict = self.data_processor.forward(data_dict)
        data_dict.pop("gt_names", None)
        data_dict.pop("_rng", None)
        return data_dict

    def generate_prediction_dicts(self, batch_dict, pred_dicts, class_names, output_path=None):
        """Model output -> per-frame annotation dicts {name, score, boxes_lidar, pred_labels, frame_id} - the record the
        reference's datasets emit (nuscenes_dataset.py:185-230) and generate_pseudo_label_samples consumes."""
        annos = []
        for index, box_dict in enumerate(pred_dicts):
            scores = box_dict["pred_scores"].detach().cpu().numpy()
            boxes = box_dict["pred_boxes"].detach().cpu().numpy()
            labels = box_dict["pred_labels"].detach().cpu().numpy().astype(np.int64)
            n = scores.shape[0]
            anno = {"name": np.array(class_names)[labels - 1] if n else np.zeros(0, dtype="<U1"), "score": scores,
                    "boxes_lidar": boxes[:, :7] if n else np.zeros((0, 7), np.float32), "pred_labels": labels,
                    "frame_id": batch_dict["frame_id"][index]}
            if "metadata" in batch_dict:
                anno["metadata"] = batch_dict["metadata"][index]
            annos.append(anno)
            if output_path is not None:
                np.save(str(output_path / f"{anno['frame_id']}.npy"), np.concatenate([anno["boxes_lidar"], scores[:, None], labels[:, None]], 1))
        return annos

    def evaluation(self, det_annos, class_names, **kwargs):
        """Dataset-specific metrics (KITTI / nuScenes / Waymo evaluators) are out of scope; a dataset with ground truth
        in `self.infos` reports BEV-centre-distance recall / precision here."""
        return "", {}

    @staticmethod
    def collate_batch(batch_list, _unused=False):
        """Concatenate voxels, prepend the batch index to points / voxel_coords, zero-pad gt_boxes
        to [B, max_gt, 8] (reference :161-233)."""
        merged = defaultdict(list)
        for sample in batch_list:
            for key, val in sample.items():
                merged[key].append(val)
        ret = {}
        for key, vals in merged.items():
            if key in ("voxels", "voxel_num_points"):
                ret[key] = np.concatenate(vals, axis=0)
            elif key in ("points", "voxel_coords") and torch.is_tensor(vals[0]):
                # device-resident samples (GPU input pipeline): the batch column is added on the device
                ret[key] = torch.cat([torch.nn.functional.pad(v, (1, 0), value=float(i)) for i, v in enumerate(vals)], dim=0)
                if key == "points":
                    ret["points_per_sample"] = [int(v.shape[0]) for v in vals]
            elif key in ("points", "voxel_coords"):
                ret[key] = np.concatenate(
                    [np.pad(v, ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, v in enumerate(vals)], axis=0)
                if key == "points":
                    ret["points_per_sample"] = [int(v.shape[0]) for v in vals]
            elif key in ("augmentation_list", "augmentation_params"):
                ret[key] = list(vals)  # per-sample python objects (stage-2 consistency step)
            elif key == "gt_boxes":
                width = vals[0].shape[-1]
                out = np.zeros((len(vals), max(len(v) for v in vals), width), dtype=np.float32)
                for k, v in enumerate(vals):
                    out[k, :len(v)] = v
                ret[key] = out
            else:
                ret[key] = np.stack(vals, axis=0)
        ret["batch_size"] = len(batch_list)
        return ret
