"""DataAugmentor (reference pcdet/datasets/augmentor/data_augmentor.py:9-257): a queue of augmentations selected by NAME
from DATA_AUGMENTOR.AUG_CONFIG_LIST.  Built: gt_sampling (database_sampler.py), random_world_flip / random_world_rotation /
random_world_scaling - all on the device for CUDA clouds.  The applied transforms are
recorded in data_dict['augmentation_list' / 'augmentation_params'] - the stage-2 consistency step undoes them on the
decoded boxes (models.reverse_transform)."""
from functools import partial

import numpy as np

from ...utils import common_utils
from . import augmentor_utils


class DataAugmentor:
    def __init__(self, root_path, augmentor_configs, class_names, logger=None):
        self.root_path, self.class_names, self.logger = root_path, class_names, logger
        cfgs = augmentor_configs if isinstance(augmentor_configs, list) else augmentor_configs.AUG_CONFIG_LIST
        disabled = [] if isinstance(augmentor_configs, list) else list(augmentor_configs.get("DISABLE_AUG_LIST", []))
        self.data_augmentor_queue = [getattr(self, c.NAME)(config=c) for c in cfgs if c.NAME not in disabled]

    def gt_sampling(self, config=None):
        from .database_sampler import DataBaseSampler
        return DataBaseSampler(root_path=self.root_path, sampler_cfg=config, class_names=self.class_names, logger=self.logger)

    @staticmethod
    def _record(data_dict, name, value):
        data_dict.setdefault("augmentation_list", []).append(name)
        data_dict.setdefault("augmentation_params", {})[name] = value

    def random_world_flip(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.random_world_flip, config=config)
        gt_boxes, points = data_dict["gt_boxes"], data_dict["points"]
        flipped = []
        for axis in config["ALONG_AXIS_LIST"]:
            assert axis in ("x", "y")
            gt_boxes, points, on = getattr(augmentor_utils, f"random_flip_along_{axis}")(gt_boxes, points, return_flip=True)
            if on:
                flipped.append(axis)
        data_dict["gt_boxes"], data_dict["points"] = gt_boxes, points
        self._record(data_dict, "random_world_flip", flipped)
        return data_dict

    def random_world_rotation(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.random_world_rotation, config=config)
        rot_range = config["WORLD_ROT_ANGLE"]
        if not isinstance(rot_range, list):
            rot_range = [-rot_range, rot_range]
        data_dict["gt_boxes"], data_dict["points"], angle = augmentor_utils.global_rotation(
            data_dict["gt_boxes"], data_dict["points"], rot_range=rot_range, return_rot=True)
        self._record(data_dict, "random_world_rotation", angle)
        return data_dict

    def random_world_scaling(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.random_world_scaling, config=config)
        data_dict["gt_boxes"], data_dict["points"], scale = augmentor_utils.global_scaling(
            data_dict["gt_boxes"], data_dict["points"], config["WORLD_SCALE_RANGE"], return_scale=True)
        self._record(data_dict, "random_world_scaling", scale)
        return data_dict

    def forward(self, data_dict):
        for step in self.data_augmentor_queue:
            data_dict = step(data_dict=data_dict)
        data_dict["gt_boxes"][:, 6] = common_utils.limit_period(data_dict["gt_boxes"][:, 6], offset=0.5, period=2 * np.pi)
        data_dict.pop("calib", None)
        data_dict.pop("road_plane", None)
        if "gt_boxes_mask" in data_dict:
            keep = data_dict.pop("gt_boxes_mask")
            data_dict["gt_boxes"] = data_dict["gt_boxes"][keep]
            data_dict["gt_names"] = data_dict["gt_names"][keep]
        return data_dict
