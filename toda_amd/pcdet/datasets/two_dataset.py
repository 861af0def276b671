"""Two-domain mixing dataset of TODA stage 1 (reference pcdet/datasets/two_dataset.py:164-290 `prepare_data` and the
sampling policy of pcdet/datasets/mix_dataset/waymo_nus_{polarmix,cutmix,lasermix}_dataset.py:154-320).

MI355X layout: both raw clouds are uploaded once and stay in HBM; feature encoding, the mix
(processor/point_mix.py), the range mask and the shuffle all run on the device and the sample leaves
`__getitem__` as a CUDA tensor that `voxelize_on_gpu` consumes - nothing of the cloud returns to the host.
Use with num_workers = 0 (the training process owns the GPU).

Config keys (as in the reference YAMLs, e.g. tools/cfgs/stage1_targetmix/*.yaml):
  MIX_TYPE: polarmix | cutmix | cutpolarmix | lasermix | pseudobbox | pseudobackground     MIX_INC_METHOD: center | corner | corner_del
  POLARMIX_PROB (CUTMIX_PROB for cutmix; the LaserMix and pseudo-mix datasets of the reference read POLARMIX_PROB too,
  waymo_nus_lasermix_dataset.py:153) = probability of a mixed sample, POLARMIX_DEGREE, POLARMIX_RC_NUM, POLARMIX_UPDATE_METHOD,
  POLARMIX_DIS (FULL | RAND), POLARMIX_USE_PITCH, LASERMIX_NUM_AREAS, LASERMIX_NUM_ANGLES (absent: the spherical variant,
  tools/cfgs/stage1_lasermix/*_pp01.yaml), LASERMIX_PITCH_ANGLE
  SYNTHETIC: {SOURCE_KIND, TARGET_KIND, NUM_SOURCE, NUM_TARGET, SEED}   (no real dataset on the box)
"""
import numpy as np
import torch

from ..config import AttrDict
from .dataset import DatasetTemplate
from .processor.inter_domain_point_cutmix import inter_domain_point_cutmix
from .processor.inter_domain_point_lasermix import inter_domain_point_lasermix
from .processor.inter_domain_point_polarmix import inter_domain_point_polarmix
from .processor.inter_domain_point_pseudomix import inter_domain_point_pseudobackground, inter_domain_point_pseudobbox
from .synthetic import synth_cloud


class SyntheticMixDataset(DatasetTemplate):
    def __init__(self, dataset_cfg, class_names, training=True, root_path=None, logger=None):
        super().__init__(dataset_cfg=dataset_cfg, class_names=class_names, training=training, root_path=root_path, logger=logger)
        syn = dataset_cfg.get("SYNTHETIC", AttrDict())
        self.source_kind, self.target_kind = syn.get("SOURCE_KIND", "waymo_toda"), syn.get("TARGET_KIND", "nuscenes_toda")
        self.num_source, self.num_target = int(syn.get("NUM_SOURCE", 32)), int(syn.get("NUM_TARGET", 32))
        self.seed = int(syn.get("SEED", 0))
        self.num_points = {self.source_kind: syn.get("NUM_POINTS_SOURCE", None), self.target_kind: syn.get("NUM_POINTS_TARGET", None)}
        self.on_device = bool(dataset_cfg.get("MIX_ON_DEVICE", True))
        self.mix_type = dataset_cfg.get("MIX_TYPE", "polarmix")
        if self.mix_type not in ("polarmix", "cutmix", "cutpolarmix", "lasermix", "pseudobbox", "pseudobackground"):
            raise NotImplementedError(self.mix_type)
        prob_key = "CUTMIX_PROB" if self.mix_type == "cutmix" else "POLARMIX_PROB"
        if self.mix_type == "lasermix" and "LASERMIX_PROB" in dataset_cfg:       # this repo's earlier spelling
            prob_key = "LASERMIX_PROB"
        self.mix_prob = float(dataset_cfg.get(prob_key, 0.5))
        self.mix_inc_method = dataset_cfg.get("MIX_INC_METHOD", "center")
        self.polarmix_rot_copy_num = int(dataset_cfg.get("POLARMIX_RC_NUM", 1))
        self.polarmix_degree = dataset_cfg.get("POLARMIX_DEGREE", 1.570796)
        self.polarmix_update_method = list(dataset_cfg.get("POLARMIX_UPDATE_METHOD", ["FIX", "FIX", "FIX"]))
        self.polarmix_dis = dataset_cfg.get("POLARMIX_DIS", "FULL")
        self.polarmix_use_pitch = bool(dataset_cfg.get("POLARMIX_USE_PITCH", False))
        # defaults of waymo_nus_lasermix_dataset.py:34-36 (no LASERMIX_NUM_ANGLES: the spherical variant)
        self.laser_pitch_angle = dataset_cfg.get("LASERMIX_PITCH_ANGLE", [-20, 0])
        self.laser_num_areas = dataset_cfg.get("LASERMIX_NUM_AREAS", 3)
        self.laser_num_angles = dataset_cfg.get("LASERMIX_NUM_ANGLES", None)
        # keep generated frames: True = resident on the device, "host" = points in pinned host memory, uploaded at every access
        self.cache_frames = dataset_cfg.get("CACHE_FRAMES", False)
        self._cache = {}
        self.train_percent = 0.0          # the trainer moves it from 0 to 1 (reference train_utils: cur_it / total_it)

    def __len__(self):
        return self.num_source + self.num_target

    def set_train_percent(self, value):
        self.train_percent = float(value)

    # ---- one domain's frame: raw points (+ upload), boxes with the class-id column, encoded features
    def _frame(self, kind, index):
        host = self.cache_frames == "host" and self.on_device
        if self.cache_frames and (kind, index) in self._cache:
            data = dict(self._cache[(kind, index)])
            if host:
                data["points"] = data["points"].cuda(non_blocking=True)      # on the caller's current stream
            return data
        data = self._make_frame(kind, index)
        if self.cache_frames:
            kept = dict(data)
            if host:
                kept["points"] = data["points"].cpu().pin_memory()
            self._cache[(kind, index)] = kept
        return data

    def _make_frame(self, kind, index):
        n_points = self.num_points.get(kind)
        points, boxes, names = synth_cloud(kind, self.seed + index, n_points, class_count=len(self.class_names))
        names = np.array([self.class_names[int(n[3:]) - 1] for n in names])
        points = points[:, :len(self.point_feature_encoder.src_feature_list)]
        keep = np.array([n in self.class_names for n in names], dtype=bool)
        ids = np.array([self.class_names.index(n) + 1 for n in names[keep]], dtype=np.float32).reshape(-1, 1)
        data = {"points": torch.from_numpy(points).cuda() if self.on_device else points,
                "gt_boxes": np.concatenate([boxes[keep], ids], axis=1).astype(np.float32), "frame_id": f"{kind}_{index:06d}"}
        return self.point_feature_encoder.forward(data)

    def mix(self, source, target):
        """The MIX_TYPE switch of the reference's prepare_data (two_dataset.py:227-268)."""
        kind = self.mix_type
        if kind == "cutpolarmix":
            kind = "cutmix" if np.random.random() < 0.5 else "polarmix"
        if kind == "cutmix":
            return inter_domain_point_cutmix(source, target, self.point_cloud_range, self.mix_inc_method)
        if kind == "polarmix":
            return inter_domain_point_polarmix(source, target, self.polarmix_rot_copy_num, self.polarmix_degree, self.train_percent,
                                               self.polarmix_update_method, self.point_cloud_range, self.polarmix_dis,
                                               self.mix_inc_method, self.polarmix_use_pitch)
        if kind == "lasermix":
            return inter_domain_point_lasermix(source, target, self.laser_pitch_angle, self.laser_num_areas, self.laser_num_angles,
                                               self.point_cloud_range, self.mix_inc_method)
        if kind == "pseudobbox":
            return inter_domain_point_pseudobbox(source, target)
        if kind == "pseudobackground":
            return inter_domain_point_pseudobackground(source, target)
        raise NotImplementedError(kind)

    def __getitem__(self, index):
        if np.random.random(1) < self.mix_prob:
            source = self._frame(self.source_kind, index % self.num_source)
            target = self._frame(self.target_kind, 100_000 + index % self.num_target)
            data = self.mix(source, target)
            if data["gt_boxes"].ndim != 2:                               # reference :271-273: draw another sample
                return self[np.random.randint(len(self))]
            data["frame_id"] = f"mix_{index:06d}"
        elif index < self.num_source:
            data = self._frame(self.source_kind, index)
        else:
            data = self._frame(self.target_kind, 100_000 + index - self.num_source)
        data = self.data_processor.forward(data)
        if self.training and len(data["gt_boxes"]) == 0:
            return self[np.random.randint(len(self))]
        return data
