"""Stage-2 dataset of TODA: (adversarial, original) pairs of target-domain frames with intra-domain MixUp
(reference pcdet/datasets/nuscenes/nuscenes_mixup_adv_dataset.py:191-274 adversarial frame, :286-588 sampling policy,
:700-760 prepare_data through intra_domain_point_mixup_cd).

Frames [0, NUM_GT) play the labelled subset (`gt_infos`), the rest the pseudo-labelled subset (`ps_infos`, labels from
PSEUDO_INFO_PATH as written by tools/generate_pseudo_labels[_perturb].py).  Per item:
  with probability 1 - MIXUP_PROB   the same frame twice: a labelled frame (probability GT_PROB) or a pseudo-labelled
                                    frame whose `adv` copy carries the adversarial point edits;
  otherwise                         a MixUp (with collision removal, on the MI355X) of two frames drawn per MIXUP_TYPE
                                    (only_gt | ps_gt | gt_gt+ps | gt+ps_gt+ps), built once for `adv` and once for `org`.
Both halves then pass the DataAugmentor independently (their flips / rotation / scaling are recorded and undone on the
predictions by the consistency loss) and the data processor.

The adversarial edit follows the reference's scheme - for every pseudo box above PSEUDO_THRESH one of modify / add /
remove (np.random.randint(3)) on a random subset of the box's points, with x' = x - eps * g, eps = 1e-3 - where g is the
stored gradient of the detection loss at the point's voxel (`p_voxel_perturb` / `p_voxel_coords` of the infos file).
The reference's lookup helpers live in a module that is not part of the repository (pcdet/utils/perturb_utils.py), so the
voxel lookup here is this build's own: key = linearised (z, y, x) voxel of the point, binary search in the stored keys."""
import numpy as np
import torch

from ..config import AttrDict
from .augmentor import augmentor_utils
from .processor.intra_domain_point_mixup import intra_domain_point_mixup_cd
from .synthetic import SyntheticPairDataset


class SyntheticMixupPairDataset(SyntheticPairDataset):
    def __init__(self, dataset_cfg, class_names, training=True, root_path=None, logger=None):
        super().__init__(dataset_cfg=dataset_cfg, class_names=class_names, training=training, root_path=root_path, logger=logger)
        self.num_gt = int(dataset_cfg.get("SYNTHETIC", AttrDict()).get("NUM_GT", self.num_samples // 2))
        self.mixup_prob = float(dataset_cfg.get("MIXUP_PROB", 0.6))
        self.mixup_type = dataset_cfg.get("MIXUP_TYPE", "gt+ps_gt+ps")
        self.gt_prob = float(dataset_cfg.get("GT_PROB", 0.5))
        self.alpha = float(dataset_cfg.get("ALPHA", 2))
        self.pseudo_thresh = float(dataset_cfg.get("PSEUDO_THRESH", 0.3))
        self.adv_eps = float(dataset_cfg.get("ADV_EPS", 1e-3))
        self.on_device = bool(dataset_cfg.get("MIX_ON_DEVICE", True))

    # ---- frames ---------------------------------------------------------------------------
    def is_gt(self, index):
        return index < self.num_gt

    def frame(self, index, adversarial):
        points, boxes, names = self.raw_sample(index, use_pseudo=not self.is_gt(index))
        points = points.copy()
        if adversarial and not self.is_gt(index) and self.pseudo_infos is not None:
            points = self.adversarial_points(points, self.pseudo_infos[self.frame_id(index)])
        return {"points": points, "gt_boxes": boxes.copy(), "gt_names": names.copy(), "frame_id": self.frame_id(index)}

    def voxel_keys(self, xyz):
        r, v = self.point_cloud_range, np.asarray(self.voxel_size, np.float32)
        cell = np.floor((xyz - r[0:3]) / v).astype(np.int64)                       # (x, y, z) cell
        gx, gy, gz = (int(g) for g in self.grid_size)
        inside = ((cell >= 0) & (cell < np.array([gx, gy, gz]))).all(1)
        return np.where(inside, (cell[:, 2] * gy + cell[:, 1]) * gx + cell[:, 0], -1)

    def adversarial_points(self, points, info):
        """Reference get_ps_adv_lidar_with_sweeps (:191-274) on one cloud."""
        if "p_voxel_perturb" not in info or len(info["gt_boxes"]) == 0:
            return points
        coords = np.asarray(info["p_voxel_coords"], np.int64)                      # [M, 3] (z, y, x)
        grads = np.asarray(info["p_voxel_perturb"], np.float32)[:, :3]
        gx, gy, _ = (int(g) for g in self.grid_size)
        keys = (coords[:, 0] * gy + coords[:, 1]) * gx + coords[:, 2]
        order = np.argsort(keys)
        keys, grads = keys[order], grads[order]
        pkeys = self.voxel_keys(points[:, :3])
        pos = np.clip(np.searchsorted(keys, pkeys), 0, max(len(keys) - 1, 0))
        found = (len(keys) > 0) & (keys[pos] == pkeys) if len(keys) else np.zeros(len(points), bool)
        perturb = np.where(found[:, None], grads[pos], 0.0).astype(np.float32)
        scores = np.asarray(info.get("p_score", np.ones(len(info["gt_boxes"]))))
        removed = []
        for box in np.asarray(info["gt_boxes"], np.float32)[scores > self.pseudo_thresh]:
            p_idx = np.nonzero(augmentor_utils.get_points_in_box(points, box)[1])[0]
            kind = np.random.randint(3)
            if kind in (0, 1) and len(p_idx) > 0:
                moved = points[p_idx, :3] - self.adv_eps * perturb[p_idx]
                k = np.random.randint(len(p_idx))
                pick = np.arange(len(p_idx))
                np.random.shuffle(pick)
                pick = pick[k:]
                if kind == 0:                                                       # modify
                    points[p_idx[pick], :3] = moved[pick]
                else:                                                               # add displaced copies
                    extra = points[p_idx[pick]].copy()
                    extra[:, :3] = moved[pick]
                    points = np.concatenate([points, extra], 0)
                    perturb = np.concatenate([perturb, np.zeros((len(extra), 3), np.float32)], 0)
            elif kind == 2 and len(p_idx) > 5:                                      # remove
                k = np.random.randint(len(p_idx))
                pick = np.arange(len(p_idx))
                np.random.shuffle(pick)
                removed.append(p_idx[pick[k:]])
        if removed:
            points = np.delete(points, np.concatenate(removed), axis=0)
        return points

    # ---- sampling policy --------------------------------------------------------------------
    def draw_mixup_indices(self):
        n, n_gt = self.num_samples, self.num_gt
        if self.mixup_type == "only_gt":
            return np.random.randint(n_gt), np.random.randint(n_gt)
        if self.mixup_type == "ps_gt":
            return n_gt + np.random.randint(n - n_gt), np.random.randint(n_gt)
        if self.mixup_type == "gt_gt+ps":
            return np.random.randint(n_gt), np.random.randint(n)
        if self.mixup_type == "gt+ps_gt+ps":
            return np.random.randint(n), np.random.randint(n)
        raise NotImplementedError(self.mixup_type)

    def _encode(self, data):
        """Class filter + class-id column + feature encoding + upload - what prepare_data does before the mix."""
        keep = np.array([n in self.class_names for n in data["gt_names"]], dtype=bool)
        ids = np.array([self.class_names.index(n) + 1 for n in data["gt_names"][keep]], dtype=np.float32).reshape(-1, 1)
        out = {"points": torch.from_numpy(data["points"]).cuda() if self.on_device else data["points"],
               "gt_boxes": np.concatenate([data["gt_boxes"][keep], ids], axis=1).astype(np.float32), "frame_id": data["frame_id"]}
        return self.point_feature_encoder.forward(out)

    def _finish(self, data):
        """DataAugmentor (records its transforms) on boxes with the class column split off, then the processor queue."""
        if self.training and self.data_augmentor is not None:
            boxes, cls = data["gt_boxes"][:, :7].copy(), data["gt_boxes"][:, 7:]
            aug = self.data_augmentor.forward({"points": data["points"], "gt_boxes": boxes})
            data.update(points=aug["points"], gt_boxes=np.concatenate([aug["gt_boxes"], cls], 1).astype(np.float32),
                        augmentation_list=aug.get("augmentation_list", []), augmentation_params=aug.get("augmentation_params", {}))
        else:
            data.setdefault("augmentation_list", [])
            data.setdefault("augmentation_params", {})
        return self.data_processor.forward(data)

    def __getitem__(self, index):
        index = index % self.num_samples
        if np.random.random(1) > self.mixup_prob or self.mixup_type == "no_mixup":
            if self.mixup_type == "no_mixup" or np.random.random(1) < self.gt_prob:
                src = index % max(self.num_gt, 1)
            else:
                src = self.num_gt + index % max(self.num_samples - self.num_gt, 1)
            adv, org = self._encode(self.frame(src, True)), self._encode(self.frame(src, False))
        else:
            i1, i2 = self.draw_mixup_indices()
            adv = intra_domain_point_mixup_cd(self._encode(self.frame(i1, True)), self._encode(self.frame(i2, True)), alpha=self.alpha)
            org = intra_domain_point_mixup_cd(self._encode(self.frame(i1, False)), self._encode(self.frame(i2, False)), alpha=self.alpha)
        adv, org = self._finish(adv), self._finish(org)
        if self.training and (len(adv["gt_boxes"]) == 0 or len(org["gt_boxes"]) == 0):
            return self[np.random.randint(len(self))]
        return adv, org
