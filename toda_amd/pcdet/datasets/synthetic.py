"""SyntheticLidarDataset: deterministic LiDAR-like clouds of the shapes BASELINE.json names
(SURVEY.md §8 d).  There is no network and no dataset on the GPU box, so every benchmark and test
runs on these; the generator is seeded per (config, sample index) and has nothing random left
after `seed` is fixed.

Point model: azimuth ~ U(-pi, pi); range r = r_min + (r_max - r_min) * u^1.5 (density falls with
distance); 70 % ground (z = z_ground + N(0, 0.05)), 20 % structure on 200 vertical segments,
10 % inside 30 vehicle-sized boxes which are also the gt boxes."""
from pathlib import Path

import numpy as np

from ..config import AttrDict
from .dataset import DatasetTemplate

SHAPES = {
    # kind: (n_points, C, r_min, r_max, z_ground, front_only)
    "waymo": (180000, 5, 2.0, 75.0, 0.0, False),
    "nuscenes": (60000, 5, 1.0, 51.0, -1.8, False),
    "nuscenes_toda": (35000, 4, 1.0, 51.0, 0.0, False),
    "waymo_toda": (180000, 4, 2.0, 54.0, 0.0, False),
    "kitti": (20000, 4, 2.0, 69.0, -1.7, True),
}


def synth_cloud(kind, seed, n_points=None, n_boxes=30, class_count=3):
    n_def, c, r_min, r_max, z_ground, front = SHAPES[kind]
    n = int(n_points or n_def)
    rng = np.random.default_rng(seed)
    n_ground, n_struct = int(0.7 * n), int(0.2 * n)
    n_box = n - n_ground - n_struct

    def polar(count):
        theta = rng.uniform(-np.pi / 2, np.pi / 2, count) if front else rng.uniform(-np.pi, np.pi, count)
        r = r_min + (r_max - r_min) * rng.uniform(0, 1, count) ** 1.5
        return r * np.cos(theta), r * np.sin(theta)

    gx, gy = polar(n_ground)
    ground = np.stack([gx, gy, z_ground + rng.normal(0, 0.05, n_ground)], 1)
    sx, sy = polar(200)
    seg = rng.integers(0, 200, n_struct)
    struct = np.stack([sx[seg] + rng.normal(0, 0.08, n_struct), sy[seg] + rng.normal(0, 0.08, n_struct),
                       rng.uniform(z_ground, z_ground + 3.0, n_struct)], 1)
    bx, by = polar(n_boxes)
    dims = np.array([4.6, 2.0, 1.7]) * rng.uniform(0.8, 1.2, (n_boxes, 3))
    yaw = rng.uniform(-np.pi, np.pi, n_boxes)
    boxes = np.concatenate([np.stack([bx, by, z_ground + dims[:, 2] / 2], 1), dims, yaw[:, None]], 1)
    owner = rng.integers(0, n_boxes, n_box)
    local = rng.uniform(-0.5, 0.5, (n_box, 3)) * dims[owner]
    cs, sn = np.cos(yaw[owner]), np.sin(yaw[owner])
    inbox = np.stack([boxes[owner, 0] + cs * local[:, 0] - sn * local[:, 1],
                      boxes[owner, 1] + sn * local[:, 0] + cs * local[:, 1],
                      boxes[owner, 2] + local[:, 2]], 1)
    xyz = np.concatenate([ground, struct, inbox], 0)
    extra = rng.uniform(0, 1, (n, c - 3))
    if kind == "waymo":
        extra[:, 0] = np.tanh(extra[:, 0] * 2)
    if kind == "nuscenes" and c >= 5:
        extra[:, 1] = rng.integers(0, 10, n) * 0.05  # sweep time lag
    points = np.concatenate([xyz, extra], 1).astype(np.float32)
    names = np.array([f"cls{1 + (i % class_count)}" for i in range(n_boxes)])
    return points, boxes.astype(np.float32), names


class SyntheticLidarDataset(DatasetTemplate):
    """dataset_cfg keys: POINT_CLOUD_RANGE, POINT_FEATURE_ENCODING, DATA_PROCESSOR (as the
    reference's dataset YAMLs) + SYNTHETIC: {KIND | KINDS, NUM_SAMPLES, SEED, NUM_POINTS}."""

    def __init__(self, dataset_cfg, class_names, training=True, root_path=None, logger=None):
        super().__init__(dataset_cfg=dataset_cfg, class_names=class_names, training=training, root_path=root_path,
                         logger=logger)
        syn = dataset_cfg.get("SYNTHETIC", AttrDict())
        self.kinds = list(syn.get("KINDS", None) or [syn.get("KIND", "waymo")])
        self.num_samples = int(syn.get("NUM_SAMPLES", 64))
        self.seed = int(syn.get("SEED", 0))
        self.num_points = syn.get("NUM_POINTS", None)
        # pseudo-label round trip (reference nuscenes_dataset.py include_nuscenes_data + INFO_PATH['pseudo']): an infos
        # pickle written by tools/generate_pseudo_labels.py replaces the frames' ground truth
        self.pseudo_infos = None
        if dataset_cfg.get("PSEUDO_INFO_PATH", None):
            import pickle
            with open(dataset_cfg.PSEUDO_INFO_PATH, "rb") as f:
                self.pseudo_infos = {Path(i["lidar_path"]).stem: i for i in pickle.load(f)}
        self._infos = None

    def __len__(self):
        return self.num_samples

    @staticmethod
    def frame_id(index):
        return f"syn_{index:06d}"

    @property
    def infos(self):
        """Per-frame records in the nuScenes infos.pkl shape the reference's pseudo-label writer edits
        (eval_utils/generate_pseudo_labels.py:12-70): lidar_path (stem = frame id), token, gt_boxes [K,7], gt_names."""
        if self._infos is None:
            self._infos = []
            for index in range(self.num_samples):
                _, boxes, names = self.raw_sample(index, labels_only=True)
                self._infos.append({"lidar_path": f"synthetic/{self.frame_id(index)}.bin", "token": self.frame_id(index),
                                    "gt_boxes": boxes, "gt_names": names})
        return self._infos

    def dump_infos(self, path):
        import pickle
        with open(path, "wb") as f:
            pickle.dump(self.infos, f)

    def evaluation(self, det_annos, class_names, **kwargs):
        """Centre-distance matching (<= 2 m in BEV, greedy by score) of predictions against the frames' boxes ->
        recall / precision per class.  Stands in for the KITTI / nuScenes evaluators (out of scope)."""
        gt_by_frame = {Path(i["lidar_path"]).stem: i for i in self.infos}
        tp = {c: 0 for c in class_names}
        n_gt, n_det = dict(tp), dict(tp)
        for anno in det_annos:
            info = gt_by_frame[str(anno["frame_id"])]
            for c in class_names:
                gt = np.asarray(info["gt_boxes"]).reshape(-1, 7)[np.asarray(info["gt_names"]) == c]
                sel = np.asarray(anno["name"]) == c
                det = anno["boxes_lidar"][sel][np.argsort(-anno["score"][sel])]
                n_gt[c] += len(gt)
                n_det[c] += len(det)
                free = np.ones(len(gt), bool)
                for d in det:
                    if not free.any():
                        break
                    dist = np.hypot(gt[:, 0] - d[0], gt[:, 1] - d[1])
                    dist[~free] = np.inf
                    j = int(dist.argmin())
                    if dist[j] <= 2.0:
                        free[j] = False
                        tp[c] += 1
        result = {}
        lines = []
        for c in class_names:
            result[f"{c}/recall_2m"] = tp[c] / max(n_gt[c], 1)
            result[f"{c}/precision_2m"] = tp[c] / max(n_det[c], 1)
            lines.append(f"{c}: recall@2m {result[f'{c}/recall_2m']:.4f} precision@2m {result[f'{c}/precision_2m']:.4f} "
                         f"({tp[c]} TP / {n_gt[c]} gt / {n_det[c]} det)")
        return "\n".join(lines), result

    def raw_sample(self, index, labels_only=False, use_pseudo=True):
        kind = self.kinds[index % len(self.kinds)]
        points, boxes, names = synth_cloud(kind, self.seed + index, self.num_points, class_count=len(self.class_names))
        names = np.array([self.class_names[int(n[3:]) - 1] for n in names])
        if self.pseudo_infos is not None and not labels_only and use_pseudo:
            info = self.pseudo_infos[self.frame_id(index)]
            boxes = np.asarray(info["gt_boxes"], dtype=np.float32).reshape(-1, 7)
            names = np.asarray(info["gt_names"]).reshape(-1)
        c = self.point_feature_encoder.num_point_features
        if points.shape[1] < c:
            raise ValueError(f"{kind} clouds have {points.shape[1]} features, config wants {c}")
        return points[:, :len(self.point_feature_encoder.src_feature_list)], boxes, names

    def __getitem__(self, index):
        points, boxes, names = self.raw_sample(index)
        data = {"points": points, "gt_boxes": boxes, "gt_names": names, "frame_id": self.frame_id(index),
                "_rng": np.random.default_rng(10_000_019 * (self.seed + 1) + index)}
        return self.prepare_data(data)


def _rotate_z(xyz, angle):
    c, s_ = np.cos(angle), np.sin(angle)
    rot = np.array([[c, s_, 0], [-s_, c, 0], [0, 0, 1]], dtype=np.float32)
    return xyz @ rot


class SyntheticPairDataset(SyntheticLidarDataset):
    """Stage-2 shaped items (reference nuscenes_mixup_adv_dataset.py:286-588 returns a pair): the
    same frame twice - `adv`: points nudged by eps * sign-noise (eps = 1e-3, the role of the stored
    voxel perturbation), `org`: the clean frame under a recorded global flip / rotation / scaling,
    which model_fn_decorator_cl undoes on the decoded boxes (reverse_transform)."""

    EPS = 1e-3

    def __getitem__(self, index):
        points, boxes, names = self.raw_sample(index)
        rng = np.random.default_rng(7_000_003 * (self.seed + 1) + index)
        adv_pts = points.copy()
        adv_pts[:, :3] -= self.EPS * np.sign(rng.standard_normal((len(points), 3))).astype(np.float32)
        adv = {"points": adv_pts, "gt_boxes": boxes.copy(), "gt_names": names.copy(), "frame_id": f"syn_{index:06d}_adv",
               "augmentation_list": [], "augmentation_params": {},
               "_rng": np.random.default_rng(10_000_019 * (self.seed + 1) + index)}
        org_pts, org_boxes = points.copy(), boxes.copy()
        aug_list, aug_params = [], {}
        if rng.random() < 0.5:  # flip along x: y -> -y
            org_pts[:, 1] = -org_pts[:, 1]
            org_boxes[:, 1] = -org_boxes[:, 1]
            org_boxes[:, 6] = -org_boxes[:, 6]
            aug_list.append("random_world_flip")
            aug_params["random_world_flip"] = ["x"]
        angle = float(rng.uniform(-0.3925, 0.3925))
        org_pts[:, :3] = _rotate_z(org_pts[:, :3], angle)
        org_boxes[:, :3] = _rotate_z(org_boxes[:, :3], angle)
        org_boxes[:, 6] += angle
        aug_list.append("random_world_rotation")
        aug_params["random_world_rotation"] = angle
        scale = float(rng.uniform(0.95, 1.05))
        org_pts[:, :3] *= scale
        org_boxes[:, :6] *= scale
        aug_list.append("random_world_scaling")
        aug_params["random_world_scaling"] = scale
        org = {"points": org_pts, "gt_boxes": org_boxes, "gt_names": names.copy(), "frame_id": f"syn_{index:06d}_org",
               "augmentation_list": aug_list, "augmentation_params": aug_params,
               "_rng": np.random.default_rng(10_000_019 * (self.seed + 1) + index)}
        return self.prepare_data(adv), self.prepare_data(org)

    @staticmethod
    def collate_batch(batch_list, _unused=False):
        adv = DatasetTemplate.collate_batch([b[0] for b in batch_list])
        org = DatasetTemplate.collate_batch([b[1] for b in batch_list])
        return adv, org
