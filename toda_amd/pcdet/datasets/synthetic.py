"""SyntheticLidarDataset: deterministic LiDAR-like clouds of the shapes BASELINE.json names
(SURVEY.md §8 d).  There is no network and no dataset on the GPU box, so every benchmark and test
runs on these; the generator is seeded per (config, sample index) and has nothing random left
after `seed` is fixed.

Point model: azimuth ~ U(-pi, pi); range r = r_min + (r_max - r_min) * u^1.5 (density falls with
distance); 70 % ground (z = z_ground + N(0, 0.05)), 20 % structure on 200 vertical segments,
10 % inside 30 vehicle-sized boxes which are also the gt boxes."""
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

    def __len__(self):
        return self.num_samples

    def raw_sample(self, index):
        kind = self.kinds[index % len(self.kinds)]
        points, boxes, names = synth_cloud(kind, self.seed + index, self.num_points, class_count=len(self.class_names))
        names = np.array([self.class_names[int(n[3:]) - 1] for n in names])
        c = self.point_feature_encoder.num_point_features
        if points.shape[1] < c:
            raise ValueError(f"{kind} clouds have {points.shape[1]} features, config wants {c}")
        return points[:, :len(self.point_feature_encoder.src_feature_list)], boxes, names

    def __getitem__(self, index):
        points, boxes, names = self.raw_sample(index)
        data = {"points": points, "gt_boxes": boxes, "gt_names": names, "frame_id": f"syn_{index:06d}",
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
