"""GT-sampling augmentation (reference pcdet/datasets/augmentor/database_sampler.py:13-252): objects cut out of other
frames are pasted into the scene when they collide with nothing.

Same config keys (DB_INFO_PATH, PREPARE {filter_by_min_points, filter_by_difficulty}, SAMPLE_GROUPS, NUM_POINT_FEATURES,
REMOVE_EXTRA_WIDTH, LIMIT_WHOLE_SCENE, DB_DATA_PATH + USE_SHARED_MEMORY), same db_infos.pkl records, same random draws.
MI355X layout: object points live in HBM - the packed database (DB_DATA_PATH, the reference's shared-memory array) is
uploaded once, per-file objects are cached on first use - and a CUDA scene is edited on the device: in-box kernel +
compaction remove the scene points under the (enlarged) pasted boxes, the objects are shifted to their box centres and
concatenated in front, as in the reference.  numpy scenes take the same path and come back as numpy."""
import pickle
from pathlib import Path

import numpy as np
import torch

from ...ops.iou3d_nms import iou3d_nms_utils
from ...utils import box_utils


class DataBaseSampler:
    def __init__(self, root_path, sampler_cfg, class_names, logger=None):
        self.root_path = Path(root_path) if root_path is not None else Path(".")
        self.class_names, self.sampler_cfg, self.logger = list(class_names), sampler_cfg, logger
        self.db_infos = {name: [] for name in self.class_names}
        for rel in sampler_cfg.DB_INFO_PATH:
            with open(str(self.root_path.resolve() / rel), "rb") as f:
                infos = pickle.load(f)
            for name in self.class_names:
                self.db_infos[name].extend(infos.get(name, []))
        for func_name, val in sampler_cfg.get("PREPARE", {}).items():
            self.db_infos = getattr(self, func_name)(self.db_infos, val)
        self.num_features = int(sampler_cfg.NUM_POINT_FEATURES)
        self.use_shared_memory = bool(sampler_cfg.get("USE_SHARED_MEMORY", False))
        self._packed = None            # whole database on the device (the reference's SharedArray)
        self._cache = {}               # per-file objects, uploaded on first use
        self.limit_whole_scene = sampler_cfg.get("LIMIT_WHOLE_SCENE", False)
        self.sample_groups, self.sample_class_num = {}, {}
        for entry in sampler_cfg.SAMPLE_GROUPS:
            name, num = entry.split(":")
            if name not in self.class_names:
                continue
            self.sample_class_num[name] = num
            self.sample_groups[name] = {"sample_num": num, "pointer": len(self.db_infos[name]),
                                        "indices": np.arange(len(self.db_infos[name]))}

    # ---- PREPARE filters (reference :86-112)
    def filter_by_difficulty(self, db_infos, removed_difficulty):
        return {k: [i for i in v if i.get("difficulty", 0) not in removed_difficulty] for k, v in db_infos.items()}

    def filter_by_min_points(self, db_infos, min_gt_points_list):
        for entry in min_gt_points_list:
            name, min_num = entry.split(":")
            if int(min_num) > 0 and name in db_infos:
                db_infos[name] = [i for i in db_infos[name] if i["num_points_in_gt"] >= int(min_num)]
        return db_infos

    def sample_with_fixed_number(self, class_name, sample_group):
        num, pointer, indices = int(sample_group["sample_num"]), sample_group["pointer"], sample_group["indices"]
        if pointer >= len(self.db_infos[class_name]):
            indices = np.random.permutation(len(self.db_infos[class_name]))
            pointer = 0
        picked = [self.db_infos[class_name][i] for i in indices[pointer:pointer + num]]
        sample_group["pointer"], sample_group["indices"] = pointer + num, indices
        return picked

    # ---- object points, resident on the device
    def object_points(self, info):
        if self.use_shared_memory:
            if self._packed is None:
                data = np.load(str(self.root_path.resolve() / self.sampler_cfg.DB_DATA_PATH[0]))
                self._packed = torch.from_numpy(np.ascontiguousarray(data, np.float32)).cuda()
            lo, hi = info["global_data_offset"]
            return self._packed[lo:hi]
        key = info["path"]
        if key not in self._cache:
            pts = np.fromfile(str(self.root_path / key), dtype=np.float32).reshape(-1, self.num_features)
            self._cache[key] = torch.from_numpy(pts).cuda()
        return self._cache[key]

    def add_sampled_boxes_to_scene(self, data_dict, sampled_gt_boxes, picked):
        mask = data_dict["gt_boxes_mask"]
        gt_boxes, gt_names = data_dict["gt_boxes"][mask], data_dict["gt_names"][mask]
        points = data_dict["points"]
        is_numpy = isinstance(points, np.ndarray)
        scene = torch.as_tensor(points, dtype=torch.float32).cuda() if not (torch.is_tensor(points) and points.is_cuda) else points
        objs = []
        for info in picked:
            obj = self.object_points(info).clone()
            obj[:, :3] += torch.from_numpy(np.asarray(info["box3d_lidar"][:3], np.float32)).to(obj.device)
            objs.append(obj)
        large = box_utils.enlarge_box3d(sampled_gt_boxes[:, 0:7], extra_width=self.sampler_cfg.REMOVE_EXTRA_WIDTH)
        scene = box_utils.remove_points_in_boxes3d(scene.contiguous(), large)
        merged = torch.cat(objs + [scene], dim=0)
        data_dict["points"] = merged.cpu().numpy() if is_numpy else merged
        data_dict["gt_names"] = np.concatenate([gt_names, np.array([i["name"] for i in picked])], axis=0)
        data_dict["gt_boxes"] = np.concatenate([gt_boxes, sampled_gt_boxes], axis=0)
        return data_dict

    def __call__(self, data_dict):
        gt_boxes = data_dict["gt_boxes"]
        gt_names = data_dict["gt_names"].astype(str)
        existing = gt_boxes
        picked_all = []
        for name, group in self.sample_groups.items():
            if self.limit_whole_scene:
                group["sample_num"] = str(int(self.sample_class_num[name]) - int(np.sum(name == gt_names)))
            if int(group["sample_num"]) <= 0:
                continue
            picked = self.sample_with_fixed_number(name, group)
            boxes = np.stack([i["box3d_lidar"] for i in picked], axis=0).astype(np.float32)
            if "shift_coor" in data_dict:
                boxes[:, :3] += data_dict["shift_coor"]
            iou_scene = iou3d_nms_utils.boxes_bev_iou_cpu(boxes[:, 0:7], existing[:, 0:7])
            iou_self = iou3d_nms_utils.boxes_bev_iou_cpu(boxes[:, 0:7], boxes[:, 0:7])
            iou_self[range(boxes.shape[0]), range(boxes.shape[0])] = 0
            iou_scene = iou_scene if iou_scene.shape[1] > 0 else iou_self
            valid = ((iou_scene.max(axis=1) + iou_self.max(axis=1)) == 0).nonzero()[0]
            boxes = boxes[valid]
            if existing.shape[1] < boxes.shape[1]:
                boxes = boxes[:, :7]
            existing = np.concatenate((existing, boxes), axis=0)
            picked_all.extend(picked[i] for i in valid)
        if picked_all:
            data_dict = self.add_sampled_boxes_to_scene(data_dict, existing[gt_boxes.shape[0]:, :], picked_all)
        data_dict.pop("gt_boxes_mask")
        return data_dict


def create_groundtruth_database(dataset, save_dir, used_classes=None, packed=True):
    """Cut every labelled object out of the dataset's frames (reference nuscenes_dataset.py:370-412 /
    waymo_dataset.py create_groundtruth_database): per object a `<frame>_<class>_<k>.bin` of its points relative to the
    box centre, `dbinfos.pkl` = {class: [{name, path, image_idx, gt_idx, box3d_lidar, num_points_in_gt, difficulty,
    global_data_offset}]} and, with `packed`, all objects in one `gt_database_global.npy` (the shared-memory layout).
    Membership = index of the first box holding the point (points_in_boxes_gpu)."""
    from ...ops.roiaware_pool3d import roiaware_pool3d_utils

    save_dir = Path(save_dir)
    db_dir = save_dir / "gt_database"
    db_dir.mkdir(parents=True, exist_ok=True)
    all_infos, chunks, offset = {}, [], 0
    for idx in range(len(dataset)):
        points, gt_boxes, gt_names = dataset.raw_sample(idx)
        owner = roiaware_pool3d_utils.points_in_boxes_gpu(torch.from_numpy(points[:, 0:3]).unsqueeze(0).float().cuda(),
                                                          torch.from_numpy(gt_boxes[:, 0:7]).unsqueeze(0).float().cuda())[0].cpu().numpy()
        for i in range(gt_boxes.shape[0]):
            obj = points[owner == i].copy()
            obj[:, :3] -= gt_boxes[i, :3]
            rel = Path("gt_database") / f"{idx}_{gt_names[i]}_{i}.bin"
            obj.astype(np.float32).tofile(str(save_dir / rel))
            if used_classes is None or gt_names[i] in used_classes:
                info = {"name": gt_names[i], "path": str(rel), "image_idx": idx, "gt_idx": i, "box3d_lidar": gt_boxes[i],
                        "num_points_in_gt": obj.shape[0], "difficulty": 0, "global_data_offset": [offset, offset + obj.shape[0]]}
                all_infos.setdefault(gt_names[i], []).append(info)
                chunks.append(obj.astype(np.float32))
                offset += obj.shape[0]
    with open(save_dir / "dbinfos.pkl", "wb") as f:
        pickle.dump(all_infos, f)
    if packed:
        np.save(str(save_dir / "gt_database_global.npy"), np.concatenate(chunks, 0) if chunks else np.zeros((0, dataset.raw_sample(0)[0].shape[1]), np.float32))
    return all_infos
