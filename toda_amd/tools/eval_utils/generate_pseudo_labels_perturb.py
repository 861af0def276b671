"""Pseudo labels + stored adversarial direction (reference tools/eval_utils/generate_pseudo_labels_perturb.py:100-194):
per unlabeled frame, (1) eval-mode detections, (2) a train-mode pass with the BatchNorm layers frozen whose loss -
against the frame's own thresholded detections - is back-propagated to the voxel tensor (MeanVFE backward runs in
toda_mean_vfe_bwd), (3) the per-voxel xyz gradient and the voxel coordinates are stored next to the pseudo boxes and
their scores: infos keys gt_boxes, gt_names, p_score, p_voxel_perturb [M, 3], p_voxel_coords [M, 3] (z, y, x).
Stage 2's dataset (SyntheticMixupPairDataset.adversarial_points) moves points against that gradient."""
import pickle
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

from toda_amd.pcdet.models import load_data_to_gpu, voxelize_on_gpu


def generate_pseudo_label_samples(unlabel_infos_path, predict_dict, output_infos_path, score_thresh={"car": 0}):
    with open(unlabel_infos_path, "rb") as f:
        infos = pickle.load(f)
    by_frame = {str(p["frame_id"]): p for p in predict_dict}
    n_boxes = 0
    for info in infos:
        info.pop("gt_boxes", None)
        info.pop("gt_names", None)
        key = Path(info["lidar_path"]).stem if "lidar_path" in info else info["point_cloud"]["lidar_idx"]
        pred = by_frame[str(key)]
        if score_thresh is not None:
            names, boxes, scores = [], [], []
            for cls, thr in score_thresh.items():
                of_cls = pred["name"] == cls
                keep = pred["score"][of_cls] > thr
                names.append(pred["name"][of_cls][keep])
                boxes.append(pred["boxes_lidar"][of_cls][keep])
                scores.append(pred["score"][of_cls][keep])
            info["gt_names"], info["gt_boxes"], info["p_score"] = np.concatenate(names), np.concatenate(boxes), np.concatenate(scores)
        else:
            info["gt_names"], info["gt_boxes"], info["p_score"] = pred["name"], pred["boxes_lidar"], pred["score"]
        info["p_voxel_perturb"], info["p_voxel_coords"] = pred["p_voxel_perturb"], pred["p_voxel_coords"]
        n_boxes += len(info["gt_names"])
    with open(output_infos_path, "wb") as f:
        pickle.dump(infos, f)
    return len(infos), n_boxes


def _freeze_batchnorm(model):
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.training = False


def voxel_gradients(model, batch_points, pred_dicts, dataset, score_thresh):
    """d loss / d voxels of one batch, reduced to one xyz vector per voxel -> per-sample (coords zyx, grad) lists."""
    bs = int(batch_points["batch_size"])
    k_max = max(1, max(int((p["pred_scores"] > score_thresh).sum()) for p in pred_dicts))
    gt = torch.zeros((bs, k_max, 8), device="cuda")
    for b, p in enumerate(pred_dicts):
        keep = p["pred_scores"] > score_thresh
        n = int(keep.sum())
        gt[b, :n, :7] = p["pred_boxes"][keep][:, :7]
        gt[b, :n, 7] = p["pred_labels"][keep].float()
    batch = {"points": batch_points["points"], "points_per_sample": batch_points.get("points_per_sample"), "batch_size": bs, "gt_boxes": gt}
    voxelize_on_gpu(batch, dataset.voxel_cfg)
    batch["voxels"].requires_grad_(True)
    voxels, coords, num = batch["voxels"], batch["voxel_coords"], batch["voxel_num_points"]
    model.train()
    _freeze_batchnorm(model)
    model.zero_grad(set_to_none=True)
    ret, _, _ = model(batch)
    ret["loss"].mean().backward()
    g = voxels.grad                                                     # [M, P, C]: every occupied slot of a voxel carries grad/num
    per_voxel = (g[:, :, :3].sum(1) / num.clamp(min=1).view(-1, 1).float()).detach()
    model.zero_grad(set_to_none=True)
    out = []
    for b in range(bs):
        sel = coords[:, 0] == b
        out.append((coords[sel][:, 1:4].int().cpu().numpy(), per_voxel[sel].cpu().numpy().astype(np.float32)))
    return out


def inference_and_generate_pseudo_labes(cfg, args, model, dataloader, logger, dist_test=False, save_to_file=False, result_dir=None,
                                        unlabel_infos_path=None, optimizer=None):
    dataset = dataloader.dataset
    result_dir.mkdir(parents=True, exist_ok=True)
    logger.info("*************** INFERENCING UNLABELD INFOS (with voxel perturbations) *****************")
    start = time.time()
    det_annos = []
    for batch_dict in dataloader:
        load_data_to_gpu(batch_dict)
        raw = {k: batch_dict[k] for k in ("points", "points_per_sample", "batch_size") if k in batch_dict}
        voxelize_on_gpu(batch_dict, dataset.voxel_cfg)
        model.eval()
        with torch.no_grad():
            pred_dicts, _ = model(batch_dict)
        annos = dataset.generate_prediction_dicts(batch_dict, pred_dicts, dataset.class_names)
        for anno, (coords, grad) in zip(annos, voxel_gradients(model, raw, pred_dicts, dataset, args.pseudo_thresh)):
            anno["p_voxel_coords"], anno["p_voxel_perturb"] = coords, grad
        det_annos += annos
    model.eval()
    if dist_test:
        from toda_amd.pcdet.utils import common_utils
        det_annos = common_utils.merge_results_dist(det_annos, len(dataset))
    logger.info("Generate label finished(sec_per_example: %.4f second)." % ((time.time() - start) / max(len(dataset), 1)))
    if cfg.LOCAL_RANK != 0:
        return None
    thresh = {name: args.pseudo_thresh for name in dataset.class_names}       # the reference thresholds car / pedestrian / bicycle alike
    out_path = result_dir / Path(f"score_{args.pseudo_thresh}_{Path(str(unlabel_infos_path)).name}")
    n_infos, n_boxes = generate_pseudo_label_samples(unlabel_infos_path, det_annos, out_path, score_thresh=thresh)
    logger.info(f"Total box num: {n_boxes}; total infos num: {n_infos}; pseudo infos file is saved to {out_path}")
    return out_path
