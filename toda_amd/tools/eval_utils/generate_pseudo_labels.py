"""Teacher role of TODA (reference tools/eval_utils/generate_pseudo_labels.py:12-147): run the stage-1 model over the
unlabeled target frames, keep detections above a per-class score threshold and write them into a copy of the frames'
infos pickle as `gt_boxes` / `gt_names` - the file stage 2 trains on."""
import pickle
import time
from pathlib import Path

import numpy as np

from .eval_utils import run_inference


def generate_pseudo_label_samples(unlabel_infos_path, predict_dict, output_infos_path, score_thresh={"car": 0}):
    """infos.pkl wire format: list of dicts with `lidar_path` (frame id = file stem) or `point_cloud.lidar_idx`; the
    entries' gt_boxes [K, 7] / gt_names [K] are replaced by the thresholded predictions, everything else is kept."""
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
            names, boxes = [], []
            for cls, thr in score_thresh.items():
                of_cls = pred["name"] == cls
                keep = pred["score"][of_cls] > thr
                names.append(pred["name"][of_cls][keep])
                boxes.append(pred["boxes_lidar"][of_cls][keep])
            info["gt_names"], info["gt_boxes"] = np.concatenate(names), np.concatenate(boxes)
        else:
            info["gt_names"], info["gt_boxes"] = pred["name"], pred["boxes_lidar"]
        n_boxes += len(info["gt_names"])
    with open(output_infos_path, "wb") as f:
        pickle.dump(infos, f)
    return len(infos), n_boxes


def inference_and_generate_pseudo_labes(cfg, args, model, dataloader, logger, dist_test=False, save_to_file=False,
                                        result_dir=None, unlabel_infos_path=None):
    dataset = dataloader.dataset
    result_dir.mkdir(parents=True, exist_ok=True)
    logger.info("*************** INFERENCING UNLABELD INFOS *****************")
    start = time.time()
    det_annos = []
    for batch_dict, pred_dicts, _ in run_inference(model, dataloader):
        det_annos += dataset.generate_prediction_dicts(batch_dict, pred_dicts, dataset.class_names,
                                                       output_path=result_dir if save_to_file else None)
    if dist_test:
        from toda_amd.pcdet.utils import common_utils
        det_annos = common_utils.merge_results_dist(det_annos, len(dataset))
    logger.info("Generate label finished(sec_per_example: %.4f second)." % ((time.time() - start) / max(len(dataset), 1)))
    if cfg.LOCAL_RANK != 0:
        return None
    n_obj = sum(len(a["name"]) for a in det_annos)
    logger.info("Average predicted number of objects(%d sample): %.3f" % (len(det_annos), n_obj / max(1, len(det_annos))))
    thresh = {name: args.pseudo_thresh for name in dataset.class_names[:1]}       # the reference thresholds 'car' only
    out_path = result_dir / Path(f"score_{args.pseudo_thresh}_{Path(str(unlabel_infos_path)).name}")
    n_infos, n_boxes = generate_pseudo_label_samples(unlabel_infos_path, det_annos, out_path, score_thresh=thresh)
    logger.info(f"Total box num: {n_boxes}; total infos num: {n_infos}; pseudo infos file is saved to {out_path}")
    return out_path
