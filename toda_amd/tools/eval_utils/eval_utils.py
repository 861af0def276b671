"""Evaluation loop (reference tools/eval_utils/eval_utils.py:22-121): eval-mode forward over a dataloader, recall
statistics from the detectors' recall record, per-frame annotation dicts, result.pkl, dataset.evaluation()."""
import pickle
import os
import time

import torch

from toda_amd.pcdet.models import load_data_to_gpu, voxelize_on_gpu
from toda_amd.pcdet.utils import common_utils


def statistics_info(cfg, ret_dict, metric, disp_dict):
    for t in cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST:
        metric[f"recall_roi_{t}"] += ret_dict.get(f"roi_{t}", 0)
        metric[f"recall_rcnn_{t}"] += ret_dict.get(f"rcnn_{t}", 0)
    metric["gt_num"] += ret_dict.get("gt", 0)
    t0 = cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST[0]
    disp_dict[f"recall_{t0}"] = "(%d, %d) / %d" % (metric[f"recall_roi_{t0}"], metric[f"recall_rcnn_{t0}"], metric["gt_num"])


def run_inference(model, dataloader, on_batch=None):
    """Eval-mode forward over the loader; yields (batch_dict, pred_dicts, recall_dict).  Raw-point batches are voxelised
    on the device exactly as in training.  LIFETIME: with the device-side input pipeline the yielded batch_dict's voxels, coordinates
    and neighbour tables (and the indices inside multi_scale_3d_features) are valid until the generator is advanced - a consumer that
    collects batches takes InputPrefetcher.keep(batch_dict) (pcdet.models); pred_dicts are ordinary tensors and stay valid."""
    dataset = dataloader.dataset
    model.eval()
    first = next(model.parameters(), None)
    net = model.module if hasattr(model, "module") and hasattr(model, "no_sync") else model
    if os.environ.get("TODA_PREFETCH", "1") == "1" and first is not None and first.is_cuda and hasattr(net, "dataset"):
        # device-side input pipeline (pcdet.models.InputPrefetcher): the next batch is uploaded, voxelised and indexed on a side
        # stream while this batch's forward and its host-side decoding run
        from toda_amd.pcdet.models import InputPrefetcher
        with torch.no_grad():
            pre = InputPrefetcher(iter(dataloader), net, first.device, eager=False, voxel_cfg=getattr(dataset, "voxel_cfg", None))
        try:
            while True:
                try:
                    with torch.no_grad():
                        batch_dict = pre.next()
                        if pre.threaded:
                            pre.kick()       # the worker thread prepares the next batch while this one is enqueued and decoded
                except StopIteration:
                    return
                with torch.no_grad():
                    pred_dicts, ret_dict = model(batch_dict)
                    pre.kick()
                yield batch_dict, pred_dicts, ret_dict
        finally:
            pre.close()                      # also when the consumer abandons the generator
    for batch_dict in dataloader:
        load_data_to_gpu(batch_dict)
        if "voxels" not in batch_dict and "points" in batch_dict:
            voxelize_on_gpu(batch_dict, dataset.voxel_cfg)
        with torch.no_grad():
            pred_dicts, ret_dict = model(batch_dict)
        yield batch_dict, pred_dicts, ret_dict


def eval_one_epoch(cfg, model, dataloader, epoch_id, logger, dist_test=False, save_to_file=False, result_dir=None):
    result_dir.mkdir(parents=True, exist_ok=True)
    final_output_dir = result_dir / "final_result" / "data"
    if save_to_file:
        final_output_dir.mkdir(parents=True, exist_ok=True)
    thresholds = cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST
    metric = {"gt_num": 0}
    for t in thresholds:
        metric[f"recall_roi_{t}"] = 0
        metric[f"recall_rcnn_{t}"] = 0
    dataset = dataloader.dataset
    class_names = dataset.class_names
    det_annos = []
    logger.info(f"*************** EPOCH {epoch_id} EVALUATION *****************")
    start = time.time()
    for batch_dict, pred_dicts, ret_dict in run_inference(model, dataloader):
        statistics_info(cfg, ret_dict, metric, {})
        det_annos += dataset.generate_prediction_dicts(batch_dict, pred_dicts, class_names,
                                                       output_path=final_output_dir if save_to_file else None)
    if dist_test:
        rank, world = common_utils.get_dist_info()
        det_annos = common_utils.merge_results_dist(det_annos, len(dataset))
        metric = common_utils.merge_results_dist([metric], world)
    logger.info(f"*************** Performance of EPOCH {epoch_id} *****************")
    logger.info("Generate label finished(sec_per_example: %.4f second)." % ((time.time() - start) / max(len(dataset), 1)))
    if cfg.LOCAL_RANK != 0:
        return {}
    if dist_test:
        total = metric[0]
        for other in metric[1:]:
            for key, val in other.items():
                total[key] += val
        metric = total
    ret = {}
    gt_num = metric["gt_num"]
    for t in thresholds:
        ret[f"recall/roi_{t}"] = metric[f"recall_roi_{t}"] / max(gt_num, 1)
        ret[f"recall/rcnn_{t}"] = metric[f"recall_rcnn_{t}"] / max(gt_num, 1)
        logger.info(f"recall_roi_{t}: {ret[f'recall/roi_{t}']:f}")
        logger.info(f"recall_rcnn_{t}: {ret[f'recall/rcnn_{t}']:f}")
    n_obj = sum(len(a["name"]) for a in det_annos)
    logger.info("Average predicted number of objects(%d samples): %.3f" % (len(det_annos), n_obj / max(1, len(det_annos))))
    with open(result_dir / "result.pkl", "wb") as f:
        pickle.dump(det_annos, f)
    result_str, result_dict = dataset.evaluation(det_annos, class_names, eval_metric=cfg.MODEL.POST_PROCESSING.get("EVAL_METRIC", None),
                                                 output_path=final_output_dir)
    logger.info(result_str)
    ret.update(result_dict)
    logger.info(f"Result is save to {result_dir}")
    logger.info("****************Evaluation done.*****************")
    return ret
