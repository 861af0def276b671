"""NMS wrappers (reference pcdet/models/model_utils/model_nms_utils.py:6-25 and
pcdet/ops/iou3d_nms/iou3d_nms_utils.py:84-101) over toda_nms_rotated."""
import torch

from toda_amd import ops


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """boxes [N, 7], scores [N] -> indices of the kept boxes in descending score order."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep, n_keep = ops.nms_rotated(boxes[order], thresh)
    return order[keep[:int(n_keep.item())]].contiguous(), None


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    src_scores = box_scores
    if score_thresh is not None:
        scores_mask = box_scores >= score_thresh
        box_scores, box_preds = box_scores[scores_mask], box_preds[scores_mask]
    selected = box_scores.new_zeros((0,), dtype=torch.long)
    if box_scores.shape[0] > 0:
        top_scores, indices = torch.topk(box_scores, k=min(nms_config.NMS_PRE_MAXSIZE, box_scores.shape[0]))
        if nms_config.NMS_TYPE != "nms_gpu":
            raise NotImplementedError(f"NMS_TYPE {nms_config.NMS_TYPE}: only nms_gpu (rotated BEV) is on this path")
        keep_idx, _ = nms_gpu(box_preds[indices][:, 0:7], top_scores, nms_config.NMS_THRESH)
        selected = indices[keep_idx[:nms_config.NMS_POST_MAXSIZE]]
    if score_thresh is not None:
        selected = scores_mask.nonzero().view(-1)[selected]
    return selected, src_scores[selected]
