"""CenterNet helpers (reference pcdet/models/model_utils/centernet_utils.py): gaussian radius
(:9-35) and the top-K heat-map decode (:136-216).  Heat-map drawing lives in the HIP kernel
toda_center_assign; circle-NMS (numba in the reference) is a small torch loop here."""
import torch
import torch.nn.functional as F

from ...utils import loss_utils


def gaussian_radius(height, width, min_overlap=0.5):
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + (b1 ** 2 - 4 * c1).sqrt()) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + (b2 ** 2 - 16 * c2).sqrt()) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + (b3 ** 2 - 4 * a3 * c3).sqrt()) / 2
    return torch.min(torch.min(r1, r2), r3)


def _nms(heat, kernel=3):
    keep = (F.max_pool2d(heat, kernel, stride=1, padding=(kernel - 1) // 2) == heat).float()
    return heat * keep


def _topk(scores, K=40):
    b, c, h, w = scores.shape
    topk_scores, topk_inds = torch.topk(scores.flatten(2, 3), K)
    topk_inds = topk_inds % (h * w)
    topk_ys = (topk_inds // w).float()
    topk_xs = (topk_inds % w).int().float()
    topk_score, topk_ind = torch.topk(topk_scores.view(b, -1), K)
    topk_classes = (topk_ind // K).int()
    pick = lambda t: loss_utils._gather_feat(t.view(b, -1, 1), topk_ind).view(b, K)  # noqa: E731
    return topk_score, pick(topk_inds), topk_classes, pick(topk_ys), pick(topk_xs)


def circle_nms(boxes, min_radius, post_max_size=83):
    """boxes [N, 3] = (x, y, score) sorted by score desc; greedy suppression by centre distance."""
    n = boxes.shape[0]
    if n == 0:
        return boxes.new_zeros((0,), dtype=torch.long)
    d2 = torch.cdist(boxes[:, :2], boxes[:, :2]).pow(2).cpu()
    alive = torch.ones(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        alive &= ~(d2[i] <= min_radius)
        if len(keep) >= post_max_size:
            break
    return torch.as_tensor(keep, dtype=torch.long, device=boxes.device)


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, point_cloud_range=None,
                             voxel_size=None, feature_map_stride=None, vel=None, K=100, circle_nms=False,
                             score_thresh=None, post_center_limit_range=None, padded=False):
    """padded=True (the stage-2 consistency step on the GPU): every sample keeps its K candidates and carries the selection as a
    boolean "mask" instead of being cut down to the selected rows - boolean-mask indexing has to read the row count back to the
    host, i.e. it drains the stream once per sample and pass (four times per stage-2 step)."""
    batch_size = heatmap.shape[0]
    if circle_nms:
        heatmap = _nms(heatmap)
    scores, inds, class_ids, ys, xs = _topk(heatmap, K=K)
    g = lambda t, d: loss_utils._transpose_and_gather_feat(t, inds).view(batch_size, K, d)  # noqa: E731
    center, rot_sin, rot_cos, center_z, dim = g(center, 2), g(rot_sin, 1), g(rot_cos, 1), g(center_z, 1), g(dim, 3)
    angle = torch.atan2(rot_sin, rot_cos)
    xs = (xs.view(batch_size, K, 1) + center[:, :, 0:1]) * feature_map_stride * voxel_size[0] + point_cloud_range[0]
    ys = (ys.view(batch_size, K, 1) + center[:, :, 1:2]) * feature_map_stride * voxel_size[1] + point_cloud_range[1]
    parts = [xs, ys, center_z, dim, angle]
    if vel is not None:
        parts.append(g(vel, 2))
    final_box_preds = torch.cat(parts, dim=-1)
    final_scores = scores.view(batch_size, K)
    final_class_ids = class_ids.view(batch_size, K)

    assert post_center_limit_range is not None
    mask = (final_box_preds[..., :3] >= post_center_limit_range[:3]).all(2)
    mask &= (final_box_preds[..., :3] <= post_center_limit_range[3:]).all(2)
    if score_thresh is not None:
        mask &= final_scores > score_thresh
    out = []
    for k in range(batch_size):
        m = mask[k]
        if padded:
            out.append({"pred_boxes": final_box_preds[k], "pred_scores": final_scores[k], "pred_labels": final_class_ids[k], "mask": m})
            continue
        out.append({"pred_boxes": final_box_preds[k, m], "pred_scores": final_scores[k, m],
                    "pred_labels": final_class_ids[k, m]})
        assert not circle_nms, "circle_nms decode is not wired on this path"
    return out
