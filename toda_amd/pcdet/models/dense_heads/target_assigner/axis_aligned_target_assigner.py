"""AxisAlignedTargetAssigner (reference target_assigner/axis_aligned_target_assigner.py:8-210):
per class, anchors are matched to that class's gt boxes by nearest-axis BEV IoU; anchors above
matched_threshold (and each gt's best anchor) become positives, below unmatched_threshold
negatives, the rest are ignored (-1).  Regression targets come from the box coder."""
import numpy as np
import torch

from ....utils import box_utils


class AxisAlignedTargetAssigner:
    def __init__(self, model_cfg, class_names, box_coder, match_height=False):
        gen_cfg = model_cfg.ANCHOR_GENERATOR_CONFIG
        tcfg = model_cfg.TARGET_ASSIGNER_CONFIG
        self.box_coder = box_coder
        if match_height:
            raise NotImplementedError("MATCH_HEIGHT needs rotated 3-D IoU (iou3d_nms): out of scope, DESIGN.md")
        self.class_names = np.array(class_names)
        self.anchor_class_names = [c["class_name"] for c in gen_cfg]
        self.pos_fraction = tcfg.POS_FRACTION if tcfg.POS_FRACTION >= 0 else None
        self.sample_size = tcfg.SAMPLE_SIZE
        self.norm_by_num_examples = tcfg.NORM_BY_NUM_EXAMPLES
        self.matched_thresholds = {c["class_name"]: c["matched_threshold"] for c in gen_cfg}
        self.unmatched_thresholds = {c["class_name"]: c["unmatched_threshold"] for c in gen_cfg}
        self.use_multihead = model_cfg.get("USE_MULTIHEAD", False)
        if self.use_multihead:
            raise NotImplementedError("AnchorHeadMulti is not on the named path (DESIGN.md)")

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        """all_anchors: list over classes of [nz,ny,nx,ns,nr,7]; gt [B, M, 8] -> dict of [B, A(, code)]."""
        code = self.box_coder.code_size
        reg_all, cls_all, w_all = [], [], []
        for gt in gt_boxes_with_classes:
            # drop the zero padding at the tail
            nonzero = (gt[:, :-1].abs().sum(dim=1) != 0).nonzero()
            count = int(nonzero[-1]) + 1 if nonzero.numel() else min(1, gt.shape[0])
            boxes, classes = gt[:count, :-1], gt[:count, -1].int()
            names = self.class_names[(classes.cpu().numpy() - 1).clip(0)] if count else np.array([])
            per_class = []
            for cls_name, anchors in zip(self.anchor_class_names, all_anchors):
                sel = torch.from_numpy(np.asarray(names == cls_name, dtype=bool).reshape(-1)).to(gt.device)
                fmap = anchors.shape[:3]
                t = self.assign_targets_single(anchors.view(-1, anchors.shape[-1]), boxes[sel], classes[sel],
                                               self.matched_thresholds[cls_name], self.unmatched_thresholds[cls_name])
                per_class.append((t, fmap))
            cls_all.append(torch.cat([t["box_cls_labels"].view(*f, -1) for t, f in per_class], dim=-1).view(-1))
            reg_all.append(torch.cat([t["box_reg_targets"].view(*f, -1, code) for t, f in per_class], dim=-2).view(-1, code))
            w_all.append(torch.cat([t["reg_weights"].view(*f, -1) for t, f in per_class], dim=-1).view(-1))
        return {"box_cls_labels": torch.stack(cls_all), "box_reg_targets": torch.stack(reg_all),
                "reg_weights": torch.stack(w_all)}

    def assign_targets_single(self, anchors, gt_boxes, gt_classes, matched_threshold=0.6, unmatched_threshold=0.45):
        n_anchor, n_gt = anchors.shape[0], gt_boxes.shape[0]
        dev = anchors.device
        labels = torch.full((n_anchor,), -1, dtype=torch.int32, device=dev)
        have = n_gt > 0 and n_anchor > 0
        if have:
            iou = box_utils.boxes3d_nearest_bev_iou(anchors[:, 0:7], gt_boxes[:, 0:7])      # [A, G]
            best_gt = iou.argmax(dim=1)
            best_iou = iou[torch.arange(n_anchor, device=dev), best_gt]
            gt_best = iou[iou.argmax(dim=0), torch.arange(n_gt, device=dev)]
            gt_best[gt_best == 0] = -1                                                       # gt that touches nothing
            forced = (iou == gt_best).nonzero()[:, 0]                                        # every gt keeps its best anchor(s)
            forced_gt = best_gt[forced]
            labels[forced] = gt_classes[forced_gt]
            pos = best_iou >= matched_threshold
            labels[pos] = gt_classes[best_gt[pos]]
            bg = (best_iou < unmatched_threshold).nonzero()[:, 0]
        else:
            bg = torch.arange(n_anchor, device=dev)
        fg = (labels > 0).nonzero()[:, 0]
        if self.pos_fraction is not None:
            num_fg = int(self.pos_fraction * self.sample_size)
            if len(fg) > num_fg:
                labels[torch.randperm(len(fg))[:len(fg) - num_fg]] = -1   # (sic) the reference indexes labels directly
                fg = (labels > 0).nonzero()[:, 0]
            num_bg = self.sample_size - int((labels > 0).sum())
            if len(bg) > num_bg:
                labels[bg[torch.randint(0, len(bg), size=(num_bg,))]] = 0
        elif not have:
            labels[:] = 0
        else:
            labels[bg] = 0
            labels[forced] = gt_classes[forced_gt]
        targets = anchors.new_zeros((n_anchor, self.box_coder.code_size))
        if have:
            targets[fg, :] = self.box_coder.encode_torch(gt_boxes[best_gt[fg], :], anchors[fg, :])
        weights = anchors.new_zeros((n_anchor,))
        if self.norm_by_num_examples:
            num = (labels >= 0).sum()
            weights[labels > 0] = 1.0 / (num if num > 1.0 else 1.0)
        else:
            weights[labels > 0] = 1.0
        return {"box_cls_labels": labels, "box_reg_targets": targets, "reg_weights": weights}
