"""AnchorHeadTemplate (reference pcdet/models/dense_heads/anchor_head_template.py:11-275): anchors,
target assigner, sigmoid-focal / smooth-L1 (sin-difference) / direction-CE losses, box decoding.
Log values stay device tensors (no .item() syncs)."""
import numpy as np
import torch
import torch.nn as nn

from ...utils import box_coder_utils, common_utils, loss_utils
from .target_assigner.anchor_generator import AnchorGenerator
from .target_assigner.axis_aligned_target_assigner import AxisAlignedTargetAssigner


class AnchorHeadTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, class_names, grid_size, point_cloud_range, predict_boxes_when_training):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.class_names = class_names
        self.predict_boxes_when_training = predict_boxes_when_training
        self.use_multihead = self.model_cfg.get("USE_MULTIHEAD", False)
        tcfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        self.box_coder = getattr(box_coder_utils, tcfg.BOX_CODER)(num_dir_bins=tcfg.get("NUM_DIR_BINS", 6),
                                                                 **tcfg.get("BOX_CODER_CONFIG", {}))
        anchors, self.num_anchors_per_location = self.generate_anchors(
            self.model_cfg.ANCHOR_GENERATOR_CONFIG, grid_size=grid_size, point_cloud_range=point_cloud_range,
            anchor_ndim=self.box_coder.code_size)
        self.anchors = anchors  # moved with the module in _apply
        self.target_assigner = self.get_target_assigner(tcfg)
        self.forward_ret_dict = {}
        self.build_losses(self.model_cfg.LOSS_CONFIG)

    def _apply(self, fn):
        super()._apply(fn)
        self.anchors = [fn(a) for a in self.anchors]
        return self

    @staticmethod
    def generate_anchors(anchor_generator_cfg, grid_size, point_cloud_range, anchor_ndim=7):
        gen = AnchorGenerator(anchor_range=point_cloud_range, anchor_generator_config=anchor_generator_cfg)
        fmaps = [np.asarray(grid_size[:2]) // c["feature_map_stride"] for c in anchor_generator_cfg]
        anchors, per_loc = gen.generate_anchors(fmaps)
        if anchor_ndim != 7:
            anchors = [torch.cat((a, a.new_zeros([*a.shape[:-1], anchor_ndim - 7])), dim=-1) for a in anchors]
        return anchors, per_loc

    def get_target_assigner(self, tcfg):
        if tcfg.NAME != "AxisAlignedTargetAssigner":
            raise NotImplementedError(f"target assigner {tcfg.NAME} is not on the named path")
        return AxisAlignedTargetAssigner(model_cfg=self.model_cfg, class_names=self.class_names, box_coder=self.box_coder,
                                         match_height=tcfg.MATCH_HEIGHT)

    def build_losses(self, losses_cfg):
        self.add_module("cls_loss_func", loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0))
        name = losses_cfg.get("REG_LOSS_TYPE", None) or "WeightedSmoothL1Loss"
        self.add_module("reg_loss_func", getattr(loss_utils, name)(code_weights=losses_cfg.LOSS_WEIGHTS["code_weights"]))
        self.add_module("dir_loss_func", loss_utils.WeightedCrossEntropyLoss())

    def assign_targets(self, gt_boxes):
        return self.target_assigner.assign_targets(self.anchors, gt_boxes)

    def _flat_anchors(self):
        a = torch.cat(self.anchors, dim=-3) if isinstance(self.anchors, list) else self.anchors
        return a.view(1, -1, a.shape[-1])

    def get_cls_layer_loss(self):
        cls_preds = self.forward_ret_dict["cls_preds"]
        labels = self.forward_ret_dict["box_cls_labels"]
        bs = int(cls_preds.shape[0])
        positives, negatives = labels > 0, labels == 0
        cls_weights = (negatives * 1.0 + 1.0 * positives).float()
        if self.num_class == 1:
            labels = labels.clone()
            labels[positives] = 1
        cls_weights = cls_weights / torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
        cls_targets = labels * (labels >= 0).type_as(labels)
        one_hot = torch.zeros(*cls_targets.shape, self.num_class + 1, dtype=cls_preds.dtype, device=cls_targets.device)
        one_hot.scatter_(-1, cls_targets.unsqueeze(-1).long(), 1.0)
        loss = self.cls_loss_func(cls_preds.view(bs, -1, self.num_class), one_hot[..., 1:], weights=cls_weights)
        cls_loss = loss.sum() / bs * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS["cls_weight"]
        return cls_loss, {"rpn_loss_cls": cls_loss.detach()}

    @staticmethod
    def add_sin_difference(boxes1, boxes2, dim=6):
        """sin(a - b) = sin a cos b - cos a sin b: encode the heading residual so it is periodic."""
        assert dim != -1
        r1 = torch.sin(boxes1[..., dim:dim + 1]) * torch.cos(boxes2[..., dim:dim + 1])
        r2 = torch.cos(boxes1[..., dim:dim + 1]) * torch.sin(boxes2[..., dim:dim + 1])
        return (torch.cat([boxes1[..., :dim], r1, boxes1[..., dim + 1:]], dim=-1),
                torch.cat([boxes2[..., :dim], r2, boxes2[..., dim + 1:]], dim=-1))

    @staticmethod
    def get_direction_target(anchors, reg_targets, one_hot=True, dir_offset=0, num_bins=2):
        bs = reg_targets.shape[0]
        anchors = anchors.view(bs, -1, anchors.shape[-1])
        rot_gt = reg_targets[..., 6] + anchors[..., 6]
        offset_rot = common_utils.limit_period(rot_gt - dir_offset, 0, 2 * np.pi)
        bins = torch.clamp(torch.floor(offset_rot / (2 * np.pi / num_bins)).long(), min=0, max=num_bins - 1)
        if not one_hot:
            return bins
        out = torch.zeros(*bins.shape, num_bins, dtype=anchors.dtype, device=bins.device)
        out.scatter_(-1, bins.unsqueeze(-1), 1.0)
        return out

    def get_box_reg_layer_loss(self):
        box_preds = self.forward_ret_dict["box_preds"]
        dir_preds = self.forward_ret_dict.get("dir_cls_preds", None)
        reg_targets = self.forward_ret_dict["box_reg_targets"]
        labels = self.forward_ret_dict["box_cls_labels"]
        bs = int(box_preds.shape[0])
        weights = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        positives = labels > 0
        reg_w = positives.float() / torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
        anchors = self._flat_anchors().repeat(bs, 1, 1)
        box_preds = box_preds.view(bs, -1, box_preds.shape[-1] // self.num_anchors_per_location)
        p_sin, t_sin = self.add_sin_difference(box_preds, reg_targets)
        loc_loss = self.reg_loss_func(p_sin, t_sin, weights=reg_w).sum() / bs * weights["loc_weight"]
        box_loss, tb = loc_loss, {"rpn_loss_loc": loc_loss.detach()}
        if dir_preds is not None:
            dir_targets = self.get_direction_target(anchors, reg_targets, dir_offset=self.model_cfg.DIR_OFFSET,
                                                    num_bins=self.model_cfg.NUM_DIR_BINS)
            logits = dir_preds.view(bs, -1, self.model_cfg.NUM_DIR_BINS)
            w = positives.type_as(logits)
            w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
            dir_loss = self.dir_loss_func(logits, dir_targets, weights=w).sum() / bs * weights["dir_weight"]
            box_loss = box_loss + dir_loss
            tb["rpn_loss_dir"] = dir_loss.detach()
        return box_loss, tb

    def get_loss(self):
        cls_loss, tb = self.get_cls_layer_loss()
        box_loss, tb_box = self.get_box_reg_layer_loss()
        tb.update(tb_box)
        rpn_loss = cls_loss + box_loss
        tb["rpn_loss"] = rpn_loss.detach()
        return rpn_loss, tb

    def generate_predicted_boxes(self, batch_size, cls_preds, box_preds, dir_cls_preds=None):
        anchors = self._flat_anchors()
        n = anchors.shape[1]
        batch_anchors = anchors.repeat(batch_size, 1, 1)
        batch_cls = cls_preds.view(batch_size, n, -1).float()
        batch_box = self.box_coder.decode_torch(box_preds.view(batch_size, n, -1), batch_anchors)
        if dir_cls_preds is not None:
            offset, limit = self.model_cfg.DIR_OFFSET, self.model_cfg.DIR_LIMIT_OFFSET
            dir_labels = torch.max(dir_cls_preds.view(batch_size, n, -1), dim=-1)[1]
            period = 2 * np.pi / self.model_cfg.NUM_DIR_BINS
            rot = common_utils.limit_period(batch_box[..., 6] - offset, limit, period)
            batch_box[..., 6] = rot + offset + period * dir_labels.to(batch_box.dtype)
        return batch_cls, batch_box

    def forward(self, **kwargs):
        raise NotImplementedError
