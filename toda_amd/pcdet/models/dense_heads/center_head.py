"""CenterHead (reference pcdet/models/dense_heads/center_head.py:11-355).

Differences from the reference, none of them numerical:
  * target assignment runs on the GPU in one kernel launch per head (toda_center_assign) instead
    of a Python loop over boxes on the CPU (reference :103-219);
  * get_loss keeps its log values as device tensors (no .item() host syncs, reference :247-250).
Module/parameter names equal the reference's (shared_conv.*, heads_list.N.<name>.*)."""
import copy

import torch
import torch.nn as nn
from torch.nn.init import kaiming_normal_

from .... import ops
from ...utils import loss_utils
from ..model_utils import centernet_utils


class SeparateHead(nn.Module):
    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for name, spec in sep_head_dict.items():
            layers = []
            for _ in range(spec["num_conv"] - 1):
                layers.append(nn.Sequential(
                    nn.Conv2d(input_channels, input_channels, 3, stride=1, padding=1, bias=use_bias),
                    nn.BatchNorm2d(input_channels), nn.ReLU()))
            layers.append(nn.Conv2d(input_channels, spec["out_channels"], 3, stride=1, padding=1, bias=True))
            branch = nn.Sequential(*layers)
            if "hm" in name:
                branch[-1].bias.data.fill_(init_bias)
            else:
                for m in branch.modules():
                    if isinstance(m, nn.Conv2d):
                        kaiming_normal_(m.weight.data)
                        if m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            setattr(self, name, branch)

    def alias_fused_buffers(self):
        """Point the hidden layers' BatchNorm running statistics at slices of two wide tensors NOW (what the fused forward
        does lazily).  common_utils.wrap_ddp calls this before a wrapper captures module.buffers(): a wrapper that broadcasts
        the tensors it saw at construction would otherwise keep syncing orphans once the first forward has re-pointed them."""
        branches = [getattr(self, name) for name in self.sep_head_dict]
        if len(branches) < 2 or not all(len(br) == 2 and type(br[0]) is nn.Sequential and len(br[0]) == 3 for br in branches):
            return False
        bns = [br[0][1] for br in branches]
        if not all(type(bn) is nn.BatchNorm2d and bn.track_running_stats and bn.num_features == bns[0].num_features for bn in bns):
            return False
        ops._aliased_running_stats(self, bns)
        return True

    def forward(self, x):
        names = list(self.sep_head_dict)
        branches = [getattr(self, name) for name in names]
        last = [br[-1] for br in branches]
        # every branch ends in Conv2d(C, out_channels <= 4, 3, padding=1): the hidden layers run branch by branch, the five
        # output convolutions (and their backward) as ONE launch per direction (toda_conv3x3_narrow_*)
        if len(names) <= 8 and all(len(br) >= 1 and ops.conv3x3_narrow_supported(x, conv) for br, conv in zip(branches, last)):
            # num_conv == 2 everywhere (every reference config): the n hidden Conv-BN-ReLU layers read the same x -> one
            # C -> n C convolution + one norm; the output convolutions then read their channel slice of that tensor in place
            if all(len(br) == 2 for br in branches) and ops.fused_branch_hidden_supported(x, [br[0] for br in branches]):
                wide = ops.fused_branch_hidden(self, x, [br[0] for br in branches])
                return dict(zip(names, ops.conv3x3_narrow_group_fused(wide, last)))
            hidden = [ops.run_dense_sequential(br[:-1], x) if len(br) > 1 else x for br in branches]
            if all(h.shape == hidden[0].shape for h in hidden):
                return dict(zip(names, ops.conv3x3_narrow_group(hidden, last)))
            return {name: last_conv(h) for name, last_conv, h in zip(names, last, hidden)}
        return {name: ops.run_dense_sequential(br, x) for name, br in zip(names, branches)}


class CenterHead(nn.Module):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size = grid_size
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.voxel_size = [float(v) for v in voxel_size]
        self.feature_map_stride = self.model_cfg.TARGET_ASSIGNER_CONFIG.get("FEATURE_MAP_STRIDE", None)
        self.class_names = list(class_names)

        self.class_names_each_head, self.class_id_mapping_each_head = [], []
        for names in self.model_cfg.CLASS_NAMES_EACH_HEAD:
            present = [n for n in names if n in self.class_names]
            self.class_names_each_head.append(present)
            self.class_id_mapping_each_head.append(torch.tensor([self.class_names.index(n) for n in present]))
        total = sum(len(n) for n in self.class_names_each_head)
        assert total == len(self.class_names), f"class_names_each_head={self.class_names_each_head}"

        shared = self.model_cfg.SHARED_CONV_CHANNEL
        use_bias = self.model_cfg.get("USE_BIAS_BEFORE_NORM", False)
        self.shared_conv = nn.Sequential(nn.Conv2d(input_channels, shared, 3, stride=1, padding=1, bias=use_bias),
                                         nn.BatchNorm2d(shared), nn.ReLU())
        self.heads_list = nn.ModuleList()
        self.separate_head_cfg = self.model_cfg.SEPARATE_HEAD_CFG
        for names in self.class_names_each_head:
            head_dict = copy.deepcopy(dict(self.separate_head_cfg.HEAD_DICT))
            head_dict["hm"] = dict(out_channels=len(names), num_conv=self.model_cfg.NUM_HM_CONV)
            self.heads_list.append(SeparateHead(shared, head_dict, init_bias=-2.19, use_bias=use_bias))
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}
        self.build_losses()

    def build_losses(self):
        self.add_module("hm_loss_func", loss_utils.FocalLossCenterNet())
        self.add_module("reg_loss_func", loss_utils.RegLossCenterNet())

    # ------------------------------------------------------------------ targets
    def _head_class_lut(self, head_idx, device):
        """lut[global class id (1-based, 0 = padding)] -> 1-based id inside this head, 0 = not here."""
        # cached per device: a fresh `.to(device)` of a pageable host tensor is a SYNCHRONOUS copy - the host sat in it until the
        # GPU had worked off the whole sparse backbone (cProfile: 6 ms of an 18 ms step inside Tensor.to), lost its launch lead
        # every step and the GPU then idled at the step boundary
        cache = self.__dict__.setdefault("_const_cache", {})
        key = ("lut", head_idx, str(device))
        if key not in cache:
            lut = torch.zeros(len(self.class_names) + 1, dtype=torch.float32)
            for local, name in enumerate(self.class_names_each_head[head_idx]):
                lut[self.class_names.index(name) + 1] = local + 1
            cache[key] = lut.to(device)
        return cache[key]

    def _device_const(self, key, device, build):
        """Small constant tensors of the head (class maps, range limits, code weights) on `device`, uploaded once."""
        cache = self.__dict__.setdefault("_const_cache", {})
        key = (key, str(device))
        if key not in cache:
            cache[key] = build().to(device)
        return cache[key]

    def _device_const_like(self, key, ref, build):
        """The same keyed on the dtype of `ref` as well (a head used in fp64 and then in fp32 must not hand the fp64 constant to the
        fp32 loss: the product would silently promote - ADVICE r3)."""
        return self._device_const((key, str(ref.dtype)), ref.device, build)

    def assign_targets(self, gt_boxes, feature_map_size=None, **kwargs):
        """gt_boxes [B, G, code+1]; feature_map_size (H, W).  Returns the reference's dict of
        per-head lists: heatmaps [B,C,H,W], target_boxes [B,500,code], inds, masks [B,500] int64."""
        fm_h, fm_w = int(feature_map_size[0]), int(feature_map_size[1])
        tcfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        ret = {"heatmaps": [], "target_boxes": [], "inds": [], "masks": [], "heatmap_masks": []}
        cls_col = gt_boxes[..., -1].long().clamp_(0, len(self.class_names))
        for head_idx, names in enumerate(self.class_names_each_head):
            head_gt = gt_boxes.clone()
            head_gt[..., -1] = self._head_class_lut(head_idx, gt_boxes.device)[cls_col]
            hm, boxes, inds, mask = ops.center_assign(
                head_gt, len(names), fm_w, fm_h, self.point_cloud_range, self.voxel_size,
                tcfg.FEATURE_MAP_STRIDE, tcfg.NUM_MAX_OBJS, tcfg.GAUSSIAN_OVERLAP, tcfg.MIN_RADIUS)
            ret["heatmaps"].append(hm)
            ret["target_boxes"].append(boxes)
            ret["inds"].append(inds)
            ret["masks"].append(mask)
        return ret

    # --------------------------------------------------------------------- loss
    @staticmethod
    def sigmoid(x):
        return torch.clamp(x.sigmoid(), min=1e-4, max=1 - 1e-4)

    def get_loss(self):
        pred_dicts = self.forward_ret_dict["pred_dicts"]
        target_dicts = self.forward_ret_dict["target_dicts"]
        weights = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        tb_dict, loss = {}, 0
        for idx, pred in enumerate(pred_dicts):
            regs = [pred[name] for name in self.separate_head_cfg.HEAD_ORDER]
            if ops.center_loss_supported(pred["hm"], regs, target_dicts["target_boxes"][idx]):
                # sigmoid-clamp + focal + gathered L1, value and gradient, in 3 + 1 launches (torch: ~110 small ones)
                hm_loss, loc_loss, prob = ops.center_loss(pred["hm"], regs, target_dicts["heatmaps"][idx], target_dicts["inds"][idx],
                                                          target_dicts["masks"][idx], target_dicts["target_boxes"][idx],
                                                          weights["code_weights"], weights["cls_weight"], weights["loc_weight"])
                pred["hm"] = prob
                loss = loss + hm_loss + loc_loss
                tb_dict[f"hm_loss_head_{idx}"] = hm_loss.detach()
                tb_dict[f"loc_loss_head_{idx}"] = loc_loss.detach()
                continue
            pred["hm"] = self.sigmoid(pred["hm"])
            hm_loss = self.hm_loss_func(pred["hm"], target_dicts["heatmaps"][idx]) * weights["cls_weight"]
            pred_boxes = torch.cat([pred[name] for name in self.separate_head_cfg.HEAD_ORDER], dim=1)
            reg = self.reg_loss_func(pred_boxes, target_dicts["masks"][idx], target_dicts["inds"][idx],
                                     target_dicts["target_boxes"][idx])
            loc_loss = (reg * self._device_const_like("code_weights", reg, lambda: torch.tensor(weights["code_weights"], dtype=reg.dtype))).sum() * weights["loc_weight"]
            loss = loss + hm_loss + loc_loss
            tb_dict[f"hm_loss_head_{idx}"] = hm_loss.detach()
            tb_dict[f"loc_loss_head_{idx}"] = loc_loss.detach()
        tb_dict["rpn_loss"] = loss.detach()
        return loss, tb_dict

    # ------------------------------------------------------------------- decode
    def generate_predicted_boxes(self, batch_size, pred_dicts):
        post = self.model_cfg.POST_PROCESSING
        ref = pred_dicts[0]["hm"]
        limit = self._device_const("limit", ref.device, lambda: torch.tensor(post.POST_CENTER_LIMIT_RANGE, dtype=torch.float32))
        ret = [{"pred_boxes": [], "pred_scores": [], "pred_labels": []} for _ in range(batch_size)]
        for idx, pred in enumerate(pred_dicts):
            decoded = centernet_utils.decode_bbox_from_heatmap(
                heatmap=pred["hm"].sigmoid(), rot_cos=pred["rot"][:, 0:1], rot_sin=pred["rot"][:, 1:2],
                center=pred["center"], center_z=pred["center_z"], dim=pred["dim"].exp(),
                vel=pred["vel"] if "vel" in self.separate_head_cfg.HEAD_ORDER else None,
                point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size,
                feature_map_stride=self.feature_map_stride, K=post.MAX_OBJ_PER_SAMPLE,
                circle_nms=(post.NMS_CONFIG.NMS_TYPE == "circle_nms"), score_thresh=post.SCORE_THRESH,
                post_center_limit_range=limit)
            mapping = self._device_const(("mapping", idx), ref.device, lambda: self.class_id_mapping_each_head[idx])
            for k, final in enumerate(decoded):
                final["pred_labels"] = mapping[final["pred_labels"].long()]
                if post.NMS_CONFIG.NMS_TYPE != "circle_nms":
                    from ..model_utils import model_nms_utils
                    sel, sel_scores = model_nms_utils.class_agnostic_nms(
                        box_scores=final["pred_scores"], box_preds=final["pred_boxes"], nms_config=post.NMS_CONFIG,
                        score_thresh=None)
                    final["pred_boxes"] = final["pred_boxes"][sel]
                    final["pred_scores"] = sel_scores
                    final["pred_labels"] = final["pred_labels"][sel]
                for key in ("pred_boxes", "pred_scores", "pred_labels"):
                    ret[k][key].append(final[key])
        for k in range(batch_size):
            ret[k]["pred_boxes"] = torch.cat(ret[k]["pred_boxes"], dim=0)
            ret[k]["pred_scores"] = torch.cat(ret[k]["pred_scores"], dim=0)
            ret[k]["pred_labels"] = torch.cat(ret[k]["pred_labels"], dim=0) + 1
        return ret

    @staticmethod
    def reorder_rois_for_refining(batch_size, pred_dicts):
        n_max = max(1, max(len(d["pred_boxes"]) for d in pred_dicts))
        ref = pred_dicts[0]["pred_boxes"]
        rois = ref.new_zeros((batch_size, n_max, ref.shape[-1]))
        roi_scores = ref.new_zeros((batch_size, n_max))
        roi_labels = ref.new_zeros((batch_size, n_max)).long()
        for b in range(batch_size):
            n = len(pred_dicts[b]["pred_boxes"])
            rois[b, :n] = pred_dicts[b]["pred_boxes"]
            roi_scores[b, :n] = pred_dicts[b]["pred_scores"]
            roi_labels[b, :n] = pred_dicts[b]["pred_labels"]
        return rois, roi_scores, roi_labels

    def forward(self, data_dict):
        feats = data_dict["spatial_features_2d"]
        x = ops.run_dense_sequential(self.shared_conv, feats)
        pred_dicts = [head(x) for head in self.heads_list]
        if self.training:
            self.forward_ret_dict["target_dicts"] = self.assign_targets(
                data_dict["gt_boxes"], feature_map_size=feats.size()[2:],
                feature_map_stride=data_dict.get("spatial_features_2d_strides", None))
        self.forward_ret_dict["pred_dicts"] = pred_dicts
        if not self.training or self.predict_boxes_when_training:
            boxes = self.generate_predicted_boxes(data_dict["batch_size"], pred_dicts)
            if self.predict_boxes_when_training:
                rois, roi_scores, roi_labels = self.reorder_rois_for_refining(data_dict["batch_size"], boxes)
                data_dict.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels, has_class_labels=True)
            else:
                data_dict["final_box_dicts"] = boxes
        return data_dict
