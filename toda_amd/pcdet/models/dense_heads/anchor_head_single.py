"""AnchorHeadSingle (reference pcdet/models/dense_heads/anchor_head_single.py:7-75): three 1x1 conv
heads (class, box, direction bin) over the BEV map."""
import numpy as np
import torch.nn as nn

from .anchor_head_template import AnchorHeadTemplate


class AnchorHeadSingle(AnchorHeadTemplate):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range,
                 predict_boxes_when_training=True, **kwargs):
        super().__init__(model_cfg=model_cfg, num_class=num_class, class_names=class_names, grid_size=grid_size,
                         point_cloud_range=point_cloud_range, predict_boxes_when_training=predict_boxes_when_training)
        self.num_anchors_per_location = sum(self.num_anchors_per_location)
        a = self.num_anchors_per_location
        self.conv_cls = nn.Conv2d(input_channels, a * self.num_class, kernel_size=1)
        self.conv_box = nn.Conv2d(input_channels, a * self.box_coder.code_size, kernel_size=1)
        if self.model_cfg.get("USE_DIRECTION_CLASSIFIER", None) is not None:
            self.conv_dir_cls = nn.Conv2d(input_channels, a * self.model_cfg.NUM_DIR_BINS, kernel_size=1)
        else:
            self.conv_dir_cls = None
        self.init_weights()

    def init_weights(self):
        pi = 0.01
        nn.init.constant_(self.conv_cls.bias, -np.log((1 - pi) / pi))
        nn.init.normal_(self.conv_box.weight, mean=0, std=0.001)

    def forward(self, data_dict):
        x = data_dict["spatial_features_2d"]
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()  # noqa: E731
        cls_preds, box_preds = nhwc(self.conv_cls(x)), nhwc(self.conv_box(x))
        self.forward_ret_dict["cls_preds"], self.forward_ret_dict["box_preds"] = cls_preds, box_preds
        dir_preds = None
        if self.conv_dir_cls is not None:
            dir_preds = nhwc(self.conv_dir_cls(x))
            self.forward_ret_dict["dir_cls_preds"] = dir_preds
        if self.training:
            self.forward_ret_dict.update(self.assign_targets(gt_boxes=data_dict["gt_boxes"]))
        if not self.training or self.predict_boxes_when_training:
            batch_cls, batch_box = self.generate_predicted_boxes(data_dict["batch_size"], cls_preds, box_preds, dir_preds)
            data_dict["batch_cls_preds"], data_dict["batch_box_preds"] = batch_cls, batch_box
            data_dict["cls_preds_normalized"] = False
        return data_dict
