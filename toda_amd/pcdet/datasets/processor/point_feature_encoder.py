"""PointFeatureEncoder (reference pcdet/datasets/processor/point_feature_encoder.py:4-61)."""
import numpy as np
import torch


class PointFeatureEncoder:
    def __init__(self, config, point_cloud_range=None):
        self.point_encoding_config = config
        assert list(config.src_feature_list[0:3]) == ["x", "y", "z"]
        self.used_feature_list = list(config.used_feature_list)
        self.src_feature_list = list(config.src_feature_list)
        self.point_cloud_range = point_cloud_range

    @property
    def num_point_features(self):
        return getattr(self, self.point_encoding_config.encoding_type)(points=None)

    def forward(self, data_dict):
        data_dict["points"], data_dict["use_lead_xyz"] = getattr(self, self.point_encoding_config.encoding_type)(
            data_dict["points"])
        return data_dict

    def absolute_coordinates_encoding(self, points=None):
        if points is None:
            return len(self.used_feature_list)
        cols = [points[:, 0:3]]
        for name in self.used_feature_list:
            if name in ("x", "y", "z"):
                continue
            i = self.src_feature_list.index(name)
            col = points[:, i:i + 1]
            if name == "intensity" and self.point_encoding_config.get("normalize_intensity", None):
                top = col.max()
                col = col / (top.clamp_min(1e-12) if torch.is_tensor(col) else max(top, 1e-12))
            cols.append(col)
        if torch.is_tensor(points):          # device-resident cloud (GPU input pipeline): same fp32 arithmetic
            return torch.cat(cols, dim=1).contiguous(), True
        return np.concatenate(cols, axis=1), True
