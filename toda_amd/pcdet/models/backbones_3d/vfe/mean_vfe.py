"""MeanVFE (reference pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31) on the HIP kernel
toda_mean_vfe_fwd/bwd: voxel_features = sum over the P slots / max(num_points, 1)."""
from toda_amd import ops
from .vfe_template import VFETemplate


class MeanVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        """voxels [M, P, C] + voxel_num_points [M]  ->  voxel_features [M, C]"""
        batch_dict["voxel_features"] = ops.mean_vfe(batch_dict["voxels"], batch_dict["voxel_num_points"])
        return batch_dict
