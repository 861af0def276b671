"""PillarVFE / PFNLayer (reference pcdet/models/backbones_3d/vfe/pillar_vfe.py:8-123): per-point
decorations (offset to the pillar's mean point and to the pillar centre), Linear -> BN1d -> ReLU
and a max over the points of each pillar.  Plain torch (the C1 "plumbing" configuration runs it
on the CPU); the sparse kernels are not involved."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .vfe_template import VFETemplate


class PFNLayer(nn.Module):
    CHUNK = 50000  # the reference splits the Linear at 50k pillars; numerically irrelevant

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not last_layer:
            out_channels //= 2
        if use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)
        self.part = self.CHUNK

    def forward(self, inputs):
        if inputs.shape[0] > self.part:
            x = torch.cat([self.linear(chunk) for chunk in inputs.split(self.part, dim=0)], dim=0)
        else:
            x = self.linear(inputs)
        if self.use_norm:
            # BN over the channel dim of [M, P, C]
            x = self.norm(x.transpose(1, 2)).transpose(1, 2)
        x = F.relu(x)
        pooled = x.max(dim=1, keepdim=True)[0]
        if self.last_vfe:
            return pooled
        return torch.cat([x, pooled.expand(-1, inputs.shape[1], -1)], dim=2)


class PillarVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.use_norm = self.model_cfg.USE_NORM
        self.with_distance = self.model_cfg.WITH_DISTANCE
        self.use_absolute_xyz = self.model_cfg.USE_ABSLOTE_XYZ
        num_point_features += 6 if self.use_absolute_xyz else 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = list(self.model_cfg.NUM_FILTERS)
        assert len(self.num_filters) > 0
        dims = [num_point_features] + self.num_filters
        self.pfn_layers = nn.ModuleList(
            PFNLayer(dims[i], dims[i + 1], self.use_norm, last_layer=(i >= len(dims) - 2)) for i in range(len(dims) - 1)
        )
        self.voxel_x, self.voxel_y, self.voxel_z = voxel_size
        self.x_offset = self.voxel_x / 2 + point_cloud_range[0]
        self.y_offset = self.voxel_y / 2 + point_cloud_range[1]
        self.z_offset = self.voxel_z / 2 + point_cloud_range[2]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    @staticmethod
    def get_paddings_indicator(actual_num, max_num, axis=0):
        """mask[m, p] = p < actual_num[m]"""
        steps = torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(1, -1)
        return actual_num.unsqueeze(axis + 1).int() > steps

    def forward(self, batch_dict, **kwargs):
        voxels, num_points, coords = batch_dict["voxels"], batch_dict["voxel_num_points"], batch_dict["voxel_coords"]
        xyz = voxels[:, :, :3]
        mean_xyz = xyz.sum(dim=1, keepdim=True) / num_points.type_as(voxels).view(-1, 1, 1)
        f_cluster = xyz - mean_xyz
        centre = torch.stack([
            coords[:, 3].to(voxels.dtype) * self.voxel_x + self.x_offset,
            coords[:, 2].to(voxels.dtype) * self.voxel_y + self.y_offset,
            coords[:, 1].to(voxels.dtype) * self.voxel_z + self.z_offset,
        ], dim=1).unsqueeze(1)
        f_center = xyz - centre
        parts = [voxels if self.use_absolute_xyz else voxels[..., 3:], f_cluster, f_center]
        if self.with_distance:
            parts.append(torch.norm(xyz, 2, 2, keepdim=True))
        features = torch.cat(parts, dim=-1)
        mask = self.get_paddings_indicator(num_points, features.shape[1]).unsqueeze(-1).type_as(voxels)
        features = features * mask
        for pfn in self.pfn_layers:
            features = pfn(features)
        batch_dict["pillar_features"] = features.squeeze(1)
        return batch_dict
