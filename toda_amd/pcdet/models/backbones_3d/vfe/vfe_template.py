"""Common base of the voxel feature encoders: what Detector3DTemplate.build_vfe relies on (a `model_cfg` attribute and
`get_output_feature_dim()`, reference detector3d_template.py:55-68)."""
import abc

from torch import nn


class VFETemplate(nn.Module, abc.ABC):
    """A VFE turns `voxels` [M, P, C] (+ `voxel_num_points`, `voxel_coords`) of the batch dict into `voxel_features` [M, C']
    (MeanVFE) or `pillar_features` (PillarVFE)."""

    def __init__(self, model_cfg, **_unused):
        nn.Module.__init__(self)
        self.model_cfg = model_cfg

    @abc.abstractmethod
    def get_output_feature_dim(self):
        """Width C' of the features the encoder emits."""

    @abc.abstractmethod
    def forward(self, batch_dict, **kwargs):
        """batch dict in, batch dict out."""
