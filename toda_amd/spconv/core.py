"""SparseConvTensor: the container the reference builds at
pcdet/models/backbones_3d/spconv_backbone.py:141-146 and reads back through .features /.indices /
.spatial_shape /.batch_size /.dense() /.replace_feature() (pcdet/utils/spconv_utils.py:28-34)."""
import torch

from .. import ops


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None,
                 indice_dict=None, benchmark=False):
        if indices.dtype != torch.int32:
            raise TypeError("SparseConvTensor indices must be int32 (the reference calls voxel_coords.int())")
        if features.dim() != 2 or indices.dim() != 2 or indices.shape[1] != 4:
            raise ValueError("features must be [N, C] and indices [N, 4] = (batch, z, y, x)")
        self.features = features
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        # rulebooks shared between layers with the same indice_key (spconv's indice_dict)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        # GridIndex of THIS index set when one is already known (output of a strided conv)
        self.grid_index = grid
        self.benchmark = benchmark
        # BatchNorm moments of `features` when the convolution that produced them took them in its epilogue (else None)
        self.bn_sums = None

    def replace_feature(self, new_features):
        out = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size,
                               indice_dict=self.indice_dict)
        out.grid_index = self.grid_index
        return out

    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        """[B, C, D, H, W] (channels first, as HeightCompression expects)."""
        out = ops.sparse_to_dense(self.features, self.indices, self.batch_size, self.spatial_shape)
        if not channels_first:
            out = out.permute(0, 2, 3, 4, 1).contiguous()
        return out

    def __repr__(self):
        return (f"SparseConvTensor(n={self.features.shape[0]}, c={self.features.shape[1]}, "
                f"shape={self.spatial_shape}, batch={self.batch_size})")
