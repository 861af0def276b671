"""PointPillarScatter (reference pointpillar_scatter.py:14-37): pillars -> [B, C, ny, nx] canvas.
On the GPU this is toda_pillar_scatter_fwd/bwd; on the CPU (the C1 plumbing configuration, pure
torch modules only) an index_put with the same `z + y*nx + x` addressing."""
import torch
import torch.nn as nn

from toda_amd import ops


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES
        self.nx, self.ny, self.nz = (int(v) for v in grid_size)
        assert self.nz == 1

    def forward(self, batch_dict, **kwargs):
        feats, coords = batch_dict["pillar_features"], batch_dict["voxel_coords"]
        batch_size = batch_dict.get("batch_size")
        if batch_size is None:
            batch_size = int(coords[:, 0].max().item()) + 1
        if feats.is_cuda:
            canvas = ops.pillar_scatter(feats, coords.int(), batch_size, self.ny, self.nx)
        else:
            c = feats.shape[1]
            canvas = feats.new_zeros((batch_size, c, self.nz * self.ny * self.nx))
            cl = coords.long()
            flat = cl[:, 1] + cl[:, 2] * self.nx + cl[:, 3]
            canvas[cl[:, 0], :, flat] = feats
            canvas = canvas.view(batch_size, c * self.nz, self.ny, self.nx)
        batch_dict["spatial_features"] = canvas
        return batch_dict
