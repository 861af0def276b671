"""HeightCompression (reference height_compression.py:10-26): densify the last sparse level with
toda_sparse_to_dense_fwd and fold D into the channels (BEV channel = c * D + d)."""
import torch.nn as nn


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES

    def forward(self, batch_dict):
        dense = batch_dict["encoded_spconv_tensor"].dense()  # [B, C, D, H, W]
        b, c, d, h, w = dense.shape
        batch_dict["spatial_features"] = dense.view(b, c * d, h, w)
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict
