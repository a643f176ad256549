"""HeightCompression (reference: opencood/models/sub_modules/height_compression.py:4-26): sparse ->
dense [N, C*D, H, W] in one streaming pass (heal_sp_to_bev)."""
import torch.nn as nn


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg["feature_num"]

    def forward(self, batch_dict):
        batch_dict["spatial_features"] = batch_dict["encoded_spconv_tensor"].dense()
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict
