"""MeanVFE (reference: opencood/models/sub_modules/mean_vfe.py:4-31) on heal_mean_vfe."""
import torch
import torch.nn as nn

from heal_amd import ops


class MeanVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        num = batch_dict["voxel_num_points"]
        if torch.is_grad_enabled() and self.training:   # gradient path (mean_vfe.py:13-31): sum over the rows / clamp_min(num, 1)
            v = batch_dict["voxel_features"]
            batch_dict["voxel_features"] = v.sum(dim=1) / num.to(v.dtype).clamp_min(1.0).unsqueeze(1)
            return batch_dict
        if num.dtype != torch.int32:
            num = num.to(torch.int32)
        batch_dict["voxel_features"] = ops.mean_vfe(batch_dict["voxel_features"], num, batch_dict.get("n_voxels_dev"))
        return batch_dict
