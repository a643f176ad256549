"""Old-style single-agent SECOND detector: host mirror of opencood/models/second.py:14-58 -- MeanVFE ->
VoxelBackBone8x (sparse conv, K3) -> HeightCompression -> BaseBEVBackbone -> 1x1 heads; output keys `psm` / `rm`.

Note on the reference: second.py:31 reads `args['anchor_num']` for the regression head while :29 reads
`args['anchor_number']`; the mirror accepts either spelling for the former."""
import torch
import torch.nn as nn

from heal_amd.opencood.models.point_pillar import head
from heal_amd.opencood.models.sub_modules.base_bev_backbone import BaseBEVBackbone
from heal_amd.opencood.models.sub_modules.height_compression import HeightCompression
from heal_amd.opencood.models.sub_modules.mean_vfe import MeanVFE
from heal_amd.opencood.models.sub_modules.sparse_backbone_3d import VoxelBackBone8x


class Second(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.mean_vfe = MeanVFE(args["mean_vfe"], 4)
        self.backbone_3d = VoxelBackBone8x(args["backbone_3d"], 4, args["grid_size"])
        self.height_compression = HeightCompression(args["height_compression"])
        self.backbone_2d = BaseBEVBackbone(args["base_bev_backbone"], 256)
        self.cls_head = nn.Conv2d(256 * 2, args["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(256 * 2, 7 * args.get("anchor_num", args["anchor_number"]), kernel_size=1)

    def forward(self, data_dict):
        lidar = data_dict["processed_lidar"]
        coords = lidar["voxel_coords"]
        batch_dict = {"voxel_features": lidar["voxel_features"], "voxel_coords": coords,
                      "voxel_num_points": lidar["voxel_num_points"],
                      "batch_size": int(coords[:, 0].max().item()) + 1}  # second.py:39
        batch_dict = self.mean_vfe(batch_dict)
        if self.training and torch.is_grad_enabled():   # gradient path (sparse_backbone_3d.py: sparse on the device)
            batch_dict = self.backbone_3d.forward_autograd(batch_dict)
        else:
            batch_dict = self.backbone_3d(batch_dict)
        batch_dict = self.height_compression(batch_dict)
        x = self.backbone_2d(batch_dict)["spatial_features_2d"]
        return {"psm": head(self.cls_head, x), "rm": head(self.reg_head, x)}
