"""Old-style intermediate-fusion PointPillars: host mirror of
opencood/models/point_pillar_baseline.py:16-135.  fusion_method max / att / v2xvit (the fusion modules on the hot
path); disconet / v2vnet are outside the scope and raise."""
import torch.nn as nn

from heal_amd.opencood.models.fuse_modules.fusion_in_one import AttFusion, MaxFusion, V2XViTFusion
from heal_amd.opencood.models.point_pillar import _PillarStem, head
from heal_amd.opencood.models.sub_modules.base_bev_backbone import BaseBEVBackbone
from heal_amd.opencood.models.sub_modules.base_bev_backbone_resnet import ResNetBEVBackbone
from heal_amd.opencood.models.sub_modules.downsample_conv import DownsampleConv
from heal_amd.opencood.models.sub_modules.naive_compress import NaiveCompressor
from heal_amd.opencood.utils.transformation_utils import normalize_pairwise_tfm


class PointPillarBaseline(_PillarStem):
    def __init__(self, args):
        super().__init__(args)
        is_resnet = args["base_bev_backbone"].get("resnet", False)
        self.backbone = (ResNetBEVBackbone if is_resnet else BaseBEVBackbone)(args["base_bev_backbone"], 64)
        method = args["fusion_method"]
        if method == "max":
            self.fusion_net = MaxFusion()
        elif method == "att":
            self.fusion_net = AttFusion(args["att"]["feat_dim"])
        elif method == "v2xvit":
            self.fusion_net = V2XViTFusion(args["v2xvit"])
        else:
            raise NotImplementedError(f"fusion_method '{method}' is outside the hot-path scope (SURVEY 2, row 2)")
        self.out_channel = sum(args["base_bev_backbone"]["num_upsample_filter"])
        self.shrink_flag = False
        if "shrink_header" in args:
            self.shrink_flag = True
            self.shrink_conv = DownsampleConv(args["shrink_header"])
            self.out_channel = args["shrink_header"]["dim"][-1]
        self.compression = False
        if "compression" in args:
            self.compression = True
            self.naive_compressor = NaiveCompressor(self.out_channel, args["compression"])
        self.cls_head = nn.Conv2d(self.out_channel, args["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(self.out_channel, 7 * args["anchor_number"], kernel_size=1)
        self.use_dir = "dir_args" in args
        if self.use_dir:
            self.dir_head = nn.Conv2d(self.out_channel, args["dir_args"]["num_bins"] * args["anchor_number"],
                                      kernel_size=1)
        if args.get("backbone_fix", False):
            self.backbone_fix()

    def backbone_fix(self):
        """point_pillar_baseline.py:71-97: freeze everything but the fusion net (fine-tuning switch)."""
        frozen = [self.pillar_vfe, self.scatter, self.backbone, self.cls_head, self.reg_head]
        if self.compression:
            frozen.append(self.naive_compressor)
        if self.shrink_flag:
            frozen.append(self.shrink_conv)
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, data_dict):
        record_len = data_dict["record_len"]
        canvas = self.encode_processed_lidar(data_dict)
        H0, W0 = canvas.shape[2:]
        affine_matrix = normalize_pairwise_tfm(data_dict["pairwise_t_matrix"], H0, W0, self.voxel_size[0])
        x = self.backbone({"spatial_features": canvas})["spatial_features_2d"]
        if self.shrink_flag:
            x = self.shrink_conv(x)
        if self.compression:
            x = self.naive_compressor(x)
        fused = self.fusion_net(x, record_len, affine_matrix)
        out = {"cls_preds": head(self.cls_head, fused), "reg_preds": head(self.reg_head, fused)}
        if self.use_dir:
            out["dir_preds"] = head(self.dir_head, fused)
        return out
