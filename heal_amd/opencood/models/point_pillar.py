"""Old-style single-agent PointPillars detector: host mirror of opencood/models/point_pillar.py:17-80.

Same constructor `args`, `forward(data_dict)` reading `data_dict['processed_lidar']`, same state_dict names
(`pillar_vfe.*`, `backbone.*`, `shrink_conv.*`, `cls_head/reg_head/dir_head.*`).  PillarVFE + PointPillarScatter run
as the fused HIP operator K2 (heal_pfn_scatter)."""
import torch
import torch.nn as nn

from heal_amd.opencood.models.heter_encoders import PointPillar as _PillarEncoder
from heal_amd.opencood.models.sub_modules.base_bev_backbone import BaseBEVBackbone
from heal_amd.opencood.models.sub_modules.base_bev_backbone_resnet import ResNetBEVBackbone
from heal_amd.opencood.models.sub_modules.bev_blocks import conv_bias_act
from heal_amd.opencood.models.sub_modules.downsample_conv import DownsampleConv


def head(conv, x):
    if torch.is_grad_enabled() and (x.requires_grad or conv.training or conv.weight.requires_grad):   # gradient path
        return conv(x)
    return conv_bias_act(x, conv.weight, conv.bias, conv.stride, conv.padding, 1, 1, False)


class _PillarStem(_PillarEncoder):
    """`pillar_vfe` + `scatter` (+ K2) reading the old `processed_lidar` key (point_pillar.py:55-66)."""

    def encode_processed_lidar(self, data_dict):
        return super().forward({"inputs_lidar": data_dict["processed_lidar"]}, "lidar")


class _PillarDetector(_PillarStem):
    """Stem + `backbone` (+ `shrink_conv`) + anchor heads: what point_pillar.py:17-52 and point_pillar_baseline.py:16-69
    construct alike.  `before_heads(args)` lets a subclass register modules that sit between the shrink header and the
    heads (and need `out_channel`)."""

    def __init__(self, args):
        super().__init__(args)
        is_resnet = args["base_bev_backbone"].get("resnet", False)
        self.backbone = (ResNetBEVBackbone if is_resnet else BaseBEVBackbone)(args["base_bev_backbone"], 64)
        self.out_channel = sum(args["base_bev_backbone"]["num_upsample_filter"])
        self.shrink_flag = "shrink_header" in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args["shrink_header"])
            self.out_channel = args["shrink_header"]["dim"][-1]
        self.before_heads(args)
        self.cls_head = nn.Conv2d(self.out_channel, args["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(self.out_channel, 7 * args["anchor_number"], kernel_size=1)
        self.use_dir = "dir_args" in args
        if self.use_dir:
            self.dir_head = nn.Conv2d(self.out_channel, args["dir_args"]["num_bins"] * args["anchor_number"],
                                      kernel_size=1)

    def before_heads(self, args):
        pass

    def bev_features(self, data_dict):
        """(canvas [n,64,ny,nx], backbone (+ shrink) output)."""
        canvas = self.encode_processed_lidar(data_dict)
        x = self.backbone({"spatial_features": canvas})["spatial_features_2d"]
        return canvas, (self.shrink_conv(x) if self.shrink_flag else x)

    def predictions(self, x):
        out = {"cls_preds": head(self.cls_head, x), "reg_preds": head(self.reg_head, x)}
        if self.use_dir:
            out["dir_preds"] = head(self.dir_head, x)
        return out


class PointPillar(_PillarDetector):
    def forward(self, data_dict):
        return self.predictions(self.bev_features(data_dict)[1])
