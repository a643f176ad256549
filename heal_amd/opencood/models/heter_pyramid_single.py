"""HeterPyramidSingle -- single-agent HEAL model (reference: opencood/models/
heter_pyramid_single.py:19-136): encoder -> backbone -> aligner -> PyramidFusion.forward_single ->
shrink -> heads."""
import torch.nn as nn

from heal_amd.opencood.models._heter_common import (anchor_heads, crop_camera_feature, wants_depth_items, detection_heads,
                                                     modality_stems)
from heal_amd.opencood.models.fuse_modules.pyramid_fuse import PyramidFusion
from heal_amd.opencood.models.sub_modules.bev_blocks import AlignNet, DownsampleConv, ResNetBEVBackbone


class HeterPyramidSingle(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.fix_modules = ["pyramid_backbone", "cls_head", "reg_head", "dir_head"]
        for m, setting in modality_stems(self, args, lambda st: ResNetBEVBackbone(st["backbone_args"])):
            setattr(self, f"aligner_{m}", AlignNet(setting["aligner_args"]))
            if args.get("fix_encoder", False):
                self.fix_modules += [f"encoder_{m}", f"backbone_{m}"]
        self.pyramid_backbone = PyramidFusion(args["fusion_backbone"])
        self.shrink_flag = "shrink_header" in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args["shrink_header"])
            self.fix_modules.append("shrink_conv")
        self.cls_head, self.reg_head, self.dir_head = anchor_heads(args["in_head"], args)
        self.model_train_init()

    def model_train_init(self):
        for name in self.fix_modules:
            for p in getattr(self, name).parameters():
                p.requires_grad_(False)

    def forward(self, data_dict):
        output_dict = {"pyramid": "single"}
        names = [k for k in data_dict.keys() if k.startswith("inputs_")]
        assert len(names) == 1
        m = names[0][len("inputs_"):]
        feature = getattr(self, f"encoder_{m}")(data_dict, m)
        feature = getattr(self, f"backbone_{m}")({"spatial_features": feature})["spatial_features_2d"]
        feature = getattr(self, f"aligner_{m}")(feature)
        feature = crop_camera_feature(self, m, feature)
        if wants_depth_items(self, m):
            output_dict[f"depth_items_{m}"] = getattr(self, f"encoder_{m}").depth_items
        feature, occ_map_list = self.pyramid_backbone.forward_single(feature)
        if self.shrink_flag:
            feature = self.shrink_conv(feature)
        cls_preds, reg_preds, dir_preds = detection_heads(feature, self.cls_head, self.reg_head, self.dir_head)
        output_dict.update({"cls_preds": cls_preds, "reg_preds": reg_preds, "dir_preds": dir_preds,
                            "occ_single_list": occ_map_list})
        return output_dict
