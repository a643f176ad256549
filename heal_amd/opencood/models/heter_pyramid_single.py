"""HeterPyramidSingle -- single-agent HEAL model (reference: opencood/models/
heter_pyramid_single.py:19-136): encoder -> backbone -> aligner -> PyramidFusion.forward_single ->
shrink -> heads."""
from collections import OrderedDict

import torch.nn as nn

from heal_amd.opencood.models._heter_common import detection_heads, center_crop, find_encoder, modality_names
from heal_amd.opencood.models.fuse_modules.pyramid_fuse import PyramidFusion
from heal_amd.opencood.models.sub_modules.bev_blocks import AlignNet, DownsampleConv, ResNetBEVBackbone


class HeterPyramidSingle(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.modality_name_list = modality_names(args)
        self.cav_range = args["lidar_range"]
        self.sensor_type_dict = OrderedDict()
        self.fix_modules = ["pyramid_backbone", "cls_head", "reg_head", "dir_head"]
        for m in self.modality_name_list:
            setting = args[m]
            sensor = setting["sensor_type"]
            self.sensor_type_dict[m] = sensor
            setattr(self, f"encoder_{m}", find_encoder(setting["core_method"])(setting["encoder_args"]))
            setattr(self, f"depth_supervision_{m}", bool(setting["encoder_args"].get("depth_supervision", False)))
            setattr(self, f"backbone_{m}", ResNetBEVBackbone(setting["backbone_args"]))
            if sensor == "camera":
                gc = setting["camera_mask_args"]["grid_conf"]
                setattr(self, f"crop_ratio_W_{m}", self.cav_range[3] / gc["xbound"][1])
                setattr(self, f"crop_ratio_H_{m}", self.cav_range[4] / gc["ybound"][1])
            setattr(self, f"aligner_{m}", AlignNet(setting["aligner_args"]))
            if args.get("fix_encoder", False):
                self.fix_modules += [f"encoder_{m}", f"backbone_{m}"]
        self.pyramid_backbone = PyramidFusion(args["fusion_backbone"])
        self.shrink_flag = "shrink_header" in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args["shrink_header"])
            self.fix_modules.append("shrink_conv")
        self.cls_head = nn.Conv2d(args["in_head"], args["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(args["in_head"], 7 * args["anchor_number"], kernel_size=1)
        self.dir_head = nn.Conv2d(args["in_head"], args["dir_args"]["num_bins"] * args["anchor_number"],
                                  kernel_size=1)
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
        if self.sensor_type_dict[m] == "camera":
            _, _, H, W = feature.shape
            feature = center_crop(feature, int(H * getattr(self, f"crop_ratio_H_{m}")),
                                  int(W * getattr(self, f"crop_ratio_W_{m}")))
            if getattr(self, f"depth_supervision_{m}"):
                output_dict[f"depth_items_{m}"] = getattr(self, f"encoder_{m}").depth_items
        feature, occ_map_list = self.pyramid_backbone.forward_single(feature)
        if self.shrink_flag:
            feature = self.shrink_conv(feature)
        cls_preds, reg_preds, dir_preds = detection_heads(feature, self.cls_head, self.reg_head, self.dir_head)
        output_dict.update({"cls_preds": cls_preds, "reg_preds": reg_preds, "dir_preds": dir_preds,
                            "occ_single_list": occ_map_list})
        return output_dict
