"""HeterModelLate -- single-agent detector used for late fusion / pre-training (reference:
opencood/models/heter_model_late.py:16-112): encoder -> light backbone -> layers 1..n of the
modality's own multiscale backbone -> deblocks -> shrink -> per-modality heads."""
from collections import OrderedDict

import torch.nn as nn

from heal_amd.opencood.models._heter_common import center_crop, find_encoder, modality_names
from heal_amd.opencood.models.sub_modules.bev_blocks import DownsampleConv, ResNetBEVBackbone


class HeterModelLate(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.modality_name_list = modality_names(args)
        self.cav_range = args["lidar_range"]
        self.sensor_type_dict = OrderedDict()
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
            setattr(self, f"layers_{m}", ResNetBEVBackbone(setting["layers_args"]))
            setattr(self, f"layers_num_{m}", len(setting["layers_args"]["num_upsample_filter"]))
            setattr(self, f"shrink_conv_{m}", DownsampleConv(setting["shrink_header"]))
            in_head = setting["head_args"]["in_head"]
            setattr(self, f"cls_head_{m}", nn.Conv2d(in_head, args["anchor_number"], kernel_size=1))
            setattr(self, f"reg_head_{m}", nn.Conv2d(in_head, args["anchor_number"] * 7, kernel_size=1))
            setattr(self, f"dir_head_{m}", nn.Conv2d(in_head, args["anchor_number"] * args["dir_args"]["num_bins"],
                                                     kernel_size=1))

    def forward(self, data_dict):
        output_dict = {}
        names = [k for k in data_dict.keys() if k.startswith("inputs_")]
        assert len(names) == 1
        m = names[0][len("inputs_"):]
        feature = getattr(self, f"encoder_{m}")(data_dict, m)
        feature = getattr(self, f"backbone_{m}")({"spatial_features": feature})["spatial_features_2d"]
        if self.sensor_type_dict[m] == "camera":
            _, _, H, W = feature.shape
            feature = center_crop(feature, int(H * getattr(self, f"crop_ratio_H_{m}")),
                                  int(W * getattr(self, f"crop_ratio_W_{m}")))
            if getattr(self, f"depth_supervision_{m}"):
                output_dict[f"depth_items_{m}"] = getattr(self, f"encoder_{m}").depth_items
        layers = getattr(self, f"layers_{m}")
        feature_list = [feature]  # layer0 of `layers_mX` is deliberately unused (heter_model_late.py:96-103)
        for i in range(1, getattr(self, f"layers_num_{m}")):
            feature = layers.get_layer_i_feature(feature, layer_i=i)
            feature_list.append(feature)
        feature = layers.decode_multiscale_feature(feature_list)
        feature = getattr(self, f"shrink_conv_{m}")(feature)
        output_dict.update({"cls_preds": getattr(self, f"cls_head_{m}")(feature),
                            "reg_preds": getattr(self, f"reg_head_{m}")(feature),
                            "dir_preds": getattr(self, f"dir_head_{m}")(feature)})
        return output_dict
