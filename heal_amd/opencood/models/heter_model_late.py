"""HeterModelLate -- single-agent detector used for late fusion / pre-training (reference:
opencood/models/heter_model_late.py:16-112): encoder -> light backbone -> layers 1..n of the
modality's own multiscale backbone -> deblocks -> shrink -> per-modality heads."""
import torch.nn as nn

from heal_amd.opencood.models._heter_common import anchor_heads, crop_camera_feature, wants_depth_items, modality_stems
from heal_amd.opencood.models.sub_modules.bev_blocks import DownsampleConv, ResNetBEVBackbone


class HeterModelLate(nn.Module):
    def __init__(self, args):
        super().__init__()
        for m, setting in modality_stems(self, args, lambda st: ResNetBEVBackbone(st["backbone_args"])):
            setattr(self, f"layers_{m}", ResNetBEVBackbone(setting["layers_args"]))
            setattr(self, f"layers_num_{m}", len(setting["layers_args"]["num_upsample_filter"]))
            setattr(self, f"shrink_conv_{m}", DownsampleConv(setting["shrink_header"]))
            for kind, head in zip(("cls", "reg", "dir"), anchor_heads(setting["head_args"]["in_head"], args)):
                setattr(self, f"{kind}_head_{m}", head)

    def forward(self, data_dict):
        output_dict = {}
        names = [k for k in data_dict.keys() if k.startswith("inputs_")]
        assert len(names) == 1
        m = names[0][len("inputs_"):]
        feature = getattr(self, f"encoder_{m}")(data_dict, m)
        feature = getattr(self, f"backbone_{m}")({"spatial_features": feature})["spatial_features_2d"]
        feature = crop_camera_feature(self, m, feature)
        if wants_depth_items(self, m):
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
