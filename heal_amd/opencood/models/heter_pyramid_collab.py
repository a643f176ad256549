"""HeterPyramidCollab -- HEAL's collaborative model (reference: opencood/models/
heter_pyramid_collab.py:21-209).  Same constructor `args`, same forward(data_dict) keys, same
state_dict key names; encoders, fusion and scatter run on the gfx950 kernels of libheal_amd.
"""
from collections import Counter

import torch
import torch.nn as nn

from heal_amd.opencood.models._heter_common import (anchor_heads, crop_camera_feature, detection_heads, encode_modalities,
                                                     modality_stems, record_len_to_list, wants_depth_items)
from heal_amd.opencood.models.fuse_modules.pyramid_fuse import PyramidFusion
from heal_amd.opencood.models.sub_modules.bev_blocks import (AlignNet, DownsampleConv, NaiveCompressor,
                                                             ResNetBEVBackbone)
from heal_amd.opencood.utils.transformation_utils import normalize_pairwise_tfm, pairwise_to_host


class HeterPyramidCollab(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.cam_crop_info = {}
        for m, setting in modality_stems(self, args, lambda st: ResNetBEVBackbone(st["backbone_args"])):
            setattr(self, f"aligner_{m}", AlignNet(setting["aligner_args"]))
            if setting["sensor_type"] == "camera":
                grid = setting["camera_mask_args"]["grid_conf"]
                setattr(self, f"xdist_{m}", grid["xbound"][1] - grid["xbound"][0])
                setattr(self, f"ydist_{m}", grid["ybound"][1] - grid["ybound"][0])
                self.cam_crop_info[m] = {f"crop_ratio_W_{m}": getattr(self, f"crop_ratio_W_{m}"),
                                         f"crop_ratio_H_{m}": getattr(self, f"crop_ratio_H_{m}")}
        self.H = self.cav_range[4] - self.cav_range[1]
        self.W = self.cav_range[3] - self.cav_range[0]
        self.fake_voxel_size = 1
        self.pyramid_backbone = PyramidFusion(args["fusion_backbone"])
        self.shrink_flag = "shrink_header" in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args["shrink_header"])
        self.cls_head, self.reg_head, self.dir_head = anchor_heads(args["in_head"], args)
        self.compress = "compressor" in args
        if self.compress:
            self.compressor = NaiveCompressor(args["compressor"]["input_dim"], args["compressor"]["compress_ratio"])
        self.model_train_init()

    def model_train_init(self):
        if self.compress:
            self.eval()
            for p in self.parameters():
                p.requires_grad_(False)
            self.compressor.train()
            for p in self.compressor.parameters():
                p.requires_grad_(True)

    def encode_modality(self, data_dict, m):
        """encoder -> light backbone -> aligner (-> camera pad) for all agents of modality m."""
        return self.encode_modality_tail(m, getattr(self, f"encoder_{m}")(data_dict, m))

    # the same in two halves, so that the lift + splat of several camera modalities can share one launch (encode_modalities)
    def encode_modality_head(self, data_dict, m):
        enc = getattr(self, f"encoder_{m}")
        return enc.forward_head(data_dict, m) if hasattr(enc, "forward_head") else enc(data_dict, m)

    def encode_modality_tail(self, m, feature):
        feature = getattr(self, f"backbone_{m}")({"spatial_features": feature})["spatial_features_2d"]
        return crop_camera_feature(self, m, getattr(self, f"aligner_{m}")(feature))

    def heads(self, fused_feature):
        if self.shrink_flag:
            fused_feature = self.shrink_conv(fused_feature)
        return detection_heads(fused_feature, self.cls_head, self.reg_head, self.dir_head)

    def forward(self, data_dict):
        output_dict = {"pyramid": "collab"}
        agent_modality_list = data_dict["agent_modality_list"]
        pairwise, grid_f64 = pairwise_to_host(data_dict["pairwise_t_matrix"])
        affine_matrix = normalize_pairwise_tfm(pairwise, self.H, self.W, self.fake_voxel_size)
        record_len = record_len_to_list(data_dict["record_len"])
        counts = Counter(agent_modality_list)
        feats = encode_modalities(self, data_dict, counts, self.encode_modality)   # concurrent streams on a HIP device
        for m in feats:
            if wants_depth_items(self, m):
                output_dict[f"depth_items_{m}"] = getattr(self, f"encoder_{m}").depth_items
        if len(feats) == 1 and len(counts) == 1:
            heter_feature_2d = next(iter(feats.values()))  # already in scene order
        else:
            cursor = {m: 0 for m in self.modality_name_list}
            parts = []
            for m in agent_modality_list:
                parts.append(feats[m][cursor[m]])
                cursor[m] += 1
            heter_feature_2d = torch.stack(parts)
        cam_boxes = None
        if self.compress:
            heter_feature_2d = self.compressor(heter_feature_2d)
        else:   # the padded camera maps reach the pyramid untouched: it may rely on their zero border (crop_camera_feature)
            cam_boxes = {m: b for m, b in self.__dict__.get("_heal_cam_boxes", {}).items() if m in feats}
        fused, occ_outputs = self.pyramid_backbone.forward_collab(
            heter_feature_2d, record_len, affine_matrix, agent_modality_list, self.cam_crop_info, grid_f64, cam_boxes=cam_boxes)
        cls_preds, reg_preds, dir_preds = self.heads(fused)
        output_dict.update({"cls_preds": cls_preds, "reg_preds": reg_preds, "dir_preds": dir_preds,
                            "occ_single_list": occ_outputs})
        return output_dict
