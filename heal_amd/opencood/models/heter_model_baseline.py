"""HeterModelBaseline -- single-scale heterogeneous collaboration with a pluggable fusion operator
(reference: opencood/models/heter_model_baseline.py:26-236).  In scope: fusion_method max / att / v2xvit
(BASELINE config 5 uses v2xvit)."""
from collections import Counter

import torch
import torch.nn as nn

from heal_amd.opencood.models._heter_common import (anchor_heads, crop_camera_feature, detection_heads, modality_stems,
                                                     wants_depth_items)
from heal_amd.opencood.models.fuse_modules.fusion_in_one import build_fusion
from heal_amd.opencood.models.sub_modules.base_bev_backbone import BaseBEVBackbone
from heal_amd.opencood.models.sub_modules.bev_blocks import DownsampleConv, NaiveCompressor
from heal_amd.opencood.utils.transformation_utils import normalize_pairwise_tfm, pairwise_to_host


class HeterModelBaseline(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.ego_modality = args["ego_modality"]
        for m, setting in modality_stems(self, args, lambda st: BaseBEVBackbone(st["backbone_args"],
                                                                                 st["backbone_args"].get("inplanes", 64))):
            setattr(self, f"shrinker_{m}", DownsampleConv(setting["shrink_header"]))
        self.H = self.cav_range[4] - self.cav_range[1]
        self.W = self.cav_range[3] - self.cav_range[0]
        self.fake_voxel_size = 1
        self.supervise_single = bool(args.get("supervise_single", False))
        if self.supervise_single:
            self.cls_head_single, self.reg_head_single, self.dir_head_single = anchor_heads(args["in_head_single"], args)
        self.fusion_net = build_fusion(args)
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
        """encoder -> backbone -> shrinker (-> camera crop) for all agents of modality m (:170-196)."""
        f = getattr(self, f"encoder_{m}")(data_dict, m)
        f = getattr(self, f"backbone_{m}")({"spatial_features": f})["spatial_features_2d"]
        return crop_camera_feature(self, m, getattr(self, f"shrinker_{m}")(f))

    def heads(self, fused):
        if self.shrink_flag:
            fused = self.shrink_conv(fused)
        return detection_heads(fused, self.cls_head, self.reg_head, self.dir_head)

    def forward(self, data_dict):
        output_dict = {}
        agent_modality_list = data_dict["agent_modality_list"]
        pairwise, _ = pairwise_to_host(data_dict["pairwise_t_matrix"])
        affine_matrix = normalize_pairwise_tfm(pairwise, self.H, self.W, self.fake_voxel_size)
        record_len = data_dict["record_len"]
        counts = Counter(agent_modality_list)
        feats = {}
        for m in self.modality_name_list:
            if m not in counts:
                continue
            feats[m] = self.encode_modality(data_dict, m)
            if wants_depth_items(self, m):
                output_dict[f"depth_items_{m}"] = getattr(self, f"encoder_{m}").depth_items
        if len(counts) == 1:      # one modality: the encoder's batch IS the agent stack (no copy)
            x = feats[agent_modality_list[0]]
        else:
            cursor = {m: 0 for m in self.modality_name_list}
            parts = []
            for m in agent_modality_list:
                parts.append(feats[m][cursor[m]])
                cursor[m] += 1
            x = torch.stack(parts)
        if self.compress:
            x = self.compressor(x)
        if self.supervise_single:
            output_dict.update({"cls_preds_single": self.cls_head_single(x), "reg_preds_single": self.reg_head_single(x),
                                "dir_preds_single": self.dir_head_single(x)})
        fused = self.fusion_net(x, record_len, affine_matrix)
        cls_preds, reg_preds, dir_preds = self.heads(fused)
        output_dict.update({"cls_preds": cls_preds, "reg_preds": reg_preds, "dir_preds": dir_preds})
        return output_dict
