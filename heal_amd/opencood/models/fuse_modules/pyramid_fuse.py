"""PyramidFusion (reference: opencood/models/fuse_modules/pyramid_fuse.py:65-168) on top of the
fused warp + occupancy-softmax kernel K5 (heal_warp_fuse)."""
import torch
import torch.nn as nn

from heal_amd import ops
from heal_amd.opencood.models.sub_modules.bev_blocks import (Bottleneck, ResNetBEVBackbone, ResNetModified)


def crop_window(H, W, ratio_h, ratio_w):
    """pyramid_fuse.py:151-158: centre window of a camera agent's map whose score is kept."""
    crop_H = H / ratio_h - 4
    crop_W = W / ratio_w - 4
    return (int(H // 2 - crop_H // 2), int(H // 2 + crop_H // 2),
            int(W // 2 - crop_W // 2), int(W // 2 + crop_W // 2))


def weighted_fuse_autograd(x, occ, record_len, affine_matrix, crops=None):
    """Gradient path of pyramid_fuse.py:17-63,139-160 with torch operators: score = sigmoid(occ) + 1e-4 (times the camera
    crop mask when one is given), features and scores of every agent sampled bilinearly in the ego frame (zeros outside),
    softmax over the agents with never-observed positions (score exactly 0) left out, weighted sum."""
    import torch.nn.functional as F
    score = torch.sigmoid(occ) + 1e-4
    if crops is not None:
        keep = torch.ones_like(score)
        for a, c in enumerate(crops):
            if c is not None:
                keep[a] = 0
                keep[a, :, c[0]:c[1], c[2]:c[3]] = 1
        score = score * keep
    out, start = [], 0
    for b, n in enumerate(record_len):
        M = torch.as_tensor(affine_matrix[b][0, :n], dtype=x.dtype, device=x.device)
        xs, ss = x[start:start + n], score[start:start + n]
        grid = F.affine_grid(M, list(xs.shape), align_corners=False)
        feat = F.grid_sample(xs, grid, align_corners=False)
        sc = F.grid_sample(ss, F.affine_grid(M, list(ss.shape), align_corners=False), align_corners=False)
        sc = sc.masked_fill(sc == 0, float("-inf")).softmax(dim=0)
        sc = torch.where(torch.isnan(sc), torch.zeros_like(sc), sc)
        out.append((feat * sc).sum(0))
        start += n
    return torch.stack(out)


class _WarpFuse(torch.autograd.Function):
    """K5 with a hand-written backward: forward = heal_warp_fuse, backward = heal_warp_fuse_backward (one thread per ego pixel
    re-samples every agent, scatters the map gradient through the bilinear taps and pushes the softmax gradient back to the
    occupancy logits).  The warped [n,C,H,W] stack of the reference's autograd is never stored."""

    @staticmethod
    def forward(ctx, x, occ, rows, grid_f64, crop):
        x, occ = x.contiguous(), occ.contiguous()
        ctx.save_for_backward(x, occ)
        ctx.args = (rows, grid_f64, crop)
        return ops.warp_fuse(x, occ, rows, grid_f64, crop)

    @staticmethod
    def backward(ctx, grad_out):
        x, occ = ctx.saved_tensors
        rows, grid_f64, crop = ctx.args
        g_x, g_occ = ops.warp_fuse_backward(x, occ, rows, grad_out.contiguous(), grid_f64, crop)
        return g_x, g_occ, None, None, None


def weighted_fuse(x, occ, record_len, affine_matrix, grid_f64=True, crops=None):
    """pyramid_fuse.py:17-63 with the score construction folded in.

    x [sum(n),C,H,W]; occ [sum(n),1,H,W] occupancy LOGITS; record_len: list of ints;
    affine_matrix: host numpy [B,L,L,2,3]; crops: per-agent (h0,h1,w0,w1) or None."""
    import os
    grad = torch.is_grad_enabled() and (x.requires_grad or occ.requires_grad)
    if grad and not (x.is_cuda and os.environ.get("HEAL_K5_BACKWARD", "1") == "1"):
        return weighted_fuse_autograd(x, occ, record_len, affine_matrix, crops)   # any device: torch operators
    out = []
    start = 0
    for b, n in enumerate(record_len):
        rows = affine_matrix[b][0, :n]
        crop = None
        if crops is not None:
            crop = [crops[start + a] if crops[start + a] is not None else (0, 0, 0, 0) for a in range(n)]
        if grad:   # training on the device: the HIP kernels in both directions
            out.append(_WarpFuse.apply(x[start:start + n], occ[start:start + n], rows, grid_f64, crop))
        else:
            out.append(ops.warp_fuse(x[start:start + n], occ[start:start + n], rows, grid_f64, crop))
        start += n
    return out[0].unsqueeze(0) if len(out) == 1 else torch.stack(out)   # (one scene: no copy)


class PyramidFusion(ResNetBEVBackbone):
    def __init__(self, model_cfg, input_channels=64):
        super().__init__(model_cfg, input_channels)
        if model_cfg["resnext"]:
            # ResNeXt stages: 32 groups, width 4 per group, Bottleneck with expansion 1
            self.resnet = ResNetModified(Bottleneck, model_cfg["layer_nums"], model_cfg["layer_strides"],
                                         model_cfg["num_filters"], inplanes=model_cfg.get("inplanes", 64),
                                         groups=32, width_per_group=4, expansion=1)
        self.align_corners = model_cfg.get("align_corners", False)
        if self.align_corners:
            raise NotImplementedError("align_corners=True fusion is not used by any HEAL config")
        for i in range(self.num_levels):
            setattr(self, f"single_head_{i}", nn.Conv2d(model_cfg["num_filters"][i], 1, kernel_size=1))

    def occupancy_head(self, i, feature):
        """single_head_i (pyramid_fuse.py:89-91): a one-output 1x1 convolution = a channel dot product (heal_channel_dot)."""
        head = getattr(self, f"single_head_{i}")
        if not torch.is_grad_enabled() and ops.channel_dot_supported(feature):
            return ops.channel_dot(feature, head.weight, head.bias)
        return head(feature)

    def forward_single(self, spatial_features):
        feature_list = self.get_multiscale_feature(spatial_features)
        occ_map_list = [self.occupancy_head(i, feature_list[i]) for i in range(self.num_levels)]
        return self.decode_multiscale_feature(feature_list), occ_map_list

    def forward_collab(self, spatial_features, record_len, affine_matrix, agent_modality_list=None,
                       cam_crop_info=None, grid_f64=True):
        """affine_matrix: host numpy [B,L,L,2,3] (normalize_pairwise_tfm of the host pairwise matrix);
        record_len: list of ints."""
        feature_list = self.get_multiscale_feature(spatial_features)
        use_crop = bool(cam_crop_info) and not self.training
        fused_feature_list, occ_map_list = [], []
        import os
        if (len(record_len) == 1 and not torch.is_grad_enabled() and feature_list[0].is_cuda and 1 <= self.num_levels <= 4
                and int(record_len[0]) <= 8 and os.environ.get("HEAL_K5_LEVELS", "1") == "1"):
            # inference, one scene: the three levels are independent once the stages ran -> ONE K5 launch for all of them
            # (heal_warp_fuse_levels; HEAL_K5_LEVELS=0 keeps one heal_warp_fuse launch per level for A/B)
            n = int(record_len[0])
            crops_all = []
            for i in range(self.num_levels):
                occ_map_list.append(self.occupancy_head(i, feature_list[i]))
                crops = None
                if use_crop:
                    _, _, H, W = occ_map_list[i].shape
                    crops = [crop_window(H, W, cam_crop_info[mod][f"crop_ratio_H_{mod}"], cam_crop_info[mod][f"crop_ratio_W_{mod}"])
                             if mod in cam_crop_info else None for mod in agent_modality_list]
                crops_all.append(crops)
            fused = ops.warp_fuse_levels([f[:n] for f in feature_list], [o[:n] for o in occ_map_list], affine_matrix[0][0, :n],
                                         grid_f64, crops_all if use_crop else None)
            return self.decode_multiscale_feature([f.unsqueeze(0) for f in fused]), occ_map_list
        for i in range(self.num_levels):
            occ_map = self.occupancy_head(i, feature_list[i])
            occ_map_list.append(occ_map)
            crops = None
            if use_crop:
                _, _, H, W = occ_map.shape
                crops = []
                for mod in agent_modality_list:
                    if mod in cam_crop_info:
                        crops.append(crop_window(H, W, cam_crop_info[mod][f"crop_ratio_H_{mod}"],
                                                 cam_crop_info[mod][f"crop_ratio_W_{mod}"]))
                    else:
                        crops.append(None)
            fused_feature_list.append(weighted_fuse(feature_list[i], occ_map, record_len, affine_matrix,
                                                    grid_f64, crops))
        return self.decode_multiscale_feature(fused_feature_list), occ_map_list
