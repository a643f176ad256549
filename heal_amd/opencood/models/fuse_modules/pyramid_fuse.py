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


def _stage_influence(lo, hi, layer, size_in):
    """[lo, hi) of the stage INPUT that differs from the all-zero input -> ([lo', hi') of the stage output that can differ from the
    zero-input output, output size).  Every Bottleneck is 1x1 -> 3x3 (stride s, padding 1) -> 1x1 (+ a 1x1 stride-s downsample):
    output o reads inputs s o - 1 .. s o + 1."""
    size = size_in
    for blk in layer:
        s_ = blk.stride
        size = (size - 1) // s_ + 1
        lo = max(0, -((1 - lo) // s_))         # ceil((lo - 1) / s)
        hi = min(size, hi // s_ + 1)           # floor((hi - 1 + 1) / s) + 1, exclusive
    return lo, hi, size


def _stage_needs(olo, ohi, layer, size_in):
    """[olo, ohi) of the stage OUTPUT -> the [ilo, ihi) of its INPUT those outputs read (the receptive field, clipped to the map)."""
    for blk in reversed(list(layer)):
        s_ = blk.stride
        olo, ohi = s_ * olo - 1, s_ * (ohi - 1) + 2
    return max(0, olo), min(size_in, ohi)


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

    # ---- camera agents: their padded maps are ZERO outside the centre box (round 6) ------------------------------------------------
    # The camera modalities' BEV maps cover the camera grid (+-51.2 m) and are zero-padded to the LiDAR range before the fusion
    # pyramid (heter_pyramid_collab.py:153-167, torchvision CenterCrop: 128^2 -> 256^2), so 75 % of a camera agent's pyramid input is
    # exactly zero -- and the reference still pushes it through every ResNeXt block (pyramid_fuse.py:71-79,139).  A convolution stack
    # is local: an output pixel whose receptive field holds only zero input equals the stack's output for an ALL-ZERO map at that
    # position.  So for the camera agents each stage runs on a CROP (the box its content can influence + the receptive-field halo,
    # aligned to 8 pixels) and the rest of the stage output is the zero-input response `BG`, computed once per weight version
    # (a frame-independent constant).  Exact: the pasted box is computed by the same kernels from the same values; only work whose
    # result is known in advance is skipped.  Level 0: 144^2 of 256^2 pixels, level 1: 88^2 of 128^2; the last level's content box
    # already reaches the borders and runs in full.  HEAL_PYRAMID_CAMCROP=0 switches it off.
    def _zero_response_key(self, like):
        return (tuple((p.data_ptr(), p._version) for p in self.resnet.parameters()) +
                tuple((b.data_ptr(), b._version) for b in self.resnet.buffers()), tuple(like.shape[1:]), str(like.device))

    def _zero_response(self, like):
        """Stage outputs of the pyramid for an all-zero [1, C, H, W] input (cached per parameter version and map size)."""
        key = self._zero_response_key(like)
        if getattr(self, "_bg_key", None) != key:
            with torch.no_grad():
                z = torch.zeros((1,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
                self._bg = [f.clone() for f in self.resnet(z)]
            self._bg_key = key
        return self._bg

    def _camcrop_plan(self, x, cam_slice, box):
        """Per stage: None (run the stage in full) or (crop of the stage input (y0, y1, x0, x1), output box to paste (global), the same
        box in the crop's output coordinates).  box = (y0, y1, x0, x1) of the non-zero region of the camera agents' input."""
        H, W = int(x.shape[2]), int(x.shape[3])
        ly, hy, lx, hx = box
        plan = []
        for i in range(self.layernum_):
            layer = getattr(self.resnet, f"layer{i}")
            s_tot = 1
            for blk in layer:
                s_tot *= blk.stride
            oly, ohy, Ho = _stage_influence(ly, hy, layer, H)
            olx, ohx, Wo = _stage_influence(lx, hx, layer, W)
            iy0, iy1 = _stage_needs(oly, ohy, layer, H)
            ix0, ix1 = _stage_needs(olx, ohx, layer, W)
            a = 8 * s_tot                      # crop origin / size: multiples of 8 output pixels (kernel tile and 16-B row constraints)
            cy0, cy1, cx0, cx1 = iy0 // a * a, min(H, -(-iy1 // a) * a), ix0 // a * a, min(W, -(-ix1 // a) * a)
            if (cy1 - cy0) * (cx1 - cx0) > 0.7 * H * W:
                plan.append(None)              # the content reaches (nearly) everywhere: no saving in cropping this stage
            else:
                plan.append(((cy0, cy1, cx0, cx1), (oly, ohy, olx, ohx),
                             (oly - cy0 // s_tot, ohy - cy0 // s_tot, olx - cx0 // s_tot, ohx - cx0 // s_tot)))
            ly, hy, lx, hx, H, W = oly, ohy, olx, ohx, Ho, Wo
        return plan

    def get_multiscale_feature_camcrop(self, x, cam_slice, box):
        """get_multiscale_feature for a scene whose agents [cam_slice] are camera agents with non-zero input inside `box` only (and
        whose other agents form ONE contiguous range)."""
        n = int(x.shape[0])
        c0, c1 = cam_slice
        other = (0, c0) if c0 > 0 else (c1, n)
        bg = self._zero_response(x)
        plan = self._camcrop_plan(x, cam_slice, box)
        feats = []
        for i in range(self.layernum_):
            layer = getattr(self.resnet, f"layer{i}")
            if plan[i] is None:
                x = self.resnet.run_stage(layer, x)
                feats.append(x)
                continue
            (cy0, cy1, cx0, cx1), (gy0, gy1, gx0, gx1), (py0, py1, px0, px1) = plan[i]
            full = torch.empty((n,) + tuple(bg[i].shape[1:]), dtype=x.dtype, device=x.device)
            if other[1] > other[0]:            # the LiDAR agents: the whole map, written straight into their slice
                xa = x[other[0]:other[1]]
                for j, blk in enumerate(layer):
                    xa = blk(xa, out=full[other[0]:other[1]]) if j == len(layer) - 1 else blk(xa)
            yc = layer(x[c0:c1, :, cy0:cy1, cx0:cx1].contiguous())
            full[c0:c1] = bg[i]                # (broadcast copy) the zero-input response everywhere ...
            full[c0:c1, :, gy0:gy1, gx0:gx1] = yc[:, :, py0:py1, px0:px1]   # ... and the box the content reaches
            x = full
            feats.append(x)
        return feats

    @property
    def layernum_(self):
        return self.resnet.layernum

    def _camcrop_args(self, spatial_features, agent_modality_list, cam_boxes):
        """(cam_slice, box) when the camera-crop walk applies: inference on a HIP device, ResNeXt stages, the camera agents form one
        contiguous range at an end of the scene and share one valid box."""
        import os
        if (not cam_boxes or agent_modality_list is None or torch.is_grad_enabled() or not spatial_features.is_cuda
                or os.environ.get("HEAL_PYRAMID_CAMCROP", "1") != "1"):
            return None
        cams = [k for k, m in enumerate(agent_modality_list) if m in cam_boxes]
        n = len(agent_modality_list)
        if not cams or cams != list(range(cams[0], cams[-1] + 1)) or (cams[0] != 0 and cams[-1] != n - 1):
            return None
        from heal_amd.opencood.models.sub_modules.bev_blocks import Bottleneck
        if not all(isinstance(b, Bottleneck) and b.conv2.kernel_size == (3, 3) and b.conv2.padding == (1, 1)
                   for i in range(self.layernum_) for b in getattr(self.resnet, f"layer{i}")):
            return None
        boxes = [cam_boxes[agent_modality_list[k]] for k in cams]
        box = (min(b[0] for b in boxes), max(b[1] for b in boxes), min(b[2] for b in boxes), max(b[3] for b in boxes))
        H, W = int(spatial_features.shape[2]), int(spatial_features.shape[3])
        if H % 32 or W % 32 or not (0 <= box[0] < box[1] <= H and 0 <= box[2] < box[3] <= W):
            return None
        if (torch.cuda.is_current_stream_capturing()
                and getattr(self, "_bg_key", None) != self._zero_response_key(spatial_features[:1])):
            return None     # the zero response would be computed INSIDE the capture (valid only after a replay): plain walk for this graph
        return (cams[0], cams[-1] + 1), box

    def multiscale(self, spatial_features, agent_modality_list=None, cam_boxes=None):
        """get_multiscale_feature, by the camera-crop walk where it applies (cam_boxes: see forward_collab)."""
        cc = self._camcrop_args(spatial_features, agent_modality_list, cam_boxes)
        if cc is not None:
            return self.get_multiscale_feature_camcrop(spatial_features, *cc)
        return self.get_multiscale_feature(spatial_features)

    def forward_collab(self, spatial_features, record_len, affine_matrix, agent_modality_list=None,
                       cam_crop_info=None, grid_f64=True, cam_boxes=None):
        """affine_matrix: host numpy [B,L,L,2,3] (normalize_pairwise_tfm of the host pairwise matrix);
        record_len: list of ints.  cam_boxes: {modality: (y0, y1, x0, x1)} -- the caller's promise that the maps of those (camera)
        modalities are exactly zero outside the box (it zero-padded them itself): enables the camera-crop stage walk."""
        feature_list = self.multiscale(spatial_features, agent_modality_list, cam_boxes if len(record_len) == 1 else None)
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
