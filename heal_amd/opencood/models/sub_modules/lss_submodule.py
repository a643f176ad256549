"""Camera feature extractors of the Lift-Splat encoder (reference: opencood/models/sub_modules/
lss_submodule.py:17-233): `Up`, `CamEncode` (EfficientNet-b0 trunk), `CamEncode_Resnet101`, and
`BevEncode` (:236-273, the resnet18-trunk BEV decoder of the old-style `lift_splat_shoot.py` model).

The reference takes the two trunks from third-party packages that are not in its tree
(efficientnet_pytorch==0.7.0, torchvision); they are restated here from the packages' published
architecture with the same parameter names, so that HEAL checkpoints (`encoder_m2.camencode.trunk.*`,
`encoder_m4.camencode.layer1.*`) load.  PARITY UNPINNED for the trunks: the build pins the
path from the (depth_logit, image feature) boundary onward (K4).  Unlike the reference, the modules
return the depth logits and the C-channel image features and never form the [BN,C,D,fH,fW] lifted
tensor -- the outer product is fused into heal_bev_pool.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from heal_amd.opencood.models.sub_modules.bev_blocks import (BasicBlock, Bottleneck, ConvBN, _FoldCache, conv_bias_act,
                                                             grad_path)


class Up(nn.Module):
    """lss_submodule.py:17-36: bilinear x2 (align_corners=True), concat, 2 x [3x3 conv + BN + ReLU]."""

    def __init__(self, in_channels, out_channels, scale_factor=2):
        super().__init__()
        self.up = nn.Upsample(scale_factor=scale_factor, mode="bilinear", align_corners=True)
        self.conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True))
        self._c = [_FoldCache(), _FoldCache()]

    def forward(self, x1, x2):
        from heal_amd import ops
        fused = x1.is_cuda and self.up.scale_factor == 2 and not grad_path(x1, self)
        up = ops.upsample2x_bilinear(x1) if fused else self.up(x1)
        x = torch.cat([x2, up], dim=1)
        x = ConvBN.run(x, self.conv[0], self.conv[1], self._c[0], relu=True)
        return ConvBN.run(x, self.conv[3], self.conv[4], self._c[1], relu=True)


# ------------------------------------------------------------------------------------------------
# EfficientNet-b0 (efficientnet_pytorch 0.7.0 layout: static "same" padding computed for a 224 px
# image at construction, BN eps 1e-3, swish, squeeze-excite ratio 0.25 of the block INPUT filters)
# ------------------------------------------------------------------------------------------------
_B0_BLOCKS = [  # repeats, kernel, stride, expand, in, out
    (1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


class _SamePadConv2d(nn.Conv2d):
    """Conv2dStaticSamePadding: TF-style SAME padding fixed at construction for `image_size`."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, groups=1, bias=False, image_size=224):
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, padding=0, groups=groups, bias=bias)
        ih = iw = image_size
        kh = kw = kernel_size
        sh = sw = stride
        oh, ow = math.ceil(ih / sh), math.ceil(iw / sw)
        pad_h = max((oh - 1) * sh + (kh - 1) + 1 - ih, 0)
        pad_w = max((ow - 1) * sw + (kw - 1) + 1 - iw, 0)
        self.same_pad = (pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2)
        self.static_padding = nn.ZeroPad2d(self.same_pad) if (pad_h > 0 or pad_w > 0) else nn.Identity()


def _conv_bn(x, conv, bn, cache, act, in_scale=None, residual=None, channel_sums=False):
    """conv + folded BatchNorm (+ SiLU).  Pointwise convolutions run on heal_conv1x1 with the squeeze-excite gate
    (in_scale, per image and input channel), the bias, the skip connection and the activation fused."""
    if grad_path(x, bn, conv):   # gradient path: the package's own composition (scale, pad, conv, BatchNorm, swish, skip)
        if in_scale is not None:
            x = in_scale * x
        y = bn(conv(conv.static_padding(x)))
        y = F.silu(y) if act else y
        return y + residual if residual is not None else y
    w, b = cache.get(conv, bn)
    if (x.is_cuda and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1
            and not any(conv.same_pad)):
        from heal_amd import ops
        if ops.conv1x1_supported(conv.in_channels, conv.out_channels, int(x.shape[2] * x.shape[3])):
            return ops.conv1x1(x, w, b, residual, 2 if act else 0, in_scale)
    if in_scale is not None:
        x = in_scale * x
    if (x.is_cuda and conv.groups == conv.in_channels == conv.out_channels and conv.kernel_size[0] in (3, 5)
            and conv.stride[0] in (1, 2) and x.shape[0] * conv.in_channels <= 65535):
        from heal_amd import ops
        return ops.depthwise_conv(x.contiguous(), w, b, conv.stride[0], conv.same_pad, "silu" if act else "none",
                                  channel_sums=channel_sums)
    if (x.is_cuda and conv.groups == 1 and conv.kernel_size == (3, 3) and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
            and conv.same_pad[0] in (0, 1) and conv.same_pad[2] in (0, 1) and conv.same_pad[1] <= 1 and conv.same_pad[3] <= 1):
        from heal_amd import ops   # the 3x3 / stride-2 stem: "same" padding, bias and SiLU in the convolution's own epilogue
        return ops.conv3x3_same(x.contiguous(), w, b, conv.stride[0], conv.same_pad, "silu" if act else "none")
    if any(conv.same_pad):
        x = F.pad(x, conv.same_pad)
    y = F.conv2d(x, w, b, conv.stride, 0, 1, conv.groups)
    y = F.silu(y, inplace=True) if act else y
    return y + residual if residual is not None else y


class _MBConv(nn.Module):
    def __init__(self, kernel, stride, expand, inp, oup, image_size):
        super().__init__()
        self.expand = expand
        self.id_skip = stride == 1 and inp == oup
        mid = inp * expand
        if expand != 1:
            self._expand_conv = _SamePadConv2d(inp, mid, 1, image_size=image_size)
            self._bn0 = nn.BatchNorm2d(mid, momentum=0.01, eps=1e-3)
        self._depthwise_conv = _SamePadConv2d(mid, mid, kernel, stride=stride, groups=mid, image_size=image_size)
        self._bn1 = nn.BatchNorm2d(mid, momentum=0.01, eps=1e-3)
        sq = max(1, int(inp * 0.25))
        out_size = math.ceil(image_size / stride)
        self._se_reduce = _SamePadConv2d(mid, sq, 1, bias=True, image_size=1)
        self._se_expand = _SamePadConv2d(sq, mid, 1, bias=True, image_size=1)
        self._project_conv = _SamePadConv2d(mid, oup, 1, image_size=out_size)
        self._bn2 = nn.BatchNorm2d(oup, momentum=0.01, eps=1e-3)
        self._c = [_FoldCache(), _FoldCache(), _FoldCache()]

    def forward(self, x):
        inp = x
        if self.expand != 1:
            x = _conv_bn(x, self._expand_conv, self._bn0, self._c[0], act=True)
        from heal_amd import ops
        dw = self._depthwise_conv
        if (x.is_cuda and not grad_path(x, self) and dw.kernel_size[0] in (3, 5) and dw.stride[0] in (1, 2)
                and x.shape[0] * dw.in_channels <= 65535):
            # the squeeze (spatial mean) rides in the depthwise launch as per-tile sums; the gate kernel adds them up and scales
            x, sums = _conv_bn(x, dw, self._bn1, self._c[1], act=True, channel_sums=True)
            gate = ops.se_gate(sums, self._se_reduce.weight, self._se_reduce.bias, self._se_expand.weight,
                               self._se_expand.bias, scale=1.0 / float(x.shape[2] * x.shape[3]), tiles=int(sums.shape[2]))
            return _conv_bn(x, self._project_conv, self._bn2, self._c[2], act=False, in_scale=gate[:, :, None, None],
                            residual=inp if self.id_skip else None)
        x = _conv_bn(x, dw, self._bn1, self._c[1], act=True)
        # efficientnet_pytorch MBConvBlock: s = expand(silu(reduce(avgpool(x)))); x = sigmoid(s) * x; project; (+ skip)
        if grad_path(x, self):
            s_ = self._se_expand(F.silu(self._se_reduce(x.mean((2, 3), keepdim=True))))
            gate = torch.sigmoid(s_)[:, :, 0, 0]
        else:
            gate = ops.se_gate(x.mean((2, 3)), self._se_reduce.weight, self._se_reduce.bias, self._se_expand.weight,
                               self._se_expand.bias)
        return _conv_bn(x, self._project_conv, self._bn2, self._c[2], act=False, in_scale=gate[:, :, None, None],
                        residual=inp if self.id_skip else None)


class EfficientNetB0(nn.Module):
    def __init__(self, image_size=224):
        super().__init__()
        self._conv_stem = _SamePadConv2d(3, 32, 3, stride=2, image_size=image_size)
        self._bn0 = nn.BatchNorm2d(32, momentum=0.01, eps=1e-3)
        size = math.ceil(image_size / 2)
        blocks = []
        for rep, k, s, e, i, o in _B0_BLOCKS:
            for r in range(rep):
                blocks.append(_MBConv(k, s if r == 0 else 1, e, i if r == 0 else o, o, size))
                if r == 0:
                    size = math.ceil(size / s)
        self._blocks = nn.ModuleList(blocks)
        self._conv_head = _SamePadConv2d(320, 1280, 1, image_size=size)  # present for checkpoint parity, unused
        self._bn1 = nn.BatchNorm2d(1280, momentum=0.01, eps=1e-3)
        self._fc = nn.Linear(1280, 1000)
        self._c0 = _FoldCache()

    def endpoints(self, x):
        """lss_submodule.py:87-107: feature maps just before every spatial reduction, plus the last."""
        out = {}
        x = _conv_bn(x, self._conv_stem, self._bn0, self._c0, act=True)
        prev = x
        for block in self._blocks:
            x = block(x)
            if prev.size(2) > x.size(2):
                out[f"reduction_{len(out) + 1}"] = prev
            prev = x
        out[f"reduction_{len(out) + 1}"] = x
        return out


def _bin_depths_lid_ud(depth_map, mode, depth_min, depth_max, num_bins):
    """camera_utils.py:137-185 with target=False (inference): -> (indices int64, valid mask)."""
    if mode == "UD":
        bin_size = (depth_max - depth_min) / num_bins
        indices = (depth_map - depth_min) / bin_size
    elif mode == "LID":
        bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
        indices = -0.5 + 0.5 * torch.sqrt(1 + 8 * (depth_map - depth_min) / bin_size)
    else:
        raise NotImplementedError(mode)
    mask = (indices < 0) | (indices >= num_bins) | (~torch.isfinite(indices))
    indices = indices.clone()
    indices[indices < 0] = 0
    indices[indices >= num_bins] = num_bins - 1
    indices[~torch.isfinite(indices)] = num_bins - 1
    return indices.type(torch.int64), ~mask


class _CamEncodeBase(nn.Module):
    def _init_common(self, D, C, downsample, ddiscr, mode, use_gt_depth, depth_supervision):
        self.D, self.C, self.downsample = D, C, downsample
        self.d_min, self.d_max, self.num_bins = ddiscr[0], ddiscr[1], ddiscr[2]
        self.mode = mode
        self.use_gt_depth = use_gt_depth
        self.depth_supervision = depth_supervision
        if use_gt_depth:
            raise NotImplementedError("use_gt_depth is a training-time ablation outside the hot path")

    def gt_depth_indices(self, x):
        """lss_submodule.py:66-85 (eval): bin the 4th image channel and sub-sample to the feature grid."""
        d = x[:, 3, :, :].clamp_max(self.d_max)
        idx, _ = _bin_depths_lid_ud(d, self.mode, self.d_min, self.d_max, self.num_bins)
        s = self.downsample
        return idx[:, s // 2::s, s // 2::s]

    def heads(self, features, x):
        x_img = self.image_head(features)
        depth_logit = self.depth_head(features)
        items = None
        if self.depth_supervision:
            items = (depth_logit, self.gt_depth_indices(x) if x.shape[1] > 3 else None)
        return items, depth_logit, x_img

    def _fused_head(self):
        """image_head | depth_head as ONE [C + D, 512] pointwise weight (cached per parameter version)."""
        hs = (self.image_head, self.depth_head)
        key = tuple((h.weight.data_ptr(), h.weight._version, h.bias.data_ptr(), h.bias._version) for h in hs)
        hit = self.__dict__.get("_heal_fused_head")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, torch.cat([h.weight for h in hs], 0).contiguous(), torch.cat([h.bias for h in hs], 0).contiguous())
            self.__dict__["_heal_fused_head"] = hit  # plain attribute: not a parameter, not in the state_dict
        return hit[1], hit[2]

    def heads_pixel_major(self, features, x):
        """The two 1x1 heads (lss_submodule.py:113,127) as one convolution that writes PIXEL-MAJOR [BN, fH*fW, C + D] (a pixel's
        C image features, then its D depth logits: one contiguous row) -- the layout K4 (heal_bev_pool_pm) reads, so the lift
        needs no transposition pass.  -> (depth_items | None, head).  `depth_items[0]` is the [BN,D,fH,fW] VIEW of the logits."""
        from heal_amd import ops
        w, b = self._fused_head()
        head = ops.conv1x1(features, w, b, None, 0, pixel_major=True)
        self.last_feature_hw = (int(features.shape[2]), int(features.shape[3]))   # what the trunk produced (ceil(H/8), not H//8)
        items = None
        if self.depth_supervision:
            BN, _, fH, fW = features.shape
            depth_logit = head[:, :, self.C:].view(BN, fH, fW, self.D).permute(0, 3, 1, 2)
            items = (depth_logit, self.gt_depth_indices(x) if x.shape[1] > 3 else None)
        return items, head

    def pixel_major_ok(self, features):
        from heal_amd import ops
        return (features.is_cuda and not grad_path(features, self) and (self.C + self.D) % 4 == 0
                and ops.conv1x1_supported(features.shape[1], self.C + self.D, int(features.shape[2] * features.shape[3]))
                and ops.bev_pool_pm_supported(self.D, int(features.shape[2]), self.C))

    def forward(self, x, pixel_major=False):
        """x [BN, 3|4, H, W] -> (depth_items | None, depth_logit [BN,D,fH,fW], x_img [BN,C,fH,fW]); with pixel_major=True and
        a shape the fused path takes: (depth_items | None, head [BN, fH*fW, C + D])."""
        f = self.features(x)
        if pixel_major and self.pixel_major_ok(f):
            return self.heads_pixel_major(f, x)
        return self.heads(f, x)


class CamEncode(_CamEncodeBase):
    """lss_submodule.py:39-138 with the EfficientNet-b0 trunk."""

    def __init__(self, D, C, downsample, ddiscr, mode, use_gt_depth=False, depth_supervision=True):
        super().__init__()
        self._init_common(D, C, downsample, ddiscr, mode, use_gt_depth, depth_supervision)
        self.trunk = EfficientNetB0()
        self.up1 = Up(320 + 112, 512)
        if downsample == 8:
            self.up2 = Up(512 + 40, 512)
        self.depth_head = nn.Conv2d(512, self.D, kernel_size=1, padding=0)
        self.image_head = nn.Conv2d(512, self.C, kernel_size=1, padding=0)

    def features(self, x):
        """lss_submodule.py:87-111: trunk endpoints -> Up (-> Up) -> [BN, 512, fH, fW]."""
        ep = self.trunk.endpoints(x[:, :3, :, :])
        f = self.up1(ep["reduction_5"], ep["reduction_4"])
        if self.downsample == 8:
            f = self.up2(f, ep["reduction_3"])
        return f


class CamEncode_Resnet101(_CamEncodeBase):
    """lss_submodule.py:140-233: torchvision resnet101 stem + layer1 + layer2 (/8, 512 channels)."""

    def __init__(self, D, C, downsample, ddiscr, mode, use_gt_depth=False, depth_supervision=True):
        super().__init__()
        self._init_common(D, C, downsample, ddiscr, mode, use_gt_depth, depth_supervision)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 64, 3, 1)
        self.layer2 = self._make_layer(256, 128, 4, 2)
        self.layer3 = nn.Identity()
        self.depth_head = nn.Conv2d(512, self.D, kernel_size=1, padding=0)
        self.image_head = nn.Conv2d(512, self.C, kernel_size=1, padding=0)
        self._c = _FoldCache()

    @staticmethod
    def _make_layer(inplanes, planes, blocks, stride):
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(inplanes, planes, stride, down, expansion=4)]
        layers += [Bottleneck(planes * 4, planes, expansion=4) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def features(self, x):
        """lss_submodule.py:196-210: conv1 -> bn1 -> relu -> maxpool -> layer1 -> layer2 -> [BN, 512, fH, fW]."""
        if x.is_cuda and not grad_path(x, self) and tuple(self.maxpool.kernel_size if isinstance(self.maxpool.kernel_size, tuple)
                                                           else (self.maxpool.kernel_size,) * 2) == (3, 3):
            # conv1 (7x7 / 2) + bn1 + relu + maxpool (3x3 / 2) in ONE kernel on the first three channels of the image tensor read
            # in place (heal_stem7x7): no channel-slice copy, no library convolution, the 64-channel half-resolution map never
            # reaches HBM
            from heal_amd import ops
            w, b = self._c.get(self.conv1, self.bn1)
            return self.layer2(self.layer1(ops.stem7x7(x, w, b, pool=True)))
        f = ConvBN.run(x[:, :3, :, :], self.conv1, self.bn1, self._c, relu=True)
        return self.layer2(self.layer1(self.maxpool(f)))


class BevEncode(nn.Module):
    """lss_submodule.py:236-273: 7x7/2 stem + torchvision resnet18 layer1..3 (BasicBlocks; restated with the package's
    parameter names `layerK.i.{conv1,bn1,conv2,bn2,downsample.0,downsample.1}`), `Up(64+256 -> 256, x4)`, then
    x2 bilinear -> 3x3 conv + BN + ReLU -> 1x1 conv.  All conv+BN pairs run folded (eval only)."""

    def __init__(self, inC, outC):
        super().__init__()
        self.conv1 = nn.Conv2d(inC, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(64, 64, 1)
        self.layer2 = self._make_layer(64, 128, 2)
        self.layer3 = self._make_layer(128, 256, 2)
        self.up1 = Up(64 + 256, 256, scale_factor=4)
        self.up2 = nn.Sequential(
            nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
            nn.Conv2d(256, 128, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
            nn.Conv2d(128, outC, kernel_size=1, padding=0))
        self._c = [_FoldCache(), _FoldCache()]

    @staticmethod
    def _make_layer(inplanes, planes, stride):
        down = None
        if stride != 1 or inplanes != planes:
            down = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        return nn.Sequential(BasicBlock(inplanes, planes, stride, down), BasicBlock(planes, planes))

    def forward(self, x):
        from heal_amd import ops
        x = ConvBN.run(x, self.conv1, self.bn1, self._c[0], relu=True)
        x1 = self.layer1(x)
        x = self.layer3(self.layer2(x1))
        x = self.up1(x, x1)
        x = ops.upsample2x_bilinear(x) if (x.is_cuda and not grad_path(x, self)) else self.up2[0](x)
        x = ConvBN.run(x, self.up2[1], self.up2[2], self._c[1], relu=True)
        last = self.up2[4]
        return conv_bias_act(x, last.weight, last.bias, last.stride, last.padding, 1, 1, False)
