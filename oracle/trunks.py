"""trunks -- TEST INFRASTRUCTURE ONLY (oracle/).

Plain torch-CPU restatements of the two third-party image trunks the reference imports but that are absent from its tree
and from this image: `efficientnet_pytorch.EfficientNet` (pinned 0.7.0, "efficientnet-b0") and
`torchvision.models.resnet.resnet101`, with the packages' parameter names.  PARITY UNPINNED: restated from the packages'
published architectures (nn.Conv2d / nn.BatchNorm2d / F.pad, un-fused), no golden vector exists for them.

Two users, both test-side:
  * tests/golden/gen_golden.py::gen_hetero_small injects them as the reference's `EfficientNet` / `resnet101` so that the
    REFERENCE's own `CamEncode` / `CamEncode_Resnet101` / `LiftSplatShoot` / `HeterPyramidCollab` run end to end on CPU in
    the build container: endpoint selection, `Up`, the 1x1 heads, depth softmax, the lift outer product, voxel pooling,
    backbones, aligners, the camera crop and the pyramid fusion are then pinned by that fixture;
  * oracle/model_ref.py uses them for the camera agents of the CPU port (bench.py's `cpu_baseline`, parity tests).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------------
# efficientnet_pytorch 0.7.0, "efficientnet-b0": width 1.0, depth 1.0, image_size 224, BN momentum 0.01 / eps 1e-3,
# drop_connect_rate 0.2 (inactive in eval mode), squeeze-excite ratio 0.25 of the block's INPUT filters.
# ---------------------------------------------------------------------------------------------------------------------
_B0 = [  # repeats, kernel, stride, expand, in, out
    (1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


class Conv2dStaticSamePadding(nn.Conv2d):
    """TF "SAME" padding computed once for `image_size` (the package fixes it at construction)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, image_size=None, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size, stride, **kwargs)
        ih = iw = image_size
        kh, kw = self.weight.size()[-2:]
        sh, sw = self.stride
        oh, ow = math.ceil(ih / sh), math.ceil(iw / sw)
        pad_h = max((oh - 1) * sh + (kh - 1) * self.dilation[0] + 1 - ih, 0)
        pad_w = max((ow - 1) * sw + (kw - 1) * self.dilation[1] + 1 - iw, 0)
        if pad_h > 0 or pad_w > 0:
            self.static_padding = nn.ZeroPad2d((pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2))
        else:
            self.static_padding = nn.Identity()

    def forward(self, x):
        x = self.static_padding(x)
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class MBConvBlock(nn.Module):
    def __init__(self, kernel, stride, expand, inp, oup, image_size):
        super().__init__()
        self.expand_ratio, self.stride, self.inp, self.oup = expand, stride, inp, oup
        mid = inp * expand
        if expand != 1:
            self._expand_conv = Conv2dStaticSamePadding(inp, mid, 1, image_size=image_size, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, momentum=0.01, eps=1e-3)
        self._depthwise_conv = Conv2dStaticSamePadding(mid, mid, kernel, stride, image_size=image_size, groups=mid,
                                                       bias=False)
        self._bn1 = nn.BatchNorm2d(mid, momentum=0.01, eps=1e-3)
        image_size = math.ceil(image_size / stride)
        sq = max(1, int(inp * 0.25))
        self._se_reduce = Conv2dStaticSamePadding(mid, sq, 1, image_size=1)
        self._se_expand = Conv2dStaticSamePadding(sq, mid, 1, image_size=1)
        self._project_conv = Conv2dStaticSamePadding(mid, oup, 1, image_size=image_size, bias=False)
        self._bn2 = nn.BatchNorm2d(oup, momentum=0.01, eps=1e-3)

    @staticmethod
    def _swish(x):
        return x * torch.sigmoid(x)

    def forward(self, inputs, drop_connect_rate=None):
        x = inputs
        if self.expand_ratio != 1:
            x = self._swish(self._bn0(self._expand_conv(x)))
        x = self._swish(self._bn1(self._depthwise_conv(x)))
        s = F.adaptive_avg_pool2d(x, 1)
        s = self._se_expand(self._swish(self._se_reduce(s)))
        x = torch.sigmoid(s) * x
        x = self._bn2(self._project_conv(x))
        if self.stride == 1 and self.inp == self.oup:
            assert not self.training, "stand-in is eval-only (drop_connect is a training-time op)"
            x = x + inputs
        return x


class EfficientNet(nn.Module):
    """The attributes lss_submodule.py:87-107 touches: _conv_stem, _bn0, _swish, _blocks, _global_params."""

    def __init__(self, image_size=224):
        super().__init__()
        self._global_params = SimpleNamespace(drop_connect_rate=0.2)
        self._conv_stem = Conv2dStaticSamePadding(3, 32, 3, 2, image_size=image_size, bias=False)
        self._bn0 = nn.BatchNorm2d(32, momentum=0.01, eps=1e-3)
        size = math.ceil(image_size / 2)
        blocks = []
        for rep, k, s, e, i, o in _B0:
            for r in range(rep):
                blocks.append(MBConvBlock(k, s if r == 0 else 1, e, i if r == 0 else o, o, size))
                if r == 0:
                    size = math.ceil(size / s)
        self._blocks = nn.ModuleList(blocks)
        self._conv_head = Conv2dStaticSamePadding(320, 1280, 1, image_size=size, bias=False)
        self._bn1 = nn.BatchNorm2d(1280, momentum=0.01, eps=1e-3)
        self._fc = nn.Linear(1280, 1000)

    @staticmethod
    def _swish(x):
        return x * torch.sigmoid(x)

    @classmethod
    def from_pretrained(cls, name):
        assert name == "efficientnet-b0"
        return cls()


# ---------------------------------------------------------------------------------------------------------------------
# torchvision.models.resnet.resnet101 (v1.5 Bottleneck: the stride sits on the 3x3 convolution); only conv1, bn1,
# maxpool, layer1, layer2 are attached by the reference (lss_submodule.py:153-161).
# ---------------------------------------------------------------------------------------------------------------------
class _TVBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class _TVResNet(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        out = [_TVBottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        out += [_TVBottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*out)


def resnet101(pretrained=False, zero_init_residual=False, **kw):
    return _TVResNet([3, 4, 23, 3])
