"""Dense 2-D BEV building blocks: ResNet / ResNeXt layers, deblocks, shrink head, ConvNeXt aligner,
channel compressor.

Parameter names and shapes follow the reference so its checkpoints load unchanged
(SURVEY 8b): opencood/models/sub_modules/resblock.py:18-219, base_bev_backbone_resnet.py:12-142,
downsample_conv.py:7-49, feature_alignnet.py:12-39, feature_alignnet_modules.py:12-31,299-361,
naive_compress.py:5-31.

Training / fine-tuning (gradients enabled, see `grad_path`): the blocks run conv -> BatchNorm -> ReLU as torch modules with
autograd.  Inference design (eval mode): every Conv+BatchNorm pair is folded into one convolution with bias
(cached, re-folded when a parameter changes), ReLU and the residual add run in place -- one pass
over each BEV map instead of three.  Pointwise, dense 3x3 (padding 1, stride 1 | 2) and 32-group 3x3 convolutions run on
the hand-written fp32-MFMA / stencil kernels of libheal_amd with that epilogue fused; what is left to the library
(MIOpen through torch) are the 7x7 stems and kernel != stride transposed convolutions.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _versions(*tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


class _FoldCache:
    """Folded (weight, bias) of a conv followed by an eval-mode BatchNorm, cached per module pair."""

    def __init__(self):
        self.key = None
        self.value = None

    def get(self, conv, bn, transposed=False):
        tensors = [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = _versions(*tensors)
        if key != self.key:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                shift = bn.bias - bn.running_mean * scale
                if transposed:  # ConvTranspose2d weight is [Cin, Cout/groups, k, k]
                    w = conv.weight * scale.view(1, -1, 1, 1)
                else:
                    w = conv.weight * scale.view(-1, 1, 1, 1)
                b = shift if conv.bias is None else shift + conv.bias * scale
                self.value = (w.contiguous(), b.contiguous())
            self.key = key
        return self.value


import os
_CONV1X1 = os.environ.get("HEAL_CONV1X1", "1") == "1"  # hand-written pointwise conv with fused epilogue (K7c)
_CONV3X3 = os.environ.get("HEAL_CONV3X3", "1") == "1"  # hand-written dense 3x3 conv on fp32 MFMA (0: MIOpen + heal_bias_act, A/B)


def conv_bias_act(x, w, b, stride, padding, dilation=1, groups=1, relu=True, residual=None, out=None):
    """conv2d + per-channel bias (+ residual) (+ ReLU); `out`: optional destination (a batch slice of a stage output).  The ResNeXt 32-group 3x3 convolutions run on the
    hand-written stencil kernel (heal_grouped_conv3x3); everything else is a library convolution WITHOUT
    bias followed by ONE fused in-place pass (heal_bias_act) instead of separate bias / add / ReLU kernels."""
    from heal_amd import ops
    # gradient path (callers that pass raw parameters): also when only the WEIGHTS are trainable -- the HIP output carries no
    # graph, so a frozen trunk in front of a trainable convolution would silently drop dL/dw
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        y = F.conv2d(x, w, b, stride, padding, dilation, groups)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    st = stride if isinstance(stride, int) else stride[0]
    pd = padding if isinstance(padding, int) else padding[0]
    dl = dilation if isinstance(dilation, int) else dilation[0]
    if (groups > 1 and w.shape[2:] == (3, 3) and pd == 1 and dl == 1 and st in (1, 2) and residual is None
            and w.shape[1] in (4, 8, 16) and w.shape[0] == x.shape[1]):
        return ops.grouped_conv3x3(x, w, b, groups, st, relu)
    if _CONV1X1 and groups == 1 and w.shape[2:] == (1, 1) and st in (1, 2) and pd == 0:
        Ho, Wo = (int(x.shape[2]) - 1) // st + 1, (int(x.shape[3]) - 1) // st + 1
        if ops.conv1x1_supported(int(w.shape[1]), int(w.shape[0]), Ho * Wo, st, Wo):
            return ops.conv1x1(x, w, b, residual, 1 if relu else 0, stride=st, out=out)
    if (_CONV3X3 and x.is_cuda and groups == 1 and tuple(w.shape[2:]) == (3, 3) and pd == 1 and dl == 1 and st in (1, 2)
            and (padding if isinstance(padding, int) else padding[1]) == 1
            and (stride if isinstance(stride, int) else stride[1]) == st):
        return ops.conv3x3(x, w, b, residual, relu, st)
    if (x.is_cuda and groups == 1 and tuple(w.shape[2:]) == (7, 7) and st == 2 and pd == 3 and dl == 1 and residual is None
            and ops.conv7x7_s2_supported(int(w.shape[1]), int(w.shape[0]), int(x.shape[3]))):
        return ops.conv7x7_s2(x, w, b, relu)      # BevEncode's stem (lss_submodule.py:242): the last library convolution of any mirrored model
    _library_fallthrough(x, w, stride, padding, dilation, groups)
    y = F.conv2d(x, w, None, stride, padding, dilation, groups)
    if b is None and residual is None and not relu:
        return y
    return ops.bias_act_(y, b, residual, relu)


LIBRARY_CONV_SHAPES = {}   # (Cin, Cout, k, stride, padding, dilation, groups, H, W) -> calls: what reached MIOpen, for inspection


def _library_fallthrough(x, w, stride, padding, dilation, groups):
    """A convolution none of the hand-written kernels takes runs on the library (MIOpen).  That is correct but silent: a config
    change can put library convolutions back on the path unnoticed.  Counted per shape, warned once per shape."""
    key = (int(w.shape[1]) * groups, int(w.shape[0]), tuple(int(v) for v in w.shape[2:]), stride, padding, dilation, groups,
           int(x.shape[2]), int(x.shape[3]))
    n = LIBRARY_CONV_SHAPES.get(key, 0)
    LIBRARY_CONV_SHAPES[key] = n + 1
    if n == 0 and x.is_cuda:
        import warnings
        warnings.warn(f"heal_amd: convolution Cin={key[0]} Cout={key[1]} k={key[2]} stride={stride} pad={padding} dil={dilation} "
                      f"groups={groups} on a {key[7]}x{key[8]} map is not covered by a hand-written kernel: library (MIOpen) path",
                      RuntimeWarning, stacklevel=3)


def grad_path(x, *modules):
    """True when this call has to record an autograd graph: gradients are enabled AND (the input already carries gradient OR
    a module of the block is in training mode OR has a parameter that requires gradient).  The fused inference operators of
    libheal_amd have no backward; on the gradient path every block runs the reference's plain composition of torch operators
    on the SAME parameters (BatchNorm with batch statistics when training), so `opencood/tools/train.py` -- forward, loss,
    backward, optimiser step -- works on the module tree unchanged.  Inference runs under `torch.no_grad()`
    (ScenePipeline, inference_utils) and never takes it."""
    if not torch.is_grad_enabled():
        return False
    if torch.is_tensor(x) and x.requires_grad:
        return True
    for m in modules:
        if m is not None and (m.training or any(p.requires_grad for p in m.parameters())):
            return True
    return False


def _require_eval(module):
    """For the operators that still have no gradient path (the sparse 3-D encoder K3)."""
    if module.training and torch.is_grad_enabled():
        raise NotImplementedError(
            f"{type(module).__name__}: no gradient path (the sparse-convolution encoder is inference-only in this build). "
            "Call model.eval() and run under torch.no_grad(), or freeze this module (eval mode, input without gradient).")


class ConvBN(nn.Module):
    """conv (no bias) + BatchNorm2d (+ ReLU) with reference-compatible child names given by the
    owner; this helper only provides the folded forward."""

    @staticmethod
    def run(x, conv, bn, cache, relu, residual=None, out=None):
        if out is not None and not grad_path(x, bn, conv):      # (inference: the agent-chunked / camera-crop stage walks)
            w, b = cache.get(conv, bn)
            y = conv_bias_act(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups, relu, residual, out=out)
            if y.data_ptr() != out.data_ptr():
                out.copy_(y)
            return out
        if out is not None:      # gradient path with a destination (CPU tests of the stage walks): compute, then copy
            return out.copy_(ConvBN.run(x, conv, bn, cache, relu, residual))
        if grad_path(x, bn, conv):   # training / fine-tuning: conv -> BatchNorm (batch statistics when training) -> + -> ReLU
            y = bn(conv(x))
            if residual is not None:
                y = y + residual
            return F.relu(y) if relu else y
        w, b = cache.get(conv, bn)
        return conv_bias_act(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups, relu, residual)


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups,
                     bias=False, dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    """resblock.py:18-64."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride
        self._c1, self._c2, self._cd = _FoldCache(), _FoldCache(), _FoldCache()

    def takes_pooled(self):
        """True if this block can read K4's sparse pixel-major map (ops.PooledBEV) directly: conv1 3x3 stride 2 pad 1 and a
        1x1 stride-2 downsample, both to 64 channels -- the opening block of the camera modalities' ResNetBEVBackbone
        (heal_bev_stem_block; the dense [C, ny, nx] canvas is then never written)."""
        d = self.downsample
        return (d is not None and isinstance(d[0], nn.Conv2d) and d[0].kernel_size == (1, 1) and d[0].stride == (2, 2)
                and d[0].padding == (0, 0) and d[0].out_channels == 64 and d[0].bias is None
                and self.conv1.stride == (2, 2) and self.conv1.out_channels == 64 and self.conv1.in_channels % 32 == 0
                and isinstance(self.bn1, nn.BatchNorm2d) and isinstance(d[1], nn.BatchNorm2d))

    def _stem_params(self, x):
        from heal_amd import ops
        w1, b1 = self._c1.get(self.conv1, self.bn1)
        wd, bd = self._cd.get(self.downsample[0], self.downsample[1])
        frag = getattr(x, "fragments", ops.stem_fragments)     # the weight layout the sparse input's kernel reads
        key = (self._c1.key, self._cd.key, getattr(x, "weight_layout", "tiles"))
        if getattr(self, "_stem_key", None) != key:
            self._stem = frag(w1, wd) + (b1, bd)
            self._stem_key = key
        return self._stem

    def forward(self, x):
        from heal_amd import ops
        if isinstance(x, (ops.PooledBEV, ops.PillarBEV)):     # a sparse stand-in for the encoder's dense map (K4 / K2)
            if (not grad_path(None, self) and self.takes_pooled() and x.channels == self.conv1.in_channels
                    and x.stem_supported(self.conv1.out_channels, self.downsample[0].out_channels)):
                wm, wd, b1, bd = self._stem_params(x)
                out, identity = x.stem_block(wm, b1, wd, bd)
                return ConvBN.run(out, self.conv2, self.bn2, self._c2, relu=True, residual=identity)
            x = x.dense()
        identity = x
        if self.downsample is not None:
            identity = ConvBN.run(x, self.downsample[0], self.downsample[1], self._cd, relu=False)
        out = ConvBN.run(x, self.conv1, self.bn1, self._c1, relu=True)
        return ConvBN.run(out, self.conv2, self.bn2, self._c2, relu=True, residual=identity)


class Bottleneck(nn.Module):
    """resblock.py:67-122.  PyramidFusion uses it with expansion 1 (pyramid_fuse.py:72)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None, expansion=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        exp = self.expansion if expansion is None else expansion
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * exp)
        self.bn3 = norm_layer(planes * exp)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self._c1, self._c2, self._c3, self._cd = _FoldCache(), _FoldCache(), _FoldCache(), _FoldCache()
        self._fused_key = None
        self._fused = None

    def _fused_params(self):
        """Folded weights of the three convolutions in the layout heal_resnext_bottleneck wants."""
        from heal_amd import ops
        w1, b1 = self._c1.get(self.conv1, self.bn1)
        w2, b2 = self._c2.get(self.conv2, self.bn2)
        w3, b3 = self._c3.get(self.conv3, self.bn3)
        key = (self._c1.key, self._c2.key, self._c3.key)
        if key != self._fused_key:
            self._fused = (ops.mfma_a_fragments(w1.reshape(w1.shape[0], w1.shape[1])), b1, w2, b2,
                           ops.mfma_a_fragments(w3.reshape(w3.shape[0], w3.shape[1])), b3)
            self._fused_key = key
        return self._fused

    def _fusable(self, x):
        # The fused kernel (heal_resnext_bottleneck) is correct but, at one workgroup per CU, still slower
        # than the un-fused sequence ("K7b"): opt-in until it wins.
        import os
        if os.environ.get("HEAL_FUSED_BOTTLENECK", "0") != "1":
            return False
        from heal_amd import ops
        if not ops.experimental_build():      # the kernel ships only in a HEAL_BUILD_EXPERIMENTAL=1 library
            return False
        c = self.conv1.in_channels
        return (x.is_cuda and self.downsample is None and self.stride == 1 and self.conv2.groups == 32
                and self.conv3.out_channels == c and self.conv1.out_channels == 2 * c and c in (64, 128, 256)
                and self.conv2.dilation == (1, 1))

    def out_shape(self, x):
        s = self.stride
        return (int(x.shape[0]), self.conv3.out_channels, (int(x.shape[2]) - 1) // s + 1, (int(x.shape[3]) - 1) // s + 1)

    def forward(self, x, out=None):
        """out: optional destination of the block's result (inference; the agent-chunked stage walk of ResNetModified)."""
        if not grad_path(x, self) and self._fusable(x):
            from heal_amd import ops
            y = ops.resnext_bottleneck(x.contiguous(), *self._fused_params())
            return y if out is None else out.copy_(y)
        identity = x
        if self.downsample is not None:
            identity = ConvBN.run(x, self.downsample[0], self.downsample[1], self._cd, relu=False)
        y = ConvBN.run(x, self.conv1, self.bn1, self._c1, relu=True)
        y = ConvBN.run(y, self.conv2, self.bn2, self._c2, relu=True)
        return ConvBN.run(y, self.conv3, self.bn3, self._c3, relu=True, residual=identity, out=out)


class ResNetModified(nn.Module):
    """resblock.py:125-219: `layer{i}` = one stage of `layers[i]` blocks with stride layer_strides[i]."""

    def __init__(self, block, layers, layer_strides, num_filters, groups=1, width_per_group=64, inplanes=64,
                 expansion=None):
        super().__init__()
        self.inplanes = inplanes
        self.groups = groups
        self.base_width = width_per_group
        self.layernum = len(num_filters)
        self._exp = block.expansion if expansion is None else expansion
        for i in range(self.layernum):
            setattr(self, f"layer{i}", self._make_layer(block, num_filters[i], layers[i], layer_strides[i]))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, block, planes, blocks, stride):
        exp = self._exp
        kw = {"expansion": exp} if block is Bottleneck else {}
        downsample = None
        if stride != 1 or self.inplanes != planes * exp:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * exp, stride), nn.BatchNorm2d(planes * exp))
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width, 1, None, **kw)]
        self.inplanes = planes * exp
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width, **kw))
        return nn.Sequential(*layers)

    @staticmethod
    def stage_chunk(layer, x):
        """Agents per chunk of the depth-first stage walk, or 0 for the plain layer-by-layer order.

        A ResNeXt stage on the wide BEV maps is HBM-bound when every launch streams ALL agents (level 0 of the fusion pyramid, 5
        agents: 84 MB in, 168 MB of 2C-wide intermediate written and read twice -- 924 MB per block against 256 MB of Infinity
        Cache), although one agent's whole block (16.8 + 33.5 + 33.5 + 16.8 MB) fits the cache.  Walking the stage agent chunk by
        agent chunk -- every block of the stage on chunk 0, then chunk 1, ... -- keeps a chunk's intermediates on the die between
        the producing and the consuming launch.  HEAL_STAGE_CHUNK_MB: the working-set budget per chunk (0: off)."""
        mb = float(os.environ.get("HEAL_STAGE_CHUNK_MB", "0"))
        if mb <= 0 or not x.is_cuda or torch.is_grad_enabled() or not all(isinstance(b, Bottleneck) for b in layer):
            return 0
        n = int(x.shape[0])
        blk = layer[-1]
        co, s0 = blk.conv3.out_channels, layer[0].stride
        ho, wo = (int(x.shape[2]) - 1) // s0 + 1, (int(x.shape[3]) - 1) // s0 + 1
        per_agent = 4.0 * ho * wo * (2 * co + 2 * blk.conv1.out_channels) / 2 ** 20      # in + out + the 2C-wide pair, MB
        chunk = max(1, int(mb // per_agent))
        return chunk if chunk < n else 0

    def run_stage(self, layer, x):
        chunk = self.stage_chunk(layer, x)
        if not chunk:
            return layer(x)
        n = int(x.shape[0])
        shape = list(x.shape)
        for blk in layer:
            shape = list(blk.out_shape(torch.empty(shape, device="meta")))
        y = torch.empty(shape, dtype=x.dtype, device=x.device)
        for a in range(0, n, chunk):
            xa = x[a:a + chunk]
            for j, blk in enumerate(layer):
                xa = blk(xa, out=y[a:a + chunk]) if j == len(layer) - 1 else blk(xa)
        return y

    def forward(self, x):
        feats = []
        for i in range(self.layernum):
            x = self.run_stage(getattr(self, f"layer{i}"), x)
            feats.append(x)
        return feats


class _Deblock(nn.Sequential):
    """ConvTranspose2d(k=s, stride=s) | Conv2d + BatchNorm2d(eps 1e-3) + ReLU, folded at inference
    (base_bev_backbone_resnet.py:49-74)."""

    def __init__(self, conv, bn):
        super().__init__(conv, bn, nn.ReLU())
        self._cache = _FoldCache()

    def out_shape(self, x):
        """(channels, H, W) of the output for input x (what decode_multiscale_feature sizes the concatenated tensor with)."""
        conv = self[0]
        if isinstance(conv, nn.ConvTranspose2d):
            return conv.out_channels, int(x.shape[2]) * conv.stride[0], int(x.shape[3]) * conv.stride[1]
        s0, s1 = conv.stride
        return conv.out_channels, (int(x.shape[2]) - conv.kernel_size[0]) // s0 + 1, (int(x.shape[3]) - conv.kernel_size[1]) // s1 + 1

    def forward(self, x, into=None):
        """into = (dst [n, Ctot, Ho, Wo], channel offset): write the result into that channel slice of the concatenated tensor
        (by the convolution's own epilogue when the deblock is a kernel == stride transposed convolution) and return the slice."""
        y = self._forward(x, into)
        if into is not None and y.data_ptr() != into[0][:, into[1]:].data_ptr():
            dst = into[0][:, into[1]:into[1] + y.shape[1]]
            dst.copy_(y)
            return dst
        return y

    def _forward(self, x, into=None):
        if grad_path(x, self):
            return super().forward(x)   # (transposed) conv -> BatchNorm -> ReLU as torch modules
        conv, bn = self[0], self[1]
        from heal_amd import ops
        if isinstance(conv, nn.ConvTranspose2d):
            w, b = self._cache.get(conv, bn, transposed=True)
            k = conv.kernel_size[0]
            if (_CONV1X1 and x.is_cuda and conv.kernel_size == conv.stride and conv.kernel_size[0] == conv.kernel_size[1]
                    and conv.padding == (0, 0) and conv.output_padding == (0, 0) and conv.groups == 1
                    and ops.conv1x1_supported(int(w.shape[0]), int(w.shape[1]) * k * k, int(x.shape[2] * x.shape[3]))):
                # kernel == stride: the transposed convolution is a pointwise convolution to Cout*k*k channels followed
                # by a depth-to-space shuffle; bias and ReLU commute with the shuffle, so they ride in the conv1x1
                # epilogue (the library path is GEMM + col2im + a bias/ReLU pass)
                key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
                if getattr(self, "_ps_key", None) != key:
                    cin, cout = int(w.shape[0]), int(w.shape[1])
                    self._ps = (w.permute(1, 2, 3, 0).reshape(cout * k * k, cin, 1, 1).contiguous(),
                                b.repeat_interleave(k * k).contiguous())
                    self._ps_key = key
                if into is not None and int(x.shape[3]) % 4 == 0 and into[0].is_contiguous():
                    # bias, ReLU, the depth-to-space shuffle AND the concatenation ride in the conv1x1 epilogue
                    return ops.conv1x1_d2s(x, self._ps[0], self._ps[1], 1, k, into[0], into[1])
                y = ops.conv1x1(x, self._ps[0], self._ps[1], None, 1)
                return y if k == 1 else F.pixel_shuffle(y, k)
            y = F.conv_transpose2d(x, w, None, conv.stride, conv.padding, conv.output_padding, conv.groups)
            return ops.bias_act_(y, b, None, True)
        w, b = self._cache.get(conv, bn)
        return conv_bias_act(x, w, b, conv.stride, conv.padding, 1, 1, True)


class ResNetBEVBackbone(nn.Module):
    """base_bev_backbone_resnet.py:12-142."""

    def __init__(self, model_cfg, input_channels=64):
        super().__init__()
        self.model_cfg = model_cfg
        if "layer_nums" in model_cfg:
            layer_nums = model_cfg["layer_nums"]
            layer_strides = model_cfg["layer_strides"]
            num_filters = model_cfg["num_filters"]
            assert len(layer_nums) == len(layer_strides) == len(num_filters)
        else:
            layer_nums = layer_strides = num_filters = []
        if "upsample_strides" in model_cfg:
            assert len(model_cfg["upsample_strides"]) == len(model_cfg["num_upsample_filter"])
            num_upsample_filters = model_cfg["num_upsample_filter"]
            upsample_strides = model_cfg["upsample_strides"]
        else:
            upsample_strides = num_upsample_filters = []
        self.resnet = ResNetModified(BasicBlock, layer_nums, layer_strides, num_filters,
                                     inplanes=model_cfg.get("inplanes", 64))
        self.num_levels = len(layer_nums)
        self.deblocks = nn.ModuleList()
        for idx in range(self.num_levels):
            if len(upsample_strides) > 0:
                stride = upsample_strides[idx]
                if stride >= 1:
                    conv = nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx], stride, stride=stride,
                                              bias=False)
                else:
                    stride = int(np.round(1 / stride))
                    conv = nn.Conv2d(num_filters[idx], num_upsample_filters[idx], stride, stride=stride, bias=False)
                self.deblocks.append(_Deblock(conv, nn.BatchNorm2d(num_upsample_filters[idx], eps=1e-3, momentum=0.01)))
        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > self.num_levels:
            self.deblocks.append(_Deblock(
                nn.ConvTranspose2d(c_in, c_in, upsample_strides[-1], stride=upsample_strides[-1], bias=False),
                nn.BatchNorm2d(c_in, eps=1e-3, momentum=0.01)))
        self.num_bev_features = c_in

    def takes_pooled(self):
        """True if the first block reads K4's sparse pixel-major map directly (BasicBlock.takes_pooled)."""
        first = getattr(self.resnet, "layer0", None)
        return first is not None and len(first) > 0 and isinstance(first[0], BasicBlock) and first[0].takes_pooled()

    def get_multiscale_feature(self, spatial_features):
        return self.resnet(spatial_features)

    def decode_multiscale_feature(self, x):
        ups = []
        if (len(self.deblocks) >= self.num_levels > 1 and x[0].is_cuda and not grad_path(x[0], self)
                and len({self.deblocks[i].out_shape(x[i])[1:] for i in range(self.num_levels)}) == 1):
            # inference: every deblock writes its channel slice of the concatenated tensor itself (no torch.cat pass)
            shapes = [self.deblocks[i].out_shape(x[i]) for i in range(self.num_levels)]
            cat = torch.empty((int(x[0].shape[0]), sum(s[0] for s in shapes), shapes[0][1], shapes[0][2]),
                              dtype=x[0].dtype, device=x[0].device)
            off = 0
            for i in range(self.num_levels):
                self.deblocks[i](x[i], into=(cat, off))
                off += shapes[i][0]
            ups = [cat]
        else:
            for i in range(self.num_levels):
                ups.append(self.deblocks[i](x[i]) if len(self.deblocks) > 0 else x[i])
        x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        if len(self.deblocks) > self.num_levels:
            x = self.deblocks[-1](x)
        return x

    def get_layer_i_feature(self, spatial_features, layer_i):
        return getattr(self.resnet, f"layer{layer_i}")(spatial_features)

    def forward(self, data_dict):
        x = self.resnet(data_dict["spatial_features"])
        data_dict["spatial_features_2d"] = self.decode_multiscale_feature(x)
        return data_dict


class DoubleConv(nn.Module):
    """downsample_conv.py:7-27: conv(k,s,p)+ReLU, conv3x3+ReLU, both with bias, no BN."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1),
            nn.ReLU(inplace=True))

    def forward(self, x):
        if grad_path(x, self):
            return self.double_conv(x)
        c0, c1 = self.double_conv[0], self.double_conv[2]
        x = conv_bias_act(x, c0.weight, c0.bias, c0.stride, c0.padding, 1, 1, True)
        return conv_bias_act(x, c1.weight, c1.bias, c1.stride, c1.padding, 1, 1, True)


class DownsampleConv(nn.Module):
    """downsample_conv.py:30-49 (config keys incl. the reference's spelling 'kernal_size')."""

    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([])
        input_dim = config["input_dim"]
        for ksize, dim, stride, padding in zip(config["kernal_size"], config["dim"], config["stride"],
                                               config["padding"]):
            self.layers.append(DoubleConv(input_dim, dim, kernel_size=ksize, stride=stride, padding=padding))
            input_dim = dim

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class LayerNorm(nn.Module):
    """feature_alignnet_modules.py:12-31."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_last":
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class ConvNeXtBlock(nn.Module):
    """feature_alignnet_modules.py:299-344 (deform=False, drop_path=0)."""

    def __init__(self, dim, layer_scale_init_value=1e-6, kernel_size=7):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=kernel_size, padding=kernel_size // 2, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)), requires_grad=True) \
            if layer_scale_init_value > 0 else None

    def _folded_pw2(self):
        """pwconv2 with the layer scale folded in: gamma * (W2 h + b2) = (gamma . W2) h + gamma . b2 (cached)."""
        t = [self.pwconv2.weight, self.pwconv2.bias] + ([self.gamma] if self.gamma is not None else [])
        key = _versions(*t)
        if getattr(self, "_pw2_key", None) != key:
            with torch.no_grad():
                w, b = self.pwconv2.weight, self.pwconv2.bias
                if self.gamma is not None:
                    w, b = w * self.gamma[:, None], b * self.gamma
                self._pw2 = (w.contiguous()[:, :, None, None], b.contiguous())
            self._pw2_key = key
        return self._pw2

    def forward(self, x):
        from heal_amd import ops
        inp = x
        k = self.dwconv.kernel_size[0]
        fused = x.is_cuda and not grad_path(x, self)   # the gradient path is the reference's composition below
        if fused and k == 7 and int(x.shape[0] * x.shape[1]) <= 65535:
            x = ops.depthwise_conv(x, self.dwconv.weight, self.dwconv.bias, 1, (3, 3, 3, 3), "none")
        else:
            x = self.dwconv(x)
        if fused and ops.conv1x1_supported(x.shape[1], 4 * x.shape[1], int(x.shape[2] * x.shape[3])):
            # NCHW all the way: channel LayerNorm in one pass, the two Linear layers as pointwise convolutions with
            # GELU / (layer scale + residual) fused -- 3 launches instead of permute, LN, 2 GEMMs, GELU, scale, add
            xn = ops.layernorm_nchw(x, self.norm.weight, self.norm.bias, self.norm.eps)
            h = ops.conv1x1(xn, self.pwconv1.weight[:, :, None, None], self.pwconv1.bias, None, 3)
            w2, b2 = self._folded_pw2()
            return ops.conv1x1(h, w2, b2, inp, 0)
        x = x.permute(0, 2, 3, 1)
        x = self.pwconv2(self.act(self.pwconv1(self.norm(x))))
        if self.gamma is not None:
            x = self.gamma * x
        return inp + x.permute(0, 3, 1, 2)


class ConvNeXt(nn.Module):
    """feature_alignnet_modules.py:346-361."""

    def __init__(self, args):
        super().__init__()
        if args.get("deform", False):
            raise NotImplementedError("deformable ConvNeXt aligner is out of scope (needs mmcv)")
        self.model = nn.Sequential(*[ConvNeXtBlock(args["dim"], kernel_size=args.get("kernel_size", 7))
                                     for _ in range(args["num_of_blocks"])])

    def forward(self, x):
        return self.model(x)


class AlignNet(nn.Module):
    """feature_alignnet.py:12-39; the HEAL configs use 'identity' and 'convnext'."""

    def __init__(self, args):
        super().__init__()
        name = args["core_method"]
        if name == "identity":
            self.channel_align = nn.Identity()
        elif name == "convnext":
            self.channel_align = ConvNeXt(args["args"])
        else:
            raise NotImplementedError(f"aligner '{name}' is outside the hot-path scope (SURVEY 2, row 6)")
        if args.get("spatial_align", False):
            raise NotImplementedError

    def forward(self, x):
        return self.channel_align(x)


class NaiveCompressor(nn.Module):
    """naive_compress.py:5-31."""

    def __init__(self, input_dim, compress_raito):
        super().__init__()
        mid = input_dim // compress_raito
        self.encoder = nn.Sequential(nn.Conv2d(input_dim, mid, 3, 1, 1), nn.BatchNorm2d(mid, eps=1e-3, momentum=0.01),
                                     nn.ReLU())
        self.decoder = nn.Sequential(nn.Conv2d(mid, input_dim, 3, 1, 1),
                                     nn.BatchNorm2d(input_dim, eps=1e-3, momentum=0.01), nn.ReLU(),
                                     nn.Conv2d(input_dim, input_dim, 3, 1, 1),
                                     nn.BatchNorm2d(input_dim, eps=1e-3, momentum=0.01), nn.ReLU())
        self._c = [_FoldCache() for _ in range(3)]

    def encode(self, x):
        """The half that runs on the SENDING agent: [n, C, H, W] -> [n, C / ratio, H, W] (what travels, naive_compress.py:25)."""
        return ConvBN.run(x, self.encoder[0], self.encoder[1], self._c[0], relu=True)

    def decode(self, z):
        """The half that runs on the RECEIVING agent (naive_compress.py:26-29)."""
        z = ConvBN.run(z, self.decoder[0], self.decoder[1], self._c[1], relu=True)
        return ConvBN.run(z, self.decoder[3], self.decoder[4], self._c[2], relu=True)

    def forward(self, x):
        return self.decode(self.encode(x))
