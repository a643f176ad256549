"""Shared construction logic of the heterogeneous models (reference: the per-modality loops at
heter_pyramid_collab.py:35-77, heter_pyramid_single.py:30-63, heter_model_late.py:26-69)."""
import importlib
from collections import OrderedDict

import torch
import torch.nn.functional as F


def modality_names(args):
    return [x for x in args.keys() if x.startswith("m") and x[1:].isdigit()]


def find_encoder(core_method):
    lib = importlib.import_module("heal_amd.opencood.models.heter_encoders")
    target = core_method.replace("_", "").lower()
    for name, cls in lib.__dict__.items():
        if name.lower() == target:
            return cls
    raise KeyError(f"encoder '{core_method}' not found in heter_encoders")


def modality_stems(model, args, make_backbone):
    """The per-modality part all heterogeneous models share: `encoder_mX`, `depth_supervision_mX`, `backbone_mX` and, for
    camera modalities, the crop ratios that bring the camera BEV grid to the LiDAR range.  A generator: after each
    modality's stem the caller registers its own per-modality modules (aligner, shrinker, heads ...), which keeps the
    registration order -- and therefore parameter order -- of the reference's loops."""
    model.modality_name_list = modality_names(args)
    model.cav_range = args["lidar_range"]
    model.sensor_type_dict = OrderedDict()
    for m in model.modality_name_list:
        setting = args[m]
        model.sensor_type_dict[m] = setting["sensor_type"]
        setattr(model, f"encoder_{m}", find_encoder(setting["core_method"])(setting["encoder_args"]))
        setattr(model, f"depth_supervision_{m}", bool(setting["encoder_args"].get("depth_supervision", False)))
        setattr(model, f"backbone_{m}", make_backbone(setting))
        enc, bb = getattr(model, f"encoder_{m}"), getattr(model, f"backbone_{m}")
        if hasattr(enc, "emit_pooled") and getattr(bb, "takes_pooled", lambda: False)():
            enc.emit_pooled = True      # K4 hands its sparse pixel-major map straight to the backbone's first block
        if setting["sensor_type"] == "camera":
            grid = setting["camera_mask_args"]["grid_conf"]
            setattr(model, f"crop_ratio_W_{m}", model.cav_range[3] / grid["xbound"][1])
            setattr(model, f"crop_ratio_H_{m}", model.cav_range[4] / grid["ybound"][1])
        yield m, setting


def crop_camera_feature(model, m, feature):
    """Camera BEV maps cover the camera grid; keep the centre that corresponds to the LiDAR range (no-op for LiDAR)."""
    if model.sensor_type_dict[m] != "camera":
        return feature
    H, W = (int(v) for v in feature.shape[-2:])
    th, tw = int(H * getattr(model, f"crop_ratio_H_{m}")), int(W * getattr(model, f"crop_ratio_W_{m}"))
    # where the map's content lies after the call when BOTH axes are zero-padded (center_crop's pad rule): everything outside this
    # box is exactly zero -- PyramidFusion's camera-crop walk (pyramid_fuse.py) skips the work whose result that fixes in advance
    boxes = model.__dict__.setdefault("_heal_cam_boxes", {})
    if th > H and tw > W:
        boxes[m] = ((th - H) // 2, (th - H) // 2 + H, (tw - W) // 2, (tw - W) // 2 + W)
    else:
        boxes.pop(m, None)
    return center_crop(feature, th, tw)


def wants_depth_items(model, m):
    """True for a camera modality configured with depth supervision: the model output then carries the encoder's
    `depth_items` (depth_logit, depth_gt_indices) under `depth_items_mX`."""
    return model.sensor_type_dict[m] == "camera" and bool(getattr(model, f"depth_supervision_{m}"))


def anchor_heads(in_channels, args):
    """cls / reg / dir 1x1 heads with the reference's channel counts."""
    import torch.nn as nn
    a = args["anchor_number"]
    return (nn.Conv2d(in_channels, a, kernel_size=1), nn.Conv2d(in_channels, 7 * a, kernel_size=1),
            nn.Conv2d(in_channels, args["dir_args"]["num_bins"] * a, kernel_size=1))


def center_crop(x, target_h, target_w):
    """torchvision.transforms.CenterCrop((th,tw)) on [...,H,W]: zero-pad when the
    target is larger (left/top floor, right/bottom ceil), then crop at round((H-th)/2)."""
    H, W = x.shape[-2:]
    if target_w > W or target_h > H:
        pl = (target_w - W) // 2 if target_w > W else 0
        pt = (target_h - H) // 2 if target_h > H else 0
        pr = (target_w - W + 1) // 2 if target_w > W else 0
        pb = (target_h - H + 1) // 2 if target_h > H else 0
        x = F.pad(x, [pl, pr, pt, pb])
        H, W = x.shape[-2:]
    top = int(round((H - target_h) / 2.0))
    left = int(round((W - target_w) / 2.0))
    return x[..., top:top + target_h, left:left + target_w]


def record_len_to_list(record_len):
    if isinstance(record_len, torch.Tensor):
        return [int(v) for v in record_len.detach().cpu().tolist()]
    return [int(v) for v in record_len]


def detection_heads(x, cls_head, reg_head, dir_head):
    """cls / reg / dir 1x1 heads (heter_pyramid_collab.py:102-107,198-200) as ONE pointwise convolution over the
    concatenated output channels (the input map is read once instead of three times); returns the three channel
    slices.  The concatenated weight is cached per parameter version."""
    from heal_amd import ops
    heads = (cls_head, reg_head, dir_head)
    if torch.is_grad_enabled() and (x.requires_grad or cls_head.training or cls_head.weight.requires_grad):   # gradient path
        return cls_head(x), reg_head(x), dir_head(x)
    if not (x.is_cuda and ops.conv1x1_supported(x.shape[1], 1, int(x.shape[2] * x.shape[3]))
            and all(h.kernel_size == (1, 1) and h.stride == (1, 1) for h in heads)):
        return cls_head(x), reg_head(x), dir_head(x)
    key = tuple((h.weight.data_ptr(), h.weight._version, h.bias.data_ptr(), h.bias._version) for h in heads)
    hit = cls_head.__dict__.get("_heal_fused_heads")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            w = torch.cat([h.weight for h in heads], 0).contiguous()
            b = torch.cat([h.bias for h in heads], 0).contiguous()
        hit = (key, w, b)
        cls_head.__dict__["_heal_fused_heads"] = hit  # plain attribute: not a parameter, not in the state_dict
    y = ops.conv1x1(x, hit[1], hit[2], None, 0)
    c0, c1 = cls_head.out_channels, cls_head.out_channels + reg_head.out_channels
    return y[:, :c0], y[:, c0:c1], y[:, c1:]


def encode_modalities(model, data_dict, present, encode):
    """Run `encode(data_dict, m)` for every modality in `present` (model order) -> {m: features}.

    The per-modality stems are independent until the fusion backbone, and two of them are long chains of small, latency-bound
    kernels (the EfficientNet / ResNet image trunks at 1/8 .. 1/32 resolution) that leave most of the chip idle: on a HIP
    device with more than one modality present each stem runs on its OWN side stream (fork: the side stream waits for the
    caller's stream; join: the caller's stream waits for every side stream), so they overlap -- also inside a captured HIP graph,
    where the fork / join become parallel branches.  Allocator discipline: a stem's tensors are allocated on its side stream
    and only its OUTPUT crosses to the caller's stream after the join; every forward starts with the fork's wait, so a block
    the side stream reuses is never still read by the caller's previous work.  `HEAL_PARALLEL_MODALITIES=0` serialises."""
    import os
    from heal_amd import ops
    mods = [m for m in model.modality_name_list if m in present]
    dev = next(model.parameters()).device
    # Round 6: camera modalities whose lift + splat (K4) can share ONE launch: their encoders run up to the heads (forward_head ->
    # PendingPool), then ops.bev_pool_pm_multi, then each modality's backbone / aligner (encode_modality_tail).  HEAL_K4_MULTI=0: off.
    # MEASURED NEGATIVE at the scene level (profiles/r06_k4_shared_launch.json): the streams have to meet for the launch, and the extra
    # branches stop the two frames in flight from overlapping (6.15 -> 7.4 ms per step) -- opt-in, and only in an experimental build.
    two = (dev.type == "cuda" and not torch.is_grad_enabled() and os.environ.get("HEAL_K4_MULTI", "0") == "1" and ops.experimental_build()
           and hasattr(model, "encode_modality_tail") and getattr(encode, "__func__", None) is getattr(model.encode_modality, "__func__", 0))
    cams = [m for m in mods if two and hasattr(getattr(model, f"encoder_{m}"), "forward_head")]
    if len(cams) < 2:
        cams = []
    pend = {}

    def first_half(m):
        if m in cams:
            pend[m] = model.encode_modality_head(data_dict, m)
            return None
        return encode(data_dict, m)

    def pool_all():
        items = [pend[m] for m in cams]
        deferred = [p for p in items if hasattr(p, "finish")]
        pooled = iter(ops.bev_pool_pm_multi([p.args for p in deferred]) if len(deferred) >= 2 else [p.finish() for p in deferred])
        return [next(pooled) if hasattr(p, "finish") else p for p in items]

    if (len(mods) < 2 or dev.type != "cuda" or os.environ.get("HEAL_PARALLEL_MODALITIES", "1") != "1"
            or torch.is_grad_enabled()):   # (a gradient path stays on one stream)
        feats = {m: first_half(m) for m in mods}
        if cams:
            for m, r in zip(cams, pool_all()):
                feats[m] = model.encode_modality_tail(m, r)
        return feats
    main = torch.cuda.current_stream(dev)
    streams = model.__dict__.setdefault("_heal_side_streams", {})
    feats = {}
    # side streams per CALLER stream: two captured copies of the step that run concurrently (pipeline.FramesInFlight) must
    # not share them -- the operators' scratch buffers are per stream
    st = {mods[0]: main}
    for m in mods[1:]:
        s = streams.get((m, dev.index, main.cuda_stream))
        if s is None:
            s = streams[(m, dev.index, main.cuda_stream)] = torch.cuda.Stream(device=dev)
        st[m] = s
        s.wait_stream(main)
        with torch.cuda.stream(s):
            feats[m] = first_half(m)
    feats[mods[0]] = first_half(mods[0])       # the first modality stays on the caller's stream
    if cams:
        # the shared launch runs on its OWN stream once every camera head is there; the camera modalities' streams then wait for it and
        # carry on with their backbones.  (A head tensor allocated on stream A and read on the pool stream is safe: A waits for the pool
        # stream before it allocates again.  HEAL_K4_MULTI_TAIL=serial runs the tails on the pool stream instead, one after the other.)
        key = ("k4pool", dev.index, main.cuda_stream)
        P = streams.get(key)
        if P is None:
            P = streams[key] = torch.cuda.Stream(device=dev)
        for m in cams:
            P.wait_stream(st[m])
        with torch.cuda.stream(P):
            results = pool_all()
        # The tails run on FRESH streams forked from the pool stream (the first on the pool stream itself), never on the head streams: a
        # stream that another capturing stream has already waited for and that then waits back and carries on made hipStreamEndCapture
        # segfault (ROCm 7.2; eager execution was fine) -- measured with the tails on the head streams and on the pool stream.
        tails = []
        for k, (m, r) in enumerate(zip(cams, results)):
            if k == 0:
                T = P
            else:
                tk = ("k4tail", m, dev.index, main.cuda_stream)
                T = streams.get(tk)
                if T is None:
                    T = streams[tk] = torch.cuda.Stream(device=dev)
                T.wait_stream(P)
                tails.append(T)
            with torch.cuda.stream(T):
                feats[m] = model.encode_modality_tail(m, r)
        for T in tails:
            main.wait_stream(T)
        main.wait_stream(P)
    for m in mods[1:]:
        main.wait_stream(st[m])
    return {m: feats[m] for m in mods}
