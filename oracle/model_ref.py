"""model_ref -- TEST INFRASTRUCTURE ONLY.

Functional CPU restatement of the reference's model forward for the PointPillars + PyramidFusion
path, driven by a plain state_dict (reference key names).  Dense convolutions are plain torch-CPU
fp32 ops (F.conv2d / F.batch_norm, un-fused, in the reference's order); the sparse / scatter / warp
parts come from oracle_np.  Used by tests and as bench.py's `cpu_baseline` (kind "port").

Follows: opencood/models/heter_pyramid_collab.py:133-209, heter_pyramid_single.py:99-136,
sub_modules/base_bev_backbone_resnet.py:88-136, resblock.py:42-122, fuse_modules/pyramid_fuse.py:
104-168, sub_modules/downsample_conv.py:22-49; for the camera agents lss_submodule.py:17-36,87-138,
196-233, heter_encoders.py:110-241, feature_alignnet_modules.py:12-31,299-361.

`heter_pyramid_collab` (any mix of m1 PointPillars / Lift-Splat camera agents) is pinned by
tests/golden/hetero_small.npz -- outputs of the REFERENCE's own HeterPyramidCollab at reduced size
(tests/test_oracle_golden.py::test_hetero_oracle_model_matches_reference).  The image trunks
(oracle/trunks.py) are restated from the third-party packages' published architectures: PARITY UNPINNED.

`heter_model_baseline` (BASELINE config 5: encoder -> BaseBEVBackbone -> shrinker -> V2X-ViT -> heads; heter_model_baseline.py:
155-236) is pinned with PointPillars agents by tests/golden/baseline_small.npz (the REFERENCE's own HeterModelBaseline); its
SECOND encoder (oracle_np.second_backbone_sparse) restates spconv 1.2.1, which is not in the reference tree: PARITY UNPINNED for
that encoder (held against the dense restatement and analytic cases only).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle_np as O


def _bn(x, sd, p, eps):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def _basic_block(x, sd, p, stride):
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1", 1e-5))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2", 1e-5)
    idt = x
    if p + ".downsample.0.weight" in sd:
        idt = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1", 1e-5)
    return F.relu(out + idt)


def _bottleneck(x, sd, p, stride, groups=32):
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1", 1e-5))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], None, stride, 1, 1, groups), sd, p + ".bn2", 1e-5))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3", 1e-5)
    idt = x
    if p + ".downsample.0.weight" in sd:
        idt = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1", 1e-5)
    return F.relu(out + idt)


def _stage(x, sd, p, n_blocks, stride, block):
    for j in range(n_blocks):
        x = block(x, sd, f"{p}.{j}", stride if j == 0 else 1)
    return x


def _deblock(x, sd, p, stride):
    y = F.conv_transpose2d(x, sd[p + ".0.weight"], None, stride)
    return F.relu(_bn(y, sd, p + ".1", 1e-3))


def _double_conv(x, sd, p, k_stride=1, k_pad=1):
    x = F.relu(F.conv2d(x, sd[p + ".double_conv.0.weight"], sd[p + ".double_conv.0.bias"], k_stride, k_pad))
    return F.relu(F.conv2d(x, sd[p + ".double_conv.2.weight"], sd[p + ".double_conv.2.bias"], 1, 1))


def pointpillar_encoder(sd, prefix, voxels, coords, num, voxel_size, lidar_range, n_agents, ny, nx):
    p = prefix + ".pillar_vfe.pfn_layers.0."
    canvas, _ = O.pfn_scatter(voxels, coords, num, sd[p + "linear.weight"].numpy(), sd[p + "norm.weight"].numpy(),
                              sd[p + "norm.bias"].numpy(), sd[p + "norm.running_mean"].numpy(),
                              sd[p + "norm.running_var"].numpy(), voxel_size, lidar_range, n_agents, ny, nx)
    return torch.from_numpy(canvas)


def heter_pyramid_collab_m1(sd, cfg_args, voxels, coords, num, n_agents, pairwise_t_matrix):
    """One scene, every agent of modality m1 (PointPillars).  Returns dict of numpy outputs."""
    sd = {k: v.detach().cpu().float() if v.dtype.is_floating_point else v.detach().cpu() for k, v in sd.items()}
    r = cfg_args["lidar_range"]
    enc = cfg_args["m1"]["encoder_args"]
    vs = enc["voxel_size"]
    nx = int(round((r[3] - r[0]) / vs[0]))
    ny = int(round((r[4] - r[1]) / vs[1]))
    with torch.no_grad():
        x = pointpillar_encoder(sd, "encoder_m1", voxels, coords, num, vs, r, n_agents, ny, nx)
        bb = cfg_args["m1"]["backbone_args"]
        for i, (nb, st) in enumerate(zip(bb["layer_nums"], bb["layer_strides"])):
            x = _stage(x, sd, f"backbone_m1.resnet.layer{i}", nb, st, _basic_block)
        fb = cfg_args["fusion_backbone"]
        feats = []
        for i, (nb, st) in enumerate(zip(fb["layer_nums"], fb["layer_strides"])):
            x = _stage(x, sd, f"pyramid_backbone.resnet.layer{i}", nb, st, _bottleneck)
            feats.append(x)
        H = r[4] - r[1]
        W = r[3] - r[0]
        aff = O.normalize_pairwise_tfm(np.asarray(pairwise_t_matrix), H, W, 1)
        fused, occs = [], []
        for i, f in enumerate(feats):
            occ = F.conv2d(f, sd[f"pyramid_backbone.single_head_{i}.weight"], sd[f"pyramid_backbone.single_head_{i}.bias"])
            occs.append(occ.numpy())
            score = O.occ_to_score(occ.numpy())
            fused.append(torch.from_numpy(O.weighted_fuse(f.numpy(), score, aff[0][0, :n_agents]))[None])
        ups = [_deblock(fused[i], sd, f"pyramid_backbone.deblocks.{i}", s) for i, s in enumerate(fb["upsample_strides"])]
        y = torch.cat(ups, dim=1)
        y = _double_conv(y, sd, "shrink_conv.layers.0")
        out = {"cls_preds": F.conv2d(y, sd["cls_head.weight"], sd["cls_head.bias"]).numpy(),
               "reg_preds": F.conv2d(y, sd["reg_head.weight"], sd["reg_head.bias"]).numpy(),
               "dir_preds": F.conv2d(y, sd["dir_head.weight"], sd["dir_head.bias"]).numpy(),
               "occ_single_list": occs}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# heterogeneous scene: LiDAR PointPillars agents + Lift-Splat camera agents (BASELINE config 4)
# ---------------------------------------------------------------------------------------------------------------------
def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _up(x1, x2, sd, p, scale=2):
    """lss_submodule.py:17-36."""
    x1 = F.interpolate(x1, scale_factor=scale, mode="bilinear", align_corners=True)
    x = torch.cat([x2, x1], dim=1)
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv.0.weight"], None, 1, 1), sd, p + ".conv.1", 1e-5))
    return F.relu(_bn(F.conv2d(x, sd[p + ".conv.3.weight"], None, 1, 1), sd, p + ".conv.4", 1e-5))


def cam_encode(sd, prefix, kind, imgs, downsample=8):
    """CamEncode / CamEncode_Resnet101 up to the (depth_logit, x_img) boundary.  imgs [BN, 3|4, H, W] torch."""
    from . import trunks
    x = imgs[:, :3]
    if kind == "EfficientNet":
        trunk = trunks.EfficientNet()
        trunk.load_state_dict(_sub(sd, prefix + ".trunk."))
        trunk.eval()
        ep = {}
        x = trunk._swish(trunk._bn0(trunk._conv_stem(x)))
        prev = x
        for blk in trunk._blocks:  # lss_submodule.py:94-101: keep the map just before every spatial reduction
            x = blk(x)
            if prev.size(2) > x.size(2):
                ep[f"reduction_{len(ep) + 1}"] = prev
            prev = x
        ep[f"reduction_{len(ep) + 1}"] = x
        f = _up(ep["reduction_5"], ep["reduction_4"], sd, prefix + ".up1")
        if downsample == 8:
            f = _up(f, ep["reduction_3"], sd, prefix + ".up2")
    elif kind == "Resnet101":
        trunk = trunks.resnet101()
        trunk.load_state_dict(_sub(sd, prefix + "."), strict=False)  # heads live beside the trunk under this prefix
        trunk.eval()
        f = trunk.layer2(trunk.layer1(trunk.maxpool(trunk.relu(trunk.bn1(trunk.conv1(x))))))
    else:
        raise NotImplementedError(kind)
    x_img = F.conv2d(f, sd[prefix + ".image_head.weight"], sd[prefix + ".image_head.bias"])
    depth_logit = F.conv2d(f, sd[prefix + ".depth_head.weight"], sd[prefix + ".depth_head.bias"])
    return depth_logit, x_img


def lift_splat_encoder(enc_args, depth_logit, x_img, cam, B, N):
    """heter_encoders.py:110-241 from the (depth_logit, x_img) boundary: frustum, geometry, lift, voxel pooling.
    cam: dict of numpy rots/trans/intrins/post_rots/post_trans [B,N,...].  -> [B, C, Y, X] torch."""
    gc = enc_args["grid_conf"]
    dx, bx, nx = O.gen_dx_bx(gc["xbound"], gc["ybound"], gc["zbound"])
    fr = O.create_frustum(enc_args["data_aug_conf"]["final_dim"], enc_args["img_downsample"], gc["ddiscr"], gc["mode"])
    geom = O.lss_geometry(fr, cam["rots"], cam["trans"], cam["intrins"], cam["post_rots"], cam["post_trans"])
    lifted = O.lift(depth_logit, x_img)
    C, D, fH, fW = lifted.shape[1:]
    x = lifted.reshape(B, N, C, D, fH, fW).transpose(0, 1, 3, 4, 5, 2)
    return torch.from_numpy(np.ascontiguousarray(O.bev_pool(geom, x, dx, bx, nx, accumulate=np.float32)))


def _convnext(x, sd, p, n_blocks):
    """feature_alignnet_modules.py:299-361 (deform False)."""
    for j in range(n_blocks):
        q = f"{p}.model.{j}"
        dim = x.shape[1]
        y = F.conv2d(x, sd[q + ".dwconv.weight"], sd[q + ".dwconv.bias"], 1, 3, 1, dim)
        y = y.permute(0, 2, 3, 1)
        y = F.layer_norm(y, (dim,), sd[q + ".norm.weight"], sd[q + ".norm.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, sd[q + ".pwconv1.weight"], sd[q + ".pwconv1.bias"])),
                     sd[q + ".pwconv2.weight"], sd[q + ".pwconv2.bias"])
        if q + ".gamma" in sd:
            y = sd[q + ".gamma"] * y
        x = x + y.permute(0, 3, 1, 2)
    return x


def _center_crop(x, th, tw):
    """torchvision CenterCrop((th,tw)) incl. zero padding when the target is larger."""
    H, W = x.shape[-2:]
    if tw > W or th > H:
        pl = (tw - W) // 2 if tw > W else 0
        pt = (th - H) // 2 if th > H else 0
        pr = (tw - W + 1) // 2 if tw > W else 0
        pb = (th - H + 1) // 2 if th > H else 0
        x = F.pad(x, [pl, pr, pt, pb])
        H, W = x.shape[-2:]
    top, left = int(round((H - th) / 2.0)), int(round((W - tw) / 2.0))
    return x[..., top:top + th, left:left + tw]


def heter_pyramid_collab(sd, cfg_args, data, boundary=None, taps=None):
    """One scene of HeterPyramidCollab with m1 (PointPillars) and Lift-Splat camera agents.

    data: the reference's collated layout with numpy arrays -- `agent_modality_list`, `pairwise_t_matrix`,
    `inputs_m1` {voxel_features, voxel_coords, voxel_num_points}, `inputs_mX` {imgs, rots, trans, intrins, post_rots,
    post_trans} [B_m, N, ...] for camera modalities.  boundary: optional {mX: (depth_logit, x_img)} numpy to start the
    camera agents of modality mX at the trunk-output boundary instead of running the trunk.  taps: optional dict that
    receives intermediate tensors.  Returns dict of numpy outputs."""
    sd = {k: v.detach().cpu().float() if v.dtype.is_floating_point else v.detach().cpu() for k, v in sd.items()}
    r = cfg_args["lidar_range"]
    agents = list(data["agent_modality_list"])
    n_agents = len(agents)
    feats = {}
    with torch.no_grad():
        for m in [k for k in cfg_args if k.startswith("m") and k[1:].isdigit()]:
            cnt = agents.count(m)
            if cnt == 0:
                continue
            st = cfg_args[m]
            enc = st["encoder_args"]
            inp = data[f"inputs_{m}"]
            if st["core_method"] == "point_pillar":
                vs = enc["voxel_size"]
                nx = int(round((r[3] - r[0]) / vs[0]))
                ny = int(round((r[4] - r[1]) / vs[1]))
                x = pointpillar_encoder(sd, f"encoder_{m}", np.asarray(inp["voxel_features"]), np.asarray(inp["voxel_coords"]),
                                        np.asarray(inp["voxel_num_points"]), vs, r, cnt, ny, nx)
            elif st["core_method"] == "lift_splat_shoot":
                imgs = torch.from_numpy(np.asarray(inp["imgs"], np.float32))
                B, N = imgs.shape[:2]
                if boundary is not None and m in boundary:
                    dl, xi = (np.asarray(t, np.float32) for t in boundary[m])
                else:
                    dl, xi = cam_encode(sd, f"encoder_{m}.camencode", enc["camera_encoder"],
                                        imgs.reshape((B * N,) + tuple(imgs.shape[2:])), enc["img_downsample"])
                    dl, xi = dl.numpy(), xi.numpy()
                if taps is not None:
                    taps[f"{m}_depth_logit"], taps[f"{m}_x_img"] = dl, xi
                x = lift_splat_encoder(enc, dl, xi, {k: np.asarray(inp[k], np.float32) for k in
                                                      ("rots", "trans", "intrins", "post_rots", "post_trans")}, B, N)
                if taps is not None:
                    taps[f"{m}_bev"] = x.numpy()
            else:
                raise NotImplementedError(st["core_method"])
            bb = st["backbone_args"]
            for i, (nb, stv) in enumerate(zip(bb["layer_nums"], bb["layer_strides"])):
                x = _stage(x, sd, f"backbone_{m}.resnet.layer{i}", nb, stv, _basic_block)
            al = st["aligner_args"]
            if al["core_method"] == "convnext":
                x = _convnext(x, sd, f"aligner_{m}.channel_align", al["args"]["num_of_blocks"])
            if taps is not None:
                taps[f"{m}_aligned"] = x.numpy()
            if st["sensor_type"] == "camera":
                grid = st["camera_mask_args"]["grid_conf"]
                H, W = x.shape[-2:]
                x = _center_crop(x, int(H * (r[4] / grid["ybound"][1])), int(W * (r[3] / grid["xbound"][1])))
            feats[m] = x
        cursor = {m: 0 for m in feats}
        parts = []
        for m in agents:
            parts.append(feats[m][cursor[m]])
            cursor[m] += 1
        x = torch.stack(parts)
        cam_crop_info = {m: {f"crop_ratio_W_{m}": r[3] / cfg_args[m]["camera_mask_args"]["grid_conf"]["xbound"][1],
                             f"crop_ratio_H_{m}": r[4] / cfg_args[m]["camera_mask_args"]["grid_conf"]["ybound"][1]}
                         for m in cfg_args if m.startswith("m") and m[1:].isdigit()
                         and cfg_args[m]["sensor_type"] == "camera"}
        fb = cfg_args["fusion_backbone"]
        levels = []
        for i, (nb, stv) in enumerate(zip(fb["layer_nums"], fb["layer_strides"])):
            x = _stage(x, sd, f"pyramid_backbone.resnet.layer{i}", nb, stv, _bottleneck)
            levels.append(x)
        aff = O.normalize_pairwise_tfm(np.asarray(data["pairwise_t_matrix"]), r[4] - r[1], r[3] - r[0], 1)
        fused, occs = [], []
        for i, f in enumerate(levels):
            occ = F.conv2d(f, sd[f"pyramid_backbone.single_head_{i}.weight"], sd[f"pyramid_backbone.single_head_{i}.bias"])
            occs.append(occ.numpy())
            mask = O.camera_crop_mask(n_agents, f.shape[2], f.shape[3], agents, cam_crop_info) if cam_crop_info else None
            score = O.occ_to_score(occ.numpy(), mask)
            fused.append(torch.from_numpy(O.weighted_fuse(f.numpy(), score, aff[0][0, :n_agents]))[None])
        ups = [_deblock(fused[i], sd, f"pyramid_backbone.deblocks.{i}", s) for i, s in enumerate(fb["upsample_strides"])]
        y = torch.cat(ups, dim=1)
        if "shrink_header" in cfg_args:
            y = _double_conv(y, sd, "shrink_conv.layers.0")
        return {"cls_preds": F.conv2d(y, sd["cls_head.weight"], sd["cls_head.bias"]).numpy(),
                "reg_preds": F.conv2d(y, sd["reg_head.weight"], sd["reg_head.bias"]).numpy(),
                "dir_preds": F.conv2d(y, sd["dir_head.weight"], sd["dir_head.bias"]).numpy(),
                "occ_single_list": occs}


def base_bev_backbone(sd, prefix, x, cfg):
    """BaseBEVBackbone.forward (base_bev_backbone.py:96-156) from a state_dict: per level ZeroPad2d(1) + Conv2d(3, stride) + BN(1e-3)
    + ReLU, `layer_nums` x (Conv2d(3, pad 1) + BN + ReLU); deblocks ConvTranspose2d(stride) + BN + ReLU; channel concat."""
    ups = []
    for i, (n, s) in enumerate(zip(cfg["layer_nums"], cfg["layer_strides"])):
        p = f"{prefix}blocks.{i}"
        x = F.relu(_bn(F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[f"{p}.1.weight"], None, s), sd, f"{p}.2", 1e-3))
        for k in range(n):
            x = F.relu(_bn(F.conv2d(x, sd[f"{p}.{4 + 3 * k}.weight"], None, 1, 1), sd, f"{p}.{5 + 3 * k}", 1e-3))
        if cfg.get("upsample_strides"):
            ups.append(_deblock(x, sd, f"{prefix}deblocks.{i}", cfg["upsample_strides"][i]))
        else:
            ups.append(x)
    return torch.cat(ups, 1) if len(ups) > 1 else ups[0]


def second_detector(sd, args, voxels, coords, num, sparse_shape, batch):
    """Old-style SECOND (opencood/models/second.py:33-58): MeanVFE -> VoxelBackBone8x + HeightCompression (the dense restatement
    of the sparse-convolution rules, oracle_np.second_backbone) -> BaseBEVBackbone -> the two 1x1 heads.  -> (psm, rm) torch CPU."""
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
    feats = O.mean_vfe(voxels, num)
    bev = torch.from_numpy(O.second_backbone({k: v.numpy() for k, v in sd.items()}, "backbone_3d.", feats, coords, sparse_shape,
                                             batch))
    x = base_bev_backbone(sd, "backbone_2d.", bev, args["base_bev_backbone"])
    psm = F.conv2d(x, sd["cls_head.weight"], sd["cls_head.bias"])
    rm = F.conv2d(x, sd["reg_head.weight"], sd["reg_head.bias"])
    return psm, rm


# ---------------------------------------------------------------------------------------------------------------------
# HeterModelBaseline with V2X-ViT fusion (BASELINE config 5)
# ---------------------------------------------------------------------------------------------------------------------
def heter_model_baseline(sd, cfg_args, data, taps=None):
    """One scene through HeterModelBaseline.forward (opencood/models/heter_model_baseline.py:198-236) with fusion_method
    'v2xvit': per modality encoder (PointPillar heter_encoders.py:22-49 | SECOND :52-80) -> BaseBEVBackbone
    (base_bev_backbone.py:96-156) -> DownsampleConv shrinker (downsample_conv.py:22-49) -> V2XViTFusion (oracle/v2xvit_ref.py)
    -> optional shrink_conv -> the three 1x1 heads.  All agents of ONE modality (what config 5 and the golden use).
    data: {"agent_modality_list", "pairwise_t_matrix" [1, L, L, 4, 4], "inputs_<m>": {"voxel_features", "voxel_coords",
    "voxel_num_points"}}.  -> dict of numpy outputs."""
    from . import v2xvit_ref
    sd = {k: v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
    sd = {k: v.float() if v.dtype.is_floating_point else v for k, v in sd.items()}
    mods = list(data["agent_modality_list"])
    m = mods[0]
    assert all(mm == m for mm in mods), "one modality per scene"
    if cfg_args["fusion_method"] != "v2xvit":
        raise NotImplementedError(cfg_args["fusion_method"])
    n = len(mods)
    r = cfg_args["lidar_range"]
    setting = cfg_args[m]
    enc = setting["encoder_args"]
    vs = enc["voxel_size"]
    inp = data[f"inputs_{m}"]
    with torch.no_grad():
        if setting["core_method"] == "point_pillar":
            nx, ny = int(round((r[3] - r[0]) / vs[0])), int(round((r[4] - r[1]) / vs[1]))
            x = pointpillar_encoder(sd, f"encoder_{m}", inp["voxel_features"], inp["voxel_coords"], inp["voxel_num_points"],
                                    vs, r, n, ny, nx)
        elif setting["core_method"] == "second":
            grid = np.round((np.array(r[3:]) - np.array(r[:3])) / np.array(vs)).astype(np.int64)
            sparse_shape = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]       # sparse_backbone_3d.py:41
            feats = O.mean_vfe(inp["voxel_features"], inp["voxel_num_points"])
            x = torch.from_numpy(O.second_backbone_sparse({k: v.numpy() for k, v in sd.items() if k.startswith(f"encoder_{m}.")},
                                                          f"encoder_{m}.spconv_block.", feats, inp["voxel_coords"],
                                                          sparse_shape, n))
        else:
            raise NotImplementedError(setting["core_method"])
        if taps is not None:
            taps["encoder"] = x.numpy()
        x = base_bev_backbone(sd, f"backbone_{m}.", x, setting["backbone_args"])
        sh = setting["shrink_header"]
        x = _double_conv(x, sd, f"shrinker_{m}.layers.0", sh["stride"][0], sh["padding"][0])
        if taps is not None:
            taps["shrunk"] = x.numpy()
        H, W = r[4] - r[1], r[3] - r[0]
        aff = O.normalize_pairwise_tfm(np.asarray(data["pairwise_t_matrix"]), H, W, 1)
        fused = v2xvit_ref.v2xvit_fusion(sd, "fusion_net.", x, [n], torch.as_tensor(np.asarray(aff)), cfg_args["v2xvit"])
        if taps is not None:
            taps["fused"] = fused.numpy()
        if "shrink_header" in cfg_args:
            fused = _double_conv(fused, sd, "shrink_conv.layers.0", cfg_args["shrink_header"]["stride"][0],
                                 cfg_args["shrink_header"]["padding"][0])
        return {"cls_preds": F.conv2d(fused, sd["cls_head.weight"], sd["cls_head.bias"]).numpy(),
                "reg_preds": F.conv2d(fused, sd["reg_head.weight"], sd["reg_head.bias"]).numpy(),
                "dir_preds": F.conv2d(fused, sd["dir_head.weight"], sd["dir_head.bias"]).numpy()}
