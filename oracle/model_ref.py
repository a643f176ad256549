"""model_ref -- TEST INFRASTRUCTURE ONLY.

Functional CPU restatement of the reference's model forward for the PointPillars + PyramidFusion
path, driven by a plain state_dict (reference key names).  Dense convolutions are plain torch-CPU
fp32 ops (F.conv2d / F.batch_norm, un-fused, in the reference's order); the sparse / scatter / warp
parts come from oracle_np.  Used by tests and as bench.py's `cpu_baseline` (kind "port").

Follows: opencood/models/heter_pyramid_collab.py:133-209, heter_pyramid_single.py:99-136,
sub_modules/base_bev_backbone_resnet.py:88-136, resblock.py:42-122, fuse_modules/pyramid_fuse.py:
104-168, sub_modules/downsample_conv.py:22-49.
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
