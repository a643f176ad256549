"""v2xvit_ref -- TEST INFRASTRUCTURE ONLY.

CPU restatement (torch fp32 on the host, functional, driven by a state_dict with the reference's parameter names) of the
V2X-ViT fusion operator of BASELINE config 5: `V2XViTFusion` (opencood/models/fuse_modules/fusion_in_one.py:320-372) and the
transformer behind it (opencood/models/sub_modules/v2xvit_basic.py, hmsa.py, mswin.py, split_attn.py, base_transformer.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; heal_amd/ (the product) never does.

[pinned] against the reference's own output: tests/golden/fusion_small.npz (`v2xvit`, produced by the imported reference's
V2XViTFusion with the deterministic fill of tests/golden/detfill.py) and, through oracle/model_ref.heter_model_baseline,
tests/golden/baseline_small.npz (the whole HeterModelBaseline); tests/test_oracle_golden.py.

Inference only (dropout = identity).  What the fusion wrapper fixes (fusion_in_one.py:346-368) is restated as such: the prior
encoding (velocity, time delay, infrastructure flag) is all zeros -> every agent has type 0 and time delay 0; the spatial correction
matrix is the identity -> the STTF resample (v2xvit_basic.py:13-34) is the identity map and the rotated-ROI mask
(torch_transformation_utils.py:13-51) is all ones, so the combined mask is the agent mask.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return a.detach().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))


def regroup_pad(x, record_len, L):
    """fuse_utils.py:13-62 `regroup` (imported as `Regroup`, fusion_in_one.py:9): [sum n, C, H, W] -> ([B, L, C, H, W] zero padded,
    mask [B, L])."""
    outs, masks = [], []
    start = 0
    for n in [int(v) for v in record_len]:
        f = x[start:start + n]
        start += n
        pad = torch.zeros((L - n,) + tuple(f.shape[1:]), dtype=f.dtype)
        outs.append(torch.cat([f, pad], 0))
        masks.append(torch.tensor([1.0] * n + [0.0] * (L - n)))
    return torch.stack(outs), torch.stack(masks)


def warp_affine_simple(src, M, dsize):
    """torch_transformation_utils.py:323-332: affine_grid + bilinear grid_sample, zero padding, align_corners False."""
    grid = F.affine_grid(M, [src.shape[0], src.shape[1], dsize[0], dsize[1]], align_corners=False).to(src)
    return F.grid_sample(src, grid, align_corners=False)


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def hgt_cav_attention(sd, p, x, mask, types, heads, dim_head):
    """hmsa.py:110-150 `HGTCavAttention.forward`.  x [B, L, H, W, C]; mask [B, H, W, 1, L] (1 = real agent, masks KEYS);
    types [B, L] int (node type of every agent).  Per-type q / k / v / output projections (:39-66, :101-108), relation matrices
    per edge type type_i * num_types + type_j (:68-99)."""
    B, L, H, W, C = x.shape
    x = x.permute(0, 2, 3, 1, 4)                                     # (B, H, W, L, C)
    num_types = 2

    def per_type(name, inp):
        return torch.stack([F.linear(inp[:, :, :, i, :], sd[f"{p}{name}.{int(types[0, i])}.weight"],
                                     sd[f"{p}{name}.{int(types[0, i])}.bias"]) for i in range(L)], 3)
    assert B == 1, "the restatement keeps the reference's per-sample loops for B = 1 (one scene per forward)"
    q, k, v = per_type("q_linears", x), per_type("k_linears", x), per_type("v_linears", x)
    rel = torch.tensor([[int(types[0, i]) * num_types + int(types[0, j]) for j in range(L)] for i in range(L)])
    w_att = sd[p + "relation_att"][rel].permute(2, 0, 1, 3, 4).unsqueeze(0)    # (1, M, L, L, c, c)
    w_msg = sd[p + "relation_msg"][rel].permute(2, 0, 1, 3, 4).unsqueeze(0)

    def split(t):                                                   # b h w l (m c) -> b m h w l c
        return t.reshape(B, H, W, L, heads, dim_head).permute(0, 4, 1, 2, 3, 5)
    q, k, v = split(q), split(k), split(v)
    att = torch.einsum("bmhwip,bmijpq,bmhwjq->bmhwij", q, w_att, k) * dim_head ** -0.5
    att = att.masked_fill(mask.unsqueeze(1) == 0, -float("inf"))
    att = att.softmax(-1)
    v_msg = torch.einsum("bmijpc,bmhwjp->bmhwijc", w_msg, v)
    out = torch.einsum("bmhwij,bmhwijc->bmhwic", att, v_msg)
    out = out.permute(0, 2, 3, 4, 1, 5).reshape(B, H, W, L, heads * dim_head)
    out = per_type("a_linears", out)
    return out.permute(0, 3, 1, 2, 4)                               # (B, L, H, W, C)


def _relative_indices(ws):
    """mswin.py:13-17: indices[j] - indices[i] + ws - 1 for the ws^2 window positions (row-major x, y)."""
    idx = torch.tensor([[a, b] for a in range(ws) for b in range(ws)])
    return idx[None, :, :] - idx[:, None, :] + ws - 1


def window_attention(sd, p, x, heads, dim_head, ws):
    """mswin.py:46-80 `BaseWindowAttention.forward` with relative position embedding.  x [B, L, H, W, C]."""
    B, L, H, W, C = x.shape
    qkv = F.linear(x, sd[p + "to_qkv.weight"]).chunk(3, -1)
    nh, nw = H // ws, W // ws

    def windows(t):   # b l (nh wh) (nw ww) (m c) -> b l m (nh nw) (wh ww) c
        t = t.reshape(B, L, nh, ws, nw, ws, heads, dim_head)
        return t.permute(0, 1, 6, 2, 4, 3, 5, 7).reshape(B, L, heads, nh * nw, ws * ws, dim_head)
    q, k, v = (windows(t) for t in qkv)
    dots = torch.einsum("blmhic,blmhjc->blmhij", q, k) * dim_head ** -0.5
    ri = _relative_indices(ws)
    dots = dots + sd[p + "pos_embedding"][ri[:, :, 0], ri[:, :, 1]]
    attn = dots.softmax(-1)
    out = torch.einsum("blmhij,blmhjc->blmhic", attn, v)
    out = out.reshape(B, L, heads, nh, nw, ws, ws, dim_head).permute(0, 1, 3, 5, 4, 6, 2, 7).reshape(B, L, H, W, heads * dim_head)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def split_attn(sd, p, windows, dim):
    """split_attn.py:30-62 `SplitAttn.forward` (radix 3, cardinality 1)."""
    sw, mw, bw = windows
    B, L = sw.shape[:2]
    gap = (sw + mw + bw).mean((2, 3), keepdim=True)
    gap = torch.relu(F.layer_norm(F.linear(gap, sd[p + "fc1.weight"]), (dim,), sd[p + "bn1.weight"], sd[p + "bn1.bias"], 1e-5))
    a = F.linear(gap, sd[p + "fc2.weight"])                           # (B, L, 1, 1, 3 dim)
    a = F.softmax(a.view(B, L, 1, 3, -1), dim=3).reshape(B, -1).view(B, L, 1, 1, -1)   # RadixSoftmax :12-27
    return sw * a[..., 0:dim] + mw * a[..., dim:2 * dim] + bw * a[..., 2 * dim:]


def pyramid_window_attention(sd, p, x, cfg):
    """mswin.py:83-122 `PyramidWindowAttention.forward`."""
    outs = [window_attention(sd, f"{p}pwmsa.{i}.", x, h, d, ws)
            for i, (h, d, ws) in enumerate(zip(cfg["heads"], cfg["dim_head"], cfg["window_size"]))]
    method = cfg["fusion_method"]
    if method == "naive":
        return sum(outs) / len(outs)
    dim = {"split_attn": 256, "split_attn128": 128, "split_attn64": 64}[method]
    return split_attn(sd, p + "split_attn.", outs, dim)


def feed_forward(sd, p, x):
    """base_transformer.py:29-41: Linear -> GELU (exact) -> Linear."""
    return F.linear(F.gelu(F.linear(x, sd[p + "net.0.weight"], sd[p + "net.0.bias"])), sd[p + "net.3.weight"], sd[p + "net.3.bias"])


def rte(sd, p, x, dts, n_hid, ratio, max_len=100):
    """v2xvit_basic.py:37-83 `RTE`: x + lin(emb[dt * ratio]); the sinusoid table is a (frozen) parameter of the state_dict."""
    out = []
    for i in range(x.shape[1]):
        e = sd[p + "emb.emb.weight"][int(dts[0, i]) * ratio]
        out.append(x[:, i] + F.linear(e, sd[p + "emb.lin.weight"], sd[p + "emb.lin.bias"])[None, None, None])
    return torch.stack(out, 1)


def v2x_transformer(sd, p, x, mask, cfg):
    """v2xvit_basic.py:121-192 `V2XTEncoder.forward` + `V2XTransformer.forward` (ego row of the result).
    x [B, L, H, W, C + 3] (the last three channels: the prior encoding), mask [B, L] -> [B, H, W, C]."""
    enc = cfg["encoder"]
    cav, pw = enc["cav_att_config"], enc["pwindow_att_config"]
    p = p + "encoder."
    prior = x[..., -3:]
    x = x[..., :-3]
    types = prior[:, :, 0, 0, 2].to(torch.int)
    if cav["use_RTE"]:
        x = rte(sd, p + "rte.", x, prior[:, :, 0, 0, 1].to(torch.int), cav["dim"], cav["RTE_ratio"])
    # STTF with the identity correction matrix: identity; ROI mask all ones (module docstring) -> com_mask = the agent mask
    B, L, H, W, C = x.shape
    com_mask = mask.view(B, 1, 1, 1, L).expand(B, H, W, 1, L)
    for d in range(enc["depth"]):
        for b in range(enc["num_blocks"]):                           # V2XFusionBlock.forward :112-118
            q = f"{p}layers.{d}.0.layers.{b}."
            if not cav["use_hetero"]:
                raise NotImplementedError("CavAttention (use_hetero False) is not part of BASELINE config 5")
            x = hgt_cav_attention(sd, q + "0.fn.", _ln(x, sd, q + "0.norm."), com_mask, types, cav["heads"], cav["dim_head"]) + x
            x = pyramid_window_attention(sd, q + "1.fn.", _ln(x, sd, q + "1.norm."), pw) + x
        q = f"{p}layers.{d}.1."
        x = feed_forward(sd, q + "fn.", _ln(x, sd, q + "norm.")) + x
    return x[:, 0]


def v2xvit_fusion(sd, prefix, x, record_len, affine_matrix, cfg):
    """fusion_in_one.py:326-372 `V2XViTFusion.forward`: x [sum n, C, H, W], affine_matrix [B, L, L, 2, 3] (normalize_pairwise_tfm)
    -> fused [B, C, H, W]."""
    sd = {k: _t(v).float() for k, v in sd.items() if k.startswith(prefix)}
    x = _t(x).float()
    affine_matrix = _t(affine_matrix).float()
    _, C, H, W = x.shape
    B, L = affine_matrix.shape[:2]
    feat, mask = regroup_pad(x, record_len, L)
    feat = torch.cat([feat, torch.zeros((B, L, 3, H, W))], 2)         # prior encoding: zeros
    feat = torch.stack([warp_affine_simple(feat[b], affine_matrix[b, 0], (H, W)) for b in range(B)])
    feat = feat.permute(0, 1, 3, 4, 2)
    fused = v2x_transformer(sd, prefix + "fusion_net.", feat, mask, cfg["transformer"])
    return fused.permute(0, 3, 1, 2)
