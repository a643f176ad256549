"""Single-scale fusion operators used by HeterModelBaseline (reference: opencood/models/fuse_modules/
fusion_in_one.py): MaxFusion (:87-124), AttFusion (:126-151, :14-45), V2XViTFusion (:320-372).
Warping to the ego frame is K5's heal_warp_agent; the per-pixel attention is K6."""
import numpy as np
import torch
import torch.nn as nn

from heal_amd import ops
from heal_amd.opencood.models._heter_common import record_len_to_list


def regroup(x, record_len):
    """fusion_in_one.py:48-51: split the agent dimension by scene."""
    lens = record_len_to_list(record_len)
    return torch.split(x, lens)


def warp_to_ego(x, affine_rows, grid_f64=True):
    """warp_affine_simple(x, t_matrix[0, :], (H, W)) for one scene: x [n,C,H,W] -> [n,C,H,W]."""
    zeros = torch.zeros((1,) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    return torch.stack([ops.warp_agent(x[a], zeros, affine_rows[a], grid_f64)[0] for a in range(x.shape[0])])


def _host_affine(affine_matrix):
    if isinstance(affine_matrix, torch.Tensor):
        a = affine_matrix.detach().cpu().numpy()
    else:
        a = np.asarray(affine_matrix)
    return a, a.dtype == np.float64


class MaxFusion(nn.Module):
    def forward(self, x, record_len, affine_matrix):
        aff, f64 = _host_affine(affine_matrix)
        out = []
        for b, feats in enumerate(regroup(x, record_len)):
            n = feats.shape[0]
            out.append(warp_to_ego(feats, aff[b][0, :n], f64).max(dim=0)[0])
        return torch.stack(out)


class AttFusion(nn.Module):
    def __init__(self, feature_dims):
        super().__init__()
        self.sqrt_dim = float(np.sqrt(feature_dims))
        self.feature_dims = feature_dims

    def forward(self, xx, record_len, affine_matrix):
        aff, f64 = _host_affine(affine_matrix)
        out = []
        for b, feats in enumerate(regroup(xx, record_len)):
            n, C, H, W = feats.shape
            x = warp_to_ego(feats, aff[b][0, :n], f64)
            x = x.view(n, C, H * W).permute(2, 0, 1).contiguous()       # [HW, n, C]
            h = ops.agent_attention(x, x, x, heads=1, scale=1.0 / self.sqrt_dim, out_rows=1)  # ego row only
            out.append(h[:, 0, :].t().reshape(C, H, W))
        return torch.stack(out)


class V2XViTFusion(nn.Module):
    def __init__(self, args):
        super().__init__()
        from heal_amd.opencood.models.sub_modules.v2xvit_basic import V2XTransformer
        self.fusion_net = V2XTransformer(args["transformer"])

    def forward(self, x, record_len, affine_matrix):
        aff, f64 = _host_affine(affine_matrix)
        out = []
        for b, feats in enumerate(regroup(x, record_len)):
            n = feats.shape[0]
            ego = warp_to_ego(feats, aff[b][0, :n], f64)                 # [n,C,H,W]; the 3 prior channels are zero
            fused = self.fusion_net(ego.permute(0, 2, 3, 1).contiguous())   # [H,W,C]
            out.append(fused.permute(2, 0, 1))
        return torch.stack(out)
