"""Single-scale fusion operators used by HeterModelBaseline (reference: opencood/models/fuse_modules/
fusion_in_one.py): MaxFusion (:87-124), AttFusion (:126-151, :14-45), V2XViTFusion (:320-372).
Warping to the ego frame is K5's heal_warp_agent; the per-pixel attention is K6."""
import os

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
    if torch.is_grad_enabled() and x.requires_grad:   # gradient path: affine_grid + grid_sample (torch_transformation_utils.py:323-332)
        import torch.nn.functional as F
        M = torch.as_tensor(affine_rows, dtype=x.dtype, device=x.device)
        return F.grid_sample(x, F.affine_grid(M, list(x.shape), align_corners=False), align_corners=False)
    zeros = torch.zeros((1,) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    return torch.stack([ops.warp_agent(x[a], zeros, affine_rows[a], grid_f64)[0] for a in range(x.shape[0])])


def _host_affine(affine_matrix):
    """-> (affine matrix, grid_is_f64).  CUDA tensors stay on the device (read by the warp kernel at run time)."""
    if isinstance(affine_matrix, torch.Tensor):
        if affine_matrix.is_cuda:
            return affine_matrix.detach(), affine_matrix.dtype == torch.float64
        a = affine_matrix.detach().numpy()
    else:
        a = np.asarray(affine_matrix)
    return a, a.dtype == np.float64


class _WarpThenFuse(nn.Module):
    """The three operators share one shape: warp every agent of a scene into the ego frame (K5), then reduce the
    ego-frame stack with `fuse_warped` ([n,C,H,W] -> [C,H,W]).  The agent-sharded path (heal_amd/dist.py) warps on the
    owning rank and calls `fuse_warped` on the gathered stack."""

    def forward(self, x, record_len, affine_matrix):
        aff, f64 = _host_affine(affine_matrix)
        out = []
        for b, feats in enumerate(regroup(x, record_len)):
            n = feats.shape[0]
            out.append(self.fuse_warped(warp_to_ego(feats, aff[b][0, :n], f64)))
        return torch.stack(out)


class MaxFusion(_WarpThenFuse):
    def fuse_warped(self, ego):
        return ego.max(dim=0)[0]


class AttFusion(_WarpThenFuse):
    def __init__(self, feature_dims):
        super().__init__()
        self.sqrt_dim = float(np.sqrt(feature_dims))
        self.feature_dims = feature_dims

    def fuse_warped(self, ego):
        n, C, H, W = ego.shape
        x = ego.reshape(n, C, H * W).permute(2, 0, 1).contiguous()      # [HW, n, C]
        if torch.is_grad_enabled() and ego.requires_grad:
            # gradient path (fusion_in_one.py:14-45,126-151): softmax(x x^T / sqrt(C)) x per pixel, the ego row
            if n <= 8 and ops.agent_attention_train_supported(x, 1) and os.environ.get("HEAL_ATTN_GRAD", "kernel") != "torch":
                # K6 forward (ego row) + heal_agent_attention_backward; x enters as q, k and v: autograd sums the three gradients
                return ops.AgentAttention.apply(x, x, x, 1, 1.0 / self.sqrt_dim, 1, False)[:, 0, :].t().reshape(C, H, W)
            attn = torch.softmax(torch.bmm(x, x.transpose(1, 2)) / self.sqrt_dim, dim=-1)
            return torch.bmm(attn, x)[:, 0, :].t().reshape(C, H, W)
        h = ops.agent_attention(x, x, x, heads=1, scale=1.0 / self.sqrt_dim, out_rows=1)  # ego row only
        return h[:, 0, :].t().reshape(C, H, W)


class V2XViTFusion(_WarpThenFuse):
    def __init__(self, args):
        super().__init__()
        from heal_amd.opencood.models.sub_modules.v2xvit_basic import V2XTransformer
        self.fusion_net = V2XTransformer(args["transformer"])

    def fuse_warped(self, ego):
        fused = self.fusion_net(ego.permute(0, 2, 3, 1).contiguous())       # [n,H,W,C] -> [H,W,C]; the 3 prior channels are zero
        return fused.permute(2, 0, 1)

    def forward(self, x, record_len, affine_matrix):
        if not x.is_cuda or x.shape[1] % 4 or (torch.is_grad_enabled() and x.requires_grad):
            return super().forward(x, record_len, affine_matrix)
        # inference on the device: every agent of a scene warped straight into the transformer's token-major layout (one launch;
        # no per-agent warp + stack + permute copy)
        aff, f64 = _host_affine(affine_matrix)
        out = []
        for b, feats in enumerate(regroup(x, record_len)):
            n = feats.shape[0]
            out.append(self.fusion_net(ops.warp_agents_pm(feats, aff[b][0, :n], f64)).permute(2, 0, 1))
        return torch.stack(out)


def build_fusion(args):
    """The single-scale fusion operator a model YAML names (`fusion_method`: max | att | v2xvit; the other methods of
    fusion_in_one.py belong to papers outside the hot-path scope)."""
    method = args["fusion_method"]
    if method == "max":
        return MaxFusion()
    if method == "att":
        return AttFusion(args["att"]["feat_dim"])
    if method == "v2xvit":
        return V2XViTFusion(args["v2xvit"])
    raise NotImplementedError(f"fusion_method '{method}' is outside the hot-path scope (SURVEY 2, row 2)")
