"""PillarVFE / PFNLayer parameter containers (names as in opencood/models/sub_modules/
pillar_vfe.py:13-100) whose inference forward is the fused HIP operator K2 (heal_pfn_scatter); `pillar_features` is the
same arithmetic as torch operators for the gradient path (training).

Only the configuration HEAL uses is implemented: one PFN layer, use_norm, use_absolute_xyz,
no distance feature (lidar_pyramid.yaml / m1m2m3m4.yaml `pillar_vfe` block).
"""
import torch
import torch.nn as nn


class PFNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not self.last_vfe:
            out_channels = out_channels // 2
        if self.use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)

    def folded_bn(self):
        """(scale, shift) of the eval-mode BatchNorm1d: y = x*scale + shift."""
        with torch.no_grad():
            if self.use_norm:
                scale = self.norm.weight / torch.sqrt(self.norm.running_var + self.norm.eps)
                shift = self.norm.bias - self.norm.running_mean * scale
            else:
                scale = torch.ones_like(self.linear.bias)
                shift = self.linear.bias.clone()
        return scale.contiguous(), shift.contiguous()


class PillarVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range):
        super().__init__()
        self.model_cfg = model_cfg
        self.use_norm = model_cfg["use_norm"]
        self.with_distance = model_cfg["with_distance"]
        self.use_absolute_xyz = model_cfg["use_absolute_xyz"]
        num_point_features += 6 if self.use_absolute_xyz else 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = list(model_cfg["num_filters"])
        if len(self.num_filters) != 1 or self.with_distance or not self.use_absolute_xyz:
            raise NotImplementedError(
                "the fused PFN kernel implements the HEAL configuration: num_filters [C], "
                "use_absolute_xyz true, with_distance false")
        self.pfn_layers = nn.ModuleList([PFNLayer(num_point_features, self.num_filters[0], self.use_norm,
                                                  last_layer=True)])
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def pillar_features(self, voxels, coords, num_points):
        """Gradient path of pillar_vfe.py:63-100: [M,P,4] points of M pillars (coords [M,4] = agent,z,y,x; num_points [M]) ->
        [M,C].  Ten inputs per point: x,y,z,intensity, offset from the pillar's mean point, offset from the pillar's centre;
        slots past num_points are zeroed; Linear -> BatchNorm1d over the channel -> ReLU -> max over the points."""
        M, P, _ = voxels.shape
        cnt = num_points.to(voxels.dtype).clamp_min(1).view(M, 1, 1)
        xyz = voxels[:, :, :3]
        from_mean = xyz - xyz.sum(1, keepdim=True) / cnt
        vs = voxels.new_tensor(self.voxel_size)
        lo = voxels.new_tensor(self.point_cloud_range[:3])
        centre = coords[:, [3, 2, 1]].to(voxels.dtype) * vs + (vs / 2 + lo)
        from_centre = xyz - centre.view(M, 1, 3)
        feats = torch.cat([voxels, from_mean, from_centre], -1)
        live = torch.arange(P, device=voxels.device).view(1, P) < num_points.view(M, 1)
        feats = feats * live.unsqueeze(-1).to(feats.dtype)
        pfn = self.pfn_layers[0]
        h = pfn.linear(feats)
        if pfn.use_norm:
            h = pfn.norm(h.permute(0, 2, 1)).permute(0, 2, 1)
        return torch.relu(h).max(dim=1)[0]

    def pillar_features_kernels(self, voxels, coords, num_points):
        """pillar_features() on the HIP kernels in BOTH directions (round 3): batch statistics from
        heal_pfn_moments, forward heal_pfn_features, backward heal_pfn_backward -- no [M,P,64] tensor in either direction.
        Updates the BatchNorm running statistics like nn.BatchNorm1d does in training mode."""
        pfn = self.pfn_layers[0]
        bn = pfn.norm
        return _PFNFunction.apply(pfn.linear.weight, bn.weight, bn.bias, voxels, coords, num_points, bn, bn.training,
                                  tuple(self.voxel_size), tuple(self.point_cloud_range))


class _PFNFunction(torch.autograd.Function):
    """Linear(10 -> 64, no bias) -> BatchNorm1d -> ReLU -> max over the points of a pillar, forward and backward on the kernels of
    heal_amd/csrc/pfn_scatter.hip.  With z = W f every statistic of the M x P rows is a closed form of s1 = sum f and
    S = sum f f^T (float64 on the device): mean = W s1 / R, E[z^2] = diag(W S W^T) / R."""

    @staticmethod
    def forward(ctx, weight, gamma, beta, voxels, coords, num_points, bn, training, voxel_size, lidar_range):
        from heal_amd import ops
        voxels, coords, num_points = voxels.detach().contiguous(), coords.detach().int().contiguous(), num_points.detach().int().contiguous()
        w = weight.detach().contiguous()
        R = float(voxels.shape[0] * voxels.shape[1])
        s1 = S = None
        if training:
            s1, S = ops.pfn_moments(voxels, coords, num_points, voxel_size, lidar_range)
            w64 = w.double()
            mean64 = (w64 @ s1) / R
            var64 = (((w64 @ S) * w64).sum(1) / R - mean64 * mean64).clamp_min(0.0)      # biased, like BatchNorm's normaliser
            with torch.no_grad():       # running statistics: momentum update with the UNBIASED variance (nn.BatchNorm1d)
                if bn.track_running_stats and bn.running_mean is not None:
                    if bn.num_batches_tracked is not None:
                        bn.num_batches_tracked.add_(1)
                    # momentum None = cumulative moving average (nn.BatchNorm1d: factor 1 / num_batches_tracked)
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(max(int(bn.num_batches_tracked), 1))
                    bn.running_mean.mul_(1.0 - mom).add_(mean64.float(), alpha=mom)
                    bn.running_var.mul_(1.0 - mom).add_((var64 * (R / max(R - 1.0, 1.0))).float(), alpha=mom)
        else:
            mean64, var64 = bn.running_mean.detach().double(), bn.running_var.detach().double()
        rstd64 = torch.rsqrt(var64 + bn.eps)
        scale = (gamma.detach().double() * rstd64).float().contiguous()
        shift = (beta.detach().double() - mean64 * gamma.detach().double() * rstd64).float().contiguous()
        out = ops.pfn_features(voxels, coords, num_points, w, scale, shift, voxel_size, lidar_range)
        ctx.save_for_backward(w, gamma.detach(), voxels, coords, num_points, scale, shift, mean64, rstd64)
        ctx.extra = (training, voxel_size, lidar_range, R, s1, S)
        return out

    @staticmethod
    def backward(ctx, g):
        from heal_amd import ops
        w, gamma, voxels, coords, num_points, scale, shift, mean64, rstd64 = ctx.saved_tensors
        training, voxel_size, lidar_range, R, s1, S = ctx.extra
        A, B, Cx = ops.pfn_backward(voxels, coords, num_points, w, scale, shift, mean64.float().contiguous(),
                                    rstd64.float().contiguous(), voxel_size, lidar_range, g.contiguous())
        g64 = gamma.double()
        if training:
            w64 = w.double()
            xf = ((w64 @ S) - mean64[:, None] * s1[None, :]) * rstd64[:, None]      # sum_rows xhat_c f
            dW = (g64 * rstd64)[:, None] * (A - (B / R)[:, None] * s1[None, :] - (Cx / R)[:, None] * xf)
        else:
            dW = (g64 * rstd64)[:, None] * A
        need = ctx.needs_input_grad
        return (dW.float() if need[0] else None, Cx.float() if need[1] else None, B.float() if need[2] else None,
                None, None, None, None, None, None, None)

