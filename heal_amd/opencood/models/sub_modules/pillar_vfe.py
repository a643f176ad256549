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
