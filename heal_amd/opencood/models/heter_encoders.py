"""Per-modality BEV encoders behind the reference's names (opencood/models/heter_encoders.py):
PointPillar (:22-50), SECOND (:52-81), LiftSplatShoot (:83-241), LiftSplatShootVoxel (:244-301).
Classes are discovered by name exactly as the reference does (heter_pyramid_collab.py:41-48).
"""
import numpy as np
import torch
import torch.nn as nn

from heal_amd import ops
from heal_amd.opencood.models.sub_modules.pillar_vfe import PillarVFE
from heal_amd.opencood.models.sub_modules.point_pillar_scatter import PointPillarScatter


class PointPillar(nn.Module):
    """voxels -> fused PFN + scatter (K2) -> [n, 64, ny, nx].

    Besides the reference input (`voxel_features`, `voxel_coords`, `voxel_num_points`) the encoder
    accepts raw device point clouds -- `inputs_mX = {'points': [tensor [N_k,4] per agent]}` -- and
    voxelises them on the GPU (K1) without a host round trip."""

    def __init__(self, args):
        super().__init__()
        grid_size = (np.array(args["lidar_range"][3:6]) - np.array(args["lidar_range"][0:3])) / \
            np.array(args["voxel_size"])
        grid_size = np.round(grid_size).astype(np.int64)
        args["point_pillar_scatter"]["grid_size"] = grid_size  # the reference mutates args the same way
        self.lidar_range = [float(v) for v in args["lidar_range"]]
        self.voxel_size = [float(v) for v in args["voxel_size"]]
        self.max_points = int(args.get("max_points_per_voxel", 32))
        self.max_voxels = int(args.get("max_voxels", 70000))
        self.pillar_vfe = PillarVFE(args["pillar_vfe"], num_point_features=4, voxel_size=args["voxel_size"],
                                    point_cloud_range=args["lidar_range"])
        self.scatter = PointPillarScatter(args["point_pillar_scatter"])
        self._fold_key = None
        self._fold = None

    def _bn(self):
        pfn = self.pillar_vfe.pfn_layers[0]
        tensors = [pfn.linear.weight] + ([pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean,
                                          pfn.norm.running_var] if pfn.use_norm else [pfn.linear.bias])
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self._fold_key:
            self._fold = pfn.folded_bn()
            self._fold_key = key
        return self._fold

    def encode_points(self, point_list):
        """Raw device point clouds -> canvas, one K1 + K2 pair per agent and NO host round trip:
        the voxel count stays on the device (K2 reads it) and every agent writes its own canvas slab."""
        scale, shift = self._bn()
        weight = self.pillar_vfe.pfn_layers[0].linear.weight.detach()
        ny, nx = self.scatter.ny, self.scatter.nx
        canvas = torch.empty((len(point_list), weight.shape[0], ny, nx), dtype=torch.float32,
                             device=point_list[0].device)
        for b, pts in enumerate(point_list):
            v, c, n, count = ops.voxelize(pts, self.lidar_range, self.voxel_size, self.max_points,
                                          self.max_voxels, batch_idx=0, sync=False)
            ops.pfn_scatter(v, c, n, weight, scale, shift, self.voxel_size, self.lidar_range, 1, ny, nx,
                            n_voxels_dev=count, out=canvas[b:b + 1])
        return canvas

    def forward(self, data_dict, modality_name):
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("heal_amd implements the inference hot path (SURVEY 8f: training is 'next')")
        inp = data_dict[f"inputs_{modality_name}"]
        if "points" in inp:
            return self.encode_points(inp["points"])
        voxels, coords, num = inp["voxel_features"], inp["voxel_coords"], inp["voxel_num_points"]
        # point_pillar_scatter.py:45 reads the batch size back from the device the same way
        n_agents = int(inp["n_agents"]) if "n_agents" in inp else int(coords[:, 0].max().item()) + 1
        if coords.dtype != torch.int32:
            coords = coords.to(torch.int32)
        if num.dtype != torch.int32:
            num = num.to(torch.int32)
        scale, shift = self._bn()
        weight = self.pillar_vfe.pfn_layers[0].linear.weight.detach()
        return ops.pfn_scatter(voxels, coords, num, weight, scale, shift, self.voxel_size, self.lidar_range,
                               n_agents, self.scatter.ny, self.scatter.nx)
