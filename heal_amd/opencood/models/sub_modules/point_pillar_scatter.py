"""PointPillarScatter (opencood/models/sub_modules/point_pillar_scatter.py:9-17): holds the grid
geometry; at inference the scatter is fused into K2 (see heter_encoders.PointPillar), `canvas` is the gradient path."""
import torch
import torch.nn as nn


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg["num_features"]
        self.nx, self.ny, self.nz = (int(v) for v in model_cfg["grid_size"])
        assert self.nz == 1

    def canvas(self, pillars, coords, n_agents):
        """Gradient path of point_pillar_scatter.py:19-52: pillars [M,C] at coords [M,4] (agent,z,y,x) -> [n,C,ny,nx]."""
        C = pillars.shape[1]
        flat = (coords[:, 0].long() * self.ny + coords[:, 2].long()) * self.nx + coords[:, 3].long()
        out = pillars.new_zeros((n_agents * self.ny * self.nx, C))
        out = out.index_copy(0, flat, pillars)      # one pillar per cell: a copy, not a sum
        return out.view(n_agents, self.ny, self.nx, C).permute(0, 3, 1, 2).contiguous()
