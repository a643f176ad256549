"""PointPillarScatter (opencood/models/sub_modules/point_pillar_scatter.py:9-17): holds the grid
geometry only; the scatter itself is fused into K2 (see heter_encoders.PointPillar)."""
import torch.nn as nn


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg["num_features"]
        self.nx, self.ny, self.nz = (int(v) for v in model_cfg["grid_size"])
        assert self.nz == 1
