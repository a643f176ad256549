"""Lift-Splat grid helpers (reference: opencood/utils/camera_utils.py:129-134 gen_dx_bx,
:187-196 depth_discretization)."""
import numpy as np
import torch


def gen_dx_bx(xbound, ybound, zbound):
    dx = torch.Tensor([row[2] for row in [xbound, ybound, zbound]])
    bx = torch.Tensor([row[0] + row[2] / 2.0 for row in [xbound, ybound, zbound]])
    nx = torch.LongTensor([(row[1] - row[0]) / row[2] for row in [xbound, ybound, zbound]])
    return dx, bx, nx


def depth_discretization(depth_min, depth_max, num_bins, mode):
    if mode == "UD":
        bin_size = (depth_max - depth_min) / num_bins
        return depth_min + bin_size * np.arange(num_bins)
    if mode == "LID":
        bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
        return depth_min + bin_size * (np.arange(num_bins) * np.arange(1, 1 + num_bins)) / 2
    raise NotImplementedError(mode)
