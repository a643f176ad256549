"""Lift-Splat grid helpers (reference: opencood/utils/camera_utils.py:129-134 gen_dx_bx, :137-185 bin_depths,
:187-196 depth_discretization, :198-207 indices_to_depth, :209-246 cumsum_trick / QuickCumsum)."""
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


def _bin_index(depth_map, mode, depth_min, depth_max, num_bins):
    """Fractional bin coordinate of a depth under the three discretisations of CaDDN (camera_utils.py:137-160)."""
    import math
    span = depth_max - depth_min
    if mode == "UD":        # uniform bins
        return (depth_map - depth_min) / (span / num_bins)
    if mode == "LID":       # linearly growing bins: d = d_min + s i (i + 1) / 2  ->  i = (-1 + sqrt(1 + 8 (d - d_min) / s)) / 2
        step = 2 * span / (num_bins * (1 + num_bins))
        return -0.5 + 0.5 * torch.sqrt(1 + 8 * (depth_map - depth_min) / step)
    if mode == "SID":       # log-spaced bins
        return num_bins * (torch.log(1 + depth_map) - math.log(1 + depth_min)) / (math.log(1 + depth_max) - math.log(1 + depth_min))
    raise NotImplementedError(mode)


def bin_depths(depth_map, mode, depth_min, depth_max, num_bins, target=True):
    """camera_utils.py:137-185: depth map -> int64 bin indices clamped to [0, num_bins - 1] (non-finite -> last bin).
    target=True returns (indices, None); target=False returns (indices, in_range_mask) with the mask taken BEFORE clamping."""
    pos = _bin_index(depth_map, mode, depth_min, depth_max, num_bins)
    outside = (pos < 0) | (pos >= num_bins) | ~torch.isfinite(pos)
    last = pos.new_full((), float(num_bins - 1))
    # the reference's rule order: below the range (-inf included) -> bin 0, at / above it (+inf included) -> last bin, NaN -> last
    pos = torch.where(pos < 0, pos.new_zeros(()), pos)
    pos = torch.where((pos >= num_bins) | torch.isnan(pos), last, pos)
    idx = pos.to(torch.int64)
    return (idx, None) if target else (idx, ~outside)


def indices_to_depth(indices, depth_min, depth_max, num_bins, mode):
    """camera_utils.py:198-207: left edge of a bin (inverse of bin_depths for UD / LID)."""
    if mode == "UD":
        return indices * ((depth_max - depth_min) / num_bins) + depth_min
    if mode == "LID":
        step = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
        return depth_min + step * (indices * (indices + 1)) / 2
    raise NotImplementedError(mode)


def _segment_tails(ranks):
    """True at the LAST element of every run of equal (sorted) ranks."""
    tail = torch.ones(ranks.shape[0], device=ranks.device, dtype=torch.bool)
    tail[:-1] = ranks[1:] != ranks[:-1]
    return tail


def cumsum_trick(x, geom_feats, ranks):
    """camera_utils.py:209-218: per-cell sums of rank-sorted rows as differences of a running sum taken at the run tails.
    (The fused kernels never form this: heal_bev_pool sums each cell directly.  Exported for code that imports the helper.)"""
    tail = _segment_tails(ranks)
    run = x.cumsum(0)[tail]
    return torch.cat((run[:1], run[1:] - run[:-1])), geom_feats[tail]


class QuickCumsum(torch.autograd.Function):
    """camera_utils.py:220-246: cumsum_trick with a hand-written backward -- every row of a run receives the gradient of its
    run's sum."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks):
        tail = _segment_tails(ranks)
        run = x.cumsum(0)[tail]
        geom_kept = geom_feats[tail]
        ctx.save_for_backward(tail)
        ctx.mark_non_differentiable(geom_kept)
        return torch.cat((run[:1], run[1:] - run[:-1])), geom_kept

    @staticmethod
    def backward(ctx, gradx, gradgeom):
        tail, = ctx.saved_tensors
        run_of_row = torch.cumsum(tail, 0)      # tails seen up to and including this row ...
        run_of_row[tail] -= 1                   # ... = index of the row's own run (a tail counts itself)
        return gradx[run_of_row], None, None
