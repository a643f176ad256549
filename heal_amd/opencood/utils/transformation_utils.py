"""Pose / affine helpers on the hot path (reference: opencood/utils/transformation_utils.py)."""
import numpy as np
import torch


def normalize_pairwise_tfm(pairwise_t_matrix, H, W, discrete_ratio, downsample_rate=1):
    """transformation_utils.py:68-92 -- [B,L,L,4,4] -> [B,L,L,2,3] for F.affine_grid; dtype kept
    (float64 when the matrix comes from the dataset's numpy array).  Accepts torch or numpy."""
    if isinstance(pairwise_t_matrix, torch.Tensor):
        a = pairwise_t_matrix[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]]
    else:
        a = np.asarray(pairwise_t_matrix)[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]].copy()
    a[..., 0, 1] = a[..., 0, 1] * H / W
    a[..., 1, 0] = a[..., 1, 0] * W / H
    a[..., 0, 2] = a[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    a[..., 1, 2] = a[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return a


def pairwise_to_host(pairwise_t_matrix):
    """The fusion kernels take the (tiny) affine matrices as launch arguments: bring the pairwise
    matrix to the host once per forward.  Returns (numpy array, grid_is_f64)."""
    if isinstance(pairwise_t_matrix, torch.Tensor):
        arr = pairwise_t_matrix.detach().cpu().numpy()
    else:
        arr = np.asarray(pairwise_t_matrix)
    return arr, arr.dtype == np.float64
