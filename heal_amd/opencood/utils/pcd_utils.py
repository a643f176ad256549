"""Host mirror of the point-cloud filters the datasets apply before voxelisation (reference:
opencood/utils/pcd_utils.py:41-130).  Same names and argument meaning.  numpy in -> numpy out (boolean indexing, like
the reference: these run in DataLoader workers); a CUDA tensor in -> the device operator heal_mask_points, which keeps
length and order and marks dropped points as NaN (the voxeliser skips NaN points, so
`preprocess_device(mask_points_by_range(mask_ego_points(pts), r))` equals the reference's
`preprocess(mask_points_by_range(mask_ego_points(pts_np), r))` without a host round trip)."""
import numpy as np
import torch

from heal_amd import ops


def mask_points_by_range(points, limit_range):
    """pcd_utils.py:41-67: keep xmin < x < xmax, ymin < y < ymax, zmin < z < zmax (strict)."""
    if isinstance(points, torch.Tensor) and points.is_cuda:
        return ops.mask_points(points, limit_range, mask_ego=False)
    keep = ((points[:, 0] > limit_range[0]) & (points[:, 0] < limit_range[3]) & (points[:, 1] > limit_range[1])
            & (points[:, 1] < limit_range[4]) & (points[:, 2] > limit_range[2]) & (points[:, 2] < limit_range[5]))
    return points[keep]


def mask_ego_points(points):
    """pcd_utils.py:70-88: drop the returns from the vehicle's own body."""
    if isinstance(points, torch.Tensor) and points.is_cuda:
        return ops.mask_points(points, None, mask_ego=True)
    body = (points[:, 0] >= -1.95) & (points[:, 0] <= 2.95) & (points[:, 1] >= -1.1) & (points[:, 1] <= 1.1)
    return points[np.logical_not(body)]


def shuffle_points(points):
    """pcd_utils.py:91-95."""
    return points[np.random.permutation(points.shape[0])]


def lidar_project(lidar_data, extrinsic):
    """pcd_utils.py:98-129: (n,4) x,y,z,intensity through a 4x4 matrix (float64 like the reference's np.dot)."""
    xyz1 = np.r_[lidar_data[:, :3].T, [np.ones(lidar_data.shape[0])]]
    xyz = np.dot(extrinsic, xyz1)[:3, :].T
    return np.hstack((xyz, np.expand_dims(lidar_data[:, 3], -1)))


def projected_lidar_stack(projected_lidar_list):
    """pcd_utils.py:132-150."""
    return np.vstack(list(projected_lidar_list))
