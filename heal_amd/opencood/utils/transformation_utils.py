"""Pose / affine helpers on the hot path (reference: opencood/utils/transformation_utils.py)."""
import numpy as np
import torch


_NORM_CONSTS = {}


def normalize_pairwise_tfm(pairwise_t_matrix, H, W, discrete_ratio, downsample_rate=1):
    """transformation_utils.py:68-92 -- [B,L,L,4,4] -> [B,L,L,2,3] for F.affine_grid; dtype kept
    (float64 when the matrix comes from the dataset's numpy array).  Accepts torch or numpy."""
    if isinstance(pairwise_t_matrix, torch.Tensor):
        # rows 0,1 x columns 0,1,3 by SLICES (an index list would become an index tensor uploaded from the host, which a
        # stream capture does not permit: the device-resident pose matrix is normalised inside the captured graph)
        t = pairwise_t_matrix
        a = torch.cat([t[..., 0:2, 0:2], t[..., 0:2, 3:4]], dim=-1)
        if t.is_cuda:
            # the four element updates below as THREE broadcast operations with constant [2,3] tensors -- (a * m1) / d * m2, the
            # reference's operation order element by element (x*1, x/1 are exact) -- instead of a dozen slice kernels at the head
            # of every captured step.  The constants are cached per (shape, dtype, device); the first (eager) call creates them.
            key = (int(H), int(W), float(discrete_ratio), float(downsample_rate), t.dtype, t.device)
            c = _NORM_CONSTS.get(key)
            if c is None:
                kw, kh = downsample_rate * discrete_ratio * W, downsample_rate * discrete_ratio * H
                mk = lambda rows: torch.tensor(rows, dtype=t.dtype, device=t.device)   # noqa: E731
                c = (mk([[1.0, H, 1.0], [W, 1.0, 1.0]]), mk([[1.0, W, kw], [H, 1.0, kh]]), mk([[1.0, 1.0, 2.0], [1.0, 1.0, 2.0]]))
                _NORM_CONSTS[key] = c
            return a * c[0] / c[1] * c[2]
    else:
        a = np.asarray(pairwise_t_matrix)[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]].copy()
    a[..., 0, 1] = a[..., 0, 1] * H / W
    a[..., 1, 0] = a[..., 1, 0] * W / H
    a[..., 0, 2] = a[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    a[..., 1, 2] = a[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return a


def pairwise_to_host(pairwise_t_matrix):
    """Where the fusion kernels read the (tiny) affine matrices from.  Returns (matrix, grid_is_f64).

    A CUDA tensor STAYS on the device (what the reference's `train_utils.to_device` hands to the model): the warp kernels
    then read the poses from device memory at run time -- no host round trip, and a captured HIP graph follows the poses
    of the frame it is replayed on.  Host tensors / numpy arrays are returned as numpy and travel as launch arguments."""
    if isinstance(pairwise_t_matrix, torch.Tensor):
        if pairwise_t_matrix.is_cuda:
            t = pairwise_t_matrix.detach()
            return t, t.dtype == torch.float64
        arr = pairwise_t_matrix.detach().numpy()
    else:
        arr = np.asarray(pairwise_t_matrix)
    return arr, arr.dtype == np.float64


# ---- pose algebra the datasets / tools call around the hot path (host numpy, 4x4 matrices) ---------------------------
def regroup(x, record_len):
    """transformation_utils.py:16-19 (and fusion_in_one.py:48-51): split the agent axis by scene."""
    cum_sum_len = torch.cumsum(record_len, dim=0)
    return torch.tensor_split(x, cum_sum_len[:-1].cpu())


def x_to_world(pose):
    """transformation_utils.py:264-307: pose [x, y, z, roll, yaw, pitch] (degrees, CARLA convention) -> T_world_x."""
    x, y, z, roll, yaw, pitch = pose[:]
    c_y, s_y = np.cos(np.radians(yaw)), np.sin(np.radians(yaw))
    c_r, s_r = np.cos(np.radians(roll)), np.sin(np.radians(roll))
    c_p, s_p = np.cos(np.radians(pitch)), np.sin(np.radians(pitch))
    m = np.identity(4)
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    m[0, 0] = c_p * c_y
    m[0, 1] = c_y * s_p * s_r - s_y * c_r
    m[0, 2] = -c_y * s_p * c_r - s_y * s_r
    m[1, 0] = s_y * c_p
    m[1, 1] = s_y * s_p * s_r + c_y * c_r
    m[1, 2] = -s_y * s_p * c_r + c_y * s_r
    m[2, 0] = s_p
    m[2, 1] = -c_p * s_r
    m[2, 2] = c_p * c_r
    return m


def x1_to_x2(x1, x2):
    """transformation_utils.py:310-334: T_x2_x1 = inv(T_world_x2) @ T_world_x1."""
    return np.dot(np.linalg.inv(x_to_world(x2)), x_to_world(x1))


def get_pairwise_transformation(base_data_dict, max_cav, proj_first):
    """transformation_utils.py:21-66: (L,L,4,4) float64, [i,j] = T_j<-i = solve(T_world_j, T_world_i); identity on the
    diagonal, for unused slots and when the clouds were projected to the ego first."""
    pairwise = np.tile(np.eye(4), (max_cav, max_cav, 1, 1))
    if proj_first:
        return pairwise
    t_list = [x_to_world(c['params']['lidar_pose']) for c in base_data_dict.values()]
    for i in range(len(t_list)):
        for j in range(len(t_list)):
            if i != j:
                pairwise[i, j] = np.linalg.solve(t_list[j], t_list[i])
    return pairwise
