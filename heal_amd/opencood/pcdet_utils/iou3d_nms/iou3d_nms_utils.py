"""pcdet 3D IoU / rotated NMS API on the MI355X kernels.

Host mirror of the reference module opencood/pcdet_utils/iou3d_nms/iou3d_nms_utils.py: same function names, argument
meaning and return values; the `iou3d_nms_cuda` extension calls are replaced by `heal_boxes_bev_matrix` /
`heal_nms_bev` (include/heal_amd.h).  Boxes are [x, y, z, dx, dy, dz, heading].
"""
import numpy as np
import torch

from heal_amd import ops


def _to_torch(x):
    """opencood/utils/common_utils.py:14-17 (check_numpy_to_torch)."""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:13-29.  The reference runs its CPU twin on host tensors; this build has no CPU compute
    path, so host boxes are staged to the current device, evaluated by the same kernel as `boxes_iou_bev`, and
    returned on the host (numpy in -> numpy out)."""
    boxes_a, is_numpy = _to_torch(boxes_a)
    boxes_b, _ = _to_torch(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    assert boxes_a.shape[1] == 7 and boxes_b.shape[1] == 7
    dev = torch.device("cuda", torch.cuda.current_device())
    ans = ops.boxes_bev_matrix(boxes_a.float().contiguous().to(dev), boxes_b.float().contiguous().to(dev), "iou").cpu()
    return ans.numpy() if is_numpy else ans


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:32-46: (N,7), (M,7) -> rotated BEV IoU (N,M)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return ops.boxes_bev_matrix(boxes_a.contiguous(), boxes_b.contiguous(), "iou")


def _heights(boxes):
    return boxes[:, 2] + boxes[:, 5] / 2, boxes[:, 2] - boxes[:, 5] / 2


def boxes_iou3d_gpu(boxes_a, boxes_b, return_union=False):
    """iou3d_nms_utils.py:152-181: BEV overlap x height overlap over the union volume (clamped at 1e-6)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max, a_min = _heights(boxes_a)
    b_max, b_min = _heights(boxes_b)
    overlaps_bev = ops.boxes_bev_matrix(boxes_a.contiguous(), boxes_b.contiguous(), "overlap")
    max_of_min = torch.max(a_min.view(-1, 1), b_min.view(1, -1))
    min_of_max = torch.min(a_max.view(-1, 1), b_max.view(1, -1))
    overlaps_h = torch.clamp(min_of_max - max_of_min, min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    union = torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)
    iou3d = overlaps_3d / union
    return (iou3d, union) if return_union else iou3d


def aligned_boxes_iou3d_gpu(boxes_a, boxes_b, return_union=False):
    """iou3d_nms_utils.py:109-149: row-aligned pairs -> (N,1).  (The reference fills the full N x N overlap matrix
    and keeps its diagonal; the values are the same.)"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    assert boxes_a.shape[0] == boxes_b.shape[0]
    a_max, a_min = _heights(boxes_a)
    b_max, b_min = _heights(boxes_b)
    full = ops.boxes_bev_matrix(boxes_a.contiguous(), boxes_b.contiguous(), "overlap")
    overlaps_bev = torch.diagonal(full).reshape(-1, 1)
    max_of_min = torch.max(a_min.view(-1, 1), b_min.view(-1, 1))
    min_of_max = torch.min(a_max.view(-1, 1), b_max.view(-1, 1))
    overlaps_h = torch.clamp(min_of_max - max_of_min, min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(-1, 1)
    union = torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)
    iou3d = overlaps_3d / union
    return (iou3d, union) if return_union else iou3d


def decode_boxes(boxes, pc_range, box_mean, box_std):
    """iou3d_nms_utils.py:66-80: normalised (N,8) [x,y,z,dx,dy,dz,sin,cos] -> (N,7)."""
    assert len(boxes.shape) == 2
    assert boxes.shape[1] == 8
    if isinstance(box_mean, list):
        box_mean = torch.tensor(box_mean, device=boxes.device)
    if isinstance(box_std, list):
        box_std = torch.tensor(box_std, device=boxes.device)
    boxes = boxes * box_std[None, :] + box_mean[None, :]
    out = torch.zeros((boxes.shape[0], 7), dtype=boxes.dtype, device=boxes.device)
    for i in range(3):
        out[:, i] = boxes[:, i] * (pc_range[i + 3] - pc_range[i]) + pc_range[i]
    out[:, 3:6] = boxes[:, 3:6].exp()
    out[:, 6] = torch.atan2(boxes[:, 6], boxes[:, 7])
    return out


def decode_boxes_and_iou3d(boxes_a, boxes_b, pc_range, box_mean, box_std):
    """iou3d_nms_utils.py:49-63."""
    return boxes_iou3d_gpu(decode_boxes(boxes_a, pc_range, box_mean, box_std),
                           decode_boxes(boxes_b, pc_range, box_mean, box_std))


def centroid_to_corners(boxes):
    """iou3d_nms_utils.py:184-192: (N,7) -> (N,8,3), numpy or torch."""
    if isinstance(boxes, np.ndarray):
        return _corners(boxes, np)
    if isinstance(boxes, torch.Tensor):
        return _corners(boxes, torch)
    raise TypeError('Input boxes should either be numpy array or torch tensor.')


def _corners(b, xp):
    """iou3d_nms_utils.py:195-240.  Corner k of the bottom face (k = 0..3: left-front, left-back, right-back,
    right-front), repeated for the top face (k + 4)."""
    sin_t, cos_t = xp.sin(b[:, -1]), xp.cos(b[:, -1])
    hx, hy = b[:, 3] / 2, b[:, 4] / 2
    xs = [b[:, 0] + hx * cos_t - hy * sin_t, b[:, 0] - hx * cos_t - hy * sin_t,
          b[:, 0] - hx * cos_t + hy * sin_t, b[:, 0] + hx * cos_t + hy * sin_t]
    ys = [b[:, 1] + hx * sin_t + hy * cos_t, b[:, 1] - hx * sin_t + hy * cos_t,
          b[:, 1] - hx * sin_t - hy * cos_t, b[:, 1] + hx * sin_t - hy * cos_t]
    zlo, zhi = b[:, 2] - b[:, 5] / 2, b[:, 2] + b[:, 5] / 2
    x = xp.stack(xs + xs, 1)
    y = xp.stack(ys + ys, 1)
    z = xp.stack([zlo] * 4 + [zhi] * 4, 1)
    return xp.stack([x, y, z], 2)


def giou3d(boxes_a_dec, boxes_b_dec):
    """iou3d_nms_utils.py:95-106."""
    corners_a = centroid_to_corners(boxes_a_dec)
    corners_b = centroid_to_corners(boxes_b_dec)
    iou, union = boxes_iou3d_gpu(boxes_a_dec, boxes_b_dec, return_union=True)
    lwh = torch.max(corners_a.max(dim=1)[0][:, None, :], corners_b.max(dim=1)[0]) \
        - torch.min(corners_a.min(dim=1)[0][:, None, :], corners_b.min(dim=1)[0])
    volume = lwh[..., 0] * lwh[..., 1] * lwh[..., 2]
    return iou - (volume - union) / volume


def decode_boxes_and_giou3d(boxes_a, boxes_b, pc_range, box_mean, box_std):
    """iou3d_nms_utils.py:83-93."""
    return giou3d(decode_boxes(boxes_a, pc_range, box_mean, box_std), decode_boxes(boxes_b, pc_range, box_mean, box_std))


def _nms(boxes, scores, thresh, pre_maxsize, rotated):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep, count = ops.nms_bev(boxes[order].contiguous(), thresh, rotated)
    return order[keep[:int(count.item())]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:252-270: rotated-BEV NMS -> (kept indices into `boxes`, None)."""
    return _nms(boxes, scores, thresh, pre_maxsize, True)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:273-289: heading ignored (axis-aligned BEV IoU)."""
    return _nms(boxes, scores, thresh, None, False)
