"""Host mirror of the opencood/utils/box_utils.py functions on the hot path (the fused operator
heal_decode_nms implements the same chain inside VoxelPostprocessor.post_process).  Same names, argument meaning and
return types; tensor math in torch with the reference's operation order, rotated NMS on the GPU (heal_nms_quads)."""
import numpy as np
import torch

from heal_amd import ops
from heal_amd.opencood.utils import common_utils


def boxes_to_corners_3d(boxes3d, order):
    """box_utils.py:152-204: (N,7) [x,y,z,h,w,l,yaw] ('hwl') or [x,y,z,l,w,h,yaw] ('lwh') -> (N,8,3) corners."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    b = boxes3d[:, [0, 1, 2, 5, 4, 3, 6]] if order == 'hwl' else boxes3d
    template = b.new_tensor(([1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1],
                             [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1])) / 2
    corners = b[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    corners = rotate_points_along_z(corners.view(-1, 8, 3), b[:, 6]).view(-1, 8, 3)
    corners = corners + b[:, None, 0:3]
    return corners.numpy() if is_numpy else corners


def rotate_points_along_z(points, angle):
    """common_utils.py:139-161 (kept next to its only hot-path caller)."""
    points, is_numpy = common_utils.check_numpy_to_torch(points)
    angle, _ = common_utils.check_numpy_to_torch(angle)
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(points.shape[0]), angle.new_ones(points.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    out = torch.cat((torch.matmul(points[:, :, 0:3].float(), rot), points[:, :, 3:]), dim=-1)
    return out.numpy() if is_numpy else out


def box3d_to_2d(box3d):
    """box_utils.py:207-222."""
    return box3d[:, :4, :2]


def corner2d_to_standup_box(box2d):
    """box_utils.py:225-248 (numpy, float64 result like np.zeros)."""
    out = np.zeros((box2d.shape[0], 4))
    out[:, 0] = np.min(box2d[:, :, 0], axis=1)
    out[:, 1] = np.min(box2d[:, :, 1], axis=1)
    out[:, 2] = np.max(box2d[:, :, 0], axis=1)
    out[:, 3] = np.max(box2d[:, :, 1], axis=1)
    return out


def corner_to_standup_box_torch(box_corner):
    """box_utils.py:251-275."""
    return torch.stack([box_corner[:, :, 0].min(1).values, box_corner[:, :, 1].min(1).values,
                        box_corner[:, :, 0].max(1).values, box_corner[:, :, 1].max(1).values], 1).float()


def project_box3d(box3d, transformation_matrix):
    """box_utils.py:278-316: (N,8,3) corners through a 4x4 homogeneous matrix."""
    assert transformation_matrix.shape == (4, 4)
    box3d, is_numpy = common_utils.check_numpy_to_torch(box3d)
    tfm, _ = common_utils.check_numpy_to_torch(transformation_matrix)
    c = box3d.transpose(1, 2)
    c = torch.cat((c, torch.ones((c.shape[0], 1, 8), device=c.device, dtype=c.dtype)), dim=1)
    out = torch.matmul(tfm.to(c.device), c)[:, :3, :].transpose(1, 2)
    return out if not is_numpy else out.numpy()


def get_mask_for_boxes_within_range_torch(boxes, gt_range):
    """box_utils.py:348-381: every corner inside [xmin,xmax] x [ymin,ymax]."""
    lo = torch.Tensor(gt_range[:2]).reshape(1, 1, -1).to(boxes.device)
    hi = torch.Tensor(gt_range[3:5]).reshape(1, 1, -1).to(boxes.device)
    return torch.all(torch.all(boxes[:, :, :2] >= lo, dim=-1) & torch.all(boxes[:, :, :2] <= hi, dim=-1), dim=-1)


def mask_boxes_outside_range_numpy(boxes, limit_range, order, min_num_corners=8, return_mask=False):
    """box_utils.py:384-421: boxes (N,7) or corners (N,8,3) as numpy; a box stays when at least `min_num_corners`
    corners lie inside [min, max] on all three axes."""
    assert boxes.shape[1] == 8 or boxes.shape[1] == 7
    new_boxes = boxes.copy()
    if boxes.shape[1] == 7:
        new_boxes = boxes_to_corners_3d(new_boxes, order)
    limit_range = np.asarray(limit_range)
    mask = ((new_boxes >= limit_range[0:3]) & (new_boxes <= limit_range[3:6])).all(axis=2)
    mask = mask.sum(axis=1) >= min_num_corners
    if return_mask:
        return boxes[mask], mask
    return boxes[mask]


def remove_large_pred_bbx(bbx_3d):
    """box_utils.py:840-869, including its quirks: the 'z' extent is computed from column 1 (y) and enters the mask as a
    truth value (non-zero), exactly as the reference does."""
    x_len = bbx_3d[:, :, 0].max(1)[0] - bbx_3d[:, :, 0].min(1)[0]
    y_len = bbx_3d[:, :, 1].max(1)[0] - bbx_3d[:, :, 1].min(1)[0]
    z_len = bbx_3d[:, :, 1].max(1)[0] - bbx_3d[:, :, 1].min(1)[0]
    index = torch.logical_and(x_len <= 6, y_len <= 6)
    return torch.logical_and(index, z_len)


def remove_bbx_abnormal_z(bbx_3d):
    """box_utils.py:872-890."""
    return torch.logical_and(bbx_3d[:, :, 2].min(1)[0] >= -3, bbx_3d[:, :, 2].max(1)[0] <= 1)


def nms_rotated(boxes, scores, threshold):
    """box_utils.py:693-738: boxes (N,4,2) or (N,8,3) torch tensor, scores (N,) -> np.int32 indices of the kept boxes in
    pick order; top-1000 by score, greedy, footprint IoU > threshold suppresses.  The greedy pass runs on the GPU
    (heal_nms_quads); exactly tied scores are ordered larger-index-first (numpy's unstable argsort leaves them
    implementation-defined in the reference)."""
    if boxes.shape[0] == 0:
        return np.array([], dtype=np.int32)
    dev = boxes.device if boxes.is_cuda else torch.device("cuda", torch.cuda.current_device())
    quads = boxes.detach()[:, :4, :2].to(device=dev, dtype=torch.float32).contiguous()
    sc = scores.detach().to(dev)
    order = torch.argsort(sc, stable=True).flip(0)[:1000]  # stable ascending reversed = descending, larger index first
    keep, count = ops.nms_quads(quads[order].contiguous(), threshold)
    return order[keep[:int(count.item())]].cpu().numpy().astype(np.int32)
