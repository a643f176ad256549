"""Host mirror of the opencood/utils/common_utils.py helpers on the hot path: same names, argument
meaning and return types.  The shapely polygon objects of the reference (`convert_format`, `compute_iou`) are replaced
by plain [n,4,2] quad arrays evaluated by heal_quad_iou (fp64 convex clip, cast to float32 like the reference)."""
import numpy as np
import torch

from heal_amd import ops


def check_numpy_to_torch(x):
    """common_utils.py:116-119: numpy arrays become float32 tensors."""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def limit_period(val, offset=0.5, period=2 * np.pi):
    """common_utils.py:104-113: val - floor(val / period + offset) * period."""
    val, is_numpy = check_numpy_to_torch(val)
    ans = val - torch.floor(val / period + offset) * period
    return ans.numpy() if is_numpy else ans


def convert_format(boxes_array):
    """common_utils.py:255-270 builds shapely Polygons from corners [i, :2], i < 4; here the same footprints as a
    float32 array [n,4,2] (the operand type of `compute_iou` below)."""
    b = boxes_array.detach().cpu().numpy() if isinstance(boxes_array, torch.Tensor) else np.asarray(boxes_array)
    return np.ascontiguousarray(b[:, :4, :2], dtype=np.float32)


def compute_iou(box, boxes):
    """common_utils.py:230-252: IoU of one footprint against a list -> float32 array.  Evaluated on the device."""
    box = np.asarray(box, np.float32).reshape(1, 4, 2)
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4, 2)
    if boxes.shape[0] == 0:
        return np.zeros((0,), np.float32)
    dev = torch.device("cuda", torch.cuda.current_device())
    iou = ops.quad_iou(torch.from_numpy(box).to(dev), torch.from_numpy(boxes).to(dev))
    return iou[0].cpu().numpy()


def torch_tensor_to_numpy(torch_tensor):
    """common_utils.py:273-288."""
    return torch_tensor.numpy() if not torch_tensor.is_cuda else torch_tensor.cpu().detach().numpy()
