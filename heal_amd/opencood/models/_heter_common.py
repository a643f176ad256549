"""Shared construction logic of the heterogeneous models (reference: the per-modality loops at
heter_pyramid_collab.py:35-77, heter_pyramid_single.py:30-63, heter_model_late.py:26-69)."""
import importlib
from collections import OrderedDict

import torch
import torch.nn.functional as F


def modality_names(args):
    return [x for x in args.keys() if x.startswith("m") and x[1:].isdigit()]


def find_encoder(core_method):
    lib = importlib.import_module("heal_amd.opencood.models.heter_encoders")
    target = core_method.replace("_", "").lower()
    for name, cls in lib.__dict__.items():
        if name.lower() == target:
            return cls
    raise KeyError(f"encoder '{core_method}' not found in heter_encoders")


def center_crop(x, target_h, target_w):
    """torchvision.transforms.CenterCrop((th,tw)) on [...,H,W]: zero-pad when the
    target is larger (left/top floor, right/bottom ceil), then crop at round((H-th)/2)."""
    H, W = x.shape[-2:]
    if target_w > W or target_h > H:
        pl = (target_w - W) // 2 if target_w > W else 0
        pt = (target_h - H) // 2 if target_h > H else 0
        pr = (target_w - W + 1) // 2 if target_w > W else 0
        pb = (target_h - H + 1) // 2 if target_h > H else 0
        x = F.pad(x, [pl, pr, pt, pb])
        H, W = x.shape[-2:]
    top = int(round((H - target_h) / 2.0))
    left = int(round((W - target_w) / 2.0))
    return x[..., top:top + target_h, left:left + target_w]


def record_len_to_list(record_len):
    if isinstance(record_len, torch.Tensor):
        return [int(v) for v in record_len.detach().cpu().tolist()]
    return [int(v) for v in record_len]
