"""Closed-form deterministic parameter fill shared by the golden-vector generator (which fills the
imported REFERENCE modules in the build container) and the tests (which fill this repo's modules /
oracle with the same numbers).  Weights never travel: only inputs and expected outputs do.
"""
import zlib

import numpy as np
import torch


def det_values(name, shape, kind):
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.float64)
    base = (zlib.crc32(name.encode()) % 1000) * 0.37
    v = np.sin(idx * 0.6180339887 + base)
    if kind == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else max(n, 1)
        v = v * np.sqrt(2.0 / max(fan_in, 1))
    elif kind == "bias":
        v = 0.1 * v
    elif kind == "bn_weight":
        v = 1.0 + 0.1 * v
    elif kind == "bn_bias":
        v = 0.1 * v
    elif kind == "running_mean":
        v = 0.1 * v
    elif kind == "running_var":
        v = 1.0 + 0.3 * v
    elif kind == "unit":
        v = 0.5 * v
    return v.reshape(shape).astype(np.float32)


def _kind(key, tensor, bn_prefixes):
    leaf = key.rsplit(".", 1)[-1]
    prefix = key.rsplit(".", 1)[0] if "." in key else ""
    if leaf == "running_mean":
        return "running_mean"
    if leaf == "running_var":
        return "running_var"
    if leaf == "num_batches_tracked":
        return None
    if prefix in bn_prefixes:
        return "bn_weight" if leaf == "weight" else "bn_bias"
    if leaf == "weight" and tensor.dim() == 1:
        return "bn_weight"  # LayerNorm-style gain
    if leaf == "weight":
        return "weight"
    if leaf == "bias":
        return "bias"
    if leaf == "gamma":
        return "unit"  # ConvNeXt layer scale: use O(1) values so the branch matters
    return "unit"


def fill_module(module):
    """Fill every parameter/buffer of `module` in place from (state_dict key, flat index)."""
    sd = module.state_dict()
    bn_prefixes = {k.rsplit(".", 1)[0] for k in sd if k.endswith("running_var")}
    with torch.no_grad():
        for k, t in sd.items():
            kind = _kind(k, t, bn_prefixes)
            if kind is None or not t.dtype.is_floating_point:
                continue
            t.copy_(torch.from_numpy(det_values(k, tuple(t.shape), kind)))
    return module
