"""heal_gconv_conv3 (grouped 3x3 + pointwise conv + identity + ReLU in one kernel) vs the two-kernel path at the PyramidFusion level-1 /
level-2 shapes, 5 agents.  Kernel-own durations (ops._Timed kernel_events).  Usage: python scripts/gc3_bench.py  (on the GPU box)"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops  # noqa: E402


def run(n, width, cout, hw):
    g = 32
    x = torch.randn((n, width, hw, hw), device="cuda")
    w2 = torch.randn((width, width // g, 3, 3), device="cuda") / (9 * width // g) ** 0.5
    b2 = torch.randn((width,), device="cuda") * 0.1
    w3 = torch.randn((cout, width, 1, 1), device="cuda") / width ** 0.5
    b3 = torch.randn((cout,), device="cuda") * 0.1
    r = torch.randn((n, cout, hw, hw), device="cuda")
    fused = lambda: ops.gconv_conv3(x, w2, b2, g, w3, b3, r, True)
    split = lambda: ops.conv1x1(ops.grouped_conv3x3(x, w2, b2, g, 1, True), w3, b3, r, 1)
    a, b = fused(), split()
    ref = torch.relu(F.conv2d(torch.relu(F.conv2d(x.double(), w2.double(), b2.double(), 1, 1, 1, g)), w3.double(), b3.double()) + r.double())
    res = {"shape": f"{n} x {width} -> {cout} @ {hw}x{hw}", "err_fused": float((a.double() - ref).abs().max() / ref.abs().max()),
           "err_split": float((b.double() - ref).abs().max() / ref.abs().max())}
    ops.TIMING = {}
    for _ in range(20):
        fused(); split()
    torch.cuda.synchronize()
    for name, (calls, ms) in ops.timing_summary().items():
        res[name + "_us"] = round(ms * 1e3, 1)
    ops.TIMING = None
    return res


if __name__ == "__main__":
    out = [run(5, 128, 64, 256), run(5, 256, 128, 128), run(3, 128, 64, 64)]
    print(json.dumps(out, indent=1))
