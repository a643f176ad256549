"""Timing experiment for heal_conv3x3_winograd (HEAL_WG_DBG bits skip parts: results INVALID)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops
x = torch.randn((1, 384, 256, 256), device="cuda"); w = torch.randn((256, 384, 3, 3), device="cuda") / 60; b = torch.randn(256, device="cuda")
res = {}
for dbg in (0, 1, 2, 3, 4, 8, 12, 5, 15):
    os.environ["HEAL_WG_DBG"] = str(dbg)
    ts = []
    for i in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv3x3(x, w, b, None, True, 1); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    res[dbg] = round(float(np.median(ts[2:])), 1)
print(json.dumps(res))
