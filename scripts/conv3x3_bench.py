"""heal_conv3x3 (fp32 MFMA implicit GEMM, fused epilogue) vs the library path it replaces (MIOpen conv without bias +
heal_bias_act) at the dense 3x3 shapes of the BASELINE scene.  Usage: python scripts/conv3x3_bench.py  (GPU box)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops  # noqa: E402

SHAPES = [  # name, n, cin, cout, H, W, stride, residual
    ("shrink0 384->256 @256^2", 1, 384, 256, 256, 256, 1, False),
    ("shrink1 256->256 @256^2", 1, 256, 256, 256, 256, 1, False),
    ("m1 block conv1 64->64 s2 @512^2 x3", 3, 64, 64, 512, 512, 2, False),
    ("m1 block conv2 64->64 @256^2 x3 +res", 3, 64, 64, 256, 256, 1, True),
    ("cam block conv1 128->64 s2 @256^2", 1, 128, 64, 256, 256, 2, False),
    ("cam block conv2 64->64 @128^2 +res", 1, 64, 64, 128, 128, 1, True),
    ("up1 432->512 @24x32 x4", 4, 432, 512, 24, 32, 1, False),
    ("up2 552->512 @48x64 x4", 4, 552, 512, 48, 64, 1, False),
    ("up2b 512->512 @48x64 x4", 4, 512, 512, 48, 64, 1, False),
    ("r101 l1 64->64 @84x112 x4", 4, 64, 64, 84, 112, 1, False),
    ("r101 l2 128->128 @42x56 x4", 4, 128, 128, 42, 56, 1, False),
    ("r101 l2 128->128 s2 @84x112 x4", 4, 128, 128, 84, 112, 2, False),
]


def timed(fn, iters=12, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def main():
    out = {}
    for name, n, cin, cout, H, W, st, res in SHAPES:
        x = torch.randn((n, cin, H, W), device="cuda")
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (9 * cin) ** 0.5
        b = torch.randn((cout,), device="cuda")
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        r = torch.randn((n, cout, Ho, Wo), device="cuda") if res else None
        flops = 2.0 * 9 * cin * cout * Ho * Wo * n

        def lib():
            y = torch.nn.functional.conv2d(x, w, None, st, 1)
            return ops.bias_act_(y, b, r, True)
        t_lib = timed(lib)
        row = {"GF": round(flops / 1e9, 1), "miopen+bias_act_us": round(t_lib, 1), "miopen_TF": round(flops / t_lib / 1e6, 1)}
        os.environ["HEAL_C3_ALGO"] = "direct"
        for th in ((16, 8, 4) if st == 1 else (8, 4)):
            os.environ["HEAL_C3_TH"] = str(th)
            t = timed(lambda: ops.conv3x3(x, w, b, r, True, st))
            row[f"direct_th{th}_us"] = round(t, 1)
            row[f"direct_th{th}_TF"] = round(flops / t / 1e6, 1)
        os.environ.pop("HEAL_C3_TH", None)
        os.environ.pop("HEAL_C3_ALGO", None)
        row["default_us"] = round(timed(lambda: ops.conv3x3(x, w, b, r, True, st)), 1)
        row["default_algo"] = ops.conv3x3_algo(st, n, cout, H, W)
        if st == 1:
            os.environ["HEAL_C3_ALGO"] = "winograd"
            os.environ["HEAL_WG_WAVES"] = "8"
            row["winograd_w8_us"] = round(timed(lambda: ops.conv3x3(x, w, b, r, True, st)), 1)
            os.environ["HEAL_WG_WAVES"] = "4"
            t = timed(lambda: ops.conv3x3(x, w, b, r, True, st))
            os.environ.pop("HEAL_C3_ALGO", None)
            os.environ.pop("HEAL_WG_WAVES", None)
            row["winograd_us"] = round(t, 1)
            row["winograd_TF_equiv"] = round(flops / t / 1e6, 1)
        err = float((ops.conv3x3(x, w, b, r, True, st) - lib()).abs().max() / lib().abs().max())
        row["rel_diff_vs_miopen"] = err
        out[name] = row
        print(name, json.dumps(row), flush=True)
    json.dump(out, open(os.path.join("gpurun_out", "r02_conv3x3_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
