"""Winograd 3x3: F(4x4,3x3) (heal_conv3x3_winograd4) vs F(2x2,3x3) (heal_conv3x3_winograd) at the stride-1 shapes of the two
BASELINE scenes.  HIP events, median.    python scripts/wino_ab.py [--json out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from heal_amd import ops

SHAPES = [("shrink0 384->256 @256^2", 1, 384, 256, 256, 256, False), ("shrink1 256->256 @256^2", 1, 256, 256, 256, 256, False),
          ("m1 conv2 64->64 @256^2 x3 +res", 3, 64, 64, 256, 256, True), ("up2 552->512 @48x64 x4", 4, 552, 512, 48, 64, False),
          ("second bev 128->128 @256^2 x8", 8, 128, 128, 256, 256, False), ("second bev 256->256 @128^2 x8", 8, 256, 256, 128, 128, False),
          ("cfg5 shrink1 256->256 @128^2 x8", 8, 256, 256, 128, 128, False), ("second bev in 256->128 @256^2 x8", 8, 256, 128, 256, 256, False)]


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
    os.environ["HEAL_C3_ALGO"] = "winograd"
    out = {}
    for name, n, cin, cout, H, W, res in SHAPES:
        x = torch.randn((n, cin, H, W), device="cuda")
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (9 * cin) ** 0.5
        b = torch.randn((cout,), device="cuda")
        r = torch.randn((n, cout, H, W), device="cuda") if res else None
        row = {}
        outs = {}
        for algo in ("winograd", "winograd4"):
            os.environ["HEAL_C3_ALGO"] = algo
            row["f22_us" if algo == "winograd" else "f44_us"] = round(timed(lambda: ops.conv3x3(x, w, b, r, True, 1)), 1)
            outs[algo] = ops.conv3x3(x, w, b, r, True, 1)
        flops = 2.0 * 9 * cin * cout * H * W * n
        row["f44_direct_equiv_TFLOPs"] = round(flops / row["f44_us"] * 1e-6, 1)
        row["f44_executed_TFLOPs"] = round(flops / 4.0 / row["f44_us"] * 1e-6, 1)
        row["speedup"] = round(row["f22_us"] / row["f44_us"], 3)
        row["max_rel_diff"] = float((outs["winograd4"] - outs["winograd"]).abs().max() / outs["winograd"].abs().max())
        out[name] = row
        print(name, json.dumps(row), flush=True)
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
