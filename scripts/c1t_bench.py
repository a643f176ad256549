"""heal_conv1x1_tiled (128 x 128 x 32 core, 32x32x2 MFMA) vs heal_conv1x1 (64 x 64 tiles, 16x16x4) at the pointwise shapes of the
two headline scenes, with the fused epilogues they run with.  HIP events around 10 back-to-back launches; max |diff| against
F.conv2d (fp32).  Usage: python scripts/c1t_bench.py [--json out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from heal_amd import ops

SHAPES = [  # n, cin, cout, H, W, residual, act   (scene5: 5 agents through the pyramid; camera trunks; config 5 shrink / heads)
    (5, 64, 128, 256, 256, False, 1), (5, 128, 64, 256, 256, True, 1), (5, 128, 256, 128, 128, False, 1),
    (5, 256, 128, 128, 128, True, 1), (5, 256, 512, 64, 64, False, 1), (5, 512, 256, 64, 64, True, 1),
    (5, 64, 256, 128, 128, False, 0), (1, 128, 512, 128, 128, False, 1), (1, 256, 2048, 64, 64, False, 1),
    (5, 256, 64, 128, 128, False, 1), (5, 512, 128, 64, 64, False, 1), (3, 64, 256, 256, 256, False, 3),
    (3, 256, 64, 256, 256, True, 0), (8, 256, 256, 128, 128, False, 1), (1, 256, 128, 256, 256, False, 1),
]


def timeit(fn, reps=10, inner=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return float(np.median(ts))


def main():
    out = []
    for n, cin, cout, H, W, res, act in SHAPES:
        g = torch.Generator().manual_seed(cin * 1000 + cout)
        x = torch.randn((n, cin, H, W), generator=g).cuda()
        w = (torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5).cuda()
        b = torch.randn((cout,), generator=g).cuda()
        r = torch.randn((n, cout, H, W), generator=g).cuda() if res else None
        ref = F.conv2d(x, w, b)
        if res:
            ref = ref + r
        ref = {0: lambda t: t, 1: torch.relu, 3: lambda t: F.gelu(t)}[act](ref)
        row = {"shape": f"{cin}->{cout} @{H}x{W} x{n}" + (" +res" if res else "") + f" act{act}"}
        flops = 2.0 * n * cin * cout * H * W
        for mode in ("0", "force"):
            os.environ["HEAL_C1_TILED"] = mode
            y = ops.conv1x1(x, w, b, r, act)
            err = float((y - ref).abs().max() / ref.abs().max())
            t = timeit(lambda: ops.conv1x1(x, w, b, r, act))
            key = "tiled" if mode == "force" else "base"
            row[f"{key}_us"] = round(t, 1)
            row[f"{key}_TF"] = round(flops / t / 1e6, 1)
            row[f"{key}_relerr"] = err
        os.environ.pop("HEAL_C1_TILED")
        row["speedup"] = round(row["base_us"] / row["tiled_us"], 3)
        row["default_takes_tiled"] = bool(ops.conv1x1_tiled_ok(n, cin, cout, H * W))
        print(row, flush=True)
        out.append(row)
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
