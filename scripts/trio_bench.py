"""The un-fused ResNeXt bottleneck (1x1 -> 32-group 3x3 -> 1x1 + identity + ReLU; resblock.py:100-122, pyramid_fuse.py:71-79) at the
level-0 / level-1 shapes of scene5 AFTER the camera crop, graph-replay timing per launch chain, with its floors:
HBM = (x + y) only (a perfect fusion), and the fp32 matrix time of its 2 x 1x1 + grouped 3x3 at the 157.3 TFLOP/s peak.
    python scripts/trio_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops

SHAPES = [("L0 lidar 3x64x256x256", 3, 64, 256, 256), ("L0 camcrop 2x64x144x144", 2, 64, 144, 144),
          ("L1 lidar 3x128x128x128", 3, 128, 128, 128), ("L1 camcrop 2x128x96x96", 2, 128, 96, 96),
          ("L2 all 5x256x64x64", 5, 256, 64, 64)]


def main():
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    rows = []
    for name, n, c, H, W in SHAPES:
        width, g = 2 * c, 32
        x = torch.randn((n, c, H, W), device="cuda")
        w1 = torch.randn((width, c, 1, 1), device="cuda") / c ** 0.5
        b1 = torch.randn((width,), device="cuda") * 0.1
        w2 = torch.randn((width, width // g, 3, 3), device="cuda") / (9 * width // g) ** 0.5
        b2 = torch.randn((width,), device="cuda") * 0.1
        w3 = torch.randn((c, width, 1, 1), device="cuda") / width ** 0.5
        b3 = torch.randn((c,), device="cuda") * 0.1

        def trio():
            t1 = ops.conv1x1(x, w1, b1, None, 1)
            t2 = ops.grouped_conv3x3(t1, w2, b2, g, 1, True)
            return ops.conv1x1(t2, w3, b3, x, 1)
        parts = {"conv1": lambda: ops.conv1x1(x, w1, b1, None, 1)}
        t1 = parts["conv1"]()
        parts["gconv"] = lambda: ops.grouped_conv3x3(t1, w2, b2, g, 1, True)
        t2 = parts["gconv"]()
        parts["conv3"] = lambda: ops.conv1x1(t2, w3, b3, x, 1)
        us = ops.graph_period_ms(trio, reps=10, iters=5) * 1e3
        px = n * H * W
        flops = 2.0 * px * (c * width * 2 + 9 * (width // g) * width)
        row = {"shape": name, "trio_us": round(us, 1),
               **{k + "_us": round(ops.graph_period_ms(f, reps=10, iters=5) * 1e3, 1) for k, f in parts.items()},
               "hbm_floor_us_at_8TBs": round(2 * px * c * 4 / 8e12 * 1e6, 1), "unfused_bytes_MB": round(px * 4 * (c + width * 4 + c * 2) / 1e6, 1),
               "mfma_floor_us": round(flops / 157.3e12 * 1e6, 1), "GFLOP": round(flops / 1e9, 2)}
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
