"""heal_conv1x1 at the low-K, wide-output pointwise layers of scene5 (cin = 64: ResNet101 layer1 of the m4 camera trunk, PyramidFusion level 0)
under the instantiated tile shapes (HEAL_C1_CFG), graph-replay timing; GB/s = (x + y [+ residual]) / time.
    python scripts/c1_lowk_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops

SHAPES = [("r101 64->256 @4x84x112", 4, 64, 256, 84, 112, False), ("r101 64->64 @4x84x112", 4, 64, 64, 84, 112, False),
          ("r101 256->64 @4x84x112", 4, 256, 64, 84, 112, False), ("L0 64->128 @3x256x256", 3, 64, 128, 256, 256, False),
          ("L0 128->64 +res @3x256x256", 3, 128, 64, 256, 256, True), ("L0cam 64->128 @2x144x144", 2, 64, 128, 144, 144, False),
          ("r101 128->512 @4x42x56", 4, 128, 512, 42, 56, False), ("L1 128->256 @3x128x128", 3, 128, 256, 128, 128, False)]


def main():
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    for name, n, cin, cout, H, W, res in SHAPES:
        x = torch.randn((n, cin, H, W), device="cuda")
        w = torch.randn((cout, cin, 1, 1), device="cuda") / cin ** 0.5
        b = torch.randn((cout,), device="cuda")
        r = torch.randn((n, cout, H, W), device="cuda") if res else None
        nbytes = 4.0 * n * H * W * (cin + cout * (2 if res else 1))
        row = {}
        for cfg in ("64,64,32", "128,64,32", "64,128,32", "128,128,32"):
            os.environ["HEAL_C1_CFG"] = cfg
            if cout % int(cfg.split(",")[0]):
                continue
            us = ops.graph_period_ms(lambda: ops.conv1x1(x, w, b, r, 1), reps=10, iters=5) * 1e3
            row[cfg] = {"us": round(us, 1), "TB/s": round(nbytes / us / 1e6, 2)}
        os.environ.pop("HEAL_C1_CFG")
        print(name, json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
