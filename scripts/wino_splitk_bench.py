"""Winograd split-K (heal_conv3x3_winograd_splitk) at the small-map, deep-reduction 3x3 layers of the camera trunk, graph-replay timing.
    python scripts/wino_splitk_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops

SHAPES = [("up1 432->512 @24x32 x4", 4, 432, 512, 24, 32), ("512->512 @24x32 x4", 4, 512, 512, 24, 32),
          ("up2b 512->512 @48x64 x4", 4, 512, 512, 48, 64), ("256->256 @32x32 x2", 2, 256, 256, 32, 32),
          ("512->512 @16x16 x5", 5, 512, 512, 16, 16)]


def timed(fn, reps=20):
    return ops.graph_period_ms(fn, reps=reps, iters=5) * 1e3


def main():
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    out = {}
    for name, n, cin, cout, H, W in SHAPES:
        x = torch.randn((n, cin, H, W), device="cuda")
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (9 * cin) ** 0.5
        b = torch.randn((cout,), device="cuda")
        r = torch.randn((n, cout, H, W), device="cuda")
        row = {}
        ref = None
        for ks in (1, 2, 3, 4, 6, 8):
            os.environ["HEAL_C3_KSPLIT"] = str(ks)
            y = ops.conv3x3(x, w, b, r, True, 1)
            if ref is None:
                ref = y
            err = float((y - ref).abs().max() / ref.abs().max())
            assert err < 1e-5, (name, ks, err)
            row[f"ksplit{ks}_us"] = round(timed(lambda: ops.conv3x3(x, w, b, r, True, 1)), 1)
        os.environ.pop("HEAL_C3_KSPLIT")
        waves = ops.conv3x3_winograd_waves(n, cout, H, W)
        row["default_ksplit"] = ops.conv3x3_winograd_ksplit(n, cin, cout, H, W, waves)
        row["waves"] = waves
        out[name] = row
        print(name, json.dumps(row), flush=True)
    return out


if __name__ == "__main__":
    main()
