import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops
os.environ["HEAL_C3_ALGO"] = "winograd"
torch.manual_seed(0)
for waves in ("8", "4"):
    os.environ["HEAL_WG_WAVES"] = waves
    for (n, cin, cout, H, W) in [(1, 8, 64, 16, 16), (1, 8, 64, 8, 16), (1, 16, 64, 32, 32), (1, 384, 256, 64, 64), (2, 64, 64, 40, 48)]:
        x = torch.randn((n, cin, H, W), device="cuda"); w = torch.randn((cout, cin, 3, 3), device="cuda") / (9 * cin) ** 0.5
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
        got = ops.conv3x3(x, w, None, None, False, 1)
        e = (got.double() - ref).abs()
        bad = (e > 1e-3).float()
        print(f"waves {waves} {n}x{cin}->{cout} {H}x{W}: max err {float(e.max()):.3e} finite {bool(torch.isfinite(got).all())} bad frac {float(bad.mean()):.4f}", end="")
        if bad.sum() > 0:
            bc = bad.sum((0, 2, 3)); br = bad.sum((0, 1, 3)); bx = bad.sum((0, 1, 2))
            print("  bad channels:", bc.nonzero().flatten()[:8].tolist(), "n", int((bc > 0).sum()), " rows:", br.nonzero().flatten()[:12].tolist(), " cols:", bx.nonzero().flatten()[:12].tolist(), end="")
        print()
