"""PMC probe target: a few launches of one kernel shape.  usage: python scripts/pmc_probe.py conv1x1|wino|grouped"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "conv1x1"
if which == "conv1x1":
    x = torch.randn((5, 256, 64, 64), device="cuda"); w = torch.randn((512, 256, 1, 1), device="cuda") / 16; b = torch.randn(512, device="cuda")
    f = lambda: ops.conv1x1(x, w, b, None, 1)
elif which == "conv1x1_hbm":
    x = torch.randn((5, 64, 256, 256), device="cuda"); w = torch.randn((128, 64, 1, 1), device="cuda") / 8; b = torch.randn(128, device="cuda")
    f = lambda: ops.conv1x1(x, w, b, None, 1)
elif which == "wino":
    x = torch.randn((1, 384, 256, 256), device="cuda"); w = torch.randn((256, 384, 3, 3), device="cuda") / 60; b = torch.randn(256, device="cuda")
    f = lambda: ops.conv3x3(x, w, b, None, True, 1)
elif which == "grouped":
    x = torch.randn((5, 512, 64, 64), device="cuda"); w = torch.randn((512, 16, 3, 3), device="cuda") / 12; b = torch.randn(512, device="cuda")
    f = lambda: ops.grouped_conv3x3(x, w, b, 32, 1, True)
for _ in range(6):
    f()
torch.cuda.synchronize()
