import os, sys, torch
sys.path.insert(0, "/root/repo")
from heal_amd import ops
n, width, cout, hw, g = 3, 128, 64, 64, 32
x = torch.randn((n, width, hw, hw), device="cuda")
w2 = torch.randn((width, width // g, 3, 3), device="cuda"); b2 = torch.randn((width,), device="cuda")
w3 = torch.randn((cout, width, 1, 1), device="cuda"); b3 = torch.randn((cout,), device="cuda")
r = torch.randn((n, cout, hw, hw), device="cuda")
for _ in range(3):
    ops.gconv_conv3(x, w2, b2, g, w3, b3, r, True)
torch.cuda.synchronize()
