"""heal_conv3x3_winograd (F(2x2,3x3)) at the stride-1 shapes of the two BASELINE scenes: HIP-event medians, executed matrix
TFLOP/s (direct / 2.25) and the error against the library convolution.  A/B of two builds: HEAL_AMD_LIB=<other .so>."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from heal_amd import ops
from scripts.wino_ab import SHAPES, timed

os.environ["HEAL_C3_ALGO"] = "winograd"
out = {}
for name, n, cin, cout, H, W, res in SHAPES:
    torch.manual_seed(0)
    x = torch.randn((n, cin, H, W), device="cuda")
    w = torch.randn((cout, cin, 3, 3), device="cuda") / (9 * cin) ** 0.5
    b = torch.randn((cout,), device="cuda")
    r = torch.randn((n, cout, H, W), device="cuda") if res else None
    us = timed(lambda: ops.conv3x3(x, w, b, r, True, 1))
    got = ops.conv3x3(x, w, b, r, True, 1)
    ref = F.conv2d(x, w, b, padding=1)
    ref = torch.relu(ref + r if res else ref)
    err = float((got - ref).abs().max() / ref.abs().max())
    fl = 2.0 * 9 * cin * cout * H * W * n
    out[name] = {"us": round(us, 1), "executed_TFLOPs": round(fl / 2.25 / us * 1e-6, 1), "rel_err_vs_library": err}
    print(name, json.dumps(out[name]), flush=True)
    assert err < 1e-4
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
