"""heal_conv_gemm against heal_conv3x3 (round-2 implicit GEMM) and MIOpen on the stride-2 3x3 layers of the two BASELINE configs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from heal_amd import ops
from scripts.k3_bench import timed
dev = torch.device("cuda:0")
torch.manual_seed(0)
out = {}
for (n, cin, cout, H, st) in ((8, 384, 256, 256, 2), (8, 128, 256, 256, 2), (5, 128, 128, 256, 2), (5, 256, 256, 128, 2), (8, 256, 256, 128, 1)):
    x = torch.randn(n, cin, H, H, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    b = torch.randn(cout, device=dev) * 0.1
    Ho = (H - 1) // st + 1
    r = torch.randn(n, cout, Ho, Ho, device=dev)
    fl = 2.0 * 9 * n * cin * cout * Ho * Ho
    ref, t_lib = timed(lambda: F.relu(F.conv2d(x, w, b, st, 1) + r), 5)
    os.environ["HEAL_CONV_GEMM"] = "0"
    old, t_old = timed(lambda: ops.conv3x3(x, w, b, r, True, st), 5)
    got, t_new = timed(lambda: ops.conv_gemm(x, w, b, r, True, st), 5)
    os.environ["HEAL_CONV_GEMM"] = "1"
    err = float((got - ref).abs().max() / ref.abs().max()); err_old = float((old - ref).abs().max() / ref.abs().max())
    key = f"{cin}->{cout} @{H}^2x{n} s{st}"
    out[key] = {"miopen_us": round(t_lib, 1), "conv3x3_r2_us": round(t_old, 1), "conv_gemm_us": round(t_new, 1),
                "conv_gemm_TFLOPs": round(fl / t_new * 1e-6, 1), "r2_TFLOPs": round(fl / t_old * 1e-6, 1), "rel_err": err}
    print(key, out[key], flush=True)
    assert err < 1e-4 and err_old < 1e-4
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
