"""32-group 3x3 convolutions of the PyramidFusion ResNeXt stages at 5 agents: matrix-core kernels (HEAL_GCONV_MFMA=1) vs the
vector-ALU stencil (=0).  HIP events, median."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from heal_amd import ops


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


for C, hw, st in ((128, 256, 1), (256, 128, 1), (512, 64, 1), (256, 256, 2), (512, 128, 2)):
    x = torch.randn((5, C, hw, hw), device="cuda"); w = torch.randn((C, C // 32, 3, 3), device="cuda") * 0.1
    b = torch.randn((C,), device="cuda")
    ref = torch.relu(torch.nn.functional.conv2d(x[:1].double(), w.double(), b.double(), st, 1, 1, 32)).float()
    line = f"C={C} ({C // 32}/group) {hw}x{hw} stride {st}:"
    for mode in ("1", "0", "16"):
        os.environ["HEAL_GCONV_MFMA"] = mode
        t = timeit(lambda: ops.grouped_conv3x3(x, w, b, 32, st, True))
        got = ops.grouped_conv3x3(x[:1].contiguous(), w, b, 32, st, True)
        err = float((got - ref).abs().max() / ref.abs().max())
        flops = 2.0 * 9 * 5 * C * (C // 32) * hw * hw / st ** 2
        line += f"  mfma={mode}: {t:7.1f} us {4 * x.numel() * (1 + 1 / st ** 2) / t / 1e3:6.0f} GB/s {flops / t / 1e6:5.1f} TF relerr {err:.1e}"
    print(line, flush=True)
