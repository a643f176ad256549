"""heal_stem7x7 (7x7/2 conv + BN + ReLU + 3x3/2 max-pool, one kernel) vs the library sequence it replaces (channel-slice copy +
MIOpen convolution + heal_bias_act + ATen max-pool) at the ResNet101 camera agent's size (4 x 336 x 448)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from heal_amd import ops


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
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return float(np.median(ts))


x = torch.randn((4, 3, 336, 448), device="cuda")
w = torch.randn((64, 3, 7, 7), device="cuda") / 12
b = torch.randn(64, device="cuda")
t_new = timeit(lambda: ops.stem7x7(x, w, b, True))
t_lib = timeit(lambda: F.max_pool2d(ops.bias_act_(F.conv2d(x[:, :3].contiguous(), w, None, 2, 3), b, None, True), 3, 2, 1))
fl = 2.0 * 4 * 64 * 147 * 168 * 224
print({"stem7x7_us": round(t_new, 1), "library_sequence_us": round(t_lib, 1), "useful_TFLOPs": round(fl / t_new / 1e6, 1)})
