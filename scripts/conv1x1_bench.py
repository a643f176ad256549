"""heal_conv1x1 (fused epilogue) vs the library path (F.conv2d without bias + heal_bias_act) at the PyramidFusion
ResNeXt shapes, 5 agents.  HIP events, median of `reps`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from heal_amd import ops


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def main():
    n = 5
    for cin, cout, hw, res in [(64, 128, 256, False), (128, 64, 256, True), (128, 256, 128, False), (256, 128, 128, True),
                               (256, 512, 64, False), (512, 256, 64, True), (64, 128, 128, False), (128, 256, 64, False)]:
        x = torch.randn((n, cin, hw, hw), device="cuda")
        w = torch.randn((cout, cin, 1, 1), device="cuda") / cin ** 0.5
        b = torch.randn((cout,), device="cuda")
        r = torch.randn((n, cout, hw, hw), device="cuda") if res else None
        t_new = timeit(lambda: ops.conv1x1(x, w, b, r, 1))
        if "--sweep" in sys.argv:
            best = []
            for bm in (128, 64):
                for bn in (128, 64):
                    for kc in (32,):
                        if cout % bm:
                            continue
                        os.environ["HEAL_C1_CFG"] = f"{bm},{bn},{kc}"
                        best.append((timeit(lambda: ops.conv1x1(x, w, b, r, 1), 10), bm, bn, kc))
            os.environ.pop("HEAL_C1_CFG")
            print("    sweep:", " ".join(f"({bm},{bn},{kc}):{t:.0f}" for t, bm, bn, kc in sorted(best)), flush=True)
        t_cg = None
        if ops.conv_gemm_supported(cin, cout, hw):
            t_cg = timeit(lambda: ops.conv_gemm(x, w, b, r, True, 1))
            err = float((ops.conv_gemm(x, w, b, r, True, 1) - ops.conv1x1(x, w, b, r, 1)).abs().max())
            print(f"    conv_gemm(1x1) {t_cg:7.1f} us ({2.0 * n * cin * cout * hw * hw / t_cg / 1e6:6.1f} TF)  max |diff| {err:.2e}", flush=True)
        t_lib = timeit(lambda: ops.bias_act_(F.conv2d(x, w), b, r, True))
        flops = 2.0 * n * cin * cout * hw * hw
        byts = 4.0 * n * hw * hw * (cin + cout * (2 if res else 1))
        print(f"{cin:>4}->{cout:<4} {hw}x{hw} res={int(res)}  fused {t_new:7.1f} us ({flops / t_new / 1e6:6.1f} TF, "
              f"{byts / t_new / 1e3:6.0f} GB/s)   library+bias_act {t_lib:7.1f} us   x{t_lib / t_new:.2f}", flush=True)


if __name__ == "__main__":
    main()
