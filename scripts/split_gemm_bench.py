"""Round 6, VERDICT r5 item 2: heal_conv1x1_split (fp32 in / out / accumulate on the bf16 matrix cores, 6 or 9 partial products) against the
exact-fp32 MFMA kernel heal_conv1x1 at the scene's shapes: kernel-own durations, useful fp32-equivalent TFLOP/s, EXECUTED bf16 TFLOP/s
against the 2.5 PFLOP/s dense bf16 roof, and the maximum error of each against an fp64 reference.  -> JSON on stdout."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops

SHAPES = [(5, 512, 256, 64, 64), (5, 256, 512, 64, 64), (5, 128, 256, 128, 128), (5, 256, 128, 128, 128), (5, 64, 128, 256, 256),
          (5, 128, 512, 64, 64), (1, 256, 2048, 64, 64), (4, 512, 128, 48, 64)]
BF16_PEAK, F32_PEAK = 2500.0, 157.3


def own_us(fn, name, reps=20):
    for _ in range(3):
        fn()
    ops.TIMING = {}
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    t = ops.timing_summary()
    ops.TIMING = None
    return t[name][1] * 1e3


rows = []
st = torch.cuda.Stream()
with torch.cuda.stream(st), torch.no_grad():
    for n, cin, cout, H, W in SHAPES:
        g = torch.Generator().manual_seed(cin + cout)
        x = torch.randn((n, cin, H, W), generator=g).cuda()
        w = (torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5).cuda()
        b = torch.randn((cout,), generator=g).cuda()
        ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), b.double()).relu()
        fl = 2.0 * n * cin * cout * H * W
        row = {"shape": {"n": n, "cin": cin, "cout": cout, "H": H, "W": W}, "gflop": round(fl / 1e9, 2)}
        for tag, env in (("f32_mfma", ""), ("bf16x6", "bf16x6"), ("bf16x9", "bf16x9")):
            os.environ["HEAL_ARITH"] = env
            y = ops.conv1x1(x, w, b, None, 1)
            err = float((y[:1].double() - ref).abs().max() / ref.abs().max())
            us = own_us(lambda: ops.conv1x1(x, w, b, None, 1), f"conv1x1_{cin}_{cout}")
            nprod = {"f32_mfma": 0, "bf16x6": 6, "bf16x9": 9}[tag]
            row[tag] = {"us": round(us, 2), "fp32_equiv_tflops": round(fl / us / 1e6, 1), "max_rel_err_vs_fp64": err}
            if nprod:
                row[tag]["executed_bf16_tflops"] = round(fl * nprod / us / 1e6, 1)
                row[tag]["frac_of_bf16_peak"] = round(fl * nprod / us / 1e6 / BF16_PEAK, 4)
            else:
                row[tag]["frac_of_fp32_mfma_peak"] = round(fl / us / 1e6 / F32_PEAK, 4)
        rows.append(row)
os.environ["HEAL_ARITH"] = ""
print(json.dumps({"what": "heal_conv1x1_split vs heal_conv1x1 (kernel-own begin / end stamps, 20 launches each)", "rows": rows}))
