"""K4 timing: the production path (heal_bev_pool_scatter on the pixel-major head tensor the fused image_head | depth_head convolution
writes, ONE launch, consumed by heal_bev_stem_block or heal_bev_pool_emit) vs the bit-reproducible sorted pipeline on the
reference's NCHW tensors, standalone.  Kernel-attached HIP events (ops._Timed kernel_events: the kernel's own begin / end, what
rocprofv3 reports) next to events recorded around the call and to the per-call period inside a captured graph.
Usage: python scripts/k4_bench.py   (on the GPU box)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops, synth  # noqa: E402
from oracle import oracle_np as O  # noqa: E402  (bench script: geometry helpers only)


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        out = fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts)), out


def timed_graph(fn, reps=20, iters=10):
    """Device time per call with launch gaps as a captured graph sees them: `reps` calls captured once, replayed."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return float(np.median(ts))


def run(final_dim, C=128, n_agents=1):
    rng = np.random.default_rng(0)
    fH, fW = final_dim[0] // 8, final_dim[1] // 8
    D, N = 48, 4
    frustum = torch.from_numpy(O.create_frustum(list(final_dim), 8, [2, 50, 48], "LID")).cuda()
    dx, bx, nx = O.gen_dx_bx([-51.2, 51.2, 0.4], [-51.2, 51.2, 0.4], [-10, 10, 20.0])
    rig = synth.camera_rig(0, N, final_dim[0], final_dim[1])
    cam = {k: torch.from_numpy(np.tile(v[None], (n_agents,) + (1,) * v.ndim).astype(np.float32)).cuda() for k, v in rig.items()}
    mats = ops.camera_matrices(cam["rots"], cam["trans"], cam["intrins"], cam["post_rots"], cam["post_trans"])
    dl = torch.from_numpy(rng.standard_normal((n_agents * N, D, fH, fW)).astype(np.float32)).cuda()
    ft = torch.from_numpy(rng.standard_normal((n_agents * N, C, fH, fW)).astype(np.float32)).cuda()
    head = torch.cat([ft, dl], 1).permute(0, 2, 3, 1).reshape(n_agents * N, fH * fW, C + D).contiguous()
    args = (n_agents, N, dx.tolist(), bx.tolist(), nx.tolist())
    res = {}
    os.environ.pop("HEAL_LSS_PATH", None)
    w1 = torch.randn((64, C, 3, 3), device="cuda") / (9 * C) ** 0.5
    wd = torch.randn((64, C, 1, 1), device="cuda") / C ** 0.5
    b1, bd = torch.randn(64, device="cuda"), torch.randn(64, device="cuda")
    wm, wdf = ops.stem_fragments(w1, wd)
    pm = lambda: ops.bev_pool_pm(head, C, D, fH, fW, frustum, mats, *args)                                     # scatter + dense emit
    prod = lambda: ops.bev_pool_pm(head, C, D, fH, fW, frustum, mats, *args, pooled=True).stem_block(wm, b1, wdf, bd)
    res["scatter_emit_us"], res["scatter_emit_min_us"], a = timed(pm)
    res["scatter_emit_in_graph_us"] = timed_graph(pm)
    res["scatter_stem_us"], _, _ = timed(prod)
    res["scatter_stem_in_graph_us"] = timed_graph(prod)
    ops.TIMING = {}
    for _ in range(30):
        prod()
        pm()
    torch.cuda.synchronize()
    for name, (calls, ms) in ops.timing_summary().items():
        res[f"kernel_events_{name}_us"] = round(ms * 1e3, 2)       # bev_pool = k_lss_scatter alone; bev_stem_block = k_bev_stem alone
    ops.TIMING = None
    os.environ["HEAL_LSS_PATH"] = "sorted"
    res["sorted_us"], _, b = timed(lambda: ops.bev_pool(dl, ft, frustum, mats, *args))
    os.environ.pop("HEAL_LSS_PATH", None)
    # the producing convolution with both epilogues (512 -> C + D at the feature resolution): what the layout change costs
    feat512 = torch.randn((n_agents * N, 512, fH, fW), device="cuda")
    w = torch.randn((C + D, 512, 1, 1), device="cuda") / 512 ** 0.5
    bias = torch.randn((C + D,), device="cuda")
    res["head_conv_nchw_us"], _, _ = timed(lambda: ops.conv1x1(feat512, w, bias, None, 0))
    res["head_conv_pm_us"], _, _ = timed(lambda: ops.conv1x1(feat512, w, bias, None, 0, pixel_major=True))
    res["max_rel_diff_vs_sorted"] = float((a - b).abs().max() / b.abs().max())
    res["nonzero_cells"] = int((a != 0).any(dim=1).sum())
    alg = (dl.numel() + ft.numel() + a.numel()) * 4            # logits + features read, canvas written
    res["alg_MB"] = alg / 1e6
    res["scatter_frac_hbm"] = alg / (res["kernel_events_bev_pool_us"] * 1e-6) / 8e12   # SURVEY 8d bytes / the scatter kernel's duration
    return res


if __name__ == "__main__":
    out = {"m2 384x512": run((384, 512)), "m4 336x448": run((336, 448)), "m2 x2 agents": run((384, 512), n_agents=2)}
    print(json.dumps(out, indent=1))
