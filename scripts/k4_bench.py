"""K4 A/B: direct splat on the matrix cores (default) vs the vector-ALU walk (HEAL_LSS_PATH=walk) vs the sorted
pipeline (HEAL_LSS_PATH=sorted), standalone launches, HIP events.
Usage: python scripts/k4_bench.py   (on the GPU box)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops, synth  # noqa: E402
from oracle import oracle_np as O  # noqa: E402  (bench script: geometry helpers only)


def run(final_dim, C=128, n_agents=1, iters=30):
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
    res = {}
    outs = {}
    for path in ("sorted", "walk", "splat"):
        os.environ["HEAL_LSS_PATH"] = path
        for _ in range(5):
            out = ops.bev_pool(dl, ft, frustum, mats, n_agents, N, dx.tolist(), bx.tolist(), nx.tolist())
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = ops.bev_pool(dl, ft, frustum, mats, n_agents, N, dx.tolist(), bx.tolist(), nx.tolist())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[path] = float(np.median(ts))
        outs[path] = out
    a, b = outs["sorted"], outs["splat"]
    res["max_rel_diff"] = float((a - b).abs().max() / a.abs().max())
    res["max_rel_diff_walk"] = float((a - outs["walk"]).abs().max() / a.abs().max())
    res["nonzero_cells"] = int((a != 0).any(dim=1).sum())
    alg = (dl.numel() + ft.numel() + a.numel()) * 4
    res["alg_MB"] = alg / 1e6
    res["splat_frac_hbm"] = alg / (res["splat"] * 1e-6) / 8e12
    return res


if __name__ == "__main__":
    out = {"m2 384x512": run((384, 512)), "m4 336x448": run((336, 448)), "m2 x2 agents": run((384, 512), n_agents=2)}
    print(json.dumps(out, indent=1))
