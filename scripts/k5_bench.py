"""K5 timing at scene5 size (5 agents, three pyramid levels): heal_warp_fuse per level (round 3: direct gathers, three launches) vs
heal_warp_fuse_levels (round 4: one launch, LDS-staged footprints).  HIP events around 10 back-to-back calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
import torch
from heal_amd import ops, synth
from oracle import oracle_np as O


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
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return float(np.median(ts))


res = {}
for n in (5, 2, 8):
    dims = [(64, 256, 256), (128, 128, 128), (256, 64, 64)]
    poses = synth.agent_poses(4, n)
    rows = O.normalize_pairwise_tfm(synth.pairwise_t_matrix(poses, 8)[None], 204.8, 204.8, 1)[0][0, :n]
    feats = [torch.randn((n, C, H, W), device="cuda") for C, H, W in dims]
    occs = [torch.randn((n, 1, H, W), device="cuda") for C, H, W in dims]
    byts = sum(4.0 * H * W * (n * (C + 1) + C) for C, H, W in dims)
    t_old = timeit(lambda: [ops.warp_fuse(f, o, rows) for f, o in zip(feats, occs)])
    t_new = timeit(lambda: ops.warp_fuse_levels(feats, occs, rows))
    per = [timeit(lambda: ops.warp_fuse(feats[l], occs[l], rows)) for l in range(3)]
    res[f"{n} agents"] = {"per_level_three_launches_us": round(t_old, 1), "per_level_us": [round(v, 1) for v in per],
                          "levels_one_launch_us": round(t_new, 1), "alg_MB": round(byts / 1e6, 1),
                          "frac_hbm_old": round(byts / t_old / 1e6 / 8.0, 3), "frac_hbm_new": round(byts / t_new / 1e6 / 8.0, 3)}
    print(n, res[f"{n} agents"], flush=True)
if "--json" in sys.argv:
    json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
