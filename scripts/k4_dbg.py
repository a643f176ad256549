"""Timing experiment: which part of K4's kernels costs what (HEAL_K4_DBG bits skip parts; results are INVALID then).
Run under rocprofv3 --kernel-trace --stats with HEAL_K4_DBG set from outside.  Both camera modalities of BASELINE config 4
(m2 384x512, m4 336x448), production hand-off: k_lss_scatter -> k_bev_stem, and the dense emit (k_lss_canvas) for comparison."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops, synth
from oracle import oracle_np as O
C, D, N = 128, 48, 4
w1 = torch.randn((64, C, 3, 3), device="cuda") / (9 * C) ** 0.5
wd = torch.randn((64, C, 1, 1), device="cuda") / C ** 0.5
b1, bd = torch.randn(64, device="cuda"), torch.randn(64, device="cuda")
wm, wdf = ops.stem_fragments(w1, wd)
dx, bx, nx = O.gen_dx_bx([-51.2, 51.2, 0.4], [-51.2, 51.2, 0.4], [-10, 10, 20.0])
for final_dim in ((384, 512), (336, 448)):
    fH, fW = final_dim[0] // 8, final_dim[1] // 8
    frustum = torch.from_numpy(O.create_frustum(list(final_dim), 8, [2, 50, 48], "LID")).cuda()
    rig = synth.camera_rig(0, N, *final_dim)
    cam = {k: torch.from_numpy(v[None].astype(np.float32)).cuda() for k, v in rig.items()}
    mats = ops.camera_matrices(cam["rots"], cam["trans"], cam["intrins"], cam["post_rots"], cam["post_trans"])
    head = torch.randn((N, fH * fW, C + D), device="cuda")
    args = (head, C, D, fH, fW, frustum, mats, 1, N, dx.tolist(), bx.tolist(), nx.tolist())
    for i in range(20):
        ops.bev_pool_pm(*args, pooled=True).stem_block(wm, b1, wdf, bd)
    for i in range(20):
        ops.bev_pool_pm(*args)
torch.cuda.synchronize()
