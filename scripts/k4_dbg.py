"""Timing experiment: which part of K4's two kernels costs what (HEAL_K4_DBG bits skip parts; results are INVALID then).
Run under rocprofv3 --kernel-trace --stats with HEAL_K4_DBG set from outside."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops, synth
from oracle import oracle_np as O
final_dim, C, D, N = (384, 512), 128, 48, 4
fH, fW = final_dim[0] // 8, final_dim[1] // 8
frustum = torch.from_numpy(O.create_frustum(list(final_dim), 8, [2, 50, 48], "LID")).cuda()
dx, bx, nx = O.gen_dx_bx([-51.2, 51.2, 0.4], [-51.2, 51.2, 0.4], [-10, 10, 20.0])
rig = synth.camera_rig(0, N, *final_dim)
cam = {k: torch.from_numpy(v[None].astype(np.float32)).cuda() for k, v in rig.items()}
mats = ops.camera_matrices(cam["rots"], cam["trans"], cam["intrins"], cam["post_rots"], cam["post_trans"])
head = torch.randn((N, fH * fW, C + D), device="cuda")
for i in range(40):
    ops.bev_pool_pm(head, C, D, fH, fW, frustum, mats, 1, N, dx.tolist(), bx.tolist(), nx.tolist())
torch.cuda.synchronize()
