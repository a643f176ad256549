"""Shader-clock stamps inside k_lss_scatter (HEAL_K4_DBG=128): cycles since kernel entry at the phase boundaries, for block 0's first
wave (a softmax wave) and last wave (keys only).  s_waitcnt(0) in front of every stamp: the phases are serialised, times are upper bounds."""
import os, sys
os.environ["HEAL_K4_DBG"] = "128"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops, synth
from oracle import oracle_np as O
C, D, N = 128, 48, 4
dx, bx, nx = O.gen_dx_bx([-51.2, 51.2, 0.4], [-51.2, 51.2, 0.4], [-10, 10, 20.0])
final_dim = (384, 512)
fH, fW = final_dim[0] // 8, final_dim[1] // 8
frustum = torch.from_numpy(O.create_frustum(list(final_dim), 8, [2, 50, 48], "LID")).cuda()
rig = synth.camera_rig(0, N, *final_dim)
cam = {k: torch.from_numpy(v[None].astype(np.float32)).cuda() for k, v in rig.items()}
mats = ops.camera_matrices(cam["rots"], cam["trans"], cam["intrins"], cam["post_rots"], cam["post_trans"])
head = torch.randn((N, fH * fW, C + D), device="cuda")
args = (head, C, D, fH, fW, frustum, mats, 1, N, dx.tolist(), bx.tolist(), nx.tolist())
names = ["loads issued", "keys done", "softmax done", "X in LDS", "barrier", "A frags", "atomics issued", "end"]
acc = np.zeros((2, 8))
reps = 20
for i in range(reps + 5):
    ops.bev_pool_pm(*args)
    torch.cuda.synchronize()
    ws = ops._ZWS[(("bev_pool_pm", 1, C, int(nx[0]), int(nx[1]), int(nx[2])), 0, torch.cuda.current_stream().cuda_stream)]
    st = ws.view(torch.int32)[16:48].cpu().numpy().reshape(2, 16)[:, :8]
    if i >= 5:
        acc += st
acc /= reps
for k, n in enumerate(names):
    print(f"{n:<16} wave0 {acc[0, k]:8.0f} cyc   last wave {acc[1, k]:8.0f} cyc")
