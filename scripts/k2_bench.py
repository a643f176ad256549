"""K2 (heal_pfn_scatter) on 3 collated LiDAR agents, HIP events; run with HEAL_CANVAS_NT=0/1 to A/B the store kind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from heal_amd import configs, ops, synth
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    return float(np.median(ts))
R = configs.FULL_RANGE
pts = [torch.from_numpy(synth.lidar_frame(4000 + k)).cuda() for k in range(3)]
v, c, n, off = ops.voxelize_collated(pts, R, [0.4, 0.4, 4], 32, 70000)
M = int(off[-1].item())
g = torch.Generator().manual_seed(0)
w = torch.randn((64, 10), generator=g).cuda(); sc = (torch.rand((64,), generator=g) + 0.5).cuda(); sh = torch.randn((64,), generator=g).cuda()
out = torch.empty((3, 64, 512, 512), device="cuda")
t = timeit(lambda: ops.pfn_scatter(v, c, n, w, sc, sh, [0.4, 0.4, 4], R, 3, 512, 512, n_voxels_dev=off[3:4], out=out))
byts = 16 * M * 32 + 20 * M + 3 * 4 * 64 * 512 * 512
print(f"NT={os.environ.get('HEAL_CANVAS_NT','0')} M={M}: {t:.1f} us  {byts/t/1e3:.0f} GB/s  {byts/t/1e3/8000:.3f} of peak")
