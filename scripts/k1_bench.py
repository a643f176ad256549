"""K1 (heal_voxelize_batch) of scene 5's three LiDAR agents as a captured graph runs it: device time per call.  A/B of the dense
cell map (default on pillar grids) against the hash grid: HEAL_VOX_DENSE=0.   python scripts/k1_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops
from heal_amd.pipeline import Scene
torch.cuda.set_stream(torch.cuda.Stream())
scene = Scene(3, seed=4, device="cuda:0", modalities=["m1", "m1", "m1"])
pts = [scene.points[k] for k in sorted(scene.points)]
rng = [-102.4, -51.2, -3, 102.4, 51.2, 1]
fn = lambda: ops.voxelize_collated(pts, rng, [0.4, 0.4, 4], 32, 32000)
out = fn()
torch.cuda.synchronize()
print("points", [int(p.shape[0]) for p in pts], "voxels", out[3].tolist(), "checksum", float(out[0][: int(out[3][-1])].double().sum()),
      "per call us", round(ops.graph_period_ms(fn) * 1e3, 1))
