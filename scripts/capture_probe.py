"""Try to capture a workload's step in a HIP graph and print the Python stack of the first op that is not capturable."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import configs
from heal_amd.pipeline import Scene, ScenePipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
small = len(sys.argv) > 2
rng = [-25.6, -25.6, -3, 25.6, 25.6, 1] if small else configs.FULL_RANGE
hypes = configs.lidar_baseline("v2xvit", rng, max_cav=max(5, n), modality="m3")
dev = torch.device("cuda:0")
ws = torch.cuda.Stream(); torch.cuda.set_stream(ws)
pipe = ScenePipeline(hypes, dev, seed=1)
scene = Scene(n, seed=4, device=dev, modalities=["m3"] * n)
for _ in range(2):
    pipe.step(scene)
torch.cuda.synchronize()
try:
    pipe.capture(scene)
    print("capture ok")
    r = pipe.replay(); torch.cuda.synchronize(); print("replay ok", None if r[0] is None else r[0].shape)
except Exception:
    tb = traceback.format_exc().splitlines()
    print("\n".join(l for l in tb if "heal_amd" in l or "Error" in l or "    " in l)[-3000:])
