"""Probe: where does hipGraph capture / replay of the step stall?  (prints progress, flushes)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from heal_amd import configs, ops
from heal_amd.pipeline import Scene, ScenePipeline
def p(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)
full = len(sys.argv) > 1 and sys.argv[1] == "full"
hy = configs.lidar_pyramid() if full else configs.lidar_pyramid([-25.6, -25.6, -3, 25.6, 25.6, 1])
pipe = ScenePipeline(hy, "cuda:0", seed=0)
scene = Scene(5 if full else 2, seed=4, device="cuda:0")
if not full:
    scene.points = {k: v[(v[:, 0].abs() < 28) & (v[:, 1].abs() < 28)][:9000].contiguous() for k, v in scene.points.items()}
pipe.calibrate_cls_bias(scene, 300)
p("eager step", pipe.step(scene)[0].shape)
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(3):
        out = pipe.model(scene.model_input())
        p("warm", i)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
p("warmup done")
g = torch.cuda.CUDAGraph()
stage = sys.argv[2] if len(sys.argv) > 2 else "model"
with torch.cuda.graph(g):
    if stage == "vox":
        v = ops.voxelize(scene.points[0], hy["model"]["args"]["lidar_range"], [0.4, 0.4, 4], 32, 70000, sync=False)
    elif stage == "enc":
        v = pipe.model.encoder_m1(scene.model_input(), "m1")
    elif stage == "step":
        pass
    else:
        v = pipe.model(scene.model_input())
if stage == "step":
    g = pipe.capture(scene)
    p("pipe.capture ok")
    for i in range(3):
        t = time.time(); r = pipe.replay(); torch.cuda.synchronize(); p("pipe.replay", i, round((time.time() - t) * 1e3, 3), "ms", None if r[0] is None else tuple(r[0].shape))
    sys.exit(0)
p("captured", stage)
for i in range(3):
    t = time.time(); g.replay(); torch.cuda.synchronize(); p("replay", i, round((time.time() - t) * 1e3, 3), "ms")
