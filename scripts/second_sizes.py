"""Active-site counts per SECOND stage for a synthetic sweep (sync mode): sizing data for the no-sync capacity policy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import configs, ops, synth
R = configs.FULL_RANGE
for seed in (4000, 4001):
    pts = torch.from_numpy(synth.lidar_frame(seed)).cuda()
    v, c, n = ops.voxelize(pts, R, [0.1, 0.1, 0.1], 5, 70000)
    x = ops.SparseTensor.from_unsorted(ops.mean_vfe(v, n), c, [41, 2048, 2048], 1)
    sizes = [x.n]
    for k, s, p in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
                    ((3, 1, 1), (2, 1, 1), (0, 0, 0))):
        oi, osh, _ = x.out_sites(k, s, p)
        x = ops.SparseTensor(torch.zeros((oi.shape[0], 4), device="cuda"), oi, osh, 1)
        sizes.append(x.n)
    print(seed, int(pts.shape[0]), sizes, [round(b / a, 2) for a, b in zip(sizes, sizes[1:])])
