"""How many bitmap-word atomics k_sp_bm_mark would save by aggregating inside a wave: for the four strided layers of config 5, the marks per
wave of 64 consecutive (sorted) input sites against the DISTINCT bitmap words they touch.   python scripts/bm_mark_dup.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import configs, ops, synth

dev = torch.device("cuda:0")
vs, cs, ns = [], [], []
for b in range(8):
    pts = torch.from_numpy(synth.lidar_frame(4000 + b)).to(dev)
    v, c, n = ops.voxelize(pts, configs.FULL_RANGE, [0.1, 0.1, 0.1], 5, 70000, batch_idx=b)
    vs.append(v); cs.append(c); ns.append(n)
x = ops.SparseTensor.from_unsorted(ops.mean_vfe(torch.cat(vs), torch.cat(ns)), torch.cat(cs), [41, 2048, 2048], 8)
LAYERS = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))]
idx, shape = x.indices.long(), list(x.spatial_shape)
for k, s, p in LAYERS:
    oshape = [(shape[d] + 2 * p[d] - k[d]) // s[d] + 1 for d in range(3)]
    keys, site = [], []
    ar = torch.arange(idx.shape[0], device=dev)
    for dz in range(2):
        for dy in range(2):
            for dx in range(2):
                o, ok = [], torch.ones(idx.shape[0], dtype=torch.bool, device=dev)
                for d, dd in zip(range(3), (dz, dy, dx)):
                    a = idx[:, 1 + d] + p[d]
                    num = a - k[d] + 1
                    lo = torch.where(num <= 0, torch.zeros_like(num), (num + s[d] - 1) // s[d])
                    hi = torch.minimum(a // s[d], torch.full_like(a, oshape[d] - 1))
                    od = lo + dd
                    ok &= od <= hi
                    o.append(od)
                key = ((idx[:, 0] * oshape[0] + o[0]) * oshape[1] + o[1]) * oshape[2] + o[2]
                keys.append(key[ok]); site.append(ar[ok])
    key, site = torch.cat(keys), torch.cat(site)
    marks = int(key.numel())
    cells = int(torch.unique(key).numel())
    ww = torch.unique(torch.stack([site // 64, key >> 5], 1), dim=0).shape[0]
    print(f"k={k} s={s}: sites {idx.shape[0]}, marks {marks} ({marks / idx.shape[0]:.2f}/site), output cells {cells}, "
          f"distinct (wave, word) pairs {ww} = {marks / ww:.2f}x fewer atomics; waves {(idx.shape[0] + 63) // 64}, words per wave {ww / ((idx.shape[0] + 63) // 64):.1f}")
    oi, oshape2, _ = x.out_sites(k, s, p)
    x = ops.SparseTensor(torch.zeros((oi.shape[0], 4), device=dev), oi, oshape2, 8)
    idx, shape = x.indices.long(), list(oshape2)
