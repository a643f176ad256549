"""Ablation timings of k_sp_conv2 on the real 64->64 layer of the 8-agent SECOND encoder (HEAL_SP_DBG bit mask: 1 no accumulator
read-modify-write, 2 no gather DMA, 4 no MFMA, 8 no weight loads, 16 setup + epilogue only): what each part costs in place."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from heal_amd import configs, ops, synth
from scripts.k3_bench import timed
dev = torch.device("cuda:0")
vs, cs, ns = [], [], []
for b in range(8):
    pts = torch.from_numpy(synth.lidar_frame(4000 + b)).to(dev)
    v, c, n = ops.voxelize(pts, configs.FULL_RANGE, [0.1, 0.1, 0.1], 5, 70000, batch_idx=b)
    vs.append(v); cs.append(c); ns.append(n)
feats = ops.mean_vfe(torch.cat(vs), torch.cat(ns))
x = ops.SparseTensor.from_unsorted(feats, torch.cat(cs), [41, 2048, 2048], 8)
k = (3, 3, 3)
def run(x, cin, cout, tag):
    nbr = x.neighbors(x.indices, x.spatial_shape, k, (1, 1, 1), (1, 1, 1))
    w = torch.randn((27, cin, cout), device=dev) * 0.05
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    xx = ops.SparseTensor(torch.randn((x.n, cin), device=dev), x.indices, x.spatial_shape, 8)
    for dbg in sys.argv[1].split(","):
        os.environ["HEAL_SP_DBG"] = dbg
        _, us = timed(lambda: xx.conv(nbr, w, sc, sh), 10)
        print(tag, "n", x.n, "dbg", dbg, round(us, 1), "us", flush=True)
run(x, 4, 16, "4->16")
for _ in range(2):
    oi, osh, _ = x.out_sites(k, (2, 2, 2), (1, 1, 1))
    x = ops.SparseTensor(torch.zeros((oi.shape[0], 4), device=dev), oi, osh, 8)
run(x, 64, 64, "64->64")
