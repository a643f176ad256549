"""heal_split_attn_weights and heal_ln_stats at the V2X-ViT shapes of BASELINE config 5 (8 agents x 128 x 128 tokens, 256
channels): the two small launches between the window attention and the merged projection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops
from scripts.k3_bench import timed
dev = torch.device("cuda:0")
L, HW, C = 8, 128 * 128, 256
torch.manual_seed(0)
br = torch.randn(3, L * HW, C, device=dev)
wo = torch.randn(3, C, C, device=dev) * 0.05; bo = torch.randn(3, C, device=dev) * 0.1
fc1 = torch.randn(C, C, device=dev) * 0.05; fc2 = torch.randn(3 * C, C, device=dev) * 0.05
lg = torch.rand(C, device=dev) + 0.5; lb = torch.randn(C, device=dev) * 0.1
(sc, bi), t = timed(lambda: ops.split_attn_weights(br, L, HW, wo, bo, fc1, lg, lb, 1e-5, fc2), 20)
print("split_attn_weights (colsum + weights) us", round(t, 1), "checksum", float(sc.double().sum()), float(bi.double().sum()))
x = br[0]
st, t = timed(lambda: ops.ln_stats(x, 1e-5), 20)
print("ln_stats us", round(t, 1), "GB/s", round(x.numel() * 4 / t * 1e-3, 1), "checksum", float(st.double().sum()))
