"""heal_linear against the library sequence it replaces (LayerNorm + addmm + GELU + add) at the V2X-ViT shapes of BASELINE
config 5 (8 agents x 128 x 128 tokens, 256 channels)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from heal_amd import ops
from scripts.k3_bench import timed
dev = torch.device("cuda:0")
T, K = 8 * 128 * 128, 256
torch.manual_seed(0)
x = torch.randn(T, K, device=dev)
res = torch.randn(T, 256, device=dev)
out = {}
for N, name in ((256, "256->256"), (768, "256->768"), (2304, "256->2304")):
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) * 0.1
    g, be = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    wf, bf = (w * g[None, :]).contiguous(), (b + w @ be).contiguous()
    ref, t_lib = timed(lambda: torch.addmm(b, F.layer_norm(x, (K,), g, be, 1e-5), w.t()), 10)
    _, t_gemm = timed(lambda: torch.addmm(b, x, w.t()), 10)
    got, t_mine = timed(lambda: ops.linear(x, wf, bf, stats=ops.ln_stats(x, 1e-5)), 10)
    _, t_mine_gemm = timed(lambda: ops.linear(x, w, b), 10)
    err = float((got - ref).abs().max() / ref.abs().max())
    fl = 2.0 * T * K * N
    out[name] = {"lib_ln+gemm_us": round(t_lib, 1), "lib_gemm_us": round(t_gemm, 1), "heal_ln+linear_us": round(t_mine, 1),
                 "heal_linear_us": round(t_mine_gemm, 1), "heal_TFLOPs": round(fl / t_mine_gemm * 1e-6, 1),
                 "lib_TFLOPs": round(fl / t_gemm * 1e-6, 1), "rel_err": err}
    print(name, out[name], flush=True)
    assert err < 1e-4
# FFN tail: GELU epilogue and residual
w1 = torch.randn(256, 256, device=dev) * 0.05; b1 = torch.randn(256, device=dev) * 0.1
ref = F.gelu(torch.addmm(b1, x, w1.t())) + res
got = ops.linear(x, w1, b1, act="gelu", residual=res)
print("gelu+res err", float((got - ref).abs().max() / ref.abs().max()))
# row map + parts
L, HW = 8, 128 * 128
w3 = torch.randn(768, 256, device=dev) * 0.05
ref = (x @ w3.t()).view(L, HW, 3, 256).permute(2, 1, 0, 3).contiguous()
got = ops.linear(x, w3, row_map=(HW, L), parts=3).view(3, HW, L, 256)
print("rowmap/parts err", float((got - ref).abs().max() / ref.abs().max()))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
