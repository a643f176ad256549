"""Window attention under autograd at the V2X-ViT shapes of BASELINE config 5 (8 agents x 128 x 128 tokens, 256 channels):
K6b forward + heal_window_attention_backward (ops.WindowAttention) against the library composition the modules use by default
(window re-layout, baddbmm + softmax + bmm, re-layout; mswin.py:64-78).  Forward + backward per call, HIP events, median; decides
whether HEAL_WATTN_GRAD=kernel becomes the default.    python scripts/wattn_bwd_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heal_amd import ops
from scripts.wino_ab import timed

L, H, W = 8, 128, 128
for ws, d, m in ((4, 16, 16), (8, 32, 8), (16, 64, 4)):
    C, T, scale = m * d, ws * ws, d ** -0.5
    torch.manual_seed(ws)
    qkv = (torch.randn(L, H, W, 3 * C, device="cuda") * 0.5).requires_grad_(True)
    bias = torch.randn(T, T, device="cuda", requires_grad=True)
    wgt = torch.randn(L, H, W, C, device="cuda")
    nh, nw = H // ws, W // ws

    def lib():
        t = qkv.view(L, nh, ws, nw, ws, 3, m, d).permute(5, 0, 6, 1, 3, 2, 4, 7).reshape(3, L * m * nh * nw, T, d)
        dots = torch.baddbmm(bias.unsqueeze(0).expand(t.shape[1], -1, -1), t[0], t[1].transpose(1, 2), beta=1.0, alpha=scale)
        o = torch.bmm(dots.softmax(-1), t[2]).view(L, m, nh, nw, ws, ws, d).permute(0, 2, 4, 3, 5, 1, 6).reshape(L, H, W, C)
        qkv.grad = bias.grad = None
        (o * wgt).sum().backward()
        return o

    def mine():
        o = ops.WindowAttention.apply(qkv, bias, m, d, ws, scale)
        qkv.grad = bias.grad = None
        (o * wgt).sum().backward()
        return o

    lib(); g_ref, b_ref = qkv.grad.clone(), bias.grad.clone()
    mine(); g_got, b_got = qkv.grad.clone(), bias.grad.clone()
    err = (float((g_got - g_ref).abs().max() / g_ref.abs().max()), float((b_got - b_ref).abs().max() / b_ref.abs().max()))
    t_lib, t_mine = timed(lib, iters=6, warm=2), timed(mine, iters=6, warm=2)
    with torch.no_grad():
        t_fwd = timed(lambda: ops.window_attention(qkv.detach(), bias.detach(), m, d, ws, scale), iters=6, warm=2)
    print(f"ws {ws:2d}: library fwd+bwd {t_lib:8.1f} us | kernels fwd+bwd {t_mine:8.1f} us (forward alone {t_fwd:7.1f}) | "
          f"grad_qkv / grad_bias rel diff {err[0]:.2e} / {err[1]:.2e}", flush=True)
