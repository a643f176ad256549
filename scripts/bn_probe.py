import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import ops
g = torch.Generator().manual_seed(0)
for C, HW in ((64, 256), (128, 128), (256, 64)):
    x = torch.randn((5, C, HW, HW), generator=g).cuda()
    w1 = torch.randn((2*C, C), generator=g).cuda()*0.1; b1 = torch.zeros(2*C).cuda()
    w2 = torch.randn((2*C, 2*C//32, 3, 3), generator=g).cuda()*0.1; b2 = torch.zeros(2*C).cuda()
    w3 = torch.randn((C, 2*C), generator=g).cuda()*0.1; b3 = torch.zeros(C).cuda()
    w1f, w3f = ops.mfma_a_fragments(w1), ops.mfma_a_fragments(w3)
    for _ in range(3): ops.resnext_bottleneck(x, w1f, b1, w2, b2, w3f, b3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.resnext_bottleneck(x, w1f, b1, w2, b2, w3f, b3)
    e1.record(); torch.cuda.synchronize()
    print("skip", os.environ.get("HEAL_BN_SKIP", "0"), "C", C, round(e0.elapsed_time(e1)/10*1e3), "us", flush=True)
