import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from heal_amd import ops
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    return float(np.median(ts))
L,H,W=8,128,128
for ws,m,d in ((4,16,16),(8,8,32),(16,4,64)):
    qkv=torch.randn((L,H,W,3*m*d),device="cuda"); T=ws*ws; bias=torch.randn((T,T),device="cuda"); sc=d**-0.5
    nh,nw=H//ws,W//ws
    def lib():
        q=qkv.view(L,nh,ws,nw,ws,3,m,d).permute(5,0,6,1,3,2,4,7).reshape(3,L*m*nh*nw,T,d)
        dots=torch.baddbmm(bias.unsqueeze(0).expand(q.shape[1],-1,-1),q[0],q[1].transpose(1,2),beta=1.0,alpha=sc)
        o=torch.bmm(dots.softmax(-1),q[2])
        return o.view(L,m,nh,nw,ws,ws,d).permute(0,2,4,3,5,1,6).reshape(L,H,W,m*d)
    t1=timeit(lambda: ops.window_attention(qkv,bias,m,d,ws,sc)); t0=timeit(lib)
    byts=4*L*H*W*m*d*4
    print(f"ws={ws} heads={m} d={d}: fused {t1:8.1f} us ({byts/t1/1e3:6.0f} GB/s)   library {t0:8.1f} us   x{t0/t1:.2f}", flush=True)
