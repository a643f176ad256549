import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from heal_amd import ops
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    return float(np.median(ts))
for C,hw in ((128,256),(256,128),(512,64)):
    x=torch.randn((5,C,hw,hw),device="cuda"); w=torch.randn((C,C//32,3,3),device="cuda")*0.1; b=torch.randn((C,),device="cuda")
    t=timeit(lambda: ops.grouped_conv3x3(x,w,b,32,1,True))
    ref=torch.relu(torch.nn.functional.conv2d(x[:1],w,b,1,1,1,32)); got=ops.grouped_conv3x3(x[:1].contiguous(),w,b,32,1,True)
    err=float((got-ref).abs().max()/ref.abs().max())
    print(f"C={C} {hw}x{hw}: {t:7.1f} us  {2*4*x.numel()/t/1e3:6.0f} GB/s  relerr {err:.1e}", flush=True)
