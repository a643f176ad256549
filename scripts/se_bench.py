import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from heal_amd import ops
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    return float(np.median(ts))
for n,C,S in ((4,1152,48),(4,672,28),(4,480,20),(4,240,10),(4,96,4),(4,32,8)):
    m=torch.randn((n,C),device="cuda"); w1=torch.randn((S,C,1,1),device="cuda"); b1=torch.randn((S,),device="cuda")
    w2=torch.randn((C,S,1,1),device="cuda"); b2=torch.randn((C,),device="cuda")
    print(n,C,S, "%.1f us" % timeit(lambda: ops.se_gate(m,w1,b1,w2,b2)), flush=True)
