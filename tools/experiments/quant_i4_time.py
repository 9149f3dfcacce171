import os, sys
sys.path.insert(0, os.getcwd())
import torch
from eetq_amd import ops
dev="cuda:0"
K=N=4096
ws=[((torch.rand(K,N,device=dev)*2-1)/K**0.5).half() for _ in range(10)]
for dt,name in ((torch.quint4x2,"int4"),(torch.int8,"int8")):
    for _ in range(3): ops.quant_weights(ws[0], dt, False)
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(50): ops.quant_weights(ws[i%10], dt, False)
    b.record(); torch.cuda.synchronize()
    print(name, "us/call", round(a.elapsed_time(b)*1e3/50,1))
