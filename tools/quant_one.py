"""One shape of quant_weights in an eager loop: the thing to put under rocprofv3 --kernel-trace --stats for the per-kernel split.
usage: python tools/quant_one.py [K N]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eetq_amd import ops
dev="cuda:0"
K, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
ws=[((torch.rand(K,N,device=dev)*2-1)/K**0.5).half() for _ in range(10)]
for _ in range(3): ops.quant_weights(ws[0], torch.int8, False)
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for i in range(50): ops.quant_weights(ws[i%10], torch.int8, False)
b.record(); torch.cuda.synchronize()
print("us/call", a.elapsed_time(b)*1e3/50)
if os.environ.get("QUANT_ONE_COPY"):
    outs = [torch.empty(K, N, dtype=torch.int8, device=dev) for _ in range(10)]
    for i in range(3): outs[0].copy_(ws[0])
    torch.cuda.synchronize()
    a.record()
    for i in range(50): outs[i % 10].copy_(ws[i % 10])
    b.record(); torch.cuda.synchronize()
    print("torch f16->int8 copy_ us/call", a.elapsed_time(b) * 1e3 / 50)
