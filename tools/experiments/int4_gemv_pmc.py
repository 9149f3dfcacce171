"""Launches of the W4A16 and W8A16 M = 1 GEMV on one 13B shape for a rocprofv3 --pmc pass (tools/experiments/int4_gemv_pmc.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from eetq_amd import ops
dev = "cuda:0"
K, N = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (5120, 13824)))
g = torch.Generator(device=dev); g.manual_seed(1)
w8 = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(4)]
w4 = [torch.randint(-128, 127, (K, N // 2), dtype=torch.int8, device=dev) for _ in range(4)]
s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
x = torch.randn(1, K, dtype=torch.float16, device=dev)
for i in range(20):
    ops.w8_a16_gemm(x, w8[i % 4], s)
    ops.w8_a16_gemm(x, w4[i % 4], s)
torch.cuda.synchronize()
