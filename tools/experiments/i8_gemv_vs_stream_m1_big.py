import json, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import eetq_amd.ops as ops
from sweep import chain_us
dev = "cuda:0"
for K, N in [(8192, 28672), (28672, 8192), (8192, 8192), (8192, 1024), (4096, 1024), (8192, 10240), (5120, 27648), (6144, 16384), (16384, 6144), (7168, 7168), (12288, 12288), (4096, 32000), (5120, 32000), (4096, 128256)]:
    L = max(3, int(640e6 // (K * N)))
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    x = torch.randn(1, K, dtype=torch.float16, device=dev)
    row = {"K": K, "N": N, "M": 1}
    for path in ("gemv", "stream"):
        def step(i, path=path):
            ops.w8_a16_gemm(x, ws[i % L], s, path=path)
        row[path] = round(chain_us(step, 2 * L, min_seconds=0.02), 2)
    row["ratio"] = round(row["stream"] / row["gemv"], 3)
    row["gemv_frac"] = round(K * N / row["gemv"] / 8e6, 3)
    print(json.dumps(row), flush=True)
    del ws
