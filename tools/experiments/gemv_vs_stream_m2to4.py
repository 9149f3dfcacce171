"""A/B: the wave-reduction GEMV kernel (rows 1..4 in one pass, kGemvMaxM) against the MFMA stream kernel at M = 2, 3, 4 -- the
reference's batched-GEMV range (weightOnlyBatchedGemv/kernelLauncher.cu:165-192).  AUTO has sent M >= 2 to the stream kernel since
round 1; the GEMV grew 8-wave bodies for many-row shapes in round 4, so the seam is measured again.  One JSON line per (shape, M):
us per launch in a graph-replayed chain over rotating weights."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import eetq_amd.ops as ops
from sweep import chain_us
dev = "cuda:0"
for K, N in [(4096, 4096), (4096, 11008), (11008, 4096), (4096, 12288), (4096, 22016), (5120, 5120), (5120, 13824), (13824, 5120),
             (5120, 15360), (8192, 8192)]:
    L = max(4, int(640e6 // (K * N)))
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    for M in (1, 2, 3, 4):
        x = torch.randn(M, K, dtype=torch.float16, device=dev)
        row = {"K": K, "N": N, "M": M}
        for path in ("gemv", "stream"):
            def step(i, path=path):
                ops.w8_a16_gemm(x, ws[i % L], s, path=path)
            row[path] = round(chain_us(step, 2 * L, min_seconds=0.02), 2)
        print(json.dumps(row), flush=True)
    del ws
