import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import eetq_amd.ops as ops
from sweep import chain_us
dev = "cuda:0"
"""The seam between the small-batch stream kernel and the split-K tile: us per call of both forced paths and AUTO, M = 8 ... 16
(round 4) -- round 5 (SEAM_R05=1): more shapes incl. the held-out ones, M = 10 ... 16, after the split-K tile's K loop got leaner."""
SHAPES = [(4096, 11008), (4096, 12288), (4096, 22016), (11008, 4096), (5120, 13824), (5120, 15360), (5120, 27648), (13824, 5120), (8192, 8192), (8192, 28672), (28672, 8192), (7168, 7168)]
MS = (8, 9, 10, 11, 12, 13, 14, 16)
if os.environ.get("SEAM_R05"):
    SHAPES += [(4096, 4096), (4096, 6144), (5120, 5120), (14336, 4096), (8192, 10240), (4096, 28672), (3584, 18944), (8192, 1024)]
    MS = (10, 12, 13, 14, 15, 16)
for K, N in SHAPES:
    L = max(4, int(640e6 // (K * N)))
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    for M in MS:
        x = torch.randn(M, K, dtype=torch.float16, device=dev)
        row = {"K": K, "N": N, "M": M}
        for path in ("auto", "stream", "splitk"):
            def step(i, path=path):
                ops.w8_a16_gemm(x, ws[i % L], s, path=path)
            row[path] = round(chain_us(step, 2 * L, min_seconds=0.02), 2)
        print(json.dumps(row), flush=True)
    del ws
