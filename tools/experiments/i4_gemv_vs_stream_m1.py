"""W4A16 at M = 1: the dot-product GEMV (gemv_kernel<BITS = 4>: exact dequantisation + v_dot2 chains, VALU-bound) against
the MFMA small-batch kernel with its activation rows in LDS (round 4) run with ONE row.  us per launch, graph-replayed chains over
rotating weights; the two paths give tier-A-equal results, not the same bits (different summation)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import eetq_amd.ops as ops
from sweep import chain_us
dev = "cuda:0"
for K, N in [(4096, 4096), (4096, 11008), (4096, 12288), (4096, 14336), (4096, 22016), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120),
             (5120, 15360), (5120, 27648), (8192, 8192), (8192, 1024), (4096, 1024), (8192, 28672), (28672, 8192)]:
    L = max(4, int(640e6 // (K * N // 2)))
    ws = [torch.randint(-128, 127, (K, N // 2), dtype=torch.int8, device=dev) for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    x = torch.randn(1, K, dtype=torch.float16, device=dev)
    row = {"K": K, "N": N, "M": 1}
    for path in ("gemv", "stream"):
        def step(i, path=path):
            ops.w8_a16_gemm(x, ws[i % L], s, path=path)
        row[path] = round(chain_us(step, 2 * L, min_seconds=0.02), 2)
    a, b = ops.w8_a16_gemm(x, ws[0], s, path="gemv").float(), ops.w8_a16_gemm(x, ws[0], s, path="stream").float()
    row["max_rel_diff"] = float(((a - b).abs().max() / a.abs().max()).item())
    print(json.dumps(row), flush=True)
    del ws
