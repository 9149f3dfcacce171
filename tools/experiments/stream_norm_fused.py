"""Round 6: RMS-norm + projection at 2 <= M <= 4 -- one launch (eetq_w8a16_gemm_rmsnorm: the small-batch kernel's block-copy form with the
norm applied to its LDS copy) against the two launches (layernorm_forward, then the projection on the rule's plan), graph-replayed chains
over rotating weight sets (> 256 MB), us per norm + projection."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import eetq_amd.ops as ops
from sweep import chain_us
SHAPES = [(5120, 15360), (5120, 27648), (4096, 12288), (4096, 22016), (4096, 4096), (8192, 10240)]
for K, N in SHAPES:
    L = max(4, int(640e6 // (K * N)))
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0") for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device="cuda:0") * 0.01
    gamma = torch.rand(K, dtype=torch.float16, device="cuda:0") + 0.5
    for M in (2, 3, 4):
        x = torch.randn(M, K, dtype=torch.float16, device="cuda:0")
        xn = torch.empty_like(x)

        def two(i):
            ops.layernorm_forward(x, gamma, xn, 1e-5)
            return ops.w8_a16_gemm(xn, ws[i % L], s)
        t2 = chain_us(two, 2 * L, min_seconds=0.02)
        t1 = chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], s, norm=(gamma, 1e-5)), 2 * L, min_seconds=0.02)
        t0 = chain_us(lambda i: ops.w8_a16_gemm(xn, ws[i % L], s), 2 * L, min_seconds=0.02)
        print(json.dumps({"K": K, "N": N, "M": M, "projection_us": round(t0, 2), "norm_then_projection_us": round(t2, 2), "one_launch_us": round(t1, 2)}), flush=True)
    del ws
