"""A/B (round 6): the normalised projections of a batched decode step (2 <= M <= 4 rows) as ONE launch -- rows through the dot-product
kernel with the activations read from LDS by row broadcast, RMS-norm of every row in the prologue (eetq_w8a16_gemv_rows) -- against
what AUTO ran before: a norm launch + the MFMA small-batch kernel.  Run once per setting (the hook is read once per process):
    EETQ_AMD_TUNING=1 EETQ_AMD_GEMV_ROWS=1 python tools/experiments/gemv_rows_ab.py     # one launch wherever supported
    EETQ_AMD_TUNING=1 EETQ_AMD_GEMV_ROWS=0 python tools/experiments/gemv_rows_ab.py     # two launches
One JSON line per (shape, M, form): us per call in a graph-replayed chain over rotating weights, tier A of the one-launch result
against norm launch + projection computed in the same process through explicit calls."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import eetq_amd.ops as ops
from sweep import chain_us
dev = "cuda:0"
mode = os.environ.get("EETQ_AMD_GEMV_ROWS", "auto")
shapes = [(5120, 15360, False), (5120, 27648, True), (4096, 12288, False), (4096, 22016, True), (4096, 6144, False), (8192, 10240, False),
          (8192, 57344, True), (2048, 8192, False)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if "%dx%d" % (s[0], s[1]) in sys.argv[1:]]
for K, N, glu in shapes:
    L = max(4, int(640e6 // (K * N)))
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01 + 0.001
    gam = (torch.rand(K, dtype=torch.float16, device=dev) + 0.5)
    for M in (1, 2, 3, 4):
        x = torch.randn(M, 1, K, dtype=torch.float16, device=dev)
        act = "silu_glu8" if glu else ""
        def step(i):
            return ops.w8_a16_gemm(x, ws[i % L], s, norm=(gam, 1e-5), activation=act)
        y = step(0).float()
        normed = torch.empty_like(x)
        ops.layernorm_forward(x, gam, normed, 1e-5)
        ref = ops.w8_a16_gemm(normed, ws[0], s, activation=act).float()
        tier_a = bool(((y - ref).abs() <= 1e-3 * ref.abs().max() + 2e-3 * ref.abs()).all())
        us = chain_us(step, 2 * L, min_seconds=0.03)
        print(json.dumps({"K": K, "N": N, "glu8": glu, "M": M, "rows_hook": mode, "us": round(us, 2), "tier_a_vs_two_launches": tier_a,
                          "max_abs": round((y - ref).abs().max().item(), 5)}), flush=True)
    del ws
