"""Which kernel path for 17 <= M <= 128 (and the M = 16 / 17 seam)?  Microseconds per call of a graph-replayed chain over rotating
weights (tools/sweep.py::chain_us) for the stream (register-streaming MFMA 16x16x32, no cross-workgroup reduction), split-K
(LDS ring + in-launch reduction), mid and tiled paths.  usage: python tools/path_compare_mid.py [--shapes 7b|13b|all]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import eetq_amd.ops as ops  # noqa: E402
from sweep import chain_us  # noqa: E402

dev = "cuda:0"
which = sys.argv[sys.argv.index("--shapes") + 1] if "--shapes" in sys.argv else "7b"
shapes = {"7b": [(4096, 4096), (4096, 11008), (11008, 4096)], "13b": [(5120, 5120), (5120, 13824), (13824, 5120)]}
shapes["all"] = shapes["7b"] + shapes["13b"]
for K, N in shapes[which]:
    L = max(4, int(640e6 // (K * N)))
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    for M in (8, 16, 17, 24, 32, 48, 64, 96, 112, 128):
        x = torch.randn(M, K, dtype=torch.float16, device=dev)
        row = {"K": K, "N": N, "M": M}
        for path in ("auto", "stream", "splitk", "mid", "mfma", "tilesplit"):
            if path == "stream" and M > 64:
                continue
            if path in ("mid", "mfma") and M < 32:
                continue
            if path == "tilesplit" and M < 96:
                continue

            def step(i, path=path):
                ops.w8_a16_gemm(x, ws[i % L], s, path=path)
            try:
                row[path] = round(chain_us(step, 2 * L, min_seconds=0.02), 2)
            except RuntimeError as e:
                row[path] = str(e)[:40]
        print(json.dumps(row), flush=True)
    del ws
    torch.cuda.empty_cache()
