"""Correctness + time of the split-K form of the tiled kernel (path "tilesplit") against the unsplit tiled kernel, the medium-batch
split-K kernel and a torch fp32 reference.  usage: python tools/experiments/tilesplit_check.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from eetq_amd import ops
from sweep import chain_us
dev = "cuda:0"
bad = 0
for (M, K, N) in [(512, 4096, 4096), (512, 11008, 4096), (384, 4096, 4096), (300, 4096, 4096), (512, 5120, 5120), (448, 8192, 4096),
                  (512, 2048, 4096), (1024, 4096, 2048), (128, 11008, 4096), (256, 11008, 4096), (512, 4096, 11008)]:
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(M + K + N)
    sets = []
    for i in range(nbuf):
        w = ((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half()
        if i == 0:
            raw, qw, sc = ops.quant_weights(w, torch.int8, True)
            wdq = (raw.float() * sc.float()[None, :]).half().float()
            sets.append((qw, sc))
        else:
            sets.append(tuple(ops.quant_weights(w, torch.int8, False)))
        del w
    x = (torch.rand(M, K, device=dev, generator=g) - 0.25).half()
    bias = torch.rand(N, device=dev, generator=g).half()
    ref = x.float() @ wdq
    tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
    y1 = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="tilesplit")
    y2 = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="tilesplit")
    y0 = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="mfma")
    yb = ops.w8_a16_bias(x, sets[0][0], sets[0][1], bias, path="tilesplit") if hasattr(ops, "w8_a16_bias") else None
    ok = bool(((y1.float() - ref).abs() <= tol).all()) and torch.equal(y1, y2)
    bad += 0 if ok else 1
    row = {"M": M, "K": K, "N": N, "ok": ok, "maxerr": round(float((y1.float() - ref).abs().max()), 5),
           "vs_unsplit_max": round(float((y1.float() - y0.float()).abs().max()), 5)}
    for path in ("tilesplit", "mfma", "splitk") if M <= 128 else ("tilesplit", "mfma"):
        row[path] = round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1], path=path), 2 * nbuf), 2)
    row["auto"] = round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1]), 2 * nbuf), 2)
    print(json.dumps(row), flush=True)
    del sets; torch.cuda.empty_cache()
print("bad:", bad)
