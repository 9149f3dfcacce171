"""Correctness + time of the wide-tile kernel (path="wide") over its plans (EETQ_AMD_WIDE_PLAN="s,ks"), against a torch fp32
matmul over the dequantised weight (tier A) and against the split-K path's time.  usage: python tools/wide_check.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from eetq_amd import ops  # noqa: E402
from sweep import chain_us  # noqa: E402

dev = "cuda:0"
bad = n = 0
timing = "--time" in sys.argv
SHAPES = ((4096, 4096), (4096, 11008), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120), (5120, 15360), (4160, 4112),
          (320, 48), (1024, 80), (2048, 22016))
for K, N in SHAPES:
    g = torch.Generator(device=dev)
    g.manual_seed(K + N)
    L = max(2, int(640e6 // (K * N))) if timing else 1
    sets = []
    for i in range(L):
        w = ((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half()
        if i == 0:
            raw, qw, s = ops.quant_weights(w, torch.int8, True)
            sets.append((qw, s))
        else:
            sets.append(tuple(ops.quant_weights(w, torch.int8, False)))
        del w
    qw, s = sets[0]
    wdq = (raw.float() * s.float()[None, :]).half().float()
    for M in (9, 17, 32, 33, 50, 64):
        x = (torch.rand(M, K, device=dev, generator=g) - 0.25).half()
        ref = x.float() @ wdq
        tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
        row = {"K": K, "N": N, "M": M}
        for S in (1, 2, 4):
            for ks in (1, 2):
                if (K // 64 + ks - 1) // ks < S:
                    continue
                os.environ["EETQ_AMD_WIDE_PLAN"] = "%d,%d" % (S, ks)
                try:
                    y1 = ops.w8_a16_gemm(x, qw, s, path="wide")
                    y2 = ops.w8_a16_gemm(x, qw, s, path="wide")
                    torch.cuda.synchronize()
                    n += 1
                    ok = bool(((y1.float() - ref).abs() <= tol).all()) and torch.equal(y1, y2)
                    if not ok:
                        bad += 1
                        print("MISMATCH K=%d N=%d M=%d S=%d ks=%d maxerr=%g same=%s" %
                              (K, N, M, S, ks, float((y1.float() - ref).abs().max()), torch.equal(y1, y2)))
                    if timing and M in (17, 32, 64) and K * N >= (1 << 24):
                        row["wide s%d k%d" % (S, ks)] = round(chain_us(
                            lambda i: ops.w8_a16_gemm(x, sets[i % L][0], sets[i % L][1], path="wide"), 2 * L, 0.01), 2)
                except RuntimeError as e:
                    print("ERR", K, N, M, S, ks, str(e)[:80])
                    bad += 1
                finally:
                    os.environ.pop("EETQ_AMD_WIDE_PLAN", None)
        if timing and M in (17, 32, 64) and K * N >= (1 << 24):
            row["splitk"] = round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % L][0], sets[i % L][1], path="splitk"), 2 * L, 0.01), 2)
            row["wide auto"] = round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % L][0], sets[i % L][1], path="wide"), 2 * L, 0.01), 2)
            print(row, flush=True)
    del sets
    torch.cuda.empty_cache()
print("checked %d (shape, M, plan) cases, %d bad" % (n, bad))
