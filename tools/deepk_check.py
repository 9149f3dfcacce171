"""Correctness of every split-K instantiation of the library, forced one by one through EETQ_AMD_SPLITK_PLAN:
tier-A against a torch fp32 matmul over the dequantised weight (the contract of tools/sweep.py), bit-equality of repeated
launches, plus odd shapes (K % 256 != 0, N % 32 != 0, M not a multiple of 32).  usage: python tools/deepk_check.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eetq_amd import ops  # noqa: E402

dev = "cuda:0"
PLANS = {1: [(1, 22), (2, 22), (1, 33), (2, 33)],
         2: [(1, 22), (2, 22), (1, 33), (2, 33)],
         3: [(1, 22), (2, 22)],
         4: [(1, 22), (2, 22)]}
bad = 0
n = 0
SHAPES = ((4096, 4096), (4096, 11008), (11008, 4096), (5120, 13824), (13824, 5120), (4160, 4112), (320, 48), (1024, 80),
          (2048, 22016), (1024, 32768))   # the last two: more workgroups than the chip holds at two per CU (the geometry of
                                          # round 2's slab-store data hazard, gemm_splitk_kernel.hpp)
for K, N in SHAPES:
    g = torch.Generator(device=dev)
    g.manual_seed(K + N)
    w = ((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half()
    raw, qw, s = ops.quant_weights(w, torch.int8, True)
    wdq = (raw.float() * s.float()[None, :]).half().float()
    for M in (9, 17, 32, 33, 50, 64, 96, 128):
        x = (torch.rand(M, K, device=dev, generator=g) - 0.25).half()
        ref = x.float() @ wdq
        tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
        for nb, ring in PLANS[(M + 31) // 32]:
            for S in (1, 2, 4):
                if S > 1 and (K // 64 + 3) // 4 // S < 1:
                    continue
                os.environ["EETQ_AMD_SPLITK_PLAN"] = "%d,%d,%d" % (nb, S, ring)
                try:
                    y1 = ops.w8_a16_gemm(x, qw, s, path="splitk")
                    y2 = ops.w8_a16_gemm(x, qw, s, path="splitk")
                    torch.cuda.synchronize()
                except RuntimeError as e:
                    print("ERR", K, N, M, nb, S, ring, str(e)[:80])
                    bad += 1
                    continue
                finally:
                    os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
                n += 1
                ok = bool(((y1.float() - ref).abs() <= tol).all()) and torch.equal(y1, y2)
                if not ok:
                    bad += 1
                    print("MISMATCH K=%d N=%d M=%d nb=%d S=%d ring=%d maxerr=%g same=%s" %
                          (K, N, M, nb, S, ring, float((y1.float() - ref).abs().max()), torch.equal(y1, y2)))
print("checked %d (shape, M, plan) cases, %d bad" % (n, bad))
sys.exit(1 if bad else 0)
