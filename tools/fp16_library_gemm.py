"""Time the vendor fp16 GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the shapes of the W8A16 metric, same box.

Comparison line for DESIGN.md only: the library multiplies fp16 x fp16 (reads 2 bytes per weight, no dequant); it is
not on the product path and nothing in the package calls it.
"""
import json
import sys

import torch


def time_mm(M, N, K, iters=200, nbuf=20):
    x = (torch.rand(M, K, device="cuda") - 0.5).half()
    ws = [(torch.rand(K, N, device="cuda") - 0.5).half() for _ in range(nbuf)]  # rotate: 32 MiB each
    for i in range(30):
        torch.matmul(x, ws[i % nbuf])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i, (a, b) in enumerate(ev):
        a.record()
        torch.matmul(x, ws[i % nbuf])
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


if __name__ == "__main__":
    out = []
    for (M, N, K) in [(1, 4096, 4096), (8, 4096, 4096), (64, 4096, 4096), (128, 4096, 4096), (1024, 4096, 4096),
                      (1024, 11008, 4096), (4096, 4096, 4096)]:
        us = time_mm(M, N, K)
        out.append({"M": M, "N": N, "K": K, "event_us_median": round(us, 2), "TFLOPs": round(2.0 * M * N * K / us / 1e6, 1),
                    "note": "torch.cuda.Event around one torch.matmul (includes launch gap)"})
        print(out[-1], flush=True)
    if len(sys.argv) > 1:
        json.dump({"what": "torch.matmul fp16 x fp16 (vendor library) on the metric shapes", "results": out},
                  open(sys.argv[1], "w"), indent=1)
