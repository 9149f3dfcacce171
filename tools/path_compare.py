"""Kernel paths against each other at small batch (which path should AUTO pick for M = 2..16?): microseconds per launch in a
HIP graph over rotating weights (cold: the set exceeds the Infinity Cache).  usage: python tools/path_compare.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eetq_amd.ops as ops  # noqa: E402

dev = "cuda:0"


def timed(fn, reps=30):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for K, N in ((4096, 4096), (5120, 15360), (5120, 27648), (13824, 5120), (5120, 5120), (4096, 22016), (11008, 4096)):
    L = max(4, int(600e6 // (K * N)))
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    for M in (1, 2, 3, 4, 8):
        x = torch.randn(M, K, dtype=torch.float16, device=dev)
        row = {"K": K, "N": N, "M": M}
        for path in ("auto", "gemv", "stream"):
            if path == "gemv" and M > 4:
                continue
            try:
                row[path] = round(timed(lambda: [ops.w8_a16_gemm(x, w, s, path=path) for w in ws]) / L, 2)
            except RuntimeError as e:
                row[path] = str(e)[:40]
        print(json.dumps(row))
