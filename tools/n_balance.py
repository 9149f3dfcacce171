"""GEMV time against N at fixed K (how much does the 8-column-unit granularity cost at N = 5120 on 256 CUs?)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eetq_amd.ops as ops
dev = "cuda:0"
def timed(fn, reps=30):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s): fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for K in (13824, 5120):
    for N in (4096, 4608, 5120, 5632, 6144, 8192):
        L = max(4, int(700e6 // (K * N)))
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(L)]
        s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
        x = torch.randn(1, K, dtype=torch.float16, device=dev)
        us = timed(lambda: [ops.w8_a16_gemm(x, w, s) for w in ws]) / L
        print(json.dumps({"K": K, "N": N, "MB": round(K * N / 1e6, 1), "us": round(us, 2), "GBps": round(K * N / us / 1e3)}))
        del ws
