"""Host time per operator call through the two bindings of the C ABI (compiled module EETQ vs ctypes).

The shape is tiny (K = 64, N = 16, M = 1: a ~2 us kernel) and the queue is drained only at the end, so the loop is bound by
the host side of a call: argument checks, output allocation, stream lookup, the launch itself.  Also times the bare
launch floor (an empty kernel through ctypes, no tensors) and W8A16Linear.forward.  usage: python tools/host_overhead.py"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def per_call(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        best = min(best, dt / n)
    return best * 1e6


def main():
    from eetq_amd import _ext, _lib, ops_ctypes
    from eetq_amd.modules.qlinear import W8A16Linear
    ext = _ext.load()
    dev = "cuda:0"
    K, N = 64, 16
    w = (torch.randn(K, N, device=dev) * 0.02).half()
    qw, s = ext.quant_weights(w, torch.int8, False)
    x = torch.rand(1, K, dtype=torch.float16, device=dev)
    y = torch.empty(1, N, dtype=torch.float16, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    L = _lib.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sp = ctypes.c_void_p(sink.data_ptr())
    lin = W8A16Linear(K, N, bias=False, dev=dev)
    lin.qweight, lin.weight_scales = qw, s
    out = {
        "ext.w8_a16_gemm (fresh output)": per_call(lambda: ext.w8_a16_gemm(x, qw, s)),
        "ext.w8_a16_gemm_ (caller's output)": per_call(lambda: ext.w8_a16_gemm_(x, qw, s, y, 1, N, K)),
        "ctypes.w8_a16_gemm (fresh output)": per_call(lambda: ops_ctypes.w8_a16_gemm(x, qw, s)),
        "ctypes.w8_a16_gemm_ (caller's output)": per_call(lambda: ops_ctypes.w8_a16_gemm_(x, qw, s, y, 1, N, K)),
        "W8A16Linear.forward (product boundary)": per_call(lambda: lin(x)),
        "torch.empty([1, N]) alone": per_call(lambda: torch.empty(1, N, dtype=torch.float16, device=dev)),
        "bare launch: empty kernel via ctypes, no tensors": per_call(lambda: L.eetq_diag_empty(sp, 1, 64, stream)),
        "torch fp16 matmul x @ W (for scale)": per_call(lambda: torch.matmul(x, w)),
    }
    print(json.dumps({"what": "host microseconds per call, M=1 K=64 N=16, queue drained at the end", **{k: round(v, 2) for k, v in out.items()}}))


if __name__ == "__main__":
    main()
