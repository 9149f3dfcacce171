"""Kernel-only time of the W4A16 GEMV / GEMM next to W8A16 on the same shapes (HIP start/stop events per dispatch)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from eetq_amd import _lib, ops

def kernel_us(run, n, per_call_max=4):
    L = _lib.lib(); cap = per_call_max * n
    _lib.check(L.eetq_prof_begin(cap)); run()
    buf = (ctypes.c_float * cap)(); cnt = ctypes.c_int(0)
    _lib.check(L.eetq_prof_end(buf, cap, ctypes.byref(cnt)))
    per = cnt.value // n
    us = np.array(buf[:per * n]).reshape(n, per).sum(axis=1)
    return float(np.median(us)), per

dev = "cuda:0"
for K, N in [(4096, 4096), (4096, 11008), (11008, 4096), (5120, 13824)]:
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(1)
    s8, s4 = [], []
    for i in range(nbuf):
        w = ((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half()
        s8.append(tuple(ops.quant_weights(w, torch.int8, False)))
        s4.append(tuple(ops.quant_weights(w, torch.quint4x2, False)))
        del w
    for M in (1, 4, 8, 64):
        x = torch.rand(M, K, device=dev, generator=g).half()
        out = {}
        for name, sets in (("w8", s8), ("w4", s4)):
            def run():
                for i in range(100):
                    ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1])
            run(); torch.cuda.synchronize()
            out[name], out[name + "_launches"] = kernel_us(run, 100)
        wbytes8 = K * N; 
        print(json.dumps({"K": K, "N": N, "M": M, "w8a16_us": round(out["w8"], 2), "w4a16_us": round(out["w4"], 2),
                          "w4_launches_per_call": out["w4_launches"], "w8_GBps": round((wbytes8 + 2*M*K + 2*N + 2*M*N) / out["w8"] / 1e3),
                          "w4_GBps": round((wbytes8 / 2 + 2*M*K + 2*N + 2*M*N) / out["w4"] / 1e3)}), flush=True)
    del s8, s4; torch.cuda.empty_cache()
