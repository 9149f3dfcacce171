import sys, json, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from eetq_amd import _lib, ops
dev = "cuda:0"
def kernel_us(run, n):
    L = _lib.lib(); _lib.check(L.eetq_prof_begin(n)); run()
    buf = (ctypes.c_float * n)(); cnt = ctypes.c_int(0)
    _lib.check(L.eetq_prof_end(buf, n, ctypes.byref(cnt)))
    us = np.array(buf[:cnt.value]); return float(np.median(us)), float(us.min())
for K, N in [(4096, 4096), (4096, 12288), (4096, 22016), (5120, 15360), (5120, 27648), (8192, 8192), (8192, 28672)]:
    nbuf = max(3, (768 << 20) // (K * N))
    sets = []
    for i in range(nbuf):
        w = torch.randint(-128, 128, (K, N), dtype=torch.int8, device=dev)
        sets.append((w, torch.rand(N, device=dev).half() * 0.01))
    for M in (1, 4):
        x = torch.rand(M, K, device=dev).half(); y = torch.empty(M, N, dtype=torch.float16, device=dev)
        it = 200
        def run():
            for i in range(it): ops.w8_a16_gemm_(x, sets[i % nbuf][0], sets[i % nbuf][1], y, M, N, K)
        run(); torch.cuda.synchronize()
        med, mn = kernel_us(run, it)
        nbytes = K * N + 2 * M * K + 2 * N + 2 * M * N
        print(json.dumps({"K": K, "N": N, "M": M, "MiB": round(K*N/2**20,1), "kernel_us_median": round(med, 2), "kernel_us_min": round(mn,2), "GBps": round(nbytes / med / 1e3, 1), "hbm_frac": round(nbytes / med / 1e3 / 8000, 3)}), flush=True)
    del sets; torch.cuda.empty_cache()
