"""Time per call of the W4A16 GEMV / GEMM next to W8A16 on the same shapes: HIP-graph-replayed chains of back-to-back calls
over rotating weight sets (tools/sweep.py::chain_us) -- NOT start/stop event pairs, whose ~4.2 us floor made the round-2
table read 4.3 us for both at 4096^2."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from eetq_amd import ops
from sweep import chain_us

dev = "cuda:0"
for K, N in [(4096, 4096), (4096, 11008), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120)]:
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(1)
    s8, s4 = [], []
    for i in range(nbuf):
        w = ((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half()
        s8.append(tuple(ops.quant_weights(w, torch.int8, False)))
        s4.append(tuple(ops.quant_weights(w, torch.quint4x2, False)))
        del w
    for M in (tuple(int(a) for a in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1, 4, 8, 64)):
        x = torch.rand(M, K, device=dev, generator=g).half()
        out = {}
        for name, sets in (("w8", s8), ("w4", s4)):
            def step(i):
                ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1])
            try:
                out[name] = chain_us(step, 2 * nbuf)
            except Exception:  # noqa: BLE001  W4A16 prefill expands into a per-stream scratch that cannot be created during
                # capture: time an eager loop instead (host launch time included)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(100):
                    step(i)
                b.record()
                torch.cuda.synchronize()
                out[name] = a.elapsed_time(b) * 10.0
                out[name + "_eager"] = True
        wbytes8 = K * N
        print(json.dumps({"K": K, "N": N, "M": M, "w8a16_us": round(out["w8"], 2), "w4a16_us": round(out["w4"], 2), "w4_timed_eagerly": bool(out.get("w4_eager", False)),
                          "w8_GBps": round((wbytes8 + 2*M*K + 2*N + 2*M*N) / out["w8"] / 1e3),
                          "w4_GBps": round((wbytes8 / 2 + 2*M*K + 2*N + 2*M*N) / out["w4"] / 1e3)}), flush=True)
    del s8, s4; torch.cuda.empty_cache()
