"""Is the M = 1024 tiled GEMM power-limited?  Same kernel, same shape: random operands vs all-zero operands (the guide's
"DVFS give-back": identical instruction streams clock higher when the data toggles fewer wires).  usage: python tools/experiments/gemm_power_check.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from eetq_amd import ops
from sweep import chain_us
dev = "cuda:0"
M = 1024
for K, N in [(4096, 4096), (5120, 13824)]:
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(1)
    rnd = [tuple(ops.quant_weights(((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half(), torch.int8, False)) for _ in range(nbuf)]
    zero_w = [(torch.full_like(rnd[0][0], -128), torch.zeros_like(rnd[0][1])) for _ in range(nbuf)]   # byte 0x80 = q 0
    xr = torch.rand(M, K, device=dev, generator=g).half() - 0.25
    xz = torch.zeros_like(xr)
    row = {"M": M, "K": K, "N": N}
    for name, x, sets in (("random", xr, rnd), ("zero_x", xz, rnd), ("zero_w", xr, zero_w), ("zero_both", xz, zero_w), ("random_again", xr, rnd)):
        row[name] = round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1], path="mfma"), nbuf), 2)
    print(json.dumps(row), flush=True)
    del rnd, zero_w; torch.cuda.empty_cache()
