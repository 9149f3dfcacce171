"""Block-owner form of the split-K tile (gemm_splitk_own_kernel.hpp) against the k-split kernel: forced plans through
EETQ_AMD_SPLITK_PLAN = "nb,s,ring,own" on the explicit split-K path, graph-replayed chains over rotating weight sets (us per
call); every forced plan is checked against the tiled kernel (tier A) and for launch-to-launch bit identity.
usage: python tools/experiments/own_blocks_scan.py
RESULT (profiles/r04_own_blocks_scan.jsonl): exact in every plan; against the k-split kernel on the SAME plan it saves 0.3-0.4 us
(4096^2 M = 64, plan 2,4,33: 10.21 vs 10.58 us) -- the cross-wave LDS add is not the 1.1-2.2 us the phase stamps of round 2
suggested -- and the shipping plan (1,2,33: 9.69 us) stays ahead; only M = 128 at 4096^2 gains (2,4,22 with two row blocks per
wave: 14.69 vs 15.71 us), M = 96 loses (13.97 vs 12.72), every other shape loses 3-40 %.  SHELVED: kernel =
tools/experiments/gemm_splitk_own_kernel.hpp (move it next to gemm_splitk.hip), launcher = own_blocks_launcher.patch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from eetq_amd import ops
from sweep import chain_us

dev = "cuda:0"
CASES = [((4096, 4096), (40, 64, 96, 128)), ((11008, 4096), (64, 128)), ((5120, 5120), (64, 128)), ((4096, 11008), (64,)),
         ((5120, 13824), (64,)), ((13824, 5120), (64, 128))]
for (K, N), Ms in CASES:
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(1)
    sets = [tuple(ops.quant_weights(((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half(), torch.int8, False)) for _ in range(nbuf)]
    for M in Ms:
        x = torch.rand(M, K, device=dev, generator=g).half()
        ref = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="mfma").float()
        tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
        row = {"K": K, "N": N, "M": M, "auto": round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1]), 2 * nbuf), 2)}
        plans = ["2,2,33,1", "2,4,33,1", "2,4,22,1", "2,2,22,1"] if M <= 64 else ["1,2,22,1", "1,4,22,1", "2,2,22,1", "2,4,22,1"]
        plans += ["2,4,33,0", "2,4,22,0", "2,2,22,0"] if M <= 64 else ["2,4,22,0", "1,2,22,0"]
        for plan in plans:
            os.environ["EETQ_AMD_SPLITK_PLAN"] = plan
            try:
                y1 = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="splitk")
                y2 = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="splitk")
                ok = bool(((y1.float() - ref).abs() <= tol).all()) and torch.equal(y1, y2)
                us = chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1], path="splitk"), 2 * nbuf)
                row[plan] = ("%.2f" % us) + ("" if ok else " WRONG")
            except RuntimeError as e:
                row[plan] = "err " + str(e)[:50]
            finally:
                os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
        print(json.dumps(row), flush=True)
    del sets
    torch.cuda.empty_cache()
