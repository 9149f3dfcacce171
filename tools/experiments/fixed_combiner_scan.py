"""Split-K hand-over forms against each other (round 4): tickets (every slice publishes, last arriver combines; lead = -1)
vs the fixed combiner with `lead` extra K steps (gemm_splitk_kernel<..., FIXED>), forced through EETQ_AMD_SPLITK_PLAN =
"nb,s,ring,lead" on the explicit split-K path.  Graph-replayed chains over rotating weight sets, us per call; every forced
plan is also checked against the tiled kernel (tier A) and launch-to-launch bit identity.
usage: python tools/experiments/fixed_combiner_scan.py [quick]
RESULT (profiles/r04_fixed_combiner_scan.jsonl): correct everywhere, slower everywhere -- 4096^2 M = 64: tickets 9.66 us, fixed
combiner 12.5 (lead 0) / 10.7 (lead 1) / 11.4 (lead 2); every extra step for the combiner costs ~0.5 us and the publish -> flag
-> read chain is not hidden by it.  SHELVED: the kernel side is tools/experiments/fixed_combiner.patch (apply to
eetq_amd/csrc/gemm_splitk_kernel.hpp + gemm_splitk.hip to re-run this scan); the library keeps the ticket hand-over."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from eetq_amd import ops
from sweep import chain_us

dev = "cuda:0"
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
CASES = [((4096, 4096), (32, 64, 128)), ((11008, 4096), (64,)), ((5120, 5120), (64,)), ((4096, 11008), (64,)), ((13824, 5120), (64,))]
if quick:
    CASES = CASES[:1]
PLANS = [(1, 2, 33), (1, 2, 22), (2, 2, 33), (2, 4, 22), (1, 4, 22), (2, 4, 33)]
LEADS = (-1, 0, 1, 2, 3, 4, 6)
for (K, N), Ms in CASES:
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(1)
    sets = [tuple(ops.quant_weights(((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half(), torch.int8, False)) for _ in range(nbuf)]
    for M in Ms:
        x = torch.rand(M, K, device=dev, generator=g).half()
        ref = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="mfma").float()
        tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
        row = {"K": K, "N": N, "M": M, "auto": round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1]), 2 * nbuf), 2)}
        for nb, s, ring in PLANS:
            if ring == 33 and (M + 31) // 32 > 2:
                continue
            for lead in LEADS:
                os.environ["EETQ_AMD_SPLITK_PLAN"] = "%d,%d,%d,%d" % (nb, s, ring, lead)
                try:
                    y1 = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="splitk")
                    y2 = ops.w8_a16_gemm(x, sets[0][0], sets[0][1], path="splitk")
                    ok = bool(((y1.float() - ref).abs() <= tol).all()) and torch.equal(y1, y2)
                    us = chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1], path="splitk"), 2 * nbuf)
                    row["%d,%d,%d,%d" % (nb, s, ring, lead)] = ("%.2f" % us) + ("" if ok else " WRONG")
                except RuntimeError as e:
                    row["%d,%d,%d,%d" % (nb, s, ring, lead)] = "err " + str(e)[:40]
                finally:
                    os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
        print(json.dumps(row), flush=True)
    del sets
    torch.cuda.empty_cache()
