"""Round 5: what would balancing K steps over ALL CUs (Stream-K style) buy the split-K medium-batch tile?
The tile's plans quantise: 4096 x 11008 at M = 64 is 172 workgroups x 16 steps (67 % of the CUs), 3584 x 18944 at M = 32 is 296
workgroups = two rounds of 14 steps.  Before building the balanced form this probe measures the two numbers its gain depends on,
with the kernel as it is (forced plans, path="splitk", graph-replayed chains):
  (1) T(steps) at fixed N, M, plan, S = 1: the per-step cost and the fixed cost (K = 256 * steps);
  (2) the hand-over cost: the same steps per workgroup at S = 1 / 2 / 4 (K = 2048 / 4096 / 8192 at N = 4096, nb = 2).
One JSON line per point."""
import json, os, sys
import torch
os.environ["EETQ_AMD_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import eetq_amd.ops as ops  # noqa: E402
from sweep import chain_us  # noqa: E402


def t(K, N, M, plan):
    L = max(2, int(640e6 // (K * N)))
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0") for _ in range(L)]
    sc = torch.rand(N, dtype=torch.float16, device="cuda:0") * 0.01
    x = torch.randn(M, K, dtype=torch.float16, device="cuda:0")
    os.environ["EETQ_AMD_SPLITK_PLAN"] = plan
    us = chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="splitk"), max(2 * L, 40), 0.012)
    os.environ.pop("EETQ_AMD_SPLITK_PLAN")
    return round(us, 2)


rows = []
for N, M, plan in ((11008, 64, "2,1,33,1"), (11008, 32, "2,1,33,1"), (18944, 32, "2,1,22,1"), (13824, 64, "2,1,33,1"), (5120, 64, "2,1,33,1"),
                   (10240, 64, "2,1,33,1")):
    for steps in (4, 6, 8, 11, 12, 16, 20, 24, 32):
        r = {"probe": "steps", "N": N, "M": M, "plan": plan, "steps": steps, "K": 256 * steps, "us": t(256 * steps, N, M, plan)}
        print(json.dumps(r), flush=True)
for M in (32, 64):
    for nb, N in ((2, 4096), (1, 2048), (2, 8192 // 2)):
        for s, K in ((1, 2048), (2, 4096), (4, 8192)):
            r = {"probe": "handover", "N": N, "M": M, "nb": nb, "S": s, "K": K, "steps_per_wg": 8,
                 "us": t(K, N, M, "%d,%d,%d,1" % (nb, s, 33 if M <= 64 else 22))}
            print(json.dumps(r), flush=True)
