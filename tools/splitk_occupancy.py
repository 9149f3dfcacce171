"""split-K plans that AUTO picks, timed as graph-replayed chains (tools/sweep.py::chain_us).  Run once as is and once with
EETQ_AMD_SPLITK_ONE_PER_CU=1 to compare one against two workgroups per CU on split (S > 1) launches."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from eetq_amd import ops
from sweep import chain_us
dev = "cuda:0"
for K, N in [(4096, 4096), (4096, 11008), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120)]:
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(1)
    sets = [tuple(ops.quant_weights(((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half(), torch.int8, False)) for _ in range(nbuf)]
    row = {"K": K, "N": N}
    for M in (17, 32, 64, 96, 128):
        x = torch.rand(M, K, device=dev, generator=g).half()
        row["M%d" % M] = round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1], path="splitk"), 2 * nbuf), 2)
    print(json.dumps(row), flush=True)
    del sets; torch.cuda.empty_cache()
