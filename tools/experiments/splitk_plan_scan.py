"""Forced split-K plans (EETQ_AMD_SPLITK_PLAN=nb,s,ring) against AUTO's choice, graph-replayed chains.  One process per plan
(the override is read once).  usage: python tools/experiments/splitk_plan_scan.py K N M [plan ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from eetq_amd import ops
    from sweep import chain_us
    K, N, M = (int(v) for v in sys.argv[2:5])
    dev = "cuda:0"
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(1)
    sets = [tuple(ops.quant_weights(((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half(), torch.int8, False)) for _ in range(nbuf)]
    x = torch.rand(M, K, device=dev, generator=g).half()
    print(round(chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1], path="splitk"), 2 * nbuf), 2))
    sys.exit(0)
K, N, M = sys.argv[1:4]
plans = sys.argv[4:] or ["", "1,1,33", "1,2,22", "1,2,33", "1,4,22", "2,1,33", "2,2,22", "2,2,33", "2,4,22"]
row = {"K": int(K), "N": int(N), "M": int(M)}
for plan in plans:
    env = dict(os.environ)
    if plan:
        env["EETQ_AMD_SPLITK_PLAN"] = plan
    out = subprocess.run([sys.executable, __file__, "--one", K, N, M], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
    row[plan or "auto"] = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "fail"
print(json.dumps(row))
