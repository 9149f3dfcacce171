"""8 + 8 + 4 column units (gemv_mixed_kernel) against the 8-column units at the Llama-2-13B O / down projections: two
processes (EETQ_AMD_GEMV_MIXED is read once), graph-replayed chains over rotating weights, us per call.
usage: python tools/experiments/mixed_units_check.py"""
import json, os, subprocess, sys
os.environ["EETQ_AMD_TUNING"] = "1"   # the A/B hooks this script sets answer only with this switch (csrc/common.hpp: tuning_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from eetq_amd import ops
    from sweep import chain_us
    dev = "cuda:0"
    out = {}
    for K, N in [(5120, 5120), (13824, 5120), (4096, 5120), (8192, 7168)]:
        nbuf = max(4, (700 << 20) // (K * N))
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(nbuf)]
        s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
        x = torch.randn(1, K, dtype=torch.float16, device=dev)
        out["%dx%d" % (K, N)] = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % nbuf], s), 2 * nbuf), 2)
        del ws
    print(json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for mixed in ("0", "1"):
        env = dict(os.environ, EETQ_AMD_GEMV_MIXED=mixed)
        r = subprocess.run([sys.executable, __file__, "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("mixed=%s" % mixed, line[-1] if line else "fail", flush=True)
