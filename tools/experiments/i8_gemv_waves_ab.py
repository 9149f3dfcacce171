"""int8 M = 1 GEMV, generic-K form: 16-wave workgroups (shipping) vs 8-wave workgroups with 2 / 4 tiles in flight
(EETQ_AMD_I8_GEMV_WAVES = 82 / 84, read once: one process per arm), graph-replayed chains, us per call."""
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
    for K, N in [(4096, 11008), (11008, 4096), (5120, 13824), (5120, 15360), (5120, 27648), (8192, 8192), (4096, 22016)]:
        nbuf = max(4, (700 << 20) // (K * N))
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(nbuf)]
        s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
        x = torch.randn(1, K, dtype=torch.float16, device=dev)
        out["%dx%d" % (K, N)] = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % nbuf], s), 2 * nbuf), 2)
        del ws
    print(json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for u in (sys.argv[1:] or ["0", "82", "84"]):
        env = dict(os.environ, EETQ_AMD_I8_GEMV_WAVES=u)
        r = subprocess.run([sys.executable, __file__, "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("arm=%s" % u, line[-1] if line else "fail", flush=True)
