import json, os, subprocess, sys
os.environ["EETQ_AMD_TUNING"] = "1"   # the A/B hooks this script sets answer only with this switch (csrc/common.hpp: tuning_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from eetq_amd import ops
    from sweep import chain_us
    dev = "cuda:0"
    K = N = 4096
    nbuf = 40
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(nbuf)]
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    x = torch.randn(1, K, dtype=torch.float16, device=dev)
    out = {}
    for G in (2, 4, 8, 16):
        def step(i):
            idx = [(i * G + j) % nbuf for j in range(G)]
            ops.w8_a16_gemv_grouped([x] * G, [ws[k] for k in idx], [s] * G)
        out["G=%d us/problem" % G] = round(chain_us(step, 40) / G, 3)
    print(json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for u in ("16", "8"):
        env = dict(os.environ, EETQ_AMD_GROUPED_WAVES=u)
        r = subprocess.run([sys.executable, __file__, "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("waves=%s" % u, line[-1] if line else "fail", flush=True)
