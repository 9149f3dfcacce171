"""Sweep of the small-batch kernel's plans (EETQ_AMD_I8_STREAM_PLAN=form,nt,waves) over shapes x M = 2..8: one process per plan
(the switch is read once), us per launch in graph-replayed chains over rotating weights, a hash of the outputs (all plans with the
same wave count must agree bit for bit).  Output: one JSON line per (shape, M) with every plan's time, and the rule's."""
import hashlib, json, os, subprocess, sys
os.environ["EETQ_AMD_TUNING"] = "1"   # the A/B hooks this script sets answer only with this switch (csrc/common.hpp: tuning_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(4096, 4096), (4096, 11008), (4096, 12288), (4096, 14336), (4096, 22016), (5120, 5120), (5120, 13824), (5120, 15360), (5120, 27648),
          (8192, 8192), (8192, 1024), (4096, 1024), (11008, 4096), (13824, 5120), (8192, 28672), (28672, 8192), (3072, 9216), (2048, 8192),
          (6144, 6144), (4096, 6144), (7168, 7168)]
MS = (2, 3, 4, 5, 6, 7, 8)
BITS = 8       # argv: bits [plan tags ...]
PLANS = [("rule", None), ("off", "off"), ("block1", "block,1,0"), ("block2", "block,2,0"), ("ring1_8", "ring,1,8"), ("ring1_16", "ring,1,16"),
         ("ring2_8", "ring,2,8"), ("ring2_16", "ring,2,16")]     # "off": EETQ_AMD_I*_STREAM_XLDS=0, the register form of rounds 1-3

def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import eetq_amd.ops as ops
    from sweep import chain_us
    dev = "cuda:0"
    for K, N in SHAPES:
        L = max(4, int(640e6 // (K * N * BITS // 8)))
        g = torch.Generator(device=dev).manual_seed(K + N)
        ws = [torch.randint(-128, 127, (K, N if BITS == 8 else N // 2), dtype=torch.int8, device=dev, generator=g) for _ in range(L)]
        s = torch.rand(N, dtype=torch.float16, device=dev, generator=g) * 0.01
        for M in MS:
            x = torch.randn(M, K, dtype=torch.float16, device=dev, generator=g)
            if BITS == 4 and K % 128:
                continue
            y = ops.w8_a16_gemm(x, ws[0], s, path="stream")
            h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
            def step(i):
                ops.w8_a16_gemm(x, ws[i % L], s, path="stream")
            print(json.dumps({"K": K, "N": N, "M": M, "us": round(chain_us(step, 2 * L, min_seconds=0.015), 2), "sha": h}), flush=True)
        del ws


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        BITS = int(sys.argv[2])
        if BITS == 4:
            MS = (2, 3, 4, 5, 6, 8, 12, 16)
        if os.environ.get("SWEEP_SHAPES"):   # "K1xN1,K2xN2"
            SHAPES = [tuple(int(v) for v in a.split("x")) for a in os.environ["SWEEP_SHAPES"].split(",")]
        if os.environ.get("SWEEP_MS"):
            MS = tuple(int(a) for a in os.environ["SWEEP_MS"].split(","))
        child()
        sys.exit(0)
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    want = sys.argv[2:]
    runs = {}
    for tag, plan in PLANS:
        if want and tag not in want:
            continue
        e = dict(os.environ)
        if plan == "off":
            e["EETQ_AMD_I%d_STREAM_XLDS" % bits] = "0"
        elif plan:
            e["EETQ_AMD_I%d_STREAM_PLAN" % bits] = plan
        out = subprocess.run([sys.executable, __file__, "child", str(bits)], env=e, capture_output=True, text=True, timeout=900)
        rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        if not rows:
            print(tag, "FAILED", out.stderr[-2000:])
        runs[tag] = {(r["K"], r["N"], r["M"]): r for r in rows}
    for key in runs.get("rule", {}):
        line = {"bits": bits, "K": key[0], "N": key[1], "M": key[2]}
        shas = {}
        for tag in runs:
            r = runs[tag].get(key)
            if r:
                line[tag] = r["us"]
                shas.setdefault(r["sha"], []).append(tag)
        line["bit_classes"] = sorted(shas.values(), key=len, reverse=True)
        print(json.dumps(line), flush=True)
