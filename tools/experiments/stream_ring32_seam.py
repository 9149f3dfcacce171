"""Round 6, verdict item 5: the small-batch stream kernel with TWO 16-row MFMA tiles per weight register tile at 17 <= M <= 32
(32-row per-wave LDS ring, streamk_kernel<..., XM = 5>) against its register form and against what AUTO runs there (split-K
plans) on the seam shapes of profiles/r05_stream_splitk_seam.jsonl.  One child process per stream plan (the plan hook is read
once per process, behind EETQ_AMD_TUNING=1); every point a graph-replayed chain over rotating weights; every stream result is
checked against a torch fp32 product of the dequantised weight (tier A).
usage: python tools/experiments/stream_ring32_seam.py  > profiles/r06_stream_ring32_seam.jsonl"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(4096, 4096), (4096, 6144), (4096, 11008), (4096, 12288), (11008, 4096), (5120, 5120), (5120, 13824), (5120, 15360),
          (13824, 5120), (8192, 8192), (8192, 10240), (7168, 7168), (14336, 4096), (3584, 18944)]
MS = (17, 24, 32)
PLANS = {"regs16": None, "regs8_nt1": "regs,1,8", "regs8_nt2": "regs,2,8", "ring32_nt1": "ring,1,8", "ring32_nt2": "ring,2,8"}


def child(path_mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import eetq_amd.ops as ops
    from sweep import chain_us
    dev = "cuda:0"
    out = {}
    for K, N in SHAPES:
        L = max(4, int(640e6 // (K * N)))
        g = torch.Generator(device=dev).manual_seed(K + N)
        raw = torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev, generator=g)
        ws = [ops.preprocess_weights(raw)] + [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev, generator=g) for _ in range(L - 1)]
        s = (torch.rand(N, dtype=torch.float16, device=dev, generator=g) * 0.01 + 0.001)
        for M in MS:
            x = torch.randn(M, K, dtype=torch.float16, device=dev, generator=g)
            y = ops.w8_a16_gemm(x, ws[0], s, path=path_mode)
            ref = x.float() @ (raw.float() * s.float()).half().float()
            tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
            ok = bool(((y.float() - ref).abs() <= tol).all())
            us = chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], s, path=path_mode), 2 * L, min_seconds=0.02)
            out["%dx%dx%d" % (K, N, M)] = [round(us, 2), ok]
        del ws
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2])
        sys.exit(0)
    res = {}
    runs = [("auto", "auto", None)] + [(name, "stream", plan) for name, plan in PLANS.items()]
    for name, mode, plan in runs:
        env = dict(os.environ)
        env.pop("EETQ_AMD_I8_STREAM_PLAN", None)
        if plan:
            env["EETQ_AMD_TUNING"] = "1"
            env["EETQ_AMD_I8_STREAM_PLAN"] = plan
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", mode], env=env, capture_output=True, text=True, timeout=1500)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(name, "FAILED", r.stderr[-2000:], file=sys.stderr)
            continue
        res[name] = json.loads(line[0][7:])
    for K, N in SHAPES:
        for M in MS:
            key = "%dx%dx%d" % (K, N, M)
            row = {"K": K, "N": N, "M": M}
            for name in res:
                row[name] = res[name][key][0]
                if not res[name][key][1]:
                    row[name + "_tier_a"] = False
            stream = {k: v for k, v in row.items() if k.startswith(("regs", "ring")) and isinstance(v, float)}
            if stream and "auto" in row:
                best = min(stream, key=stream.get)
                row["best_stream"] = best
                row["best_stream_vs_auto_pct"] = round((stream[best] / row["auto"] - 1) * 100, 1)
            print(json.dumps(row), flush=True)
