"""Round 6: the tall tile (256 x 128 on eight waves, gemm_tile_kernel<..., RH = 2>; hook value 1) and the deep tile (256 x 128 on four
waves with 256 accumulators per lane, RB = 2; hook value 2) against the shipping 128 x 128 tile on the explicit
MFMA path at M >= 1024: bit-identity of the two outputs and graph-replayed chain times.  One child process per arm (the hook
EETQ_AMD_TILE_TALL is read once per process, behind EETQ_AMD_TUNING=1)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(4096, 4096), (4096, 11008), (11008, 4096), (5120, 15360), (5120, 27648), (13824, 5120)]
MS = (1024, 2048, 4096, 3000)


def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hashlib, torch
    import eetq_amd.ops as ops
    from sweep import chain_us
    out = {}
    for K, N in SHAPES:
        L = max(2, int(320e6 // (K * N)))
        g = torch.Generator(device="cuda:0").manual_seed(K + N)
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0", generator=g) for _ in range(L)]
        s = torch.rand(N, dtype=torch.float16, device="cuda:0", generator=g) * 0.01
        for M in MS:
            x = torch.randn(M, K, dtype=torch.float16, device="cuda:0", generator=g)
            y = ops.w8_a16_gemm(x, ws[0], s, path="mfma")
            digest = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]
            t = chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], s, path="mfma"), 2 * L, min_seconds=0.05)
            out["%dx%dx%d" % (K, N, M)] = [round(t, 2), digest]
        del ws
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    res = {}
    for name, tall in (("tile128", None), ("tall256", "1"), ("deep256", "2")):
        env = dict(os.environ)
        if tall:
            env["EETQ_AMD_TUNING"] = "1"
            env["EETQ_AMD_TILE_TALL"] = tall
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=1500)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if line:
            res[name] = json.loads(line[0][7:])
        else:
            print(name, "FAILED", r.stderr[-1500:], file=sys.stderr)
    for key in res.get("tile128", {}):
        a = res["tile128"][key]
        row = {"point": key, "tile128_us": a[0]}
        for arm in ("tall256", "deep256"):
            b = res.get(arm, {}).get(key)
            if b:
                row[arm + "_us"] = b[0]
                row[arm + "_ratio"] = round(b[0] / a[0], 3)
                row[arm + "_bit_identical"] = a[1] == b[1]
        print(json.dumps(row), flush=True)
