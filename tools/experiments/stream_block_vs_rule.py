"""Round 6: what would forcing the block-copy form (activations copied once per workgroup into LDS -- the form a fused RMS-norm
prologue needs) cost against the rule's pick on the two norm-consuming projections of a 13B decoder layer at M = 2, 3, 4, 8?
One child process per plan (the plan hook is read once per process, behind EETQ_AMD_TUNING=1)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(5120, 15360), (5120, 27648), (4096, 12288), (4096, 22016)]
MS = (2, 3, 4, 8)
PLANS = {"rule": None, "block_nt1_w16": "block,1,16", "block_nt1_w8": "block,1,8", "block_nt2_w8": "block,2,8", "block_nt2_w16": "block,2,16"}


def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import eetq_amd.ops as ops
    from sweep import chain_us
    out = {}
    for K, N in SHAPES:
        L = max(4, int(640e6 // (K * N)))
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0") for _ in range(L)]
        s = torch.rand(N, dtype=torch.float16, device="cuda:0") * 0.01
        for M in MS:
            x = torch.randn(M, K, dtype=torch.float16, device="cuda:0")
            out["%dx%dx%d" % (K, N, M)] = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], s, path="stream"), 2 * L, min_seconds=0.02), 2)
        del ws
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    res = {}
    for name, plan in PLANS.items():
        env = dict(os.environ)
        if plan:
            env["EETQ_AMD_TUNING"] = "1"
            env["EETQ_AMD_I8_STREAM_PLAN"] = plan
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if line:
            res[name] = json.loads(line[0][7:])
        else:
            print(name, "FAILED", r.stderr[-800:], file=sys.stderr)
    for key in res["rule"]:
        print(json.dumps(dict({"point": key}, **{n: res[n][key] for n in res})), flush=True)
