"""A/B of two builds of the library on ONE box: the working tree against a second copy of the package (e.g. `git archive HEAD
eetq_amd include tools/... | tar -x -C .ab/old` + make there).  Each arm runs in its own process (ctypes binding), the arms
alternate (old, new, old, new), every point is a graph-replayed chain over rotating weights (tools/sweep.py::chain_us); printed:
per point the best of each arm's passes and new / old - 1.
usage: python tools/experiments/ab_lib.py --old .ab/old [--points KxNxM[:path],...] [--passes 2]"""
import argparse, json, os, subprocess, sys

DEFAULT = ("4096x4096x17,4096x4096x32,4096x4096x64,4096x4096x128,4096x4096x256,4096x11008x32,4096x11008x64,11008x4096x64,"
           "5120x5120x64,5120x13824x64,13824x5120x64,4096x6144x24,8192x8192x64,8192x1024x16,4096x4096x8,4096x4096x1024")


def child(root, points):
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
    import torch
    import eetq_amd.ops as ops
    from sweep import chain_us
    assert os.path.realpath(ops.__file__).startswith(os.path.realpath(root)), ops.__file__
    out, cache = {}, {}
    for p in points:
        shape, _, path = p.partition(":")
        K, N, M = (int(v) for v in shape.split("x"))
        if (K, N) not in cache:
            cache.clear(); torch.cuda.empty_cache()
            L = max(2, int(640e6 // (K * N)))
            g = torch.Generator(device="cuda:0").manual_seed(K + N)
            cache[(K, N)] = ([torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0", generator=g) for _ in range(L)],
                             [torch.rand(N, dtype=torch.float16, device="cuda:0", generator=g) * 0.01 for _ in range(L)])
        ws, scs = cache[(K, N)]   # one scale vector per weight set: like the weights they come from HBM (a model's layers do)
        L = len(ws)
        x = torch.randn(M, K, dtype=torch.float16, device="cuda:0")
        kw = {"path": path} if path else {}
        out[p] = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], scs[i % L], **kw), max(2 * L, 40 if M <= 256 else 8), 0.015), 3)
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], sys.argv[3].split(","))
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--old", required=True)
    ap.add_argument("--points", default=DEFAULT)
    ap.add_argument("--passes", type=int, default=2)
    a = ap.parse_args()
    new_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    arms = {"old": os.path.abspath(a.old), "new": new_root}
    res = {"old": [], "new": []}
    env = dict(os.environ, EETQ_AMD_BOUNDARY="ctypes")
    for _ in range(a.passes):
        for arm in ("old", "new"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", arms[arm], a.points], env=env, capture_output=True,
                               text=True, timeout=1200)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(arm, "FAILED", r.stderr[-1500:])
                sys.exit(1)
            res[arm].append(json.loads(line[0][7:]))
    for p in a.points.split(","):
        o, n = min(r[p] for r in res["old"]), min(r[p] for r in res["new"])
        print(json.dumps({"point": p, "old": o, "new": n, "delta_pct": round((n / o - 1) * 100, 1),
                          "old_all": [r[p] for r in res["old"]], "new_all": [r[p] for r in res["new"]]}), flush=True)
