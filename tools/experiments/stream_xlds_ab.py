"""A/B of the stream kernel's LDS-staged activations (EETQ_AMD_I8_STREAM_XLDS = byte limit of M*K*2; experiment of round 4).
Runs itself once per setting (the switch is read once per process), prints us per launch (graph-replayed chain over rotating
weights) and a hash of the outputs: the LDS form feeds the same fragments to the same MFMAs in the same order, so the hashes must
be equal to the register form's."""
import hashlib, json, os, subprocess, sys
os.environ["EETQ_AMD_TUNING"] = "1"   # the A/B hooks this script sets answer only with this switch (csrc/common.hpp: tuning_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(4096, 4096), (4096, 11008), (4096, 12288), (4096, 22016), (5120, 5120), (5120, 13824), (5120, 15360), (8192, 8192),
          (11008, 4096), (13824, 5120)]
MS = (2, 4, 5, 8, 12, 16)


def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import eetq_amd.ops as ops
    from sweep import chain_us
    dev = "cuda:0"
    for K, N in SHAPES:
        L = max(4, int(640e6 // (K * N)))
        g = torch.Generator(device=dev).manual_seed(K + N)
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev, generator=g) for _ in range(L)]
        s = torch.rand(N, dtype=torch.float16, device=dev, generator=g) * 0.01
        for M in MS:
            x = torch.randn(M, K, dtype=torch.float16, device=dev, generator=g)
            y = ops.w8_a16_gemm(x, ws[0], s, path="stream")
            h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
            def step(i):
                ops.w8_a16_gemm(x, ws[i % L], s, path="stream")
            print(json.dumps({"K": K, "N": N, "M": M, "us": round(chain_us(step, 2 * L, min_seconds=0.02), 2), "sha": h}), flush=True)
        del ws


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    runs = {}
    for tag, env in (("regs", {}), ("lds64k", {"EETQ_AMD_I8_STREAM_XLDS": "65536"}), ("lds128k", {"EETQ_AMD_I8_STREAM_XLDS": "131072"}),
                     ("lds64k_nt1", {"EETQ_AMD_I8_STREAM_XLDS": "65536", "EETQ_AMD_I8_STREAM_XLDS_NT": "1"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=600)
        rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        if not rows:
            print(tag, "FAILED", out.stderr[-2000:])
        runs[tag] = {(r["K"], r["N"], r["M"]): r for r in rows}
    for key in runs.get("regs", {}):
        line = {"K": key[0], "N": key[1], "M": key[2]}
        for tag in runs:
            r = runs[tag].get(key)
            if r:
                line[tag] = r["us"]
                if tag != "regs":
                    line[tag + "_same_bits"] = r["sha"] == runs["regs"][key]["sha"]
        print(json.dumps(line), flush=True)
