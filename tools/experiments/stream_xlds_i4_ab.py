"""A/B of LDS-staged activations in the W4A16 stream kernel (EETQ_AMD_I4_STREAM_XLDS = byte limit of M*K*2) and a check of the
adopted W8A16 rule (EETQ_AMD_I8_STREAM_XLDS = 0 against the default) through the AUTO dispatcher.  One process per setting;
us per launch in graph-replayed chains over rotating weights + a hash of the outputs (must not depend on the setting)."""
import hashlib, json, os, subprocess, sys
os.environ["EETQ_AMD_TUNING"] = "1"   # the A/B hooks this script sets answer only with this switch (csrc/common.hpp: tuning_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES4 = [(4096, 4096), (4096, 11008), (4096, 12288), (4096, 22016), (5120, 5120), (5120, 13824), (8192, 8192), (11008, 4096), (13824, 5120)]
SHAPES8 = [(4096, 4096), (4096, 11008), (4096, 12288), (4096, 14336), (4096, 22016), (4096, 1024), (8192, 1024), (5120, 5120),
           (5120, 13824), (5120, 15360), (8192, 8192), (8192, 28672), (11008, 4096), (13824, 5120), (2048, 8192), (3072, 9216)]


def child(bits):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import eetq_amd.ops as ops
    from sweep import chain_us
    dev = "cuda:0"
    for K, N in (SHAPES4 if bits == 4 else SHAPES8):
        L = max(4, int(640e6 // (K * N * bits // 8)))
        g = torch.Generator(device=dev).manual_seed(K + N)
        if bits == 4:
            ws = [torch.randint(-128, 127, (K, N // 2), dtype=torch.int8, device=dev, generator=g) for _ in range(L)]
        else:
            ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev, generator=g) for _ in range(L)]
        s = torch.rand(N, dtype=torch.float16, device=dev, generator=g) * 0.01
        for M in ((2, 4, 8, 12, 16) if bits == 4 else (2, 3, 4, 5, 8)):
            x = torch.randn(M, K, dtype=torch.float16, device=dev, generator=g)
            y = ops.w8_a16_gemm(x, ws[0], s)
            h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
            def step(i):
                ops.w8_a16_gemm(x, ws[i % L], s)
            print(json.dumps({"K": K, "N": N, "M": M, "us": round(chain_us(step, 2 * L, min_seconds=0.02), 2), "sha": h}), flush=True)
        del ws


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
        sys.exit(0)
    for bits, settings in ((4, (("regs", {}), ("lds64k", {"EETQ_AMD_I4_STREAM_XLDS": "65536"}), ("lds160k", {"EETQ_AMD_I4_STREAM_XLDS": "147456"}),
                                ("lds64k_nt1", {"EETQ_AMD_I4_STREAM_XLDS": "65536", "EETQ_AMD_I4_STREAM_XLDS_NT": "1"}),
                                ("lds64k_nt2", {"EETQ_AMD_I4_STREAM_XLDS": "65536", "EETQ_AMD_I4_STREAM_XLDS_NT": "2"}))),
                           (8, (("off", {"EETQ_AMD_I8_STREAM_XLDS": "0"}), ("rule", {})))):
        if len(sys.argv) > 1 and int(sys.argv[1]) != bits:
            continue
        runs = {}
        for tag, env in settings:
            e = dict(os.environ); e.update(env)
            out = subprocess.run([sys.executable, __file__, "child", str(bits)], env=e, capture_output=True, text=True, timeout=900)
            rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
            if not rows:
                print(tag, "FAILED", out.stderr[-2000:])
            runs[tag] = {(r["K"], r["N"], r["M"]): r for r in rows}
        first = settings[0][0]
        for key in runs.get(first, {}):
            line = {"bits": bits, "K": key[0], "N": key[1], "M": key[2]}
            for tag in runs:
                r = runs[tag].get(key)
                if r:
                    line[tag] = r["us"]
                    if tag != first:
                        line[tag + "_same_bits"] = r["sha"] == runs[first][key]["sha"]
            print(json.dumps(line), flush=True)
