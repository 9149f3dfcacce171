"""Round 5: deeper WEIGHT rings in the split-K tile, re-measured now that the K loop's steady steps use a constant wait (round 3
measured them 0-10 % slower through the per-step compare-and-branch chain, whose length grows with the ring depths).
Forced plans EETQ_AMD_SPLITK_PLAN="nb,s,ring,1" with ring = 10*SA + SB: 33 (shared), 34 / 36 (64-column blocks), 38 (32-column
blocks); every plan checked against AUTO's output.  usage: python tools/experiments/splitk_deep_rings.py [--out file]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import eetq_amd.ops as ops  # noqa: E402
from sweep import chain_us  # noqa: E402

SHAPES = [(4096, 11008), (5120, 13824), (4096, 4096), (11008, 4096), (8192, 10240), (4096, 28672), (5120, 5120)]
MS = (24, 32, 48, 64)


def main():
    out = open(sys.argv[sys.argv.index("--out") + 1], "w") if "--out" in sys.argv else None
    for K, N in SHAPES:
        L = max(2, int(640e6 // (K * N)))
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0") for _ in range(L)]
        sc = torch.rand(N, dtype=torch.float16, device="cuda:0") * 0.01
        for M in MS:
            x = torch.randn(M, K, dtype=torch.float16, device="cuda:0")
            os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
            ref = ops.w8_a16_gemm(x, ws[0], sc, path="splitk").float()
            tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
            row = {"K": K, "N": N, "M": M, "planned": round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="splitk"), max(2 * L, 40), 0.012), 2)}
            mt = -(-M // 32)
            for nb, s in ((1, 1), (1, 2), (2, 1), (2, 2), (2, 4)):
                rings = [33] + ([38] if nb == 1 else [34] + ([36] if mt == 1 else []))
                for ring in rings:
                    key = "%d,%d,%d,1" % (nb, s, ring)
                    os.environ["EETQ_AMD_SPLITK_PLAN"] = key
                    try:
                        y = ops.w8_a16_gemm(x, ws[0], sc, path="splitk").float()
                        ok = bool(((y - ref).abs() <= tol).all())
                        t = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="splitk"), max(2 * L, 40), 0.012), 2)
                        row[key] = t if ok else "WRONG(%.3g)" % float((y - ref).abs().max())
                    except RuntimeError as e:
                        row[key] = "err:" + str(e)[:40]
            os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
            line = json.dumps(row)
            print(line, flush=True)
            if out:
                out.write(line + "\n"); out.flush()
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
