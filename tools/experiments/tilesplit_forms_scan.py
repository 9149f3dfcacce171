"""Round 5: ring depth / occupancy forms of the K-sliced 128 x 64 tile (gemm_tile_splitk_kernel<1, ST, OCC>).  Forces
EETQ_AMD_TILESPLIT_PLAN="S,ring,occ" on path="tilesplit" (read per call on the forced path) for S in {2, 4}, (ring, occ) in
{(6,1), (4,1), (3,1), (4,2), (3,2)}, times each as a graph-replayed chain next to AUTO and the unsplit tiled kernel, and compares
every forced plan's output with the unsplit kernel's (tier A).  One JSON line per point.
usage: python tools/experiments/tilesplit_forms_scan.py [--shapes KxN,...] [--ms 128,192,...] [--out file]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import eetq_amd.ops as ops  # noqa: E402
from sweep import chain_us  # noqa: E402

SHAPES = [(4096, 4096), (5120, 5120), (4096, 11008), (11008, 4096), (8192, 8192), (4096, 6144)]
MS = (97, 128, 192, 256, 384, 512)
FORMS = ((6, 1), (4, 1), (3, 1), (4, 2), (3, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--ms", default=",".join(str(m) for m in MS))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    shapes = SHAPES if not a.shapes else [tuple(int(v) for v in t.split("x")) for t in a.shapes.split(",")]
    out = open(a.out, "w") if a.out else None
    for K, N in shapes:
        L = max(2, int(640e6 // (K * N)))
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0") for _ in range(L)]
        sc = torch.rand(N, dtype=torch.float16, device="cuda:0") * 0.01
        for M in (int(m) for m in a.ms.split(",")):
            x = torch.randn(M, K, dtype=torch.float16, device="cuda:0")
            os.environ.pop("EETQ_AMD_TILESPLIT_PLAN", None)
            ref = ops.w8_a16_gemm(x, ws[0], sc, path="mfma").float()
            tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
            calls = max(2 * L, 40)
            row = {"K": K, "N": N, "M": M,
                   "auto": round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc), calls, 0.012), 2),
                   "mfma": round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="mfma"), calls, 0.012), 2)}
            best = None
            tiles = -(-M // 128) * -(-N // 64)
            for S in (2, 4):
                if tiles * S > 2 * 256 + 64:
                    continue
                for ring, occ in FORMS:
                    if (K // 64) // S < ring - 1:
                        continue
                    if occ == 1 and tiles * S > 256 + 64:
                        continue
                    key = "%d,%d,%d" % (S, ring, occ)
                    os.environ["EETQ_AMD_TILESPLIT_PLAN"] = key
                    try:
                        y = ops.w8_a16_gemm(x, ws[0], sc, path="tilesplit").float()
                        ok = bool(((y - ref).abs() <= tol).all())
                        y2 = ops.w8_a16_gemm(x, ws[0], sc, path="tilesplit").float()
                        same = bool(torch.equal(y, y2))
                        t = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="tilesplit"), calls, 0.012), 2)
                        row[key] = t if (ok and same) else "WRONG(%.3g,%s)" % (float((y - ref).abs().max()), same)
                        if ok and same and (best is None or t < best[1]):
                            best = (key, t)
                    except RuntimeError as e:
                        row[key] = "err:" + str(e)[:50]
            os.environ.pop("EETQ_AMD_TILESPLIT_PLAN", None)
            if best:
                row["best"], row["best_us"], row["gain_vs_auto"] = best[0], best[1], round(1 - best[1] / row["auto"], 4)
            line = json.dumps(row)
            print(line, flush=True)
            if out:
                out.write(line + "\n"); out.flush()
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
