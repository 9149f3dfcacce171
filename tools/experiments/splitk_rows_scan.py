"""Round 5: row groups (R) in the split-K medium-batch tile.  For every (K, N) x M the script forces plans through
EETQ_AMD_SPLITK_PLAN="nb,s,ring,r" on path="splitk" (read per call on the forced path) -- nb column blocks of 32, s K slices,
r row groups of 32 * ceil(M / (32 r)) rows -- and times each as a graph-replayed chain next to AUTO; every forced plan's output is
compared with AUTO's (tier A: another summation order).  One JSON line per point.
usage: python tools/experiments/splitk_rows_scan.py [--shapes KxN,...] [--ms 24,32,...] [--out file]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import eetq_amd.ops as ops  # noqa: E402
from sweep import chain_us  # noqa: E402

SHAPES = [(4096, 4096), (4096, 6144), (4096, 11008), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120), (8192, 8192), (7168, 7168)]
MS = (24, 32, 48, 64, 96, 128)
NCU = 256


def plans_for(M, N):
    out = []
    for r in (1, 2, 3, 4, 6, 8, 12, 16):
        if r > 1 and M <= 32 * (r // 2):
            continue                      # an empty row group
        mt = -(-M // (32 * r))
        if mt > 4:
            continue
        if -(-M // (32 * mt)) != r:
            continue                      # this r collapses to a smaller one
        for nb in (1, 2):
            for s in (1, 2, 4):
                wgs = -(-N // (32 * nb)) * s * r
                if wgs > 3 * NCU or (wgs * 4 < NCU):
                    continue
                ring = 33 if (mt <= 2 and wgs <= NCU) else 22
                out.append((nb, s, ring, r))
    return out


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
            os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
            ref = ops.w8_a16_gemm(x, ws[0], sc).float()
            tol = 1e-3 * ref.abs().max() + 2e-3 * ref.abs()
            row = {"K": K, "N": N, "M": M, "auto": round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc), max(2 * L, 40), 0.012), 2)}
            best = None
            for plan in plans_for(M, N):
                key = "%d,%d,%d,%d" % plan
                os.environ["EETQ_AMD_SPLITK_PLAN"] = key
                try:
                    y = ops.w8_a16_gemm(x, ws[0], sc, path="splitk").float()
                    ok = bool(((y - ref).abs() <= tol).all())
                    t = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="splitk"), max(2 * L, 40), 0.012), 2)
                    row[key] = t if ok else "WRONG(%.3g)" % float((y - ref).abs().max())
                    if ok and (best is None or t < best[1]):
                        best = (key, t)
                except RuntimeError as e:
                    row[key] = "err:" + str(e)[:50]
            os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
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
