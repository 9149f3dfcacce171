"""Round 5: the balanced (Stream-K) form of the split-K medium-batch tile against AUTO.
For every (K, N) x M: AUTO's time and path, then the balanced launch forced through EETQ_AMD_SPLITK_PLAN="nb,0,0,1,q" on
path="splitk" (q = 0: the planned ceil(units / CUs); other q on request) -- graph-replayed chains over rotating weights; every
balanced output is compared with AUTO's (another summation order: |d| <= 1e-3 max|y| + 2e-3 |y|) and must be bit-identical
between two launches.  One JSON line per point.
usage: python tools/experiments/splitk_balanced_scan.py [--shapes KxN,...] [--ms 17,32,...] [--q 0,..] [--nb 2,1] [--out file]"""
import argparse, ctypes, json, os, sys
import torch
os.environ["EETQ_AMD_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import eetq_amd.ops as ops  # noqa: E402
from eetq_amd import _lib  # noqa: E402
from sweep import chain_us  # noqa: E402

SHAPES = [(4096, 4096), (4096, 6144), (4096, 11008), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120), (8192, 8192), (7168, 7168),
          (4096, 28672), (14336, 4096), (8192, 10240), (3584, 18944), (8192, 1024), (4096, 12288), (5120, 15360), (6144, 6144), (2048, 8192)]
MS = (17, 24, 32, 48, 64, 96, 128)
NAMES = {1: "gemv", 2: "mfma", 3: "stream", 4: "mid", 5: "splitk", 6: "tilesplit"}


def auto_path(M, N, K):
    p, d = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.lib().eetq_diag_auto_path(8, M, N, K, ctypes.byref(p), ctypes.byref(d)))
    return NAMES.get(p.value, str(p.value)) + ("/%d" % d.value if d.value else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--ms", default=",".join(str(m) for m in MS))
    ap.add_argument("--q", default="0")
    ap.add_argument("--nb", default="2")
    ap.add_argument("--plans", default="", help="K-slice plans to time next to the balanced ones: nb.s.ring[.r];...")
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
            T, tiles = -(-K // 256), -(-N // 64)
            row = {"K": K, "N": N, "M": M, "auto_path": auto_path(M, N, K), "tiles": tiles, "steps": T,
                   "auto": round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc), max(2 * L, 40), 0.012), 2)}
            row["splitk"] = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="splitk"), max(2 * L, 40), 0.012), 2)
            for plan in (p for p in a.plans.split(";") if p):
                os.environ["EETQ_AMD_SPLITK_PLAN"] = plan.replace(".", ",")
                try:
                    row["plan_" + plan] = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="splitk"), max(2 * L, 40), 0.012), 2)
                except RuntimeError as e:
                    row["plan_" + plan] = str(e)[-40:]
                os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
            for nb in (int(v) for v in a.nb.split(",")):
                for q in (int(v) for v in a.q.split(",")):
                    key = "%d,0,0,1,%d" % (nb, q)
                    os.environ["EETQ_AMD_SPLITK_PLAN"] = key
                    try:
                        y1 = ops.w8_a16_gemm(x, ws[0], sc, path="splitk")
                        y2 = ops.w8_a16_gemm(x, ws[0], sc, path="splitk")
                        ok = bool(((y1.float() - ref).abs() <= tol).all()) and bool(torch.equal(y1, y2))
                        t = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="splitk"), max(2 * L, 40), 0.012), 2)
                        row["bal_nb%d_q%d" % (nb, q)] = t if ok else "WRONG"
                        if not ok:
                            row["maxerr"] = float((y1.float() - ref).abs().max())
                    except RuntimeError as e:
                        row["bal_nb%d_q%d" % (nb, q)] = str(e)[-50:]
                    os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
            line = json.dumps(row)
            print(line, flush=True)
            if out:
                out.write(line + "\n"); out.flush()
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
