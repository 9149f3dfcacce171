"""Round 5: the persistent form of the wide tile (gemm_tile_persistent_kernel) against one workgroup per tile.
For every (K, N, M): the persistent launch must give the SAME BITS as the one-tile-per-workgroup kernel run on row chunks small
enough to have at most one tile per CU (a tile's arithmetic does not depend on the launch that ran it), with and without the fused
bias + residual epilogue; then both forms are timed as graph-replayed chains (the non-persistent one in a child process:
EETQ_AMD_TUNING=1 EETQ_AMD_TILE_PERSIST=0).  One JSON line per case.
usage: python tools/experiments/persist_check.py [--out file]"""
import json, os, subprocess, sys, zlib
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import eetq_amd.ops as ops  # noqa: E402
from sweep import chain_us  # noqa: E402

CASES = [(4096, 4096, 2048), (4096, 4096, 3000), (4096, 4096, 4096), (4096, 4096, 8192), (4096, 11008, 1024), (4096, 11008, 2049),
         (11008, 4096, 4096), (5120, 5120, 2048), (5120, 13824, 1024), (13824, 5120, 4096), (5120, 15360, 4096), (1024, 4096, 4096)]


def crc(t):
    return zlib.crc32(t.cpu().contiguous().view(torch.uint8).numpy().tobytes()) & 0xFFFFFFFF


def run(child):
    out = {}
    for K, N, M in CASES:
        g = torch.Generator(device="cuda:0"); g.manual_seed(K + N + M)
        L = max(2, int(400e6 // (K * N)))
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0", generator=g) for _ in range(L)]
        sc = torch.rand(N, dtype=torch.float16, device="cuda:0", generator=g) * 0.01
        x = torch.randn(M, K, dtype=torch.float16, device="cuda:0", generator=g)
        bias = torch.randn(N, dtype=torch.float16, device="cuda:0", generator=g)
        res = torch.randn(M, N, dtype=torch.float16, device="cuda:0", generator=g)
        y = ops.w8_a16_gemm(x, ws[0], sc, path="mfma")
        yf = ops.w8_a16_gemm(x, ws[0], sc, path="mfma", bias=bias, residual=res)
        row = {"K": K, "N": N, "M": M, "crc": crc(y), "crc_fused": crc(yf)}
        if not child:
            # reference: row chunks with <= 256 wide tiles each (the one-tile-per-workgroup kernel), same bits expected
            rows_per = max(128, (256 // -(-N // 128)) * 128)
            ref = torch.cat([ops.w8_a16_gemm(x[m:m + rows_per], ws[0], sc, path="mfma") for m in range(0, M, rows_per)])
            row["bit_identical_to_chunks"] = bool(torch.equal(y, ref))
            row["fused_equals_separate_adds"] = bool(torch.equal(yf, (y + bias) + res))
            if not row["bit_identical_to_chunks"]:
                d = (y.float() - ref.float()).abs()
                row["max_abs_diff"] = float(d.max()); row["wrong_elements"] = int((d > 0).sum())
                bad = (d > 0).nonzero()[:6].tolist(); row["first_wrong"] = bad
        calls = max(2 * L, 8)
        row["us"] = round(chain_us(lambda i: ops.w8_a16_gemm(x, ws[i % L], sc, path="mfma"), calls, 0.03), 2)
        out["%d_%d_%d" % (K, N, M)] = row
        del ws, x, res
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    if "--child" in sys.argv:
        print(json.dumps(run(True)))
        sys.exit(0)
    dst = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    mine = run(False)
    env = dict(os.environ, EETQ_AMD_TUNING="1", EETQ_AMD_TILE_PERSIST="0")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    other = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else {}
    f = open(dst, "w") if dst else None
    for key, row in mine.items():
        o = other.get(key, {})
        row["one_tile_per_wg_us"] = o.get("us")
        row["same_bits_as_one_tile_per_wg"] = (o.get("crc") == row["crc"] and o.get("crc_fused") == row["crc_fused"]) if o else None
        if o.get("us"):
            row["speedup"] = round(o["us"] / row["us"], 4)
            row["tflops"] = round(2.0 * row["M"] * row["N"] * row["K"] / row["us"] / 1e6, 1)
        line = json.dumps(row)
        print(line, flush=True)
        if f:
            f.write(line + "\n")
    if f:
        f.close()
