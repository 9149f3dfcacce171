"""int4 M = 1 GEMV: several tile rows per workgroup (gemv_rows_kernel) against the whole-tile-row kernels, two processes
(EETQ_AMD_I4_ROWS is read once), graph-replayed chains over rotating weights, us per call; W8A16 on the same shape beside it.
RESULT (profiles/r04_i4_rows.txt): exact, but no general gain -- 5120x13824 11.58 -> 11.59 us, 5120x27648 19.7 -> 19.4,
4096x11008 7.94 -> 9.61 (worse: two k tiles per wave and row), only 8192x28672 33.8 -> 28.1.  The int4 GEMV is bound by VALU
issue + per-wave latency, not by the per-workgroup overheads this form removes (profiles/r04_int4_gemv_pmc.txt: VALU 60 % busy,
2.4 wave-instructions per 16 weight bytes).  SHELVED: the kernel is tools/experiments/i4_rows_kernel.patch (apply to
eetq_amd/csrc/gemv_kernel.hpp + gemv.hip to re-run)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from eetq_amd import ops
    from sweep import chain_us
    dev = "cuda:0"
    out = {}
    for K, N in [(4096, 11008), (5120, 13824), (5120, 15360), (5120, 27648), (4096, 22016), (8192, 28672)]:
        nbuf = max(4, (700 << 20) // (K * N))
        w4 = [torch.randint(-128, 127, (K, N // 2), dtype=torch.int8, device=dev) for _ in range(nbuf)]
        w8 = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev) for _ in range(nbuf)]
        s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
        x = torch.randn(1, K, dtype=torch.float16, device=dev)
        out["%dx%d" % (K, N)] = [round(chain_us(lambda i: ops.w8_a16_gemm(x, w4[i % nbuf], s), 2 * nbuf), 2),
                                 round(chain_us(lambda i: ops.w8_a16_gemm(x, w8[i % nbuf], s), 2 * nbuf), 2)]
        del w4, w8
    print(json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for rows in ("0", "1"):
        env = dict(os.environ, EETQ_AMD_I4_ROWS=rows)
        r = subprocess.run([sys.executable, __file__, "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("rows_kernel=%s [w4, w8] us" % rows, line[-1] if line else "fail", flush=True)
