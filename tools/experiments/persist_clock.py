"""Round 5: why the persistent form of the wide tile buys nothing -- time, shader cycles and the clock the chip held, for the
persistent form and one workgroup per tile (child process: EETQ_AMD_TUNING=1 EETQ_AMD_TILE_PERSIST=0), on BASELINE-like random
operands and on all-zero operands (same instruction stream, nothing toggles).  bench.stamped_chain: a graph of dependent launches
between two clock-stamp launches (s_memtime / s_memrealtime per XCD).  usage: python tools/experiments/persist_clock.py"""
import json, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import eetq_amd.ops as ops  # noqa: E402
from eetq_amd.utils.replicas import ReplicaGroup  # noqa: E402


def run():
    grp = ReplicaGroup()
    dev = grp.device
    out = {}
    for (K, N, M) in ((4096, 4096, 4096), (4096, 11008, 1024)):
        g = torch.Generator(device=dev); g.manual_seed(7)
        L = 8
        ws = [ops.quant_weights(((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half(), torch.int8, False) for _ in range(L)]
        x = torch.rand(M, K, device=dev, generator=g).half()
        y = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(2)]
        zw = torch.full((K, N), -128, dtype=torch.int8, device=dev)
        zs = torch.ones(N, dtype=torch.float16, device=dev)
        zx = torch.zeros(M, K, dtype=torch.float16, device=dev)

        def rnd(first, count):
            for i in range(first, first + count):
                ops.w8_a16_gemm_(x, ws[i % L][0], ws[i % L][1], y[i % 2], M, N, K)

        def zero(first, count):
            for i in range(first, first + count):
                ops.w8_a16_gemm_(zx, zw, zs, y[i % 2], M, N, K)
        for name, fn in (("random", rnd), ("zeros", zero)):
            fn(0, 20)
            torch.cuda.synchronize()
            us, mhz, nx = bench.stamped_chain(grp, fn, 100, 0.05)
            out["%dx%d_M%d_%s" % (K, N, M, name)] = {"us": round(us, 2), "mhz": round(mhz) if mhz else None,
                                                    "kcycles": round(us * mhz / 1e3, 1) if mhz else None}
    return out


if __name__ == "__main__":
    if "--child" in sys.argv:
        print(json.dumps(run()))
        sys.exit(0)
    mine = run()
    env = dict(os.environ, EETQ_AMD_TUNING="1", EETQ_AMD_TILE_PERSIST="0")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    other = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else {}
    for k in mine:
        print(json.dumps({"case": k, "persistent": mine[k], "one_tile_per_workgroup": other.get(k)}))
