"""Driver for rocprofv3 --pmc passes (kernel-trace only): the round-1 tile (path="mid") and the split-K tile forced to the SAME
decomposition (EETQ_AMD_SPLITK_PLAN=1,1,33,1), 20 launches each on rotating weights, plus the row-group plan AUTO takes at
M = 128 / 256 on 4096^2 (HBM traffic: every weight tile is pulled by r workgroups -- from L2, or again from HBM?).
usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python tools/experiments/pmc_mid_vs_splitk.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eetq_amd.ops as ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
for (K, N, M, mode) in ((4096, 6144, 24, "mid"), (4096, 6144, 24, "splitk11"), (4096, 4096, 128, "auto"), (4096, 4096, 256, "auto"),
                        (4096, 4096, 64, "auto")):
    L = 24
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev, generator=g) for _ in range(L)]
    sc = torch.rand(N, dtype=torch.float16, device=dev, generator=g) * 0.01
    x = torch.randn(M, K, dtype=torch.float16, device=dev, generator=g)
    if mode == "splitk11":
        os.environ["EETQ_AMD_SPLITK_PLAN"] = "1,1,33,1"
    for i in range(20):
        ops.w8_a16_gemm(x, ws[i % L], sc, path={"mid": "mid", "splitk11": "splitk", "auto": "auto"}[mode])
    os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
    torch.cuda.synchronize()
    del ws
