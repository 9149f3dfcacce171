"""Driver for rocprofv3 --pmc passes (kernel-trace only): the split-K tile's K-slice plans against the balanced form on the SAME
decomposition (4096^2, M = 64: "2,1,33" vs q = 16 -- one whole tile per workgroup, no hand-over; "2,4,33" vs q = 4), 20 launches
each on rotating weights.  usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python <this>"""
import os, sys
import torch
os.environ["EETQ_AMD_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eetq_amd.ops as ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
K = N = 4096; M = 64; L = 24
ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev, generator=g) for _ in range(L)]
sc = torch.rand(N, dtype=torch.float16, device=dev, generator=g) * 0.01
x = torch.randn(M, K, dtype=torch.float16, device=dev, generator=g)
for plan in ("2,1,33,1", "2,0,0,1,16", "2,4,33,1", "2,0,0,1,4"):
    os.environ["EETQ_AMD_SPLITK_PLAN"] = plan
    for i in range(20):
        ops.w8_a16_gemm(x, ws[i % L], sc, path="splitk")
    torch.cuda.synchronize()
os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
