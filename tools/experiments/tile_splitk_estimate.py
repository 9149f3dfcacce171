import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from eetq_amd import ops
from sweep import chain_us
dev = "cuda:0"
for (M, K, N, path) in [(128, 1024, 16384, "mfma"), (128, 2048, 8192, "mfma"), (128, 4096, 4096, "mfma"), (128, 4096, 4096, "splitk"),
                        (96, 1024, 16384, "mfma"), (64, 1024, 16384, "mfma"), (128, 2752, 16384, "mfma"), (128, 11008, 4096, "splitk"),
                        (128, 1280, 20480, "mfma"), (128, 5120, 5120, "splitk")]:
    nbuf = max(2, (640 << 20) // (K * N))
    g = torch.Generator(device=dev); g.manual_seed(1)
    sets = [tuple(ops.quant_weights(((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half(), torch.int8, False)) for _ in range(nbuf)]
    x = torch.rand(M, K, device=dev, generator=g).half()
    t = chain_us(lambda i: ops.w8_a16_gemm(x, sets[i % nbuf][0], sets[i % nbuf][1], path=path), 2 * nbuf)
    print(json.dumps({"M": M, "K": K, "N": N, "path": path, "us": round(t, 2)}), flush=True)
    del sets; torch.cuda.empty_cache()
