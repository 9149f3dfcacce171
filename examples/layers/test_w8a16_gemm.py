"""The reference's operator check (examples/layers/test_w8a16_gemm.py:21-61) on this library's EETQ module: M = 1, N = 13824,
K = 5120 (a Llama-2-13B projection), weights ~ U[0, 1) on the CPU.  quant_weights(w, int8, True) -> (raw, processed, scales);
preprocess_weights(raw) must equal `processed`; both feed w8_a16_gemm; the result is compared with the fp16 matmul on the
dequantised weight and timed over 500 calls like the reference does.  Exit status 1 when a check fails.
usage: python examples/layers/test_w8a16_gemm.py"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from EETQ import preprocess_weights, quant_weights, w8_a16_gemm  # noqa: E402  (the reference's import line)


def set_random_seed(seed):
    random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


if __name__ == "__main__":
    M, N, K = 1, 13824, 5120
    set_random_seed(1)
    x = torch.rand(M, K, dtype=torch.float16).cuda()
    w_cpu = torch.rand(K, N, dtype=torch.float16)
    raw, processed, scales = quant_weights(w_cpu, torch.int8, True)            # CPU in, CPU out, like the reference
    assert raw.device.type == "cpu" and processed.shape == (K, N) and scales.shape == (N,)
    out1 = w8_a16_gemm(x, processed.cuda(), scales.cuda())
    processed_again = preprocess_weights(raw)
    same_bytes = torch.equal(processed_again, processed)
    out2 = w8_a16_gemm(x, processed_again.cuda(), scales.cuda())
    print("preprocess_weights(raw) == processed:", same_bytes, "| out1 == out2:", torch.equal(out1, out2))
    w_deq = (raw.to(torch.float16).cuda() * scales.cuda())                     # the reference's "ref_torch_weights fp16"
    out_deq = torch.matmul(x.float(), w_deq.float())
    out_torch = torch.matmul(x, w_cpu.cuda())
    err_deq = (out1.float() - out_deq).abs().max().item()
    err_fp16 = (out1.float() - out_torch.float()).abs().max().item()
    print("max |err| vs matmul on the dequantised weight: %.3e (|y| up to %.1f); vs fp16 matmul on the original weight: %.3e"
          % (err_deq, out_deq.abs().max().item(), err_fp16))
    ok = same_bytes and torch.equal(out1, out2) and err_deq <= 1e-3 * out_deq.abs().max().item() + 2e-3 * out_deq.abs().max().item()
    pw, sc = processed.cuda(), scales.cuda()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(500):
        out = w8_a16_gemm(x, pw, sc)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("time per call (eager loop, host launch included): %.2f us" % ((t2 - t1) / 500 * 1e6))
    sys.exit(0 if ok else 1)
