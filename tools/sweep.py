#!/usr/bin/env python
"""BASELINE.json configs[3]: Llama-2-7B shape sweep, (K,N) in {(4096,4096), (4096,11008), (11008,4096)} x M in
{1, 8, 64, 1024}.  Prints one JSON object per case: kernel-only microseconds (HIP start/stop events on each dispatch),
algorithmic GB/s and TFLOP/s, and tier-A parity against a torch fp32 matmul over the dequantised weight.
Usage: python tools/sweep.py [--out profiles/r01_sweep.json]"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from eetq_amd import _lib, ops  # noqa: E402


def kernel_us(run, n):
    """Per-call kernel time: a call may be more than one launch (the tiled GEMM splits ragged shapes into two), so the
    recorded per-launch durations are summed per call."""
    L = _lib.lib()
    cap = 4 * n
    _lib.check(L.eetq_prof_begin(cap))
    run()
    buf = (ctypes.c_float * cap)()
    cnt = ctypes.c_int(0)
    _lib.check(L.eetq_prof_end(buf, cap, ctypes.byref(cnt)))
    per_call = cnt.value // n
    assert per_call >= 1 and per_call * n == cnt.value, (cnt.value, n)
    us = np.array(buf[:cnt.value]).reshape(n, per_call).sum(axis=1)
    return float(np.median(us)), float(us.mean()), float(us.min())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--extra", action="store_true", help="also Llama-2-13B shapes")
    ap.add_argument("--ms", default="1,8,64,1024", help="comma-separated batch sizes M")
    args = ap.parse_args()
    dev = "cuda:0"
    shapes = [(4096, 4096), (4096, 11008), (11008, 4096)]
    if args.extra:
        shapes += [(5120, 5120), (5120, 13824), (13824, 5120)]
    results = []
    for K, N in shapes:
        nbuf = max(2, (640 << 20) // (K * N))
        g = torch.Generator(device=dev)
        g.manual_seed(K + N)
        sets = []
        for i in range(nbuf):
            w = ((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half()
            sets.append(tuple(ops.quant_weights(w, torch.int8, False)))
            if i == 0:
                raw0 = ops.quant_weights(w, torch.int8, True)[0]
            del w
        wdq = (raw0.float() * sets[0][1].float()[None, :]).half().float()   # fp16(q*s) exactly as the contract
        for M in [int(v) for v in args.ms.split(",")]:
            x = (torch.rand(M, K, device=dev, generator=g) - 0.25).half()
            y = torch.empty(M, N, dtype=torch.float16, device=dev)
            iters = 200 if M <= 64 else 60

            def run():
                for i in range(iters):
                    ops.w8_a16_gemm_(x, sets[i % nbuf][0], sets[i % nbuf][1], y, M, N, K)
            run()
            torch.cuda.synchronize()
            med, mean, mn = kernel_us(run, iters)
            ops.w8_a16_gemm_(x, sets[0][0], sets[0][1], y, M, N, K)
            ref = x.float() @ wdq
            err = (y.float() - ref).abs()
            ok = bool((err <= 1e-3 * ref.abs().max() + 2e-3 * ref.abs()).all())
            nbytes = K * N + 2 * M * K + 2 * N + 2 * M * N
            flops = 2.0 * M * N * K
            results.append({"K": K, "N": N, "M": M, "kernel_us_median": round(med, 2), "kernel_us_mean": round(mean, 2),
                            "kernel_us_min": round(mn, 2), "GBps": round(nbytes / med / 1e3, 1),
                            "TFLOPs": round(flops / med / 1e6, 2), "hbm_frac": round(nbytes / med / 1e3 / 8000, 4),
                            "mfma_frac": round(flops / med / 1e6 / 2500, 4), "tier_a_ok": ok,
                            "max_abs_err": float(err.max())})
            print(json.dumps(results[-1]), flush=True)
        del sets
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"what": "Llama-2 shape sweep, kernel-only time per dispatch, MI355X", "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
