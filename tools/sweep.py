#!/usr/bin/env python
"""BASELINE.json configs[3]: Llama-2-7B shape sweep, (K,N) in {(4096,4096), (4096,11008), (11008,4096)} x M in
{1, 8, 64, 1024}.  Prints one JSON object per case: microseconds per call of a HIP-graph-replayed chain of back-to-back
calls (rotating weight sets; the kernel's begin->end plus the <= 0.1 us inter-dispatch gap -- NOT start/stop event pairs,
whose own floor of ~4.2 us hid every kernel shorter than that in the round-1/2 sweeps), algorithmic GB/s and TFLOP/s, and
tier-A parity against a torch fp32 matmul over the dequantised weight.
Usage: python tools/sweep.py [--out profiles/r01_sweep.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from eetq_amd import ops  # noqa: E402


def chain_us(step, calls_per_graph, min_seconds=0.03):
    """Microseconds per call: `calls_per_graph` back-to-back calls captured as ONE HIP graph (step(i) issues call i),
    replayed until >= min_seconds have been timed; best of 3 timed regions."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            step(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # the warm-up stream took a split-K scratch region if `step` runs a K-sliced kernel: hand it back (16 regions per device, never
    # reclaimed by the library on its own; the captured launches below run on torch's one capture stream, which keeps its region)
    ops.release_stream_workspace(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(calls_per_graph):
            step(i)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    reps = max(2, int(min_seconds / max(time.perf_counter() - t0, 1e-6)))
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (reps * calls_per_graph))
    return best * 1e6


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
            calls = (4 if M <= 128 else 1) * nbuf     # whole passes over the rotating weight sets

            def step(i):
                ops.w8_a16_gemm_(x, sets[i % nbuf][0], sets[i % nbuf][1], y, M, N, K)
            med = chain_us(step, calls)
            ops.w8_a16_gemm_(x, sets[0][0], sets[0][1], y, M, N, K)
            ref = x.float() @ wdq
            err = (y.float() - ref).abs()
            ok = bool((err <= 1e-3 * ref.abs().max() + 2e-3 * ref.abs()).all())
            nbytes = K * N + 2 * M * K + 2 * N + 2 * M * N
            flops = 2.0 * M * N * K
            results.append({"K": K, "N": N, "M": M, "us_per_call": round(med, 2), "GBps": round(nbytes / med / 1e3, 1),
                            "TFLOPs": round(flops / med / 1e6, 2), "hbm_frac": round(nbytes / med / 1e3 / 8000, 4),
                            "mfma_frac": round(flops / med / 1e6 / 2500, 4), "tier_a_ok": ok,
                            "max_abs_err": float(err.max())})
            print(json.dumps(results[-1]), flush=True)
        del sets
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"what": "Llama-2 shape sweep, microseconds per call of a graph-replayed chain of back-to-back calls "
                               "(rotating weight sets), MI355X", "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
