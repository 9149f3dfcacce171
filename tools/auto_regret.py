#!/usr/bin/env python
"""AUTO's regret on shapes its thresholds were NOT fitted on (round-4 verdict, weak 6).

For every (K, N) x M the script times, in ONE process and on the same rotating weight sets, EETQ_PATH_AUTO and every kernel
path that can run the point (graph-replayed chains, tools/sweep.py::chain_us, best of 3 regions) and prints one JSON line per
point: microseconds per path, the path AUTO took (eetq_diag_auto_path), the best forced path and
regret = t(auto) / t(best forced path) - 1.  AUTO runs one of the forced paths, so regret >= 0 up to timing noise (the same
kernel timed twice differs by ~1 %; box to box the same binary differs by 2-4 %: differences below that are not evidence).

Held-out shapes (K x N): Llama-3-8B 4096x6144 (fused q|k|v), 4096x28672 (gate|up), 14336x4096 (down); Llama-3-70B 8192x10240,
8192x57344, 28672x8192; Qwen2-7B 3584x18944; 7168^2.  `--shapes fitted` runs the 7B / 13B shapes the rules were read off.
usage: python tools/auto_regret.py [--shapes heldout|fitted|all] [--ms 1,2,...] [--out profiles/r05_auto_regret.jsonl]
The reference picks its CUTLASS tile with occupancy queries at run time (cutlass_heuristic.cc:123-206); this library with one
rule (abi.hip::auto_path_i8) -- this is the measurement of what the rule costs."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import eetq_amd.ops as ops  # noqa: E402
from eetq_amd import _lib  # noqa: E402
from sweep import chain_us  # noqa: E402

HELDOUT = [(4096, 6144), (4096, 28672), (14336, 4096), (8192, 10240), (8192, 57344), (28672, 8192), (3584, 18944), (7168, 7168)]
FITTED = [(4096, 4096), (4096, 11008), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120)]
MS = (1, 2, 4, 8, 12, 16, 17, 24, 32, 48, 64, 96, 128, 192, 256, 512, 1024)
PATH_NAMES = {1: "gemv", 2: "mfma", 3: "stream", 4: "mid", 5: "splitk", 6: "tilesplit"}


def auto_path(M, N, K):
    p, d = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.lib().eetq_diag_auto_path(8, M, N, K, ctypes.byref(p), ctypes.byref(d)))
    return PATH_NAMES.get(p.value, str(p.value)), d.value


def candidate_paths(M):
    """every explicit path that accepts this M (the launchers refuse the rest; `mid` is the round-1 tile, kept as a fallback)"""
    out = []
    if M <= 4:
        out.append("gemv")
    if M <= 64:
        out.append("stream")
    if 2 <= M <= 1024:
        out.append("splitk")
    if 17 <= M <= 128:
        out.append("mid")
    if M >= 17:
        out.append("mfma")
    if M >= 65:
        out.append("tilesplit")
    return out


def measure(K, N, M, ws, s, min_seconds):
    L = len(ws)
    x = torch.randn(M, K, dtype=torch.float16, device="cuda:0")
    row = {"K": K, "N": N, "M": M}
    name, detail = auto_path(M, N, K)
    row["auto_path"] = name + ("/S=%d" % detail if name == "tilesplit" else "/rows=%d" % detail if name == "splitk" and detail else "")
    calls = max(2 * L, 40 if M <= 256 else 8)
    for path in ["auto"] + candidate_paths(M):
        def step(i, path=path):
            ops.w8_a16_gemm(x, ws[i % L], s, path=path)
        try:
            row[path] = round(chain_us(step, calls, min_seconds=min_seconds), 2)
        except RuntimeError as e:
            row[path] = None
            row.setdefault("errors", {})[path] = str(e)[:60]
    forced = {p: t for p, t in row.items() if p in PATH_NAMES.values() and isinstance(t, float)}
    if forced and isinstance(row.get("auto"), float):
        best = min(forced, key=forced.get)
        row["best"] = best
        row["regret"] = round(row["auto"] / forced[best] - 1.0, 4)
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="heldout")
    ap.add_argument("--ms", default=",".join(str(m) for m in MS))
    ap.add_argument("--out", default=None)
    ap.add_argument("--min-ms", type=float, default=15.0)
    args = ap.parse_args()
    shapes = {"heldout": HELDOUT, "fitted": FITTED, "all": HELDOUT + FITTED}.get(args.shapes)
    if shapes is None:
        shapes = [tuple(int(v) for v in a.split("x")) for a in args.shapes.split(",")]
    ms = [int(m) for m in args.ms.split(",")]
    out = open(args.out, "w") if args.out else None
    rows = []
    for K, N in shapes:
        L = max(2, int(640e6 // (K * N)))
        ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0") for _ in range(L)]
        s = torch.rand(N, dtype=torch.float16, device="cuda:0") * 0.01
        for M in ms:
            row = measure(K, N, M, ws, s, args.min_ms * 1e-3)
            rows.append(row)
            line = json.dumps(row)
            print(line, flush=True)
            if out:
                out.write(line + "\n")
                out.flush()
        del ws
        torch.cuda.empty_cache()
    reg = [r for r in rows if "regret" in r]
    worst = sorted(reg, key=lambda r: -r["regret"])[:10]
    summary = {"summary": True, "points": len(reg), "regret_gt_5pct": sum(r["regret"] > 0.05 for r in reg),
               "regret_gt_2pct": sum(r["regret"] > 0.02 for r in reg),
               "mean_regret": round(sum(max(r["regret"], 0.0) for r in reg) / max(len(reg), 1), 4),
               "worst": [{k: r[k] for k in ("K", "N", "M", "auto_path", "best", "regret")} for r in worst]}
    print(json.dumps(summary), flush=True)
    if out:
        out.write(json.dumps(summary) + "\n")
        out.close()


if __name__ == "__main__":
    main()
