"""Decode-step attention at Llama-2-13B shapes (40 heads x 128, one new token), as the graph decoder issues it: `layers`
independent caches (together larger than the 256 MB Infinity Cache) stepped back to back inside one HIP graph.
Prints microseconds per layer-step for the one-launch form (eetq_rope_decode_attention_f16) over a sweep of chunk counts,
and for the two-launch pair.  usage: python tools/attn_bench.py [--filled 1024] [--rows 1082] [--batch 1]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eetq_amd.ops as ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--filled", type=int, default=1024)
ap.add_argument("--rows", type=int, default=1082)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--heads", type=int, default=40)
ap.add_argument("--kv-heads", type=int, default=40)
ap.add_argument("--layers", type=int, default=20)
ap.add_argument("--splits", default="default,4,8,12,17,25,34")
ap.add_argument("--stamps", action="store_true", help="per-phase device-clock decomposition of the one-launch form")
ap.add_argument("--stamps-fused", action="store_true", help="device-clock decomposition of the fused attention + o projection launch")
ap.add_argument("--pair", action="store_true", help="attention + the o projection behind it (H*D x H*D int8), with and without "
                                                    "the L2 prefetch of the projection's weight inside the attention launch")
args = ap.parse_args()
dev = "cuda:0"
B, H, Hkv, D, S, L = args.batch, args.heads, args.kv_heads, 128, args.rows, args.layers
torch.manual_seed(0)
inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
fr = torch.einsum("i,j->ij", torch.arange(S + 64).float(), inv)
table = torch.cat([fr.cos(), fr.sin()], -1).half().to(dev)
kc = [torch.randn(B, Hkv, S, D, dtype=torch.float16, device=dev) for _ in range(L)]
vc = [torch.randn(B, Hkv, S, D, dtype=torch.float16, device=dev) for _ in range(L)]
qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, dtype=torch.float16, device=dev)
q = qkv[..., : H * D].unflatten(-1, (H, D))[:, 0]
k = qkv[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))[:, 0]
v = qkv[..., (H + Hkv) * D:].unflatten(-1, (Hkv, D))[:, 0]
pos = torch.full((B,), args.filled, dtype=torch.int64, device=dev)
tickets = [torch.zeros(B * H + 1, dtype=torch.int32, device=dev) for _ in range(L)]
counters = [torch.tensor(args.filled, dtype=torch.int64, device=dev) for _ in range(L)]
kv_mb = 2 * B * Hkv * (args.filled + 1) * D * 2 / 1e6


def one_launch(i, splits):
    return ops.rope_decode_attention(pos, q, k, v, table, kc[i], vc[i], tickets[i], slots=counters[i], splits=splits,
                                     kv_len=counters[i], kv_len_bias=1)


def two_launch(i, splits):
    ops.rotary_embedding_neox_kvcache(pos, q, k, v, D, table, kc[i], vc[i], slots=counters[i])
    return ops.decode_attention(q, kc[i], vc[i], splits=splits, kv_len=counters[i], kv_len_bias=1)


def timed(fn, splits, reps=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(L):
            fn(i, splits)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(L):
                fn(i, splits)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * L)


if args.pair:
    C = H * D
    gw = torch.Generator(device=dev).manual_seed(3)
    ow = [torch.randint(-128, 127, (C, C), dtype=torch.int8, device=dev, generator=gw) for _ in range(L)]
    osc = [(torch.rand(C, dtype=torch.float16, device=dev, generator=gw) * 0.01 + 0.001) for _ in range(L)]
    res = torch.zeros(B, C, dtype=torch.float16, device=dev)

    def pair(i, hint):
        kw = {"next_weight": ow[i]} if hint else {}   # (the hint exists only with tools/experiments/attn_l2_prefetch.patch applied)
        att = ops.rope_decode_attention(pos, q, k, v, table, kc[i], vc[i], tickets[i], slots=counters[i], kv_len=counters[i],
                                        kv_len_bias=1, **kw)
        return ops.w8_a16_gemm(att.view(B, C), ow[i], osc[i], residual=res)

    def attn_only(i, _):
        return ops.rope_decode_attention(pos, q, k, v, table, kc[i], vc[i], tickets[i], slots=counters[i], kv_len=counters[i], kv_len_bias=1)

    def attn_hint_only(i, _):
        return ops.rope_decode_attention(pos, q, k, v, table, kc[i], vc[i], tickets[i], slots=counters[i], kv_len=counters[i], kv_len_bias=1,
                                         next_weight=ow[i])

    try:
        attn_hint_only(0, None)
        have_hint = True
    except TypeError:
        have_hint = False

    def proj_only(i, _):
        return ops.w8_a16_gemm(res, ow[i], osc[i], residual=res)

    tickets2 = [torch.zeros(ops.rope_decode_attention_oproj_tickets(H) if getattr(ops, 'rope_decode_attention_oproj_tickets', None) else B * H + 2, dtype=torch.int32, device=dev) for _ in range(L)]
    have_fused = (getattr(ops, "rope_decode_attention_oproj", None) is not None and B == 1
                  and ops.rope_decode_attention_oproj_supported(1, H, D, C, False))

    def fused(i, _):
        return ops.rope_decode_attention_oproj(pos, q, k, v, table, kc[i], vc[i], tickets2[i], ow[i], osc[i], None, res.view(-1),
                                               slots=counters[i], kv_len=counters[i], kv_len_bias=1)

    y0 = pair(0, False).clone()
    same = bool(torch.equal(y0, pair(0, True).clone())) if have_hint else None
    out = {"form": "attention + o projection", "batch": B, "heads": H, "filled": args.filled, "rows": S, "o_proj": "%dx%d" % (C, C),
           "same_bits_with_hint": same}
    if have_fused:
        yf = fused(0, None).clone().view(B, C)
        torch.cuda.synchronize()
        err = (yf.float() - y0.float()).abs().max().item()
        out["fused_vs_pair_max_abs"] = err
        out["fused_vs_pair_tier_a"] = bool(((yf.float() - y0.float()).abs() <= 1e-3 * y0.float().abs().max() + 2e-3 * y0.float().abs()).all())
        out["fused_timeouts"] = ops.attn_oproj_timeouts()
    for rep in range(2):
        out["attention_us"] = round(timed(attn_only, None), 2)
        if have_hint:
            out["attention_with_prefetch_us"] = round(timed(attn_hint_only, None), 2)
        out["o_proj_us"] = round(timed(proj_only, None), 2)
        out["pair_us"] = round(timed(pair, False), 2)
        if have_hint:
            out["pair_with_prefetch_us"] = round(timed(pair, True), 2)
        if have_fused:
            out["fused_us"] = round(timed(fused, None), 2)
            out["fused_timeouts_after"] = ops.attn_oproj_timeouts()
        print(json.dumps(out), flush=True)
    sys.exit(0)

if args.stamps_fused:
    import ctypes
    from eetq_amd import _lib
    lib = _lib.lib()
    C = H * D
    gw = torch.Generator(device=dev).manual_seed(3)
    ow = [torch.randint(-128, 127, (C, C), dtype=torch.int8, device=dev, generator=gw) for _ in range(L)]
    osc = [(torch.rand(C, dtype=torch.float16, device=dev, generator=gw) * 0.01 + 0.001) for _ in range(L)]
    res = torch.zeros(B, C, dtype=torch.float16, device=dev)
    tk = [torch.zeros(ops.rope_decode_attention_oproj_tickets(H), dtype=torch.int32, device=dev) for _ in range(L)]
    splits = lib.eetq_decode_attention_splits(B, H, S)
    A, R = H * splits, C // 16
    buf = torch.zeros((A + R) * 8, dtype=torch.int64, device=dev)

    def fused(i):
        return ops.rope_decode_attention_oproj(pos, q, k, v, table, kc[i], vc[i], tk[i], ow[i], osc[i], None, res.view(-1),
                                               slots=counters[i], kv_len=counters[i], kv_len_bias=1)

    lib.eetq_diag_attn_stamps(ctypes.c_void_p(buf.data_ptr()))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(L):
            fused(i)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(L):
                fused(i)
    torch.cuda.current_stream().wait_stream(side)
    lib.eetq_diag_attn_stamps(None)
    rows = []
    for it in range(12):
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        st = buf.view(A + R, 8).cpu().double()
        t0 = st[:, 0][st[:, 0] > 0].min()
        rel = (st - t0) / 100.0
        rel[st == 0] = float("nan")
        rows.append(rel)
    allr = torch.stack(rows[2:])
    an = ["entry", "scalar reads", "q rotated + first trip landed", "chunk done", "record published", "ticket drawn",
          "merge done (last of head)", "output stored (last of head)"]
    pn = ["entry", "weights requested", "weights landed (wave 0)", "flag seen", "dot products done", "output stored"]
    print("fused attention + o projection, %d attention + %d projection workgroups; us since the first workgroup entered (mean / min / max)" % (A, R))
    for title, sl, names in (("attention workgroups", slice(0, A), an), ("projection workgroups", slice(A, A + R), pn)):
        print(title + ":")
        for i, n in enumerate(names):
            col = allr[:, sl, i]
            col = col[~col.isnan()]
            if col.numel():
                print("  %-34s mean %6.2f   min %6.2f   max %6.2f   (n=%d)" % (n, col.mean(), col.min(), col.max(), col.numel()))
    print("launch span (first entry -> last projection output stored): %.2f us" %
          torch.nan_to_num(allr[:, A:, 5], nan=0.0).amax(1).mean())
    sys.exit(0)

if args.stamps:
    import ctypes
    from eetq_amd import _lib
    lib = _lib.lib()
    splits = lib.eetq_decode_attention_splits(B, H, S)
    if args.splits.split(',')[0].isdigit():
        splits = int(args.splits.split(',')[0])
    nwg = B * H * splits
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    names = ["entry", "scalar reads", "q rotated + first trip landed", "chunk done", "record published", "ticket drawn",
             "merge done (last of head)", "output stored (last of head)"]
    rows = []
    # the stamped launch is the LAST of a graph-replayed chain of L launches (one per cache): warm clocks, warm TLBs, the
    # previous launch's tail in front of it -- the conditions the chain timing below measures (every launch of the graph
    # writes the same stamp rows; the last one's survive)
    lib.eetq_diag_attn_stamps(ctypes.c_void_p(buf.data_ptr()))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(L):
            one_launch(i, splits)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(L):
                one_launch(i, splits)
    torch.cuda.current_stream().wait_stream(side)
    lib.eetq_diag_attn_stamps(None)
    for it in range(12):
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        st = buf.view(nwg, 8).cpu().double()
        t0 = st[:, 0].min()
        rel = (st - t0) / 100.0                    # microseconds since the first workgroup's entry
        rel[st == 0] = float("nan")
        rows.append(rel)
    allr = torch.stack(rows[2:])                   # [launch, workgroup, stamp]
    print("one-launch decode attention, %d workgroups (%d splits), device clock, us since the first workgroup entered" % (nwg, splits))
    print("(mean over %d launches and the workgroups that reach the stamp):" % allr.shape[0])
    for i, n in enumerate(names):
        col = allr[:, :, i]
        col = col[~col.isnan()]
        print("  %-34s mean %6.2f   min %6.2f   max %6.2f   (n=%d)" % (n, col.mean(), col.min(), col.max(), col.numel()))
    d = allr[:, :, 1:] - allr[:, :, :-1]           # per launch and workgroup: no mixing of launches
    print("phase lengths (same launch, same workgroup):")
    for i in range(7):
        col = d[:, :, i]
        col = col[~col.isnan()]
        if col.numel():
            print("  %-34s -> %-34s mean %6.2f  min %6.2f  max %6.2f us" % (names[i], names[i + 1], col.mean(), col.min(), col.max()))
    last = allr[:, :, 7]
    print("kernel span (first entry -> last output stored), mean over launches: %.2f us" % torch.nan_to_num(last, nan=0.0).amax(1).mean())
    sys.exit(0)

for name, fn in (("one_launch", one_launch), ("two_launch", two_launch)):
    for s in args.splits.split(","):
        splits = None if s == "default" else int(s)
        us = timed(fn, splits)
        print(json.dumps({"form": name, "splits": s, "batch": B, "heads": H, "kv_heads": Hkv, "filled": args.filled,
                          "rows": S, "us_per_layer_step": round(us, 2), "kv_MB": round(kv_mb, 1),
                          "kv_GBps": round(kv_mb / us * 1e3, 0)}))
