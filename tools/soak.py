"""Soak test of the in-launch hand-offs (write-through records + tickets): thousands of launches of the split-K GEMM and of
the one-launch decode attention on fixed inputs, interleaved with launches that dirty the caches, every result compared
bit for bit with the first.  A lost or early-read record shows up as a mismatch.  usage: python tools/soak.py [iterations]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eetq_amd.ops as ops  # noqa: E402

dev = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
torch.manual_seed(0)
t0 = time.time()

# ---- split-K GEMM, several shapes / slice counts (AUTO picks S from its plan) ----
bad = 0
cases = []
for M, K, N in ((64, 4096, 4096), (32, 4096, 4096), (128, 4096, 4096), (64, 11008, 4096), (48, 5120, 5120), (100, 2048, 1024)):
    w = torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev)
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    x = torch.randn(M, K, dtype=torch.float16, device=dev)
    cases.append((x, w, s, ops.w8_a16_gemm(x, w, s, path="splitk").clone()))
# the same tile on int4 weight tiles (W4A16, gemm_splitk_kernel<BITS = 4>): same slabs, same ticket arrays, interleaved
for M, K, N in ((64, 4096, 4096), (40, 5120, 5120), (128, 2048, 2048)):
    w4 = torch.randint(-128, 127, (K, N // 2), dtype=torch.int8, device=dev)      # packed nibbles: any byte is a valid pair
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    x = torch.randn(M, K, dtype=torch.float16, device=dev)
    cases.append((x, w4, s, ops.w8_a16_gemm(x, w4, s, path="splitk").clone()))
junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
streams = [torch.cuda.Stream() for _ in range(3)]
for it in range(iters):
    x, w, s, ref = cases[it % len(cases)]
    st = streams[it % 3] if it % 5 == 0 else torch.cuda.current_stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        y = ops.w8_a16_gemm(x, w, s, path="splitk")
    torch.cuda.current_stream().wait_stream(st)
    if it % 7 == 0:
        junk.add_(1)            # churn L2 / Infinity Cache between launches
    if not torch.equal(y, ref):
        bad += 1
print("split-K: %d launches, %d mismatches" % (iters, bad))

# ---- K slices of the tiled kernel (shares the split-K tickets): 4 and 2 slices, interleaved with split-K launches ----
bad_t, n_t = 0, max(1, iters // 4)
tcases = []
for M, K, N in ((128, 8192, 4096), (256, 11008, 4096), (100, 5120, 5120)):
    w = torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev)
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    x = torch.randn(M, K, dtype=torch.float16, device=dev)
    tcases.append((x, w, s, ops.w8_a16_gemm(x, w, s, path="tilesplit").clone()))
for it in range(n_t):
    x, w, s, ref = tcases[it % len(tcases)]
    st = streams[it % 3] if it % 4 == 0 else torch.cuda.current_stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        y = ops.w8_a16_gemm(x, w, s, path="tilesplit")
    torch.cuda.current_stream().wait_stream(st)
    if it % 3 == 0:
        xs, ws, ss, rs = cases[it % len(cases)]
        if not torch.equal(ops.w8_a16_gemm(xs, ws, ss, path="splitk"), rs):
            bad_t += 1
    if it % 7 == 0:
        junk.add_(1)
    if not torch.equal(y, ref):
        bad_t += 1
print("K-sliced tiled kernel: %d launches, %d mismatches" % (n_t, bad_t))
bad += bad_t

# ---- small-batch kernel with the activation rows in LDS (block copy, per-wave rings; W8A16 and W4A16): the ring slots are rewritten
# by LDS-DMA while earlier reads of the same slot have only just retired -- a read that saw a half-written slot shows up as a mismatch
bad_s, n_s = 0, max(1, iters // 2)
scases = []
for bits, M, K, N in ((8, 2, 4096, 11008), (8, 6, 4096, 11008), (8, 4, 4096, 4096), (8, 12, 4096, 4096), (8, 16, 5120, 13824), (8, 3, 11008, 4096),
                      (8, 5, 8192, 8192), (4, 3, 4096, 11008), (4, 7, 11008, 4096), (4, 16, 4096, 4096), (4, 4, 5120, 27648)):
    w = torch.randint(-128, 127, (K, N if bits == 8 else N // 2), dtype=torch.int8, device=dev)
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    x = torch.randn(M, K, dtype=torch.float16, device=dev)
    scases.append((x, w, s, ops.w8_a16_gemm(x, w, s).clone(), "auto"))
# round 6: the 32-row ring (17 <= M <= 32 on the explicit stream path: four LDS-DMAs per k tile and wave, two MFMA row tiles)
for M, K, N in ((17, 4096, 4096), (32, 4096, 11008), (24, 5120, 5120)):
    w = torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev)
    s = torch.rand(N, dtype=torch.float16, device=dev) * 0.01
    x = torch.randn(M, K, dtype=torch.float16, device=dev)
    scases.append((x, w, s, ops.w8_a16_gemm(x, w, s, path="stream").clone(), "stream"))
for it in range(n_s):
    x, w, s, ref, pth = scases[it % len(scases)]
    st = streams[it % 3] if it % 4 == 0 else torch.cuda.current_stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        y = ops.w8_a16_gemm(x, w, s, path=pth)
    torch.cuda.current_stream().wait_stream(st)
    if it % 7 == 0:
        junk.add_(1)
    if not torch.equal(y, ref):
        bad_s += 1
print("small-batch kernel (LDS forms): %d launches, %d mismatches" % (n_s, bad_s))
bad += bad_s

# ---- one-launch decode attention: fixed cache and token, the counter walks and is reset; tickets must stay zero ----
B, H, Hkv, D, S = 2, 40, 8, 128, 600
inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
fr = torch.einsum("i,j->ij", torch.arange(S + 8).float(), inv)
table = torch.cat([fr.cos(), fr.sin()], -1).half().to(dev)
kc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=dev)
vc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=dev)
qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, dtype=torch.float16, device=dev)
q = qkv[..., : H * D].unflatten(-1, (H, D))[:, 0]
k = qkv[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))[:, 0]
v = qkv[..., (H + Hkv) * D:].unflatten(-1, (Hkv, D))[:, 0]
pos = torch.tensor([500, 480], device=dev)
tickets = torch.zeros(B * H + 1, dtype=torch.int32, device=dev)
counter = torch.tensor(500, dtype=torch.int64, device=dev)
refs, bad2 = {}, 0
for it in range(iters):
    splits = (None, 3, 9, 16)[it % 4]
    counter.fill_(500)
    out = ops.rope_decode_attention(pos, q, k, v, table, kc, vc, tickets, slots=counter, splits=splits, kv_len=counter,
                                    kv_len_bias=1, advance=counter)
    if it % 7 == 0:
        junk.add_(1)
    key = splits
    if key not in refs:
        refs[key] = out.clone()
    elif not torch.equal(out, refs[key]):
        bad2 += 1
    if it % 97 == 0 and (int(counter.item()) != 501 or int(tickets.abs().sum().item()) != 0):
        bad2 += 1000
print("decode attention: %d launches, %d mismatches; tickets zero: %s" % (iters, bad2, int(tickets.abs().sum().item()) == 0))

# ---- the same launch on a long cache (round 6: chunks of more than 16 blocks take the kernel's second-batch trips), 13B head count
B, H, Hkv, S = 1, 40, 40, 2300
fr = torch.einsum("i,j->ij", torch.arange(S + 8).float(), inv)
table = torch.cat([fr.cos(), fr.sin()], -1).half().to(dev)
kc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=dev)
vc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=dev)
qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, dtype=torch.float16, device=dev)
q = qkv[..., : H * D].unflatten(-1, (H, D))[:, 0]
k = qkv[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))[:, 0]
v = qkv[..., (H + Hkv) * D:].unflatten(-1, (Hkv, D))[:, 0]
pos = torch.tensor([2200], device=dev)
tickets = torch.zeros(B * H + 1, dtype=torch.int32, device=dev)
counter = torch.tensor(2200, dtype=torch.int64, device=dev)
refs, bad3, n3 = {}, 0, max(1, iters // 2)
for it in range(n3):
    splits = (None, 4, 16)[it % 3]
    counter.fill_(2200)
    out = ops.rope_decode_attention(pos, q, k, v, table, kc, vc, tickets, slots=counter, splits=splits, kv_len=counter,
                                    kv_len_bias=1, advance=counter)
    if it % 7 == 0:
        junk.add_(1)
    if splits not in refs:
        refs[splits] = out.clone()
    elif not torch.equal(out, refs[splits]):
        bad3 += 1
    if it % 97 == 0 and (int(counter.item()) != 2201 or int(tickets.abs().sum().item()) != 0):
        bad3 += 1000
print("decode attention, long cache: %d launches, %d mismatches; tickets zero: %s" % (n3, bad3, int(tickets.abs().sum().item()) == 0))
bad2 += bad3
print("elapsed %.1f s" % (time.time() - t0))
sys.exit(1 if bad or bad2 else 0)
