"""Prompt attention (eetq_prefill_attention_f16) against a float32 softmax(q k^T) v of the same inputs (max abs error per case) and,
chain-timed, against torch's scaled_dot_product_attention at Llama-2-13B shapes (40 heads x 128, 1 024 tokens), batch 1 and 4.
usage: python tools/prefill_attn_bench.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import eetq_amd.ops as ops
from sweep import chain_us
dev = "cuda:0"
torch.manual_seed(0)
def run(B, T, H, Hkv, S, D=128, base=0):
    qkv = torch.randn(B, T, (H + 2 * Hkv) * D, dtype=torch.float16, device=dev)
    q = qkv[..., :H * D].unflatten(-1, (H, D))
    kc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=dev)
    vc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=dev)
    keys = base + T
    out = ops.prefill_attention(q, kc, vc, keys)
    qq = q.transpose(1, 2).float()
    k, v = kc[:, :, :keys].float(), vc[:, :, :keys].float()
    if Hkv != H:
        k, v = k.repeat_interleave(H // Hkv, 1), v.repeat_interleave(H // Hkv, 1)
    sc = torch.matmul(qq, k.transpose(2, 3)) / D ** 0.5
    mask = torch.arange(keys, device=dev)[None, :] > (torch.arange(T, device=dev)[:, None] + base)
    sc = sc.masked_fill(mask, float("-inf"))
    ref = torch.matmul(torch.softmax(sc, -1), v).transpose(1, 2)
    err = (out.float() - ref).abs().max().item()
    print("B %d T %d H %d Hkv %d S %d base %d: max abs err %.5f (ref max %.3f) nan %s" % (B, T, H, Hkv, S, base, err, ref.abs().max().item(), bool(out.isnan().any())), flush=True)
    return q, kc, vc, keys
run(1, 128, 2, 2, 160)
run(1, 200, 4, 2, 256)
run(2, 130, 4, 4, 300, base=37)
run(1, 64, 2, 1, 64)
q, kc, vc, keys = run(1, 1024, 40, 40, 1082)
def f(i): return ops.prefill_attention(q, kc, vc, keys)
print("B1 T1024 H40: %.1f us" % chain_us(f, 10))
qt = q.transpose(1, 2)
def g(i): return torch.nn.functional.scaled_dot_product_attention(qt, kc[:, :, :keys], vc[:, :, :keys], is_causal=True)
print("torch sdpa:   %.1f us" % chain_us(g, 10))
q, kc, vc, keys = run(4, 1024, 40, 40, 1082)
print("B4 T1024 H40: %.1f us" % chain_us(f, 10))
