"""Per-kernel breakdown of ONE prefill of the config-5 model (Llama-2-13B shapes, 1024 tokens) with torch.profiler.
usage: python tools/prefill_breakdown.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, transformers
from eetq_amd.utils import eet_accelerator
dev = "cuda:0"
cfg = transformers.LlamaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                               num_key_value_heads=40, vocab_size=32000, max_position_embeddings=4096)
torch.manual_seed(0)
torch.set_default_dtype(torch.float16)
with torch.device(dev):
    model = transformers.LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True, fused_residual=True)
prompt = torch.randint(0, 32000, (1, 1024), generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    for _ in range(2):
        model(prompt)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); model(prompt); b.record(); torch.cuda.synchronize()
    print("prefill wall (events): %.2f ms" % a.elapsed_time(b))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        model(prompt)
        torch.cuda.synchronize()
rows = [(e.key, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.count) for e in prof.key_averages()]
rows = [r for r in rows if r[1] > 0]
rows.sort(key=lambda r: -r[1])
tot = sum(r[1] for r in rows if not r[0].startswith("aten::") and not r[0].startswith("eetq") or True)
seen = 0
for k, t, c in rows[:28]:
    print("%9.2f ms %5d x  %s" % (t / 1e3, c, k[:100]))
