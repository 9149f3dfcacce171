"""Where the prefill of the config-5 workload goes (13B shapes, prompt 1024, batch B): device time per kernel family under the torch
profiler, next to the wall time of the eager prefill call.  usage: prefill_timeline.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, transformers
from eetq_amd.utils import GraphDecoder, eet_accelerator
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = transformers.LlamaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                               num_key_value_heads=40, vocab_size=32000, max_position_embeddings=4096)
torch.manual_seed(0); torch.set_default_dtype(torch.float16)
with torch.device("cuda:0"):
    model = transformers.LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True, fused_residual=True)
prompt = torch.randint(0, 32000, (B, 1024), generator=torch.Generator().manual_seed(1)).cuda()
dec = GraphDecoder(model, B, 1024 + 58)


def prefill():   # what GraphDecoder.generate runs before its first replay
    return dec.prefill(prompt).logits[:, -1].argmax(-1)


with torch.no_grad():
    for _ in range(3):
        prefill()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        prefill()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        prefill()
        torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type == torch.autograd.DeviceType.CUDA]
tot = sum(e.device_time_total for e in ev)
print("batch %d prefill wall %.2f ms, device kernel time %.2f ms, %d launches" % (B, wall, tot / 1e3, sum(e.count for e in ev)))
for e in sorted(ev, key=lambda e: -e.device_time_total)[:28]:
    print("%9.1f us %5d x  %s" % (e.device_time_total, e.count, e.key[:120]))
