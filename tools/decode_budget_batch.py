import sys, os, json, ctypes, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, transformers
from eetq_amd.utils import GraphDecoder, eet_accelerator
from eetq_amd import _lib
B = int(sys.argv[1])
cfg = transformers.LlamaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                               num_key_value_heads=40, vocab_size=32000, max_position_embeddings=4096)
torch.manual_seed(0); torch.set_default_dtype(torch.float16)
with torch.device("cuda:0"):
    model = transformers.LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True, fused_residual=True)
prompt = torch.randint(0, 32000, (B, 1024), generator=torch.Generator().manual_seed(1)).cuda()
dec = GraphDecoder(model, B, 1024 + 58)
L = _lib.lib()
with torch.no_grad():
    dec.generate(prompt, 8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dec.graph.replay()
    torch.cuda.synchronize(); step = (time.perf_counter() - t0) / 20 * 1e6
    cap = 40 * 10
    _lib.check(L.eetq_prof_begin(cap)); dec._step()
    buf = (ctypes.c_float * cap)(); cnt = ctypes.c_int(0)
    _lib.check(L.eetq_prof_end(buf, cap, ctypes.byref(cnt)))
us = np.array(buf[:cnt.value]); per = cnt.value // 40
body = us[:40 * per].reshape(40, per)
print("batch", B, "step_us", round(step, 1), "library launches per layer", per, "per-launch means", [round(float(v), 2) for v in body.mean(0)], "sum/token", round(float(body.sum()), 1))
from torch.profiler import profile, ProfilerActivity
with torch.no_grad():
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        dec._step()
        torch.cuda.synchronize()
rows = [(e.key, e.device_time_total, e.count) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1])
for k, t, c in rows[:16]:
    print("%9.1f us %5d x  %s" % (t, c, k[:110]))
