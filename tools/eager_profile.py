"""Where the host time of an eager decode goes: cProfile of transformers' generate on the accelerated 13B-shape model
(static cache, compiled layer step).  Usage: python tools/eager_profile.py [layers]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformers  # noqa: E402

from eetq_amd.utils import eet_accelerator  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = transformers.LlamaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=layers, num_attention_heads=40,
                               num_key_value_heads=40, vocab_size=32000, max_position_embeddings=4096)
torch.manual_seed(0)
torch.set_default_dtype(torch.float16)
with torch.device("cuda:0"):
    model = transformers.LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True, fused_residual=True)
prompt = torch.randint(0, 32000, (1, 1024), generator=torch.Generator().manual_seed(1)).cuda()
kw = dict(max_new_tokens=50, min_new_tokens=50, do_sample=False, pad_token_id=0, cache_implementation="static",
          disable_compile=True)
with torch.no_grad():
    model.generate(prompt[:, :64], max_new_tokens=4, min_new_tokens=4, do_sample=False, pad_token_id=0,
                   cache_implementation="static", disable_compile=True)
    for name in ("generate",):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(prompt, **kw)
        torch.cuda.synchronize()
        print("static-cache eager generate: %.1f ms  (%.1f tokens/s)" % ((time.perf_counter() - t0) * 1e3, 50 / (time.perf_counter() - t0)))
    pr = cProfile.Profile()
    pr.enable()
    model.generate(prompt, **kw)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))
