"""Where the end-to-end time of config 5 (Llama-2-13B shapes, prompt 1024 + 50 tokens, batch 1) goes outside the decode graph:
cache reset, prefill with the static cache, the graph replays, the final concatenation -- each timed with a synchronize on both
sides.  usage: python tools/decode_timeline.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformers  # noqa: E402

from eetq_amd.utils import GraphDecoder, eet_accelerator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = transformers.LlamaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                               num_key_value_heads=40, vocab_size=32000, max_position_embeddings=4096)
torch.manual_seed(0)
torch.set_default_dtype(torch.float16)
with torch.device("cuda:0"):
    model = transformers.LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True, fused_residual=True)
P, NEW = 1024, 50
prompt = torch.randint(0, 32000, (B, P), generator=torch.Generator().manual_seed(1)).cuda()
dec = GraphDecoder(model, B, P + NEW + 8)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


with torch.no_grad():
    dec.generate(prompt[:, :64], 4)
    print("generate(prompt, 50) end to end      %8.2f ms" % timed(lambda: dec.generate(prompt, NEW)))
    print("generate(prompt, 1) (reset + prefill) %8.2f ms" % timed(lambda: dec.generate(prompt, 1)))
    print("cache.reset()                         %8.2f ms" % timed(dec.cache.reset))
    print("model(prompt) without a cache         %8.2f ms" % timed(lambda: model(prompt)))

    def prefill_cache():
        dec.cache.reset()
        model(prompt, past_key_values=dec.cache, cache_position=torch.arange(P, device=prompt.device), use_cache=True)
    print("reset + model(prompt, static cache)   %8.2f ms" % timed(prefill_cache))
    dec.generate(prompt, 8)
    print("7 single-step graph replays           %8.2f ms per step" % (timed(lambda: [dec.graph.replay() for _ in range(7)]) / 7))
    dec.generate(prompt, 8)
    print("one 7-step graph replay               %8.2f ms per step" % (timed(lambda: dec.graph_n.replay()) / 7))
    dec.generate(prompt, 8)
    print("four 7-step graph replays             %8.2f ms per step" % (timed(lambda: [dec.graph_n.replay() for _ in range(4)], 1) / 28))
