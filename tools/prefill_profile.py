"""One 1024-token prefill of the accelerated 13B-shape model (8 layers) on a static cache -- run under
`rocprofv3 --kernel-trace --stats` to see what the prefill launches besides the GEMMs.  Prints wall time of the prefill."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformers  # noqa: E402

from eetq_amd.utils import GraphDecoder, eet_accelerator  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = transformers.LlamaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=L, num_attention_heads=40,
                               num_key_value_heads=40, vocab_size=32000, max_position_embeddings=4096)
torch.manual_seed(0)
torch.set_default_dtype(torch.float16)
with torch.device("cuda:0"):
    model = transformers.LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True, fused_residual=True)
prompt = torch.randint(0, 32000, (1, 1024), generator=torch.Generator().manual_seed(1)).cuda()
dec = GraphDecoder(model, 1, 1100, capture=False)
with torch.no_grad():
    dec.generate(prompt, 1)
    torch.cuda.synchronize()
    print("MARK begin")
    t0 = time.perf_counter()
    for _ in range(5):
        dec.generate(prompt, 1)
    torch.cuda.synchronize()
    print("prefill (+1 token), %d layers: %.2f ms each" % (L, (time.perf_counter() - t0) * 200))
