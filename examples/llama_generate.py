#!/usr/bin/env python
"""BASELINE.json configs[4]: Llama-2-13B-shaped model, eet_quantize(model), greedy generate prompt=1024 new=50,
replicated on 1/2/4/8 MI355X (one process per GPU: torchrun --nproc-per-node N examples/llama_generate.py).

No checkpoints are available offline, so the model is a random-init LlamaForCausalLM with the 13B shapes (hidden 5120,
intermediate 13824, 40 layers, 40 heads, vocab 32000), fp16, fixed seed -- the same recipe the reference's
examples/models/llama_transformers_example.py:22-90 applies to a real checkpoint (load fp16 -> quantise -> warm-up
generate -> one timed generate with synchronize on both sides).  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eetq_amd.utils.quantizer import eet_quantize  # noqa: E402
from eetq_amd.utils.replicas import ReplicaGroup  # noqa: E402
from eetq_amd.utils.graph_decoder import GraphDecoder  # noqa: E402


def build_model(args, dev):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=args.hidden, intermediate_size=args.inter, num_hidden_layers=args.layers,
                      num_attention_heads=args.heads, num_key_value_heads=args.heads, vocab_size=32000,
                      max_position_embeddings=4096)
    torch.manual_seed(0)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    try:
        with torch.device(dev):
            model = LlamaForCausalLM(cfg)
    finally:
        torch.set_default_dtype(old)
    return model.eval()


def fuse_rmsnorm(model):
    """Route every LlamaRMSNorm through the library's T5/RMS layernorm op (the reference's layernorm_forward,
    csrc/layernorm_kernels/layernorm.cu): one kernel instead of ~6 elementwise launches per norm."""
    import types

    from eetq_amd.ops import layernorm_forward

    def forward(self, hidden_states):
        x = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        out = torch.empty_like(x)
        layernorm_forward(x, self.weight, out, self.variance_epsilon)
        return out

    n = 0
    for m in model.modules():
        if type(m).__name__ == "LlamaRMSNorm":
            m.forward = types.MethodType(forward, m)
            n += 1
    return n


class _SharedInputProj(torch.nn.Module):
    """Stand-in for q/k/v_proj (or gate/up_proj): the first sibling runs ONE fused W8A16 GEMM over the concatenated
    output channels and parks the parts; the others hand out their part.  Keeps the stock transformers block intact."""

    def __init__(self, fused, index, box):
        super().__init__()
        self.fused, self.index, self.box = (fused if index == 0 else None), index, box

    def forward(self, x):
        if self.index == 0:
            self.box[:] = self.fused(x)
        return self.box[self.index]


def fuse_projections(model):
    """q/k/v and gate/up share their input: 7 -> 4 GEMM launches per decoder layer (eetq_amd.utils.fuse)."""
    from eetq_amd.utils.fuse import fuse_w8a16_linears
    n = 0
    for layer in model.model.layers:
        att, mlp = layer.self_attn, layer.mlp
        for owner, names in ((att, ("q_proj", "k_proj", "v_proj")), (mlp, ("gate_proj", "up_proj"))):
            fused = fuse_w8a16_linears([getattr(owner, nm) for nm in names])
            box = [None] * len(names)
            for i, nm in enumerate(names):
                setattr(owner, nm, _SharedInputProj(fused, i, box))
            n += 1
    torch.cuda.empty_cache()
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--prompt", type=int, default=1024)
    ap.add_argument("--new", type=int, default=50)
    ap.add_argument("--hidden", type=int, default=5120)
    ap.add_argument("--inter", type=int, default=13824)
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--heads", type=int, default=40)
    ap.add_argument("--no-quant", action="store_true", help="fp16 nn.Linear baseline (rocBLAS) for comparison")
    ap.add_argument("--fuse-norm", action="store_true", help="LlamaRMSNorm -> eetq layernorm_forward kernel")
    ap.add_argument("--fuse-proj", action="store_true", help="one W8A16 launch for q/k/v and one for gate/up")
    ap.add_argument("--accelerate", action="store_true",
                    help="eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True): fused-QKV "
                         "attention blocks with the in-place rotary kernel, gate/up in one launch, RMS-norm kernel")
    ap.add_argument("--decode-attn", default="kernel", choices=["kernel", "math", "off"],
                    help="single-token attention of the EET blocks: the library's split-KV kernel, batched matrix-vector "
                         "products, or the stock attention call")
    ap.add_argument("--gated-fusion", action="store_true", help="silu*mul inside the down GEMV launch instead of its own launch")
    ap.add_argument("--two-launch-step", action="store_true",
                    help="decode step as rotary+cache-write launch and two attention launches instead of the one-launch form")
    ap.add_argument("--python-layer-step", action="store_true",
                    help="decode step of a layer through the Python blocks instead of one call into the compiled module")
    ap.add_argument("--no-glu8", action="store_true", help="gate|up in plain column order: silu*mul as its own launch")
    ap.add_argument("--static-cache", action="store_true",
                    help="eager generate(cache_implementation='static', disable_compile=True): pre-allocated cache, so the "
                         "accelerated blocks take their one-launch decode step / the compiled layer step")
    ap.add_argument("--graph", action="store_true",
                    help="greedy decode with a static KV cache and ONE captured HIP graph per token (launch-bound "
                         "inner loop -> hipGraph) instead of transformers' eager generate()")
    args = ap.parse_args()

    grp = ReplicaGroup()
    dev = grp.device
    t0 = time.perf_counter()
    model = build_model(args, dev)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    if args.accelerate:
        from eetq_amd.utils import eet_accelerator
        eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True, fused_residual=True,
                        glu8=not args.no_glu8)
        for layer in model.model.layers:
            layer.self_attn.decode_math_attention = {"kernel": True, "math": "always", "off": False}[args.decode_attn]
            layer.mlp.fuse_activation = args.gated_fusion
            layer.self_attn.fused_decode_step = not args.two_launch_step
            layer.fused_layer_step = layer.fused_layer_step and not args.python_layer_step
    elif not args.no_quant:
        eet_quantize(model)
    if args.fuse_norm:
        fuse_rmsnorm(model)
    if args.fuse_proj and not args.no_quant:
        fuse_projections(model)
    torch.cuda.synchronize()
    t_quant = time.perf_counter() - t0

    # identical prompts on every replica (rank 0 draws them, one broadcast)
    g = torch.Generator().manual_seed(1)
    prompt = torch.randint(0, 32000, (args.batch, args.prompt), generator=g).to(dev)
    grp.fan_out(prompt)
    kw = dict(max_new_tokens=args.new, min_new_tokens=args.new, do_sample=False, pad_token_id=0)
    if args.static_cache:
        kw.update(cache_implementation="static", disable_compile=True)

    with torch.no_grad():
        out = model.generate(prompt[:, :64], max_new_tokens=4, min_new_tokens=4, do_sample=False, pad_token_id=0)  # warm-up
        torch.cuda.synchronize()
        # prefill-only timing (one forward over the prompt)
        grp.barrier()
        t0 = time.perf_counter()
        model(prompt)
        torch.cuda.synchronize()
        t_prefill = time.perf_counter() - t0
        holder = {}

        decoder = GraphDecoder(model, args.batch, args.prompt + args.new + 8) if args.graph else None
        if decoder is not None:
            decoder.generate(prompt[:, :64], 4)  # warm-up
            torch.cuda.synchronize()

        def run():
            if args.graph:
                holder["out"] = decoder.generate(prompt, args.new)
            else:
                holder["out"] = model.generate(prompt, **kw)
        secs = grp.timed(run)
    out = holder["out"]
    crcs = grp.gather_checksums(out[:, args.prompt:].to(torch.int32))
    if grp.rank == 0:
        new_tokens = args.batch * args.new
        line = {"config": "Llama-2-13B shapes, random init fp16, %s, prompt=%d new=%d batch=%d, %s" %
                          ("fp16 nn.Linear" if args.no_quant else "eet_quantize (W8A16)", args.prompt, args.new, args.batch,
                           ("hipGraph decode" if args.graph else "transformers eager generate" + (" (static cache)" if args.static_cache else "")) +
                           (", fused rmsnorm" if args.fuse_norm else "") + (", fused qkv + gate/up" if args.fuse_proj else "") +
                           (", eet_accelerator(fused_attn, fused_mlp, fused_norm)" if args.accelerate else "") +
                           (", two-launch decode step" if args.two_launch_step else "") +
                           (", python layer step" if args.python_layer_step else "") + (", no glu8" if args.no_glu8 else "")),
                "n_gpus": grp.world_size, "end_to_end_s": round(secs, 4), "prefill_s": round(t_prefill, 4),
                "tokens_per_s_per_replica": round(new_tokens / secs, 2),
                "tokens_per_s_aggregate": round(grp.world_size * new_tokens / secs, 2),
                "decode_tokens_per_s_per_replica": round(new_tokens / max(secs - t_prefill, 1e-9), 2),
                "replicas_identical_tokens": len(set(crcs)) == 1, "build_s": round(t_build, 1),
                "quantize_s": round(t_quant, 2),
                "max_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}
        print(json.dumps(line))
    grp.close()


if __name__ == "__main__":
    main()
