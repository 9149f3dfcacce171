"""``eet_accelerator``: fused-attention rewrite of a transformers Llama model.

Host-side mirror of /root/reference/python/eetq/utils/accelerator.py:15-78 (``eet_accelerator``,
``replace_with_eet_fp16_fused_attn``, ``replace_with_eet_quant_fused_attn``) and of ``replace_with_eet_qlinear``
(python/eetq/utils/quantizer.py:13-38) for the interface of the installed transformers (see
eetq_amd/modules/llama_modules.py).  ``fused_mlp`` / ``fused_norm`` are extensions (default off): gate/up as one launch,
and every ``LlamaRMSNorm`` through ``layernorm_forward``.
"""
import types

import torch
import torch.nn as nn

from ..modules.llama_modules import EETLlamaAttention, EETLlamaMLP, EETQuantLlamaAttention, _rope_base
from ..modules.qlinear import W8A16Linear
from .quantizer import _progress, eet_quantize, find_layers, get_named_linears, set_op_by_name

__all__ = ["eet_accelerator", "replace_with_eet_fp16_fused_attn", "replace_with_eet_quant_fused_attn",
           "replace_with_eet_qlinear", "replace_with_eet_fused_mlp", "replace_with_eet_rmsnorm",
           "replace_with_eet_fused_residual"]


def _llama_attention_type():
    from transformers.models.llama.modeling_llama import LlamaAttention
    return LlamaAttention


def _attn_geometry(m):
    cfg = getattr(m, "config", None)
    if cfg is not None:
        heads, kv_heads, hidden = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.hidden_size
        max_pos, theta = getattr(cfg, "max_position_embeddings", 2048), _rope_base(cfg)
    else:  # transformers 4.3x attribute names, as the reference reads them (accelerator.py:39)
        heads, hidden = m.num_heads, m.hidden_size
        kv_heads, max_pos, theta = getattr(m, "num_key_value_heads", heads), 2048, 10000.0
    return dict(hidden_size=hidden, num_heads=heads, num_key_value_heads=kv_heads, layer_idx=getattr(m, "layer_idx", 0),
                config=cfg, max_position_embeddings=max_pos, rope_theta=theta)


def replace_with_eet_fp16_fused_attn(model):
    """Every LlamaAttention -> EETLlamaAttention over ONE fp16 nn.Linear holding [q; k; v] (reference :22-51)."""
    named = find_layers(model, include=[_llama_attention_type()], exclude=[])
    for name, m in _progress(list(named.items()), "[EET][INFO] attention fusion processiong..."):
        q, k, v = m.q_proj, m.k_proj, m.v_proj
        w = torch.cat([q.weight, k.weight, v.weight], dim=0)
        bias = torch.cat([q.bias, k.bias, v.bias], dim=0) if q.bias is not None else None
        qkv = nn.Linear(q.in_features, w.shape[0], bias is not None, dtype=w.dtype, device=w.device)
        qkv.weight = nn.Parameter(w, requires_grad=False)
        qkv.bias = nn.Parameter(bias, requires_grad=False) if bias is not None else None
        attn = EETLlamaAttention(qkv_proj=qkv, o_proj=m.o_proj, dev=w.device, **_attn_geometry(m))
        set_op_by_name(model, name, attn)
    return model


def replace_with_eet_quant_fused_attn(model, dev="cuda:0"):
    """Every LlamaAttention whose projections are already W8A16Linear -> EETQuantLlamaAttention (reference :54-78)."""
    for name, m in list(model.named_modules()):
        if not isinstance(m, _llama_attention_type()):
            continue
        attn = EETQuantLlamaAttention(q_proj=m.q_proj, k_proj=m.k_proj, v_proj=m.v_proj, o_proj=m.o_proj,
                                      dev=m.q_proj.qweight.device, **_attn_geometry(m))
        set_op_by_name(model, name, attn)
    return model


def replace_with_eet_qlinear(model, init_only=False, target_model="llama", device="cuda:0"):
    """W8A16 for every nn.Linear inside the decoder layers (lm_head untouched); reference quantizer.py:13-38, whose
    ``structure_mapping`` table resolves to ``model.model.layers`` for llama."""
    if target_model != "llama":
        raise ValueError("replace_with_eet_qlinear: only target_model='llama' is mapped")
    base = getattr(model, "model", model)
    for layer in _progress(list(base.layers), "[EET][INFO] replace with eet weight quantize only linear..."
                           + ("(init only)" if init_only else "")):
        for name, linear in get_named_linears(layer).items():
            if linear.weight.dtype == torch.float16:
                q_linear = W8A16Linear.from_torch(linear, scales=None, init_only=init_only)
            elif linear.weight.dtype == torch.int8:
                q_linear = W8A16Linear.from_torch(linear, scales=torch.div(linear.state_dict()["SCB"], 127.0),
                                                  init_only=init_only)
            else:
                raise ValueError("Unsupported data type: {}".format(linear.weight.dtype))
            set_op_by_name(layer, name, q_linear)
            del linear  # (the reference's `linear.cpu()` first is a PCIe copy nothing reads)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return model


def replace_with_eet_fused_mlp(model, glu8=True):
    """Extension: every Llama MLP whose projections are W8A16Linear -> EETLlamaMLP (gate/up in one launch; ``glu8``: with the
    activation in that launch's epilogue for single-token steps)."""
    n = 0
    for name, m in list(model.named_modules()):
        if type(m).__name__ == "LlamaMLP" and all(isinstance(getattr(m, p, None), W8A16Linear)
                                                   for p in ("gate_proj", "up_proj", "down_proj")):
            set_op_by_name(model, name, EETLlamaMLP(m.gate_proj, m.up_proj, m.down_proj, glu8=glu8))
            n += 1
    return n


def replace_with_eet_rmsnorm(model):
    """Extension: route every LlamaRMSNorm through ``layernorm_forward`` (the reference binds that op but never calls
    it from Python; csrc/layernorm_kernels/layernorm.cu:98-113)."""
    from ..ops import layernorm_forward

    def forward(self, hidden_states):
        x = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        out = torch.empty_like(x)
        layernorm_forward(x, self.weight, out, self.variance_epsilon)
        return out

    n = 0
    for m in model.modules():
        if type(m).__name__ == "LlamaRMSNorm" and m.weight.dtype == torch.float16:
            m.forward = types.MethodType(forward, m)
            n += 1
    return n


def replace_with_eet_fused_residual(model):
    """Extension: decoder layers whose attention and MLP are the EET blocks add their residuals inside the o_proj /
    down_proj epilogues (``eetq_w8a16_gemm_fused``) instead of two elementwise kernels per layer, and hand their two RMS-norms
    to the QKV and gate/up projections (inside the GEMV launch for single-token steps, ``eetq_w8a16_gemv_rmsnorm``)."""
    from .. import ops
    from ..modules.llama_modules import EETLlamaAttention
    layer_step = ops.llama_decode_layer   # None under the ctypes binding

    def step_weights(self, attn, mlp, n1, n2):
        """The layer's tensors in llama_decode_layer's order, gathered once (module attribute lookups cost more than the call
        itself otherwise); rebuilt when a projection's weight buffer has been replaced (.to(), load_state_dict(assign=True))."""
        qkv, o, gu, down = attn.qkv_proj, attn.o_proj, mlp.gate_up_proj, mlp.down_proj
        cached = self._step_weights
        if (cached is not None and cached[0][1] is qkv._buffers["qweight"] and cached[1][0] is o._buffers["qweight"]
                and cached[1][4] is gu._buffers["qweight"] and cached[1][7] is down._buffers["qweight"]
                and cached[0][0][0] is n1._parameters["weight"]):
            return cached
        cached = (((n1.weight, n1.variance_epsilon), qkv.qweight, qkv.weight_scales, qkv.bias),
                  (o.qweight, o.weight_scales, o.bias, (n2.weight, n2.variance_epsilon), gu.qweight, gu.weight_scales, gu.bias,
                   down.qweight, down.weight_scales, down.bias))
        self._step_weights = cached
        return cached

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                position_embeddings=None, **kwargs):
        attn, mlp, n1, n2 = self._eet_blocks
        if (hidden_states.shape[1] == 1 and layer_step is not None and self.fused_layer_step and attn.fused_decode_step
                and not mlp.fuse_activation and not kwargs.get("output_attentions", False)):
            # single-token step on a static cache: the whole layer (six launches) as one call into the compiled module
            ready = attn.decode_step_state(hidden_states, attention_mask, position_ids, past_key_values)
            if ready is not None:
                positions, table, cache, tickets, add = ready
                head, tail = step_weights(self, attn, mlp, n1, n2)
                return layer_step(hidden_states, *head, positions, table, cache.keys, cache.values, tickets,
                                  cache.cumulative_length, add, attn.scaling, attn.num_heads, attn.num_key_value_heads, *tail,
                                  mlp.glu8)
        h, _ = attn(hidden_states=hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                    past_key_values=past_key_values, use_cache=use_cache, position_embeddings=position_embeddings,
                    residual=hidden_states, input_norm=(n1.weight, n1.variance_epsilon), **kwargs)
        return mlp(h, residual=h, norm=(n2.weight, n2.variance_epsilon))

    n = 0
    for m in model.modules():
        if (type(m).__name__ == "LlamaDecoderLayer" and isinstance(m.self_attn, EETLlamaAttention)
                and isinstance(m.mlp, EETLlamaMLP) and isinstance(m.self_attn.o_proj, W8A16Linear)
                and m.input_layernorm.weight.dtype == torch.float16 and hasattr(m.input_layernorm, "variance_epsilon")):
            eligible = (isinstance(m.self_attn.qkv_proj, W8A16Linear) and isinstance(m.mlp.down_proj, W8A16Linear)
                        and m.mlp.intermediate_size % 8 == 0)
            m.fused_layer_step = bool(eligible)
            m._eet_blocks = (m.self_attn, m.mlp, m.input_layernorm, m.post_attention_layernorm)  # plain attribute: no __getattr__
            m._step_weights = None
            m.forward = types.MethodType(forward, m)
            n += 1
    return n


def eet_accelerator(model, quantize=False, fused_attn=False, dev="cuda:0", fused_mlp=False, fused_norm=False,
                    fused_residual=False, static_cache=False, glu8=True):
    """Reference semantics (accelerator.py:15-19): ``fused_attn`` first builds fp16 fused-QKV attention blocks, then
    ``quantize`` turns every decoder nn.Linear -- the fused QKV included -- into W8A16.  ``static_cache`` (extension): make
    ``model.generate`` default to a pre-allocated KV cache (``cache_implementation="static"`` in the model's generation
    config), the form on which the accelerated blocks run a decode step as one launch per attention / one call per layer.
    Any rewrite also sets ``disable_compile=True`` there: the operators break torch.compile's graph, so a compiled static-
    cache forward would capture nothing.  ``glu8``: see ``replace_with_eet_fused_mlp``."""
    if fused_attn:
        replace_with_eet_fp16_fused_attn(model)
    if quantize:
        replace_with_eet_qlinear(model, init_only=False, target_model="llama", device=dev)
    if fused_mlp:
        replace_with_eet_fused_mlp(model, glu8=glu8)
    if fused_norm:
        replace_with_eet_rmsnorm(model)
    if fused_residual:
        replace_with_eet_fused_residual(model)
    gen = getattr(model, "generation_config", None)
    if gen is not None and (quantize or fused_attn or fused_mlp or fused_norm or fused_residual):
        # the operators are opaque to torch.compile (every one breaks the graph): with a static cache transformers would
        # compile the forward, capture an EMPTY CUDA graph ("The CUDA Graph is empty" warning) and run eagerly anyway
        gen.disable_compile = True
    if static_cache and gen is not None:
        gen.cache_implementation = "static"
    return model
