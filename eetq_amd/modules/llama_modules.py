"""Llama attention / MLP blocks built on the library's ops.

Host-side counterpart of /root/reference/python/eetq/modules/llama_modules.py: ``EETRotaryEmbedding`` (:19-65,
cos|sin cache + the in-place ``rotary_embedding_neox`` op), ``EETLlamaAttention`` (:68-148, one fused QKV projection) and
``EETQuantLlamaAttention`` (:151-240, W8A16 projections).  The reference targets the transformers 4.3x attention
interface (``past_key_value`` tuples, flash-attn); these modules speak the interface of the installed transformers
(``position_embeddings`` / ``past_key_values`` cache objects, the registered attention functions), support grouped-query
models, rotate q and k *in place inside the fused QKV output* (``rotary_embedding_neox_strided``), and run the three W8A16
projections as ONE launch over concatenated channels (bit-identical to three launches, eetq_amd/utils/fuse.py).
``EETLlamaMLP`` (gate/up as one launch) has no counterpart in llama_modules.py; the reference fuses gate/up only in its
offline export layer (python/eetq/models/llama.py:39-77).
"""
import contextlib
import threading

import torch
import torch.nn as nn

from .. import ops

__all__ = ["EETRotaryEmbedding", "EETLlamaAttention", "EETQuantLlamaAttention", "EETLlamaMLP", "fresh_static_prefill"]

_prefill_promise = threading.local()


@contextlib.contextmanager
def fresh_static_prefill():
    """The caller's promise for the forwards run inside (per thread): the static KV cache holds NO tokens yet and the prompt is
    unpadded, so a prompt's attention is plain causal attention over its own T tokens.  The attention blocks then write cache
    rows 0 .. T - 1 and SET the token counter to T (the rows need no reset first; the counters do where the model derives the
    prompt's positions from them), attend those T rows with the causal
    flag of the library attention (half the work of a masked pass over the whole pre-allocated cache) and ignore the
    model-level mask.  ``GraphDecoder.generate`` makes this promise; without it a prompt
    on a static cache is attended as transformers does it: every cache row, under the model's mask."""
    prev = getattr(_prefill_promise, "fresh", False)
    _prefill_promise.fresh = True
    try:
        yield
    finally:
        _prefill_promise.fresh = prev


class EETRotaryEmbedding(nn.Module):
    """cos|sin cache [max_position, dim] in fp16 and the in-place NeoX rotation (reference :19-65)."""

    def __init__(self, dim, max_position_embeddings=2048, base=10000, device=None):
        super().__init__()
        self.dim = dim
        self.max_position_embeddings = max_position_embeddings
        self.base = base
        inv_freq = 1.0 / (self.base ** (torch.arange(0, self.dim, 2, dtype=torch.float32, device=device) / self.dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)
        self._set_cos_sin_cache(max_position_embeddings, self.inv_freq.device)

    def _set_cos_sin_cache(self, seq_len, device):
        self.max_seq_len_cached = seq_len
        t = torch.arange(seq_len, device=device, dtype=torch.float32)
        freqs = torch.outer(t, self.inv_freq.to(device=device, dtype=torch.float32))
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1)
        self.register_buffer("cos_sin_cache", cache.half().contiguous(), persistent=False)

    def forward(self, query, key, positions):
        """query [..., q_heads, dim], key [..., k_heads, dim] (dense last two dims, any token stride), positions int64
        with one entry per token.  Rotates in place and returns (query, key)."""
        ops.rotary_embedding_neox_strided(positions, query, key, self.dim, self.cos_sin_cache)
        return query, key


def _rope_base(config):
    params = getattr(config, "rope_parameters", None) or {}
    rope_type = params.get("rope_type", "default") if isinstance(params, dict) else "default"
    if rope_type not in (None, "default"):
        raise NotImplementedError("EETRotaryEmbedding implements the default RoPE only (got rope_type=%r)" % (rope_type,))
    base = params.get("rope_theta") if isinstance(params, dict) else None
    return float(base if base is not None else getattr(config, "rope_theta", 10000.0))


class _EETAttentionBase(nn.Module):
    def _setup(self, hidden_size, num_heads, dev, num_key_value_heads, layer_idx, config, max_position_embeddings,
               rope_theta):
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_dim = getattr(config, "head_dim", None) or hidden_size // num_heads
        self.num_key_value_heads = num_key_value_heads or num_heads
        if config is None and self.head_dim * num_heads != hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {hidden_size}"
                             f" and `num_heads`: {num_heads}).")
        if num_heads % self.num_key_value_heads:
            raise ValueError("num_heads must be a multiple of num_key_value_heads")
        self.layer_idx = layer_idx
        self.config = config
        # attributes the registered attention functions of transformers read from the module
        self.num_key_value_groups = num_heads // self.num_key_value_heads
        self.scaling = self.head_dim ** -0.5
        self.attention_dropout = 0.0
        self.is_causal = True
        self.decode_math_attention = True
        self.fused_decode_step = True   # rotary + cache write + attention of a static-cache decode step as one launch
        self.static_prefill = True      # rotary + cache write of a PROMPT on an initialised static cache as one launch
        self.mfma_prefill = True        # a promised-fresh prompt's causal attention through ops.prefill_attention (else torch SDPA)
        self._tickets = None            # that launch's arrival counters (zero between launches)
        self.rotary_emb = EETRotaryEmbedding(self.head_dim, max_position_embeddings=max_position_embeddings,
                                             base=rope_theta, device=dev)

    def _positions(self, position_ids, past_key_values, batch, q_len, device):
        if position_ids is None:
            seen = past_key_values.get_seq_length(self.layer_idx) if past_key_values is not None else 0
            position_ids = (torch.arange(q_len, device=device) + seen).unsqueeze(0)
        if position_ids.shape[0] != batch:
            position_ids = position_ids.expand(batch, q_len)
        return position_ids.to(torch.int64).contiguous()

    def _static_cache_layer(self, past_key_values):
        """The transformers StaticLayer of this block if the fused decode path applies, else None."""
        layers = getattr(past_key_values, "layers", None)
        if layers is None or self.layer_idx >= len(layers) or self.head_dim not in (64, 128):
            return None
        layer = layers[self.layer_idx]
        if (type(layer).__name__ != "StaticLayer" or not getattr(layer, "is_initialized", False)
                or not all(hasattr(layer, a) for a in ("keys", "values", "cumulative_length"))
                or layer.keys.dtype != torch.float16 or layer.keys.dim() != 4):
            return None
        return layer

    # (key, additive rows, the mask itself) of the most recent conversion: the model hands every layer the same mask.  ONE
    # immutable tuple, replaced whole: a reader in another thread sees the old entry or the new one, never a mixture; the
    # entry keeps the mask alive, so its id() cannot be recycled while the entry stands.
    _mask_memo = [(None, None, None)]

    @classmethod
    def _decode_mask_rows(cls, attention_mask, batch, s_len, dtype, device):
        """Additive fp16 mask rows [B or 1, S] for a single-token step, ``None`` for "no mask", or ``False`` when the mask
        has a form this path does not understand (the caller then takes the stock attention path).  Accepts the 4-D
        [B, 1, 1, S'] masks transformers builds (bool or additive) and 2-D [B, S'] padding masks (bool / integer keep
        flags or additive floats).  The conversion is done once per forward: the result is remembered against the mask's
        identity AND version counter, so a caller that updates one mask buffer in place never gets a stale copy."""
        if attention_mask is None:
            return None
        if attention_mask.dim() == 4:
            if attention_mask.shape[1] != 1 or attention_mask.shape[2] != 1:
                return False
            rows = attention_mask[:, 0, 0]
        elif attention_mask.dim() == 2:
            rows = attention_mask
        else:
            return False
        if rows.shape[0] not in (1, batch) or rows.shape[-1] < s_len or rows.device != device:
            return False
        if rows.dtype == dtype and rows.is_floating_point():
            return rows[..., :s_len] if rows.shape[-1] != s_len else rows
        key = (id(attention_mask), attention_mask._version, attention_mask.data_ptr(), tuple(attention_mask.shape),
               attention_mask.dtype, dtype)
        entry = cls._mask_memo[0]
        if entry[0] != key or entry[2] is not attention_mask:
            if rows.is_floating_point():
                add = rows.to(dtype)
            else:  # bool / integer: non-zero = attend
                add = torch.zeros(rows.shape, dtype=dtype, device=device).masked_fill_(rows == 0, float("-inf"))
            entry = cls._mask_memo[0] = (key, add, attention_mask)
        add = entry[1]
        return add[..., :s_len] if add.shape[-1] != s_len else add

    def _grow_table(self, need):
        """The rotation indexes the cos|sin cache by position: grow it (host-side, never during graph capture -- the first
        eager pass already sees the same lengths) when the cache or the sequence is longer than the table."""
        if need > self.rotary_emb.max_seq_len_cached:
            self.rotary_emb._set_cos_sin_cache(max(need, 2 * self.rotary_emb.max_seq_len_cached),
                                               self.rotary_emb.cos_sin_cache.device)

    def _step_tickets(self, batch, device):
        if self._tickets is None or self._tickets.numel() < batch * self.num_heads + 1 or self._tickets.device != device:
            self._tickets = torch.zeros(batch * self.num_heads + 1, dtype=torch.int32, device=device)
        return self._tickets

    # what every layer of a model derives from the step's position_ids / attention_mask, computed by the first layer of the
    # step and reused by the others: [position_ids, its version, attention_mask, its version, batch, rows, positions, add].
    # The tensors themselves are held (an id() could be recycled) and their version counters checked (in-place updates).
    # Per thread: two models stepping in two threads must not read each other's entry between the check and the use.
    _step_memo = threading.local()

    def decode_step_state(self, hidden_states, attention_mask, position_ids, past_key_values):
        """What the one-call decoder-layer step (ops.llama_decode_layer) needs beyond the weights, or None when this step
        is not a single-token step on an initialised static cache with a mask this path understands:
        (positions [B] int64, cos|sin table, the cache layer, ticket buffer, additive mask rows or None)."""
        layer = self._static_cache_layer(past_key_values)
        if layer is None or self.decode_math_attention is not True or position_ids is None or not hidden_states.is_cuda:
            return None
        bsz, rows = hidden_states.shape[0], layer.keys.shape[2]
        memo = getattr(_EETAttentionBase._step_memo, "entry", None)
        if memo is None:
            memo = _EETAttentionBase._step_memo.entry = [None] * 8
        if not (memo[0] is position_ids and memo[1] == position_ids._version and memo[2] is attention_mask
                and (attention_mask is None or memo[3] == attention_mask._version) and memo[4] == bsz and memo[5] == rows):
            add = self._decode_mask_rows(attention_mask, bsz, rows, hidden_states.dtype, hidden_states.device)
            positions = None
            if add is not False:
                positions = self._positions(position_ids, past_key_values, bsz, 1, hidden_states.device)[:, 0]
                if not positions.is_contiguous():
                    positions = positions.contiguous()
            memo[:] = [position_ids, position_ids._version, attention_mask,
                       None if attention_mask is None else attention_mask._version, bsz, rows, positions, add]
        positions, add = memo[6], memo[7]
        if add is False:
            return None
        if rows > self.rotary_emb.max_seq_len_cached:
            self._grow_table(rows)
        table = self.rotary_emb.cos_sin_cache
        if table.shape[-1] != self.head_dim or positions.device != hidden_states.device:
            return None
        return positions, table, layer, self._step_tickets(bsz, hidden_states.device), add

    def _attend(self, q, k, v, attention_mask, past_key_values, input_shape, kwargs, cached=None):
        """q [B, T, H, D], k/v [B, T, Hkv, D] (views into the projection output) -> [B, T, H*D].  ``cached``: the (keys, values)
        of a static cache that already hold this call's rotated k and v (then k, v and past_key_values are unused)."""
        q = q.transpose(1, 2)
        if cached is not None:
            k, v = cached
            if (getattr(_prefill_promise, "fresh", False) and q.shape[2] > 1 and not kwargs.get("output_attentions", False)):
                # promised: the cache was empty and the prompt is unpadded -> causal attention over the T rows just written
                t_len = q.shape[2]
                if (self.mfma_prefill and ops.prefill_attention is not None and q.is_cuda and q.dtype == torch.float16
                        and ops.prefill_attention_supported(self.head_dim) and k.stride(-1) == 1 and v.stride(-1) == 1):
                    # the library's own flash kernel on the matrix cores: the strided query view of the QKV projection, the
                    # cache rows as they lie (ops.prefill_attention; 37 us per layer at 13B shapes against torch's 76)
                    out = ops.prefill_attention(q.transpose(1, 2), k, v, t_len, scaling=self.scaling)
                    return out.reshape(*input_shape, -1), None
                out = torch.nn.functional.scaled_dot_product_attention(
                    q, k[:, :, :t_len], v[:, :, :t_len], is_causal=True, scale=self.scaling,
                    enable_gqa=self.num_key_value_groups > 1).transpose(1, 2)
                return out.reshape(*input_shape, -1).contiguous(), None
        else:
            k, v = k.transpose(1, 2), v.transpose(1, 2)
            if past_key_values is not None:
                k, v = past_key_values.update(k, v, self.layer_idx)
        # One query token.  The library attention kernels launch one workgroup per head (40 workgroups streaming the whole
        # KV cache: 68 us per layer at Llama-13B shapes, S = 1.2 k).  decode_math_attention: True -> the library's own
        # split-KV kernel (ops.decode_attention, ~10 us) whenever it applies, and a batched matrix-vector fallback while
        # a HIP graph is being captured; "always" -> the fallback also in eager mode; False -> stock attention.
        mode     = self.decode_math_attention
        kernel_ok = self.head_dim in (64, 128) and q.is_cuda
        use_math = mode == "always" or (mode and q.is_cuda and (kernel_ok or torch.cuda.is_current_stream_capturing()))
        if q.shape[2] == 1 and use_math and not kwargs.get("output_attentions", False):
            bsz, heads, _, s_len = q.shape[0], q.shape[1], q.shape[2], k.shape[2]
            add = self._decode_mask_rows(attention_mask, bsz, s_len, q.dtype, q.device)
            if add is not False:
                static = self._static_cache_layer(past_key_values)
                if kernel_ok and mode != "always":
                    # split-KV decode kernel of the library: the whole chip streams the cache once (a static cache is
                    # returned whole by update(): only its filled rows are attended, whatever the mask says)
                    out = ops.decode_attention(q[:, :, 0], k, v, mask=add, scaling=self.scaling,
                                               kv_len=static.cumulative_length if static is not None else None
                                               ).unsqueeze(1)  # [B, 1, H, D]
                    return out.reshape(*input_shape, -1), None
                if static is not None and add is None:
                    add = torch.zeros(1, s_len, dtype=q.dtype, device=q.device).masked_fill_(
                        torch.arange(s_len, device=q.device)[None, :] >= static.cumulative_length, float("-inf"))
                if self.num_key_value_groups > 1:
                    k = k.repeat_interleave(self.num_key_value_groups, dim=1)
                    v = v.repeat_interleave(self.num_key_value_groups, dim=1)
                scores = torch.matmul(q, k.transpose(2, 3))                                # [B, H, 1, S]
                if add is not None:
                    scores = torch.add(add[:, None, None, :], scores, alpha=self.scaling)
                else:
                    scores = scores * self.scaling
                probs = torch.softmax(scores, dim=-1)      # fp16 in/out, fp32 accumulation inside
                out = torch.matmul(probs, v).transpose(1, 2)
                return out.reshape(*input_shape, -1).contiguous(), None
        fn = None
        impl = getattr(self.config, "_attn_implementation", None) if self.config is not None else None
        if impl is not None:
            from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
            from transformers.models.llama.modeling_llama import eager_attention_forward
            fn = ALL_ATTENTION_FUNCTIONS.get_interface(impl, eager_attention_forward)
        if fn is not None:
            out, weights = fn(self, q, k, v, attention_mask, dropout=0.0, scaling=self.scaling, **kwargs)
        else:  # stand-alone use without a transformers config: plain SDPA, causal when there is no cache
            if self.num_key_value_groups > 1:
                k = k.repeat_interleave(self.num_key_value_groups, dim=1)
                v = v.repeat_interleave(self.num_key_value_groups, dim=1)
            out = torch.nn.functional.scaled_dot_product_attention(
                q, k, v, attn_mask=attention_mask, is_causal=attention_mask is None and q.shape[2] > 1,
                scale=self.scaling).transpose(1, 2)
            weights = None
        return out.reshape(*input_shape, -1).contiguous(), weights


class EETLlamaAttention(_EETAttentionBase):
    """Multi-headed attention over ONE fused QKV projection (reference :68-148).

    ``qkv_proj`` maps hidden -> (num_heads + 2 * num_key_value_heads) * head_dim (an ``nn.Linear`` or, after
    ``eet_quantize``, a ``W8A16Linear``)."""

    def __init__(self, hidden_size, num_heads, qkv_proj, o_proj, dev, num_key_value_heads=None, layer_idx=0, config=None,
                 max_position_embeddings=2048, rope_theta=10000.0):
        super().__init__()
        self._setup(hidden_size, num_heads, dev, num_key_value_heads, layer_idx, config, max_position_embeddings,
                    rope_theta)
        self.qkv_proj = qkv_proj
        self.o_proj = o_proj

    def _qkv(self, hidden_states, input_norm=None):
        if input_norm is None:
            return self.qkv_proj(hidden_states)
        if hasattr(self.qkv_proj, "qweight"):  # W8A16Linear: RMS-norm inside the launch for single-token steps
            return self.qkv_proj(hidden_states, norm=input_norm)
        x = hidden_states.contiguous()
        normed = torch.empty_like(x)
        ops.layernorm_forward(x, input_norm[0], normed, input_norm[1])
        return self.qkv_proj(normed)

    def forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None,
                position_ids=None, residual=None, input_norm=None, **kwargs):
        """Input shape: Batch x Time x Channel.  ``position_embeddings`` (the model-level cos/sin) is accepted for
        interface compatibility and unused: the rotation reads this module's fp16 cos|sin cache.  ``residual``
        (extension): added to the output projection inside its epilogue when ``o_proj`` is a W8A16Linear.
        ``input_norm=(gamma, eps)`` (extension): ``hidden_states`` is the un-normalised residual stream and the block's
        input RMS-norm is applied here (fused into the QKV launch for single-token steps)."""
        bsz, q_len, _ = hidden_states.shape
        h, hkv, d = self.num_heads, self.num_key_value_heads, self.head_dim
        qkv = self._qkv(hidden_states, input_norm)          # [B, T, (H + 2 Hkv) * D]
        q = qkv[..., : h * d].unflatten(-1, (h, d))
        k = qkv[..., h * d: (h + hkv) * d].unflatten(-1, (hkv, d))
        v = qkv[..., (h + hkv) * d:].unflatten(-1, (hkv, d))
        positions = self._positions(position_ids, past_key_values, bsz, q_len, hidden_states.device)
        kwargs.pop("use_cache", None)
        layer = self._static_cache_layer(past_key_values) if q_len == 1 else None
        need = q_len
        if layer is not None:
            need = layer.keys.shape[2]
        elif past_key_values is not None:
            seen = past_key_values.get_seq_length(self.layer_idx)
            if isinstance(seen, int):
                need = seen + q_len
        self._grow_table(need)
        add = False
        if layer is not None and self.decode_math_attention is True and not kwargs.get("output_attentions", False):
            add = self._decode_mask_rows(attention_mask, bsz, layer.keys.shape[2], hidden_states.dtype, hidden_states.device)
        if add is not False:
            # decode step on an initialised static cache: ONE launch rotates q in place (by its position) and writes the
            # rotated k and v straight into the cache row the cache's own token counter names -- for a left-padded batch
            # the position (real tokens so far) is smaller than that row -- instead of the stock arange + add + two
            # index_copy launches plus the rotary launch.  Then the split-KV attention kernel reads the cache: rows beyond
            # counter + 1 are never attended, mask or no mask, and the kernel's last launch advances the counter (the
            # cache's bookkeeping: the next step's positions and mask come from it).
            counter = layer.cumulative_length
            pos = positions[:, 0].contiguous()
            table = self.rotary_emb.cos_sin_cache
            if self.fused_decode_step and table.shape[-1] == self.head_dim:
                # ... and both as ONE launch: q and k rotated in registers, the new row taken from registers, the chunk
                # merge done by the last workgroup of each head (bit-identical to the pair of launches below)
                out = ops.rope_decode_attention(pos, q[:, 0], k[:, 0], v[:, 0], table, layer.keys, layer.values,
                                                self._step_tickets(bsz, q.device), slots=counter, mask=add, scaling=self.scaling,
                                                kv_len=counter, kv_len_bias=1, advance=counter).reshape(bsz, q_len, -1)
            else:
                ops.rotary_embedding_neox_kvcache(pos, q[:, 0], k[:, 0], v[:, 0], self.head_dim, table, layer.keys,
                                                  layer.values, slots=counter)
                out = ops.decode_attention(q[:, 0], layer.keys, layer.values, mask=add, scaling=self.scaling,
                                           kv_len=counter, kv_len_bias=1, advance=counter).reshape(bsz, q_len, -1)
            weights = None
        else:
            player = self._static_cache_layer(past_key_values) if (q_len > 1 and q.is_cuda and self.static_prefill) else None
            if (player is not None and not getattr(past_key_values, "offloading", False) and player.keys.shape[0] == bsz
                    and player.keys.shape[1] == hkv and q_len <= player.keys.shape[2]
                    and player.cumulative_length.device == q.device):
                # a prompt on an initialised static cache: ONE launch rotates q in place and writes the rotated k and v into
                # the cache rows counter .. counter + T - 1, one more advances the counter -- the stock update() is arange +
                # add + counter add + two index_copy launches, behind the separate rotary launch.  Same bits in the cache.
                self._grow_table(player.keys.shape[2])
                table = self.rotary_emb.cos_sin_cache
                if table.shape[-1] == d:
                    counter = player.cumulative_length
                    if getattr(_prefill_promise, "fresh", False):   # promised empty: rows 0 .. T - 1, whatever the counter held
                        ops.rotary_embedding_neox_kvcache_prefill(positions, q, k, v, d, table, player.keys, player.values)
                        counter.fill_(q_len)
                    else:
                        ops.rotary_embedding_neox_kvcache_prefill(positions, q, k, v, d, table, player.keys, player.values,
                                                                  first_row_dev=counter)
                        counter.add_(q_len)
                else:
                    player = None
            else:
                player = None
            if player is None:
                self.rotary_emb(q, k, positions)
                out, weights = self._attend(q, k, v, attention_mask, past_key_values, (bsz, q_len), kwargs)
            else:
                out, weights = self._attend(q, None, None, attention_mask, None, (bsz, q_len), kwargs,
                                            cached=(player.keys, player.values))
        if residual is None:
            return self.o_proj(out), weights
        if hasattr(self.o_proj, "qweight"):
            return self.o_proj(out, residual=residual), weights
        return residual + self.o_proj(out), weights


class EETQuantLlamaAttention(EETLlamaAttention):
    """W8A16 attention (reference :151-240).  The reference keeps q/k/v as three W8A16 linears; here they become one
    launch over the concatenated output channels when all three are ``W8A16Linear`` (bit-identical results)."""

    def __init__(self, hidden_size, num_heads, q_proj, k_proj, v_proj, o_proj, dev, num_key_value_heads=None,
                 layer_idx=0, config=None, max_position_embeddings=2048, rope_theta=10000.0):
        from ..utils.fuse import fuse_w8a16_linears
        from .qlinear import W8A16Linear
        if not all(isinstance(p, W8A16Linear) for p in (q_proj, k_proj, v_proj)):
            raise TypeError("EETQuantLlamaAttention expects W8A16Linear q/k/v projections (run eet_quantize first)")
        fused = fuse_w8a16_linears([q_proj, k_proj, v_proj]).fused
        super().__init__(hidden_size, num_heads, fused, o_proj, dev, num_key_value_heads=num_key_value_heads,
                         layer_idx=layer_idx, config=config, max_position_embeddings=max_position_embeddings,
                         rope_theta=rope_theta)


class EETLlamaMLP(nn.Module):
    """down(silu(gate(x)) * up(x)) with gate and up as ONE W8A16 launch over concatenated channels.

    ``glu8`` (default): the fused weight's columns are interleaved in groups of 8 gate + 8 up, so for a single token the
    activation is computed in that launch's epilogue (``eetq_w8a16_gemv_glu8``) and the block is two launches; larger inputs
    run the projection and one ``silu_mul`` launch on the interleaved output.  Bit-identical to the plain order."""

    def __init__(self, gate_proj, up_proj, down_proj, glu8=True):
        super().__init__()
        from ..utils.fuse import fuse_w8a16_linears
        self.intermediate_size = gate_proj.out_features
        self.glu8 = bool(glu8) and self.intermediate_size % 16 == 0 and gate_proj.in_features % 64 == 0
        self.gate_up_proj = fuse_w8a16_linears([gate_proj, up_proj], glu8=self.glu8).fused
        self.down_proj = down_proj
        # silu(gate) * up inside the down projection's GEMV launch (eetq_w8a16_gemv_silu_gated) is available but off: every
        # workgroup recomputes the activation of the whole vector, which costs more than the one launch it saves
        # (Llama-13B decode on one box: 241 vs 247 tokens/s)
        self.fuse_activation = False

    def forward(self, x, residual=None, norm=None):
        if self.glu8 and x.is_cuda:
            return self.down_proj(self.gate_up_proj(x, norm=norm, activation="silu_glu8"), residual=residual)
        gu = self.gate_up_proj(x, norm=norm)
        if self.glu8:   # (CPU tensors never get here through the operators; kept for shape-only uses)
            gu = gu.unflatten(-1, (-1, 2, 8))
            return self.down_proj(torch.nn.functional.silu(gu[..., 0, :]).flatten(-2) * gu[..., 1, :].flatten(-2),
                                  residual=residual)
        if self.intermediate_size % 8 == 0 and gu.is_cuda:
            if self.fuse_activation:
                # silu(gate) * up inside the down projection's launch for a single token, one silu_mul launch otherwise
                return self.down_proj(gu, residual=residual, gated=True)
            return self.down_proj(ops.silu_mul(gu), residual=residual)
        gate, up = gu[..., : self.intermediate_size], gu[..., self.intermediate_size:]
        return self.down_proj(torch.nn.functional.silu(gate) * up, residual=residual)
