"""Greedy decode with a static KV cache and captured HIP graphs: one replay per generated token, or per several.

The reference's end-to-end recipe (examples/models/llama_transformers_example.py:68-79) calls transformers'
``generate``; with ~360 short kernels per token at Llama-13B shapes that loop is bound by host launch time, not by the
GPU.  The decode step is launch-bound inner-loop work, so it is captured once (token id and position live in static device
tensors; the static cache's own token counter is advanced on the device by the attention kernel) and replayed per token.
Prefill stays eager.  Works with any transformers causal LM whose forward accepts ``past_key_values`` / ``cache_position``
(stock Llama, ``eet_quantize``-d or ``eet_accelerator``-ed).
"""
import contextlib
import inspect

import torch

__all__ = ["GraphDecoder"]


class GraphDecoder:
    def __init__(self, model, batch, max_len, capture=True, lean=True, steps_per_graph=7):
        """``max_len``: cache rows (prompt + new tokens).  ``capture=False`` keeps the same static-cache stepping but runs
        every step eagerly (used to check the graph against the launches it was captured from).  ``lean``: step fully
        accelerated models layer by layer instead of through the stock model forward (see ``_lean``).
        ``steps_per_graph``: a second graph holds that many consecutive steps (one replay = that many tokens); the token
        hand-over between steps (next input id, position, the output row) is done by launches INSIDE the graphs, so a
        generated token costs the host one ``replay`` per ``steps_per_graph`` tokens and nothing else -- with one graph
        per token plus three eager bookkeeping launches the host side was 0.125 ms of every 2.96 ms token at 13B shapes
        (``profiles/r06_bench_*.json``: decode_budget.step_us vs the end-to-end time)."""
        self.lean = bool(lean)
        from transformers import StaticCache
        self.model = model
        self.batch = int(batch)
        self.max_len = int(max_len)
        dev = next(model.parameters()).device
        self.device = dev
        try:
            self.cache = StaticCache(config=model.config, max_cache_len=self.max_len)
        except TypeError:  # older constructor
            self.cache = StaticCache(config=model.config, max_batch_size=self.batch, max_cache_len=self.max_len, device=dev,
                                     dtype=torch.float16)
        self.s_tok = torch.zeros(self.batch, 1, dtype=torch.long, device=dev)
        self.s_pos = torch.zeros(1, dtype=torch.long, device=dev)
        # generated tokens: step i of a generate() writes column s_idx (a device counter the graphs advance themselves)
        self.s_idx = torch.zeros(1, 1, dtype=torch.long, device=dev)
        self.out_buf = torch.zeros(self.batch, self.max_len + 1, dtype=torch.long, device=dev)
        self.graph = None
        self.graph_n = None
        # the prompt pass needs the LAST position's logits only (the stock forward projects all of them onto the vocabulary)
        params = inspect.signature(model.forward).parameters
        self._last_logits_only = {"logits_to_keep": 1} if "logits_to_keep" in params else {}
        self.steps_per_graph = max(1, int(steps_per_graph))
        with torch.no_grad():
            # the cache tensors are allocated lazily by the first forward: run one tiny prefill before capturing
            model(self.s_tok, past_key_values=self.cache, cache_position=self.s_pos, use_cache=True)
            if capture:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._advance()
                torch.cuda.current_stream().wait_stream(side)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self._advance()
                if self.steps_per_graph > 1:
                    self.graph_n = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_n):
                        for _ in range(self.steps_per_graph):
                            self._advance()
        self.cache.reset()

    def _fresh_prefill_ok(self):
        """True when the prompt may run under ``fresh_static_prefill``: the lean layer-by-layer model, every attention block
        able to take its static-cache prompt path (initialised cache layer, supported head size)."""
        if not self._lean():
            return False
        for layer in self.model.model.layers:
            attn = getattr(layer, "self_attn", None)
            if (attn is None or not getattr(attn, "static_prefill", False) or not hasattr(attn, "_static_cache_layer")
                    or attn._static_cache_layer(self.cache) is None):
                return False
        return True

    def _lean(self):
        """True when every decoder layer is an accelerated one (eet_accelerator with fused_attn / fused_mlp / fused_residual):
        those read positions and the cache's own token counter and need neither the causal mask nor the cos / sin tensors
        the stock ``LlamaModel.forward`` builds on every step (~20 small launches per token)."""
        base = getattr(self.model, "model", None)
        layers = getattr(base, "layers", None)
        return (self.lean and layers is not None and len(layers) > 0 and hasattr(base, "embed_tokens") and hasattr(base, "norm")
                and hasattr(self.model, "lm_head") and all(getattr(l, "fused_layer_step", False) for l in layers))

    def _logits(self):
        """The last position's logits [batch, vocab] of one decode step on the static tensors."""
        if self._lean():
            base = self.model.model
            h = base.embed_tokens(self.s_tok)
            pos = self.s_pos.view(1, 1).expand(self.batch, 1)
            for layer in base.layers:
                h = layer(h, attention_mask=None, position_ids=pos, past_key_values=self.cache, use_cache=True)
                if isinstance(h, tuple):
                    h = h[0]
            lg = self.model.lm_head(base.norm(h))
        else:
            lg = self.model(self.s_tok, past_key_values=self.cache, cache_position=self.s_pos, use_cache=True).logits
        return lg[:, -1]

    def _step(self):
        return self._logits().argmax(-1, keepdim=True)

    def _advance(self):
        """One step and its hand-over, all on the device: the new token goes to its output column and becomes the next
        input id; position and output column move on -- one library launch (``ops.greedy_handover``: torch.argmax's answer)
        where the logits are fp16 on the GPU, otherwise argmax + scatter_ + copy_ + two add_."""
        lg = self._logits()
        if lg.is_cuda and lg.dtype == torch.float16 and lg.stride(-1) == 1:
            from .. import ops
            ops.greedy_handover(lg, self.out_buf, self.s_idx, self.s_tok, self.s_pos)
            return
        nxt = lg.argmax(-1, keepdim=True)
        self.out_buf.scatter_(1, self.s_idx.expand(self.batch, 1), nxt)
        self.s_tok.copy_(nxt)
        self.s_pos.add_(1)
        self.s_idx.add_(1)

    @torch.no_grad()
    def generate(self, prompt, new_tokens, return_prefill_logits=False):
        """prompt [batch, P] int64 on the model's device -> [batch, P + new_tokens]."""
        B, P = prompt.shape
        if B != self.batch or P + new_tokens > self.max_len:
            raise ValueError("GraphDecoder: built for batch %d and %d cache rows" % (self.batch, self.max_len))
        out = self.prefill(prompt)
        tok = out.logits[:, -1].argmax(-1, keepdim=True)
        self.out_buf[:, :1].copy_(tok)
        self.s_tok.copy_(tok)
        self.s_pos.fill_(P)
        self.s_idx.fill_(1)
        left = new_tokens - 1
        while left > 0:
            if self.graph_n is not None and left >= self.steps_per_graph:
                self.graph_n.replay()
                left -= self.steps_per_graph
            elif self.graph is not None:
                self.graph.replay()
                left -= 1
            else:
                self._advance()
                left -= 1
        tokens = torch.cat([prompt, self.out_buf[:, :new_tokens]], dim=1)
        return (tokens, out.logits[:, -1]) if return_prefill_logits else tokens

    def prefill(self, prompt):
        """Empty the cache and run the (unpadded) prompt [batch, P] eagerly; returns the model output, whose logits hold the LAST
        position only where the model's forward can be told so."""
        P = prompt.shape[1]
        fresh = self._fresh_prefill_ok()
        if fresh:
            # every layer is an accelerated block on an initialised static cache: only the token counters are reset (the model
            # derives the prompt's positions from them; the blocks never read a row at or beyond the counter, so the stock
            # reset's zero-fill of 2 x layers cache tensors buys nothing), and the prompt is attended causally over its own
            # rows (llama_modules.fresh_static_prefill)
            torch._foreach_zero_([layer.cumulative_length for layer in self.cache.layers])   # one launch for all layers
            from ..modules.llama_modules import fresh_static_prefill
            ctx = fresh_static_prefill()
        else:
            self.cache.reset()
            ctx = contextlib.nullcontext()
        with ctx:
            out = self.model(prompt, past_key_values=self.cache, cache_position=torch.arange(P, device=prompt.device),
                             use_cache=True, **self._last_logits_only)
        return out
