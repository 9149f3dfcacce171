"""Fusing several W8A16 linears that share their input into one launch.

Per-channel scales make concatenation along the output dimension exact, and in the gfx950 layout a [K, N] weight is
the byte sequence of its N/16 tile rows, so the fused processed weight is simply the concatenation of the parts'
bytes (every part's N must be a multiple of 16).  One launch over N = sum(N_i) instead of len(parts) launches: fewer
kernel boundaries and more bytes per dispatch -- the lever that lifts decode above the fixed ~2 us cost of a 16 MiB
launch.  Counterpart of the reference's fused-qkv / gate-up helpers
(python/eetq/utils/base.py:40-130, python/eetq/models/llama.py:39-77), which re-quantise the fused fp16 weight instead.
"""
import torch
import torch.nn as nn

from ..modules.qlinear import W8A16Linear

__all__ = ["fuse_w8a16_linears", "FusedW8A16Linear"]


class FusedW8A16Linear(nn.Module):
    """One W8A16 GEMM over the concatenated output channels; forward returns the per-part outputs (views)."""

    def __init__(self, fused, splits):
        super().__init__()
        self.fused = fused
        self.splits = list(splits)

    @torch.no_grad()
    def forward(self, x):
        return torch.split(self.fused(x), self.splits, dim=-1)


def _glu8_interleave_columns(gate, up):
    """[.., I] x 2 -> [.., 2I] in "glu8" order: groups of 16 = 8 gate entries followed by the 8 matching up entries."""
    return torch.stack([gate.unflatten(-1, (-1, 8)), up.unflatten(-1, (-1, 8))], dim=-2).flatten(-3)


def _glu8_interleave_tiles(gate, up, k):
    """The same for two processed (gfx950) int8 weights [K, I]: tile row t' of a part = 16 columns x K = K/64 tiles of
    [4 k groups][16 columns][16 bytes]; output tile row 2t' + h takes columns 8h..8h+7 of gate's and of up's tile row t'."""
    def split(w):   # -> [tile rows, K/64, 4, half, 8, 16]
        return w.contiguous().reshape(-1, k // 64, 4, 2, 8, 16)
    g, u = split(gate), split(up)
    both = torch.stack([g, u], dim=4)                      # [t', kt, grp, half, gate|up, 8, 16]
    return both.permute(0, 3, 1, 2, 4, 5, 6).contiguous()  # [t', half, kt, grp, gate|up, 8, 16] = tile rows 2t' + half


def fuse_w8a16_linears(parts, glu8=False):
    """parts: W8A16Linear modules with the same in_features, device and bias-ness -> FusedW8A16Linear.
    ``glu8`` (two parts = gate and up of a gated MLP, out_features % 16 == 0): columns interleaved in groups of 8 + 8 so one
    16-column tile carries both operands of 8 outputs and the activation can ride in the projection's epilogue
    (``w8_a16_gemm(..., activation="silu_glu8")``); the fused module then has ``glu8 = True``."""
    parts = list(parts)
    if not parts or not all(isinstance(p, W8A16Linear) for p in parts):
        raise TypeError("fuse_w8a16_linears expects W8A16Linear modules")
    k = parts[0].in_features
    dev = parts[0].qweight.device
    has_bias = parts[0].bias is not None
    for p in parts:
        if p.in_features != k or p.qweight.device != dev or (p.bias is not None) != has_bias:
            raise ValueError("parts must share in_features, device and bias-ness")
        if p.out_features % 16:
            raise ValueError("every part's out_features must be a multiple of 16")
    n = sum(p.out_features for p in parts)
    fused = W8A16Linear(k, n, bias=has_bias, dev=dev)
    fused.glu8 = bool(glu8)
    if glu8:
        if len(parts) != 2 or parts[0].out_features != parts[1].out_features or k % 64:
            raise ValueError("glu8 fuses exactly two parts (gate, up) of equal width")
        gate, up = parts
        fused.qweight = _glu8_interleave_tiles(gate.qweight, up.qweight, k).reshape(k, n)
        fused.weight_scales = _glu8_interleave_columns(gate.weight_scales, up.weight_scales)
        if has_bias:
            fused.bias = _glu8_interleave_columns(gate.bias, up.bias)
        return FusedW8A16Linear(fused, [n])
    # [K, N_i] int8 tensors hold N_i/16 tile rows of K*16 bytes each: concatenate the raw bytes
    flat = torch.cat([p.qweight.contiguous().reshape(-1) for p in parts])
    fused.qweight = flat.reshape(k, n)
    fused.weight_scales = torch.cat([p.weight_scales for p in parts])
    if has_bias:
        fused.bias = torch.cat([p.bias for p in parts])
    return FusedW8A16Linear(fused, [p.out_features for p in parts])
