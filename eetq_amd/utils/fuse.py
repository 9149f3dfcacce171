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


def fuse_w8a16_linears(parts):
    """parts: W8A16Linear modules with the same in_features, device and bias-ness -> FusedW8A16Linear."""
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
    # [K, N_i] int8 tensors hold N_i/16 tile rows of K*16 bytes each: concatenate the raw bytes
    flat = torch.cat([p.qweight.contiguous().reshape(-1) for p in parts])
    fused.qweight = flat.reshape(k, n)
    fused.weight_scales = torch.cat([p.weight_scales for p in parts])
    if has_bias:
        fused.bias = torch.cat([p.bias for p in parts])
    return FusedW8A16Linear(fused, [p.out_features for p in parts])
