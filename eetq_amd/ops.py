"""The operator surface of the package: the reference's module ``EETQ`` (csrc/eetpy.cpp:7-19) plus this library's extensions.

Two bindings of the same C ABI (include/eetq_amd.h) exist:
  * ``ext``    -- the compiled module ``EETQ`` (eetq_amd/csrc/torch_ext.cpp, pybind11 + torch tensors): the product boundary,
                  what ``import EETQ`` gives, a few microseconds of host time per call;
  * ``ctypes`` -- eetq_amd/ops_ctypes.py, pure Python: fallback when the module cannot be built (no C++ compiler) and an
                  independent second binding for the tests.
``EETQ_AMD_BOUNDARY=ext|ctypes`` forces one; by default the compiled module is used whenever it loads.  Neither has a CPU
fallback: without libeetq_amd.so or without a GPU the operators raise.
"""
import os

_want = os.environ.get("EETQ_AMD_BOUNDARY", "").strip().lower()
if _want not in ("", "ext", "ctypes"):
    raise ImportError("EETQ_AMD_BOUNDARY must be 'ext' or 'ctypes' (got %r)" % _want)

_C = None
_ext_error = None
if _want != "ctypes":
    try:
        from . import _ext
        _C = _ext.load()
    except Exception as e:  # no compiler / stale module that cannot be rebuilt: fall back unless the caller insisted
        if _want == "ext":
            raise
        _ext_error = e

if _C is not None:
    BOUNDARY = "ext"
    quant_weights = _C.quant_weights
    preprocess_weights = _C.preprocess_weights
    unprocess_weights = _C.unprocess_weights
    w8_a16_gemm = _C.w8_a16_gemm
    w8_a16_gemm_ = _C.w8_a16_gemm_
    layernorm_forward = _C.layernorm_forward
    rotary_embedding_neox = _C.rotary_embedding_neox
    rotary_embedding_neox_strided = _C.rotary_embedding_neox_strided
    rotary_embedding_neox_kvcache = _C.rotary_embedding_neox_kvcache
    rotary_embedding_neox_kvcache_prefill = _C.rotary_embedding_neox_kvcache_prefill
    greedy_handover = _C.greedy_handover
    decode_attention = _C.decode_attention
    rope_decode_attention = _C.rope_decode_attention
    silu_mul = _C.silu_mul
    prefill_attention = _C.prefill_attention                        # compiled boundary only (None -> torch's attention)
    prefill_attention_supported = _C.prefill_attention_supported
    w8_a16_gemv_grouped = _C.w8_a16_gemv_grouped
    llama_decode_layer = _C.llama_decode_layer   # compiled boundary only: its point is the interpreter time it saves
else:
    BOUNDARY = "ctypes"
    from .ops_ctypes import (decode_attention, greedy_handover, layernorm_forward, preprocess_weights, quant_weights,  # noqa: F401
                             rope_decode_attention, rotary_embedding_neox, rotary_embedding_neox_kvcache,
                             rotary_embedding_neox_kvcache_prefill, rotary_embedding_neox_strided, silu_mul,
                             unprocess_weights, w8_a16_gemm, w8_a16_gemm_, w8_a16_gemv_grouped)
    llama_decode_layer = None
    prefill_attention = None
    prefill_attention_supported = None

__all__ = ["quant_weights", "preprocess_weights", "unprocess_weights", "w8_a16_gemm", "w8_a16_gemm_", "layernorm_forward",
           "rotary_embedding_neox", "rotary_embedding_neox_strided", "rotary_embedding_neox_kvcache", "rotary_embedding_neox_kvcache_prefill",
           "greedy_handover", "decode_attention", "rope_decode_attention", "silu_mul", "convert_layout", "w8_a16_gemv_grouped", "decode_dropped_steps",
           "release_stream_workspace", "release_workspace", "BOUNDARY"]


def decode_dropped_steps(reset=True, device=None):
    """Decode steps the KV-cache kernels skipped because the cache row lay outside the cache (a full static cache): 0 unless
    generation overran ``max_cache_len`` -- then every token after that point is wrong.  Synchronises the device."""
    import ctypes

    import torch

    from . import _lib
    n = ctypes.c_ulonglong(0)
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _lib.check(_lib.lib().eetq_decode_dropped_steps(ctypes.byref(n), 1 if reset else 0))
    return int(n.value)


def release_stream_workspace(stream=None):
    """Hand the split-K scratch region a stream owns (40 MiB; split-K launches -- medium-batch GEMMs, 9 <= M <= 128 on some
    shapes, and the K-sliced tiled kernel -- take one per launch stream, at most 16 per device) back to the pool -- the library never reclaims one on its own, so a server that
    creates and destroys streams calls this before destroying one (unless HIP graphs captured on it are still replayed).
    Synchronises the stream.  ``stream``: a ``torch.cuda.Stream`` (default: the current one).  eetq_release_stream_workspace."""
    import ctypes

    import torch

    from . import _lib
    stream = stream if stream is not None else torch.cuda.current_stream()
    with torch.cuda.device(stream.device):
        _lib.check(_lib.lib().eetq_release_stream_workspace(ctypes.c_void_p(stream.cuda_stream)))


def release_workspace():
    """Free ALL library-owned scratch on every device (split-K regions, W4A16 expansion buffers, the quantiser's NULL-workspace
    buffer); returns the bytes freed.  Not while a HIP graph that captured a split-K / W4A16 launch is still going to be
    replayed.  eetq_release_workspace."""
    import ctypes

    from . import _lib
    n = ctypes.c_size_t(0)
    _lib.check(_lib.lib().eetq_release_workspace(ctypes.byref(n)))
    return int(n.value)


def convert_layout(weight, src_layout, dst_layout, is_int4=False):
    """Re-encode a processed weight, e.g. an NVIDIA-written EETQ checkpoint ('sm80') -> 'gfx950'."""
    if is_int4:
        return preprocess_weights(unprocess_weights(weight, src_layout, True), True, dst_layout)
    return preprocess_weights(unprocess_weights(weight, src_layout), False, dst_layout)
