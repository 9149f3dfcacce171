"""The reference's native operator surface, re-hosted on libeetq_amd.so -- the ctypes boundary.

The product boundary is the compiled module ``EETQ`` (eetq_amd/csrc/torch_ext.cpp); this pure-Python twin calls the same
C ABI through ctypes and is kept as the fallback when no C++ compiler is available and as an independent second binding
for the tests (``EETQ_AMD_BOUNDARY=ctypes``).  Both expose identical names, arguments and error behaviour.

Mirrors the pybind module ``EETQ`` (/root/reference/csrc/eetpy.cpp:7-19): same function names, argument
order, defaults and error type (RuntimeError).  torch is used for tensors, device memory and the current
stream only; all arithmetic happens in the HIP kernels behind the C ABI (include/eetq_amd.h).  There is
no CPU fallback: without the shared library or without a GPU these functions raise.

Differences from the reference, all supersets or documented:
  * ``quant_weights`` / ``preprocess_weights`` also accept GPU tensors (results stay on that GPU); CPU tensors
    in -> CPU tensors out exactly like the reference (fpA_intB_gemm_wrapper.cu:33,113), staged through the GPU.
  * the processed layout is this library's gfx950 layout (the reference's is a function of the CUDA SM
    version, cutlass_preprocessors.cc:113-128); ``layout="sm80"`` reproduces the reference's sm75..sm89 bytes.
  * int4 (quint4x2) goes through the compiled module only.
"""
import ctypes

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_IDENTITY, ACT_RELU, ACT_SILU, DTYPE_F16, DTYPE_F32, LAYOUT_GFX950, LAYOUT_ROW_MAJOR,
                   LAYOUT_SM80, PATH_AUTO, PATH_GEMV, PATH_MFMA, check)

__all__ = ["quant_weights", "preprocess_weights", "unprocess_weights", "w8_a16_gemm", "w8_a16_gemm_",
           "layernorm_forward", "rotary_embedding_neox", "rotary_embedding_neox_strided", "rotary_embedding_neox_kvcache", "rotary_embedding_neox_kvcache_prefill", "greedy_handover", "decode_attention", "rope_decode_attention", "silu_mul", "convert_layout", "w8_a16_gemv_grouped"]

_LAYOUTS = {"gfx950": LAYOUT_GFX950, "native": LAYOUT_GFX950, "sm80": LAYOUT_SM80, "row_major": LAYOUT_ROW_MAJOR,
            LAYOUT_GFX950: LAYOUT_GFX950, LAYOUT_SM80: LAYOUT_SM80, LAYOUT_ROW_MAJOR: LAYOUT_ROW_MAJOR}
_PATHS = {"auto": PATH_AUTO, "gemv": PATH_GEMV, "mfma": PATH_MFMA, "stream": _lib.PATH_STREAM, "mid": _lib.PATH_MID,
          "splitk": _lib.PATH_SPLITK, "tilesplit": _lib.PATH_TILESPLIT}
_ACTS = {"": ACT_IDENTITY, "identity": ACT_IDENTITY, "none": ACT_IDENTITY, "relu": ACT_RELU, "gelu": ACT_GELU,
         "silu": ACT_SILU}


def _eager_only(fn):
    """The operators call the library through ctypes with raw pointers: keep torch.compile (which transformers turns on by
    itself for `generate(cache_implementation="static")`) from tracing into them -- it breaks the graph around the call."""
    disable = getattr(getattr(torch, "compiler", None), "disable", None)
    return disable(fn) if disable is not None else fn


def _layout_id(layout):
    try:
        return _LAYOUTS[layout]
    except KeyError:
        raise RuntimeError("unknown weight layout %r (expected 'gfx950', 'sm80' or 'row_major')" % (layout,))


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("eetq_amd: no HIP device available; the W8A16 path has no CPU implementation")


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _work_device(t):
    if t.is_cuda:
        return t.device
    _require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


@_eager_only
def quant_weights(origin_weight, quant_type, return_unprocessed_quantized_tensor=False, layout="gfx950"):
    """Per-column symmetric int8 quantisation of a [K, N] fp16/fp32 weight.

    Reference: symmetric_quantize_last_axis_of_tensor, fpA_intB_gemm_wrapper.cu:28-107.  Returns
    ``[processed_int8 [K, N], scales [N]]`` or, with the flag, ``[raw_int8, processed_int8, scales]``
    (scales have the dtype of the weight).  raw int8 and scales are bit-exact with the reference.
    """
    weight = origin_weight
    if not isinstance(weight, torch.Tensor):
        raise RuntimeError("quant_weights(): origin_weight must be a torch.Tensor")
    if not weight.is_contiguous():
        raise RuntimeError("weight must be contiguous")
    if weight.numel() == 0:
        raise RuntimeError("weight should not be empty tensor")
    if weight.dim() not in (2, 3):
        raise RuntimeError("Invalid dim. The dim of weight should be 2 or 3")
    if weight.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("Invalid datatype. Weight must be FP16 or FP32")
    if quant_type == torch.quint4x2:
        raise RuntimeError("eetq_amd: int4 (quint4x2) weight-only quantization is not implemented (W8A16 only)")
    if quant_type != torch.int8:
        raise RuntimeError("Must be int4 or int8 quantization")
    if weight.dim() == 3:
        # [E, K, N] expert stack: the reference allocates [E, K, N] / [E, N] outputs and quantises expert 0 only
        # (fpA_intB_gemm_wrapper.cu:45-66, :82, :90 pass the 2-D shape); here every expert is quantised
        parts = [quant_weights(weight[e], quant_type, return_unprocessed_quantized_tensor, layout)
                 for e in range(weight.shape[0])]
        return [torch.stack([p[i] for p in parts], 0) for i in range(len(parts[0]))]
    lay = _layout_id(layout)
    K, N = weight.shape
    dev = _work_device(weight)
    with torch.cuda.device(dev):
        w_dev = weight if weight.is_cuda else weight.to(dev, non_blocking=False)
        raw = torch.empty((K, N), dtype=torch.int8, device=dev) if return_unprocessed_quantized_tensor else None
        processed = torch.empty((K, N), dtype=torch.int8, device=dev)
        scales = torch.empty((N,), dtype=weight.dtype, device=dev)
        colmax = torch.empty((_lib.lib().eetq_quantize_workspace_floats(K, N),), dtype=torch.float32, device=dev)
        check(_lib.lib().eetq_quantize_i8_ws(
            _ptr(w_dev), DTYPE_F16 if weight.dtype == torch.float16 else DTYPE_F32, K, N,
            _ptr(raw) if raw is not None else None, _ptr(processed), lay, _ptr(scales), _ptr(colmax), colmax.numel(),
            _stream_ptr()))
        if not weight.is_cuda:
            processed, scales = processed.cpu(), scales.cpu()
            raw = raw.cpu() if raw is not None else None
    if return_unprocessed_quantized_tensor:
        return [raw, processed, scales]
    return [processed, scales]


def _relayout(src, layout, pack):
    if not isinstance(src, torch.Tensor) or src.dtype != torch.int8:
        raise RuntimeError("expected an int8 tensor")
    if src.dim() < 2:
        raise RuntimeError("Shape must be 2-D")
    if not src.is_contiguous():
        src = src.contiguous()
    lay = _layout_id(layout)
    K, N = src.shape[-2], src.shape[-1]
    if src.dim() != 2:
        raise RuntimeError("[FT][ERROR] Shape must be 2-D")
    dev = _work_device(src)
    with torch.cuda.device(dev):
        s_dev = src if src.is_cuda else src.to(dev)
        out = torch.empty_like(s_dev)
        fn = _lib.lib().eetq_pack_i8 if pack else _lib.lib().eetq_unpack_i8
        check(fn(_ptr(s_dev), K, N, _ptr(out), lay, _stream_ptr()))
        return out if src.is_cuda else out.cpu()


@_eager_only
def preprocess_weights(origin_weight, is_int4=False, layout="gfx950"):
    """Row-major int8 [K, N] -> processed layout (same shape, re-ordered bytes).

    Reference: preprocess_weights_cuda, fpA_intB_gemm_wrapper.cu:109-128.
    """
    if is_int4:
        raise RuntimeError("eetq_amd: int4 weights are not implemented (W8A16 only)")
    return _relayout(origin_weight, layout, pack=True)


@_eager_only
def unprocess_weights(processed_weight, layout="gfx950"):
    """Inverse of :func:`preprocess_weights` (no reference counterpart; needed for checkpoint interop)."""
    return _relayout(processed_weight, layout, pack=False)


def convert_layout(weight, src_layout, dst_layout):
    """Re-encode a processed int8 weight, e.g. an NVIDIA-written EETQ checkpoint ('sm80') -> 'gfx950'."""
    return preprocess_weights(unprocess_weights(weight, src_layout), False, dst_layout)


@_eager_only
def _gemm_launch(input, weight, scale, output, m, n, k, path, bias=None, residual=None, act=ACT_IDENTITY):
    if input.dtype != torch.float16:
        raise RuntimeError("w8_a16_gemm: input must be float16 (got %s)" % input.dtype)
    if not input.is_cuda:
        raise RuntimeError("input must be a CUDA tensor")
    if weight.dtype != torch.int8 or scale.dtype != torch.float16:
        raise RuntimeError("w8_a16_gemm: weight must be int8 and scale float16")
    if weight.device != input.device or scale.device != input.device or output.device != input.device:
        raise RuntimeError("w8_a16_gemm: input, weight, scale and output must be on the same device")
    if not weight.is_contiguous() or not scale.is_contiguous() or not output.is_contiguous():
        raise RuntimeError("w8_a16_gemm: weight, scale and output must be contiguous")
    x = input if input.is_contiguous() else input.contiguous()
    if bias is not None:
        if bias.dtype != torch.float16 or bias.device != input.device or bias.numel() != n or not bias.is_contiguous():
            raise RuntimeError("w8_a16_gemm: bias must be a contiguous float16 [N] tensor on the input's device")
    if residual is not None:
        if (residual.dtype != torch.float16 or residual.device != input.device or residual.numel() != m * n
                or not residual.is_contiguous() or residual.shape[-1] != n):
            raise RuntimeError("w8_a16_gemm: residual must be a contiguous float16 [..., N] tensor with the output's "
                               "element count, on the input's device")
    with torch.cuda.device(input.device):
        check(_lib.lib().eetq_w8a16_gemm_act(_ptr(x), _ptr(weight), _ptr(scale), _ptr(bias) if bias is not None else None,
                                             _ptr(residual) if residual is not None else None, _ptr(output), m, n, k, path,
                                             act, _stream_ptr()))
    return output


def w8_a16_gemm(input, weight, scale, path="auto", bias=None, residual=None, norm=None, gated=False, activation=""):
    """``y = input @ dequant(weight, scale) (+ bias)``: fp16 [..., K] x int8 [K, N] (processed) -> fp16 [..., N].

    Reference: w8_a16_gemm_forward_cuda, fpA_intB_gemm_wrapper.cu:130-173 (fresh output tensor, current
    stream, asynchronous).  ``path`` ("auto" | "gemv" | "stream" | "mfma" | "mid" | "splitk" | "tilesplit") is a testing hook.  ``bias`` (extension,
    SURVEY 8f row 3) fuses the reference's separate ``output + bias`` into the kernel epilogue, bit-identically;
    ``residual`` (same shape as the output) is added after it, again in fp16 -- the decoder block's ``residual + proj(x)``.
    ``norm=(gamma, eps)`` (extension) RMS-normalises the input first: inside the GEMV launch for a single row, as a
    separate ``layernorm_forward`` otherwise.  ``gated=True`` (extension): ``input`` is a fused gate|up row block [..., 2K]
    and the GEMM runs on ``silu(gate) * up`` -- inside the GEMV launch for a single row, through ``silu_mul`` otherwise.
    ``activation`` ("relu" | "gelu" | "silu"; extension): FT's bias + activation epilogue, ``fp16(act(acc + bias))``
    (csrc/cutlass_kernels/fpA_intB_gemm.cu:35-62), which the reference compiles but never binds.
    """
    if activation == "silu_glu8":
        # gated MLP over a weight in "glu8" column order (groups of 16 = 8 gate + the 8 matching up columns): one launch for
        # a single row (activation in the GEMV epilogue, RMS-norm in its prologue), projection + silu_mul(glu8) otherwise
        k, n = weight.shape[-2], scale.numel()
        if gated or residual is not None or n % 16 or weight.shape[-1] != n:
            raise RuntimeError("w8_a16_gemm: silu_glu8 takes an int8 [K, N] weight with N % 16 == 0, no residual")
        rows = input.numel() // k if k else 0
        gamma = norm[0] if norm is not None else None
        if (rows == 1 and path == "auto" and input.shape[-1] == k and input.is_cuda and input.dtype == torch.float16
                and (gamma is None or (gamma.dtype == torch.float16 and gamma.is_contiguous() and gamma.numel() == k
                                       and gamma.device == input.device))):
            return _gemv_glu8_launch(input, gamma, norm[1] if norm is not None else 0.0, weight, scale, bias, n, k)
        if (rows >= 2 and path == "auto" and input.shape[-1] == k and input.is_cuda and input.dtype == torch.float16):
            # batched decode: the small-batch kernel's epilogue; prompts: the tiled MFMA kernel's gated write-out where AUTO
            # runs the shape on it (None: no such form, two launches)
            out = _gemm_glu8_launch(input, norm, weight, scale, bias, rows, n, k)
            if out is not None:
                return out
        return silu_mul(w8_a16_gemm(input, weight, scale, path, bias, None, norm), glu8=True)
    if activation not in _ACTS:
        raise RuntimeError("unknown activation %r (identity, relu, gelu, silu; silu_glu8 for gated weights)" % (activation,))
    act = _ACTS[activation]
    if gated:
        k2 = input.shape[-1]
        k = weight.shape[-2]
        if k2 != 2 * k:
            raise RuntimeError("w8_a16_gemm: gated input must be [..., 2K] for a [K, N] weight")
        rows = input.numel() // k2 if k2 else 0
        if (rows == 1 and path == "auto" and norm is None and input.is_cuda and input.dtype == torch.float16
                and input.is_contiguous() and k % 8 == 0 and not act):
            n = weight.shape[-1]
            output = torch.empty(tuple(input.shape[:-1]) + (n,), dtype=input.dtype, device=input.device)
            return _gemv_gated_launch(input, weight, scale, output, n, k, bias, residual)
        input = silu_mul(input if input.is_contiguous() else input.contiguous())
    k = input.shape[-1]
    n = weight.shape[-1]
    if weight.shape[-2] != k:
        raise RuntimeError("w8_a16_gemm: weight is [%d, %d] but input has K=%d" % (weight.shape[-2], n, k))
    m = input.numel() // k if k else 0
    output = torch.empty(tuple(input.shape[:-1]) + (n,), dtype=input.dtype, device=input.device)
    if m == 0:
        return output
    if norm is not None:
        gamma, eps = norm
        if (m == 1 and path == "auto" and gamma.dtype == torch.float16 and gamma.is_contiguous() and gamma.numel() == k
                and not act):
            return _gemv_rmsnorm_launch(input, gamma, eps, weight, scale, output, n, k, bias, residual)
        normed = torch.empty_like(input if input.is_contiguous() else input.contiguous())
        layernorm_forward(input if input.is_contiguous() else input.contiguous(), gamma, normed, eps)
        input = normed
    return _gemm_launch(input, weight, scale, output, m, n, k, _PATHS[path], bias, residual, act)


@_eager_only
def _gemm_glu8_launch(input, norm, weight, scale, bias, rows, n, k):
    if weight.dtype != torch.int8 or scale.dtype != torch.float16 or not weight.is_contiguous():
        raise RuntimeError("w8_a16_gemm: weight must be contiguous int8 and scale float16")
    for t in (weight, scale) + ((bias,) if bias is not None else ()):
        if t.device != input.device:
            raise RuntimeError("w8_a16_gemm: all tensors must be on the input's device")
    if bias is not None and (bias.dtype != torch.float16 or bias.numel() != n or not bias.is_contiguous()):
        raise RuntimeError("w8_a16_gemm: bias must be a contiguous float16 [N] tensor")
    x = input if input.is_contiguous() else input.contiguous()
    if norm is not None:
        normed = torch.empty_like(x)
        layernorm_forward(x, norm[0], normed, norm[1])
        x = normed
    output = torch.empty(tuple(input.shape[:-1]) + (n // 2,), dtype=input.dtype, device=input.device)
    with torch.cuda.device(input.device):
        st = _lib.lib().eetq_w8a16_gemm_glu8(_ptr(x), _ptr(weight), _ptr(scale), _ptr(bias) if bias is not None else None,
                                             _ptr(output), rows, n, k, _stream_ptr())
    if st == _lib.ERR_UNSUPPORTED and rows > 16:
        return None
    check(st)
    return output


@_eager_only
def _gemv_glu8_launch(input, gamma, eps, weight, scale, bias, n, k):
    if weight.dtype != torch.int8 or scale.dtype != torch.float16 or not weight.is_contiguous():
        raise RuntimeError("w8_a16_gemm: weight must be contiguous int8 and scale float16")
    for t in (weight, scale) + ((bias,) if bias is not None else ()):
        if t.device != input.device:
            raise RuntimeError("w8_a16_gemm: all tensors must be on the input's device")
    if bias is not None and (bias.dtype != torch.float16 or bias.numel() != n or not bias.is_contiguous()):
        raise RuntimeError("w8_a16_gemm: bias must be a contiguous float16 [N] tensor")
    x = input if input.is_contiguous() else input.contiguous()
    output = torch.empty(tuple(input.shape[:-1]) + (n // 2,), dtype=input.dtype, device=input.device)
    with torch.cuda.device(input.device):
        check(_lib.lib().eetq_w8a16_gemv_glu8(_ptr(x), _ptr(gamma) if gamma is not None else None, float(eps), _ptr(weight),
                                              _ptr(scale), _ptr(bias) if bias is not None else None, _ptr(output), n, k,
                                              _stream_ptr()))
    return output


@_eager_only
def _gemv_gated_launch(gate_up, weight, scale, output, n, k, bias, residual):
    if weight.dtype != torch.int8 or scale.dtype != torch.float16 or not weight.is_contiguous():
        raise RuntimeError("w8_a16_gemm: weight must be contiguous int8 and scale float16")
    for t in (weight, scale, output) + ((bias,) if bias is not None else ()) + ((residual,) if residual is not None else ()):
        if t.device != gate_up.device:
            raise RuntimeError("w8_a16_gemm: all tensors must be on the input's device")
    if bias is not None and (bias.dtype != torch.float16 or bias.numel() != n or not bias.is_contiguous()):
        raise RuntimeError("w8_a16_gemm: bias must be a contiguous float16 [N] tensor")
    if residual is not None and (residual.dtype != torch.float16 or residual.numel() != n or not residual.is_contiguous()):
        raise RuntimeError("w8_a16_gemm: residual must be a contiguous float16 tensor with the output's element count")
    with torch.cuda.device(gate_up.device):
        check(_lib.lib().eetq_w8a16_gemv_silu_gated(_ptr(gate_up), _ptr(weight), _ptr(scale),
                                                    _ptr(bias) if bias is not None else None,
                                                    _ptr(residual) if residual is not None else None, _ptr(output), n, k,
                                                    _stream_ptr()))
    return output


@_eager_only
def _gemv_rmsnorm_launch(input, gamma, eps, weight, scale, output, n, k, bias, residual):
    if input.dtype != torch.float16 or not input.is_cuda:
        raise RuntimeError("w8_a16_gemm: input must be a float16 CUDA tensor")
    if weight.dtype != torch.int8 or scale.dtype != torch.float16 or not weight.is_contiguous():
        raise RuntimeError("w8_a16_gemm: weight must be contiguous int8 and scale float16")
    for t in (weight, scale, gamma, output) + ((bias,) if bias is not None else ()) + ((residual,) if residual is not None else ()):
        if t.device != input.device:
            raise RuntimeError("w8_a16_gemm: all tensors must be on the input's device")
    if bias is not None and (bias.dtype != torch.float16 or bias.numel() != n or not bias.is_contiguous()):
        raise RuntimeError("w8_a16_gemm: bias must be a contiguous float16 [N] tensor")
    if residual is not None and (residual.dtype != torch.float16 or residual.numel() != n or not residual.is_contiguous()):
        raise RuntimeError("w8_a16_gemm: residual must be a contiguous float16 tensor with the output's element count")
    x = input if input.is_contiguous() else input.contiguous()
    with torch.cuda.device(input.device):
        check(_lib.lib().eetq_w8a16_gemv_rmsnorm(_ptr(x), _ptr(gamma), float(eps), _ptr(weight), _ptr(scale),
                                                 _ptr(bias) if bias is not None else None,
                                                 _ptr(residual) if residual is not None else None, _ptr(output), n, k,
                                                 _stream_ptr()))
    return output


def w8_a16_gemm_(input, weight, scale, output, m, n, k):
    """In-place variant writing into ``output`` (reference: w8_a16_gemm_forward_cuda_, :176-202)."""
    return _gemm_launch(input, weight, scale, output, int(m), int(n), int(k), PATH_AUTO)


@_eager_only
def layernorm_forward(input, gamma, out, eps):
    """T5/RMS layernorm into ``out`` (reference: layernorm_forward_cuda, layernorm.cu:98-113). Returns None."""
    if input.dtype != torch.float16 or gamma.dtype != torch.float16 or out.dtype != torch.float16:
        raise RuntimeError("layernorm_forward: expected scalar type Half")
    if not (input.is_cuda and gamma.is_cuda and out.is_cuda):
        raise RuntimeError("layernorm_forward: tensors must be CUDA tensors")
    if not (input.is_contiguous() and gamma.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("layernorm_forward: tensors must be contiguous")
    cols = input.shape[-1]
    rows = input.numel() // cols if cols else 0
    if gamma.numel() != cols or out.numel() != input.numel():
        raise RuntimeError("layernorm_forward: shape mismatch")
    with torch.cuda.device(input.device):
        check(_lib.lib().eetq_rmsnorm_f16(_ptr(input), _ptr(gamma), _ptr(out), float(eps), rows, cols, _stream_ptr()))
    return None


class _GemvProblem(ctypes.Structure):   # eetq_gemv_problem (include/eetq_amd.h)
    _fields_ = [("x", ctypes.c_void_p), ("w_packed", ctypes.c_void_p), ("scales", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("N", ctypes.c_int), ("K", ctypes.c_int)]


@_eager_only
def w8_a16_gemv_grouped(inputs, weights, scales, biases=None, residuals=None):
    """Independent single-row W8A16 problems in as few dispatches as possible (extension, eetq_w8a16_gemv_grouped):
    inputs[i] fp16 with K_i elements, weights[i] processed int8 [K_i, N_i], scales[i] fp16 [N_i]; returns fresh outputs."""
    n = len(inputs)
    if len(weights) != n or len(scales) != n:
        raise RuntimeError("w8_a16_gemv_grouped: inputs, weights and scales must have one entry per problem")
    if (biases is not None and len(biases) != n) or (residuals is not None and len(residuals) != n):
        raise RuntimeError("w8_a16_gemv_grouped: one bias / residual entry (or None) per problem")
    if n == 0:
        return []
    dev = inputs[0].device
    probs = (_GemvProblem * n)()
    outs, keep = [], []
    for i, (x, w, s) in enumerate(zip(inputs, weights, scales)):
        if x.dtype != torch.float16:
            raise RuntimeError("w8_a16_gemm: input must be float16 (got %s)" % x.dtype)
        if not x.is_cuda:
            raise RuntimeError("input must be a CUDA tensor")
        if w.dtype != torch.int8 or s.dtype != torch.float16:
            raise RuntimeError("w8_a16_gemm: weight must be int8 and scale float16")
        if w.dim() != 2 or not w.is_contiguous() or not s.is_contiguous():
            raise RuntimeError("w8_a16_gemm: weight [K, N] and scale must be contiguous")
        if x.device != dev or w.device != dev or s.device != dev:
            raise RuntimeError("w8_a16_gemv_grouped: all tensors must be on one device")
        K, N = w.shape
        if x.numel() != K or x.shape[-1] != K:
            raise RuntimeError("w8_a16_gemv_grouped: every input must be ONE row of K elements")
        if s.numel() != N:
            raise RuntimeError("w8_a16_gemm: scale must have N elements")
        xc = x if x.is_contiguous() else x.contiguous()
        y = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float16, device=dev)
        b = biases[i] if biases is not None else None
        r = residuals[i] if residuals is not None else None
        if b is not None and (b.dtype != torch.float16 or b.device != dev or b.numel() != N or not b.is_contiguous()):
            raise RuntimeError("w8_a16_gemm: bias must be a contiguous float16 [N] tensor on the input's device")
        if r is not None and (r.dtype != torch.float16 or r.device != dev or r.numel() != N or not r.is_contiguous()
                              or r.shape[-1] != N):
            raise RuntimeError("w8_a16_gemm: residual must be a contiguous float16 [..., N] tensor with the output's "
                               "element count, on the input's device")
        keep.append(xc)
        outs.append(y)
        probs[i] = _GemvProblem(xc.data_ptr(), w.data_ptr(), s.data_ptr(), y.data_ptr(),
                                b.data_ptr() if b is not None else None, r.data_ptr() if r is not None else None, N, K)
    with torch.cuda.device(dev):
        check(_lib.lib().eetq_w8a16_gemv_grouped(probs, n, _stream_ptr()))
    return outs


@_eager_only
def rotary_embedding_neox(positions, query, key, head_size, cos_sin_cache):
    """In-place NeoX rotary embedding of query/key (reference: pos_encoding_kernels.cu:55-87): float16, float32, float64."""
    dts = {torch.float16: 0, torch.float32: 1, torch.float64: 2}
    if query.dtype not in dts:
        raise RuntimeError("eetq_amd: rotary_embedding_neox is implemented for float16, float32 and float64")
    if key.dtype != query.dtype or cos_sin_cache.dtype != query.dtype:
        raise RuntimeError("rotary_embedding_neox: query, key and cos_sin_cache must share one dtype")
    if positions.dtype != torch.int64:
        raise RuntimeError("rotary_embedding_neox: positions must be int64")
    if not (query.is_contiguous() and key.is_contiguous() and cos_sin_cache.is_contiguous()
            and positions.is_contiguous()):
        raise RuntimeError("rotary_embedding_neox: tensors must be contiguous")
    tokens = query.shape[0] * query.shape[1]
    rot_dim = cos_sin_cache.shape[1]
    heads = query.shape[-2]
    with torch.cuda.device(query.device):
        check(_lib.lib().eetq_rotary_neox(_ptr(positions), _ptr(query), _ptr(key), _ptr(cos_sin_cache), dts[query.dtype],
                                          tokens, heads, int(head_size), rot_dim, _stream_ptr()))
    return None


@_eager_only
def rotary_embedding_neox_strided(positions, query, key, head_size, cos_sin_cache):
    """In-place NeoX rotary embedding of strided views: query [..., q_heads, head_size] and key [..., k_heads, head_size]
    whose last two dimensions are dense and whose leading (token) dimensions share one stride -- e.g. the q and k
    slices of a fused QKV projection output.  k_heads may differ from q_heads (grouped-query attention)."""
    if query.dtype != torch.float16 or key.dtype != torch.float16 or cos_sin_cache.dtype != torch.float16:
        raise RuntimeError("eetq_amd: rotary_embedding_neox is implemented for float16 only")
    if positions.dtype != torch.int64 or not positions.is_contiguous() or not cos_sin_cache.is_contiguous():
        raise RuntimeError("rotary_embedding_neox_strided: positions must be contiguous int64, the cache contiguous")

    def token_view(t):
        heads, hs = t.shape[-2], t.shape[-1]
        if hs != head_size or t.stride(-1) != 1 or t.stride(-2) != hs:
            raise RuntimeError("rotary_embedding_neox_strided: the last two dimensions must be dense [heads, head_size]")
        tokens = 1
        for d in t.shape[:-2]:
            tokens *= d
        # the leading dimensions must collapse to one stride
        stride, expect = None, None
        for d, st in zip(reversed(t.shape[:-2]), reversed(t.stride()[:-2])):
            if d == 1:
                continue
            if stride is None:
                stride, expect = st, st * d
            elif st != expect:
                raise RuntimeError("rotary_embedding_neox_strided: leading dimensions do not collapse to one stride")
            else:
                expect = st * d
        return tokens, heads, (stride if stride is not None else heads * hs)

    tq, hq, sq = token_view(query)
    tk, hk, sk = token_view(key)
    if tq != tk or positions.numel() != tq:
        raise RuntimeError("rotary_embedding_neox_strided: query, key and positions disagree on the token count")
    with torch.cuda.device(query.device):
        check(_lib.lib().eetq_rotary_neox_strided_f16(_ptr(positions), _ptr(query), _ptr(key), _ptr(cos_sin_cache), tq,
                                                      hq, hk, int(head_size), cos_sin_cache.shape[1], sq, sk,
                                                      _stream_ptr()))
    return None


@_eager_only
def decode_attention(query, key_cache, value_cache, mask=None, scaling=None, splits=None, kv_len=None, kv_len_bias=0,
                     advance=None):
    """Single-query attention over a KV cache (extension; the decode step of the EET attention blocks).

    query [B, H, D] (any batch/head strides, dense D), key_cache / value_cache [B, Hkv, S, D] (dense D), mask: additive
    float16 [B, S] or [1, S] (or broadcastable [B, 1, 1, S]) with -inf at masked positions, or None.  ``kv_len`` (int64
    device scalar): only rows below ``kv_len + kv_len_bias`` are attended (the filled part of a static cache); ``advance``
    (int64 device scalar, may be the same tensor): incremented by one when the attention has been computed.  Returns
    float16 [B, H, D].  fp32 softmax and accumulation; D must be 64 or 128."""
    if query.dtype != torch.float16 or key_cache.dtype != torch.float16 or value_cache.dtype != torch.float16:
        raise RuntimeError("decode_attention: query and caches must be float16")
    if query.dim() != 3 or key_cache.dim() != 4 or value_cache.shape != key_cache.shape:
        raise RuntimeError("decode_attention: expected query [B, H, D] and caches [B, Hkv, S, D]")
    B, H, D = query.shape
    Bk, Hkv, S, Dk = key_cache.shape
    if Bk != B or Dk != D or H % Hkv or query.stride(-1) != 1 or key_cache.stride(-1) != 1 or value_cache.stride(-1) != 1:
        raise RuntimeError("decode_attention: shape / stride mismatch")
    mrow, m_sb = None, 0
    if mask is not None:
        mrow = mask.reshape(mask.shape[0], -1) if mask.dim() != 2 else mask
        if mrow.dtype != torch.float16 or mrow.shape[-1] < S or mrow.stride(-1) != 1 or mrow.device != query.device:
            raise RuntimeError("decode_attention: mask must be additive float16 with a dense last dimension >= S")
        if mrow.shape[0] not in (1, B):
            raise RuntimeError("decode_attention: the mask needs one row per batch entry (or a single shared row)")
        m_sb = mrow.stride(0) if mrow.shape[0] == B and B > 1 else 0
    for name, t in (("kv_len", kv_len), ("advance", advance)):
        if t is not None and (t.dtype != torch.int64 or t.numel() != 1 or t.device != query.device):
            raise RuntimeError("decode_attention: %s must be a one-element int64 tensor on the query's device" % name)
    if scaling is None:
        scaling = D ** -0.5
    if splits is None:  # the library's tuned default (about two workgroups per CU, at most 8 chunks)
        splits = _lib.lib().eetq_decode_attention_splits(B, H, S)
    out = torch.empty((B, H, D), dtype=torch.float16, device=query.device)
    ws = torch.empty((B * H * splits * (D + 4),), dtype=torch.float32, device=query.device)
    strides = (ctypes.c_long * 11)(query.stride(0), query.stride(1), key_cache.stride(0), key_cache.stride(1),
                                   key_cache.stride(2), value_cache.stride(0), value_cache.stride(1), value_cache.stride(2),
                                   m_sb, out.stride(0), out.stride(1))
    with torch.cuda.device(query.device):
        check(_lib.lib().eetq_decode_attention_f16(_ptr(query), _ptr(key_cache), _ptr(value_cache),
                                                   _ptr(mrow) if mrow is not None else None, _ptr(out), _ptr(ws), B, H,
                                                   Hkv, S, D, int(splits), float(scaling), strides,
                                                   _ptr(kv_len) if kv_len is not None else None, int(kv_len_bias),
                                                   _ptr(advance) if advance is not None else None, _stream_ptr()))
    return out


@_eager_only
def rope_decode_attention(positions, query, key, value, cos_sin_cache, key_cache, value_cache, tickets, slots=None,
                          mask=None, scaling=None, splits=None, kv_len=None, kv_len_bias=0, advance=None):
    """``rotary_embedding_neox_kvcache`` followed by ``decode_attention`` as ONE launch (extension; the decode step of a
    static KV cache).  Bit-identical to that pair in the cache rows written and in the returned [B, H, D] output; the
    rotated query is not written back.  ``tickets``: an int32 device tensor of at least B * H + 1 zeros that the launch
    leaves zeroed (launches that may overlap need their own).  The rotation must cover the whole head."""
    for t in (query, key, value, cos_sin_cache, key_cache, value_cache):
        if t.dtype != torch.float16:
            raise RuntimeError("rope_decode_attention: float16 tensors expected")
    if positions.dtype != torch.int64 or not positions.is_contiguous():
        raise RuntimeError("rope_decode_attention: positions must be contiguous int64")
    if query.dim() != 3 or key.dim() != 3 or value.dim() != 3 or key_cache.dim() != 4 or value_cache.shape != key_cache.shape:
        raise RuntimeError("rope_decode_attention: expected query [B, H, D], key / value [B, Hkv, D], caches [B, Hkv, S, D]")
    B, H, D = query.shape
    Hkv, S = key.shape[1], key_cache.shape[2]
    if (key.shape[0] != B or key.shape[2] != D or value.shape != key.shape or key_cache.shape[0] != B
            or key_cache.shape[1] != Hkv or key_cache.shape[3] != D or H % Hkv or positions.numel() != B):
        raise RuntimeError("rope_decode_attention: shape mismatch")
    for t in (query, key, value):
        if t.stride(-1) != 1 or t.stride(-2) != D:
            raise RuntimeError("rope_decode_attention: [heads, head_size] must be dense")
    if (key_cache.stride(-1) != 1 or value_cache.stride(-1) != 1 or not cos_sin_cache.is_contiguous()
            or cos_sin_cache.shape[-1] != D):
        raise RuntimeError("rope_decode_attention: cache rows must be dense and the rotation must cover the whole head "
                           "(rot_dim == D)")
    if (tickets.dtype != torch.int32 or not tickets.is_contiguous() or tickets.numel() < B * H + 1
            or tickets.device != query.device):
        raise RuntimeError("rope_decode_attention: tickets must be a zeroed int32 tensor of at least B * H + 1 elements "
                           "on the device")
    slot_stride = 0
    if slots is not None:
        if (slots.dtype != torch.int64 or slots.device != query.device or slots.numel() not in (1, B)
                or not slots.is_contiguous()):
            raise RuntimeError("rope_decode_attention: slots must be contiguous int64 on the device, 1 or B elements")
        slot_stride = 1 if (slots.numel() == B and B > 1) else 0
    mrow, m_sb = None, 0
    if mask is not None:
        mrow = mask.reshape(mask.shape[0], -1) if mask.dim() != 2 else mask
        if mrow.dtype != torch.float16 or mrow.shape[-1] < S or mrow.stride(-1) != 1 or mrow.device != query.device:
            raise RuntimeError("rope_decode_attention: mask must be additive float16 with a dense last dimension >= S")
        if mrow.shape[0] not in (1, B):
            raise RuntimeError("rope_decode_attention: the mask needs one row per batch entry (or a single shared row)")
        m_sb = mrow.stride(0) if mrow.shape[0] == B and B > 1 else 0
    for name, t in (("kv_len", kv_len), ("advance", advance)):
        if t is not None and (t.dtype != torch.int64 or t.numel() != 1 or t.device != query.device):
            raise RuntimeError("rope_decode_attention: %s must be a one-element int64 tensor on the query's device" % name)
    if scaling is None:
        scaling = D ** -0.5
    if splits is None:
        splits = _lib.lib().eetq_decode_attention_splits(B, H, S)
    out = torch.empty((B, H, D), dtype=torch.float16, device=query.device)
    ws = torch.empty((B * H * splits * (D + 4),), dtype=torch.float32, device=query.device)
    strides = (ctypes.c_long * 12)(query.stride(0), key.stride(0), value.stride(0), key_cache.stride(0),
                                   key_cache.stride(1), key_cache.stride(2), value_cache.stride(0), value_cache.stride(1),
                                   value_cache.stride(2), m_sb, out.stride(0), out.stride(1))
    with torch.cuda.device(query.device):
        check(_lib.lib().eetq_rope_decode_attention_f16(
            _ptr(positions), _ptr(slots) if slots is not None else None, slot_stride, _ptr(query), _ptr(key), _ptr(value),
            _ptr(cos_sin_cache), _ptr(key_cache), _ptr(value_cache), _ptr(mrow) if mrow is not None else None, _ptr(out),
            _ptr(ws), _ptr(tickets), B, H, Hkv, S, D, int(splits), float(scaling), strides,
            _ptr(kv_len) if kv_len is not None else None, int(kv_len_bias),
            _ptr(advance) if advance is not None else None, _stream_ptr()))
    return out


@_eager_only
def rotary_embedding_neox_kvcache(positions, query, key, value, head_size, cos_sin_cache, key_cache, value_cache,
                                  slots=None):
    """Decode step: rotate ``query`` [B, H, D] in place by ``positions`` [B] (int64), write the rotated ``key`` [B, Hkv, D]
    and ``value`` [B, Hkv, D] into the caches [B, Hkv, S, D] at row ``slots`` (int64 on the device: one element = the same
    row for the whole batch, e.g. a static cache's token counter, or [B]); ``slots=None`` writes at ``positions`` (they
    differ for left-padded batches).  One launch instead of rotary + two cache copies."""
    for t in (query, key, value, cos_sin_cache, key_cache, value_cache):
        if t.dtype != torch.float16:
            raise RuntimeError("rotary_embedding_neox_kvcache: float16 tensors expected")
    if positions.dtype != torch.int64 or not positions.is_contiguous():
        raise RuntimeError("rotary_embedding_neox_kvcache: positions must be contiguous int64")
    B, H, D = query.shape
    Hkv = key.shape[1]
    if (key.shape != (B, Hkv, D) or value.shape != (B, Hkv, D) or D != head_size or key_cache.dim() != 4
            or key_cache.shape[0] != B or key_cache.shape[1] != Hkv or key_cache.shape[3] != D
            or value_cache.shape != key_cache.shape or value_cache.stride() != key_cache.stride()
            or positions.numel() != B):
        raise RuntimeError("rotary_embedding_neox_kvcache: shape mismatch")
    for t in (query, key, value):
        if t.stride(-1) != 1 or t.stride(-2) != D:
            raise RuntimeError("rotary_embedding_neox_kvcache: [heads, head_size] must be dense")
    if key_cache.stride(-1) != 1 or not cos_sin_cache.is_contiguous():
        raise RuntimeError("rotary_embedding_neox_kvcache: cache rows must be dense")
    slot_stride = 0
    if slots is not None:
        if (slots.dtype != torch.int64 or slots.device != query.device or slots.numel() not in (1, B)
                or not slots.is_contiguous()):
            raise RuntimeError("rotary_embedding_neox_kvcache: slots must be contiguous int64 on the device, 1 or B elements")
        slot_stride = 1 if (slots.numel() == B and B > 1) else 0
    strides = (ctypes.c_long * 6)(query.stride(0), key.stride(0), value.stride(0), key_cache.stride(0),
                                  key_cache.stride(1), key_cache.stride(2))
    with torch.cuda.device(query.device):
        check(_lib.lib().eetq_rotary_neox_kvcache_f16(_ptr(positions), _ptr(slots) if slots is not None else None,
                                                      slot_stride, _ptr(query), _ptr(key), _ptr(value),
                                                      _ptr(cos_sin_cache), _ptr(key_cache), _ptr(value_cache), B, H, Hkv,
                                                      int(head_size), cos_sin_cache.shape[1], strides,
                                                      key_cache.shape[2], _stream_ptr()))
    return None


@_eager_only
def rotary_embedding_neox_kvcache_prefill(positions, query, key, value, head_size, cos_sin_cache, key_cache, value_cache,
                                          first_row=0, first_row_dev=None):
    """Prefill on a pre-allocated KV cache: rotate ``query`` [B, T, H, D] in place by ``positions`` [B, T] (int64), write the
    rotated ``key`` [B, T, Hkv, D] and ``value`` into the caches [B, Hkv, S, D] at rows ``base + t``, base = ``first_row_dev``
    (one int64 on the device, e.g. a static cache's token counter; read, not advanced) or ``first_row``.  One launch instead
    of rotary + two index_copy launches and their index arithmetic (eetq_rotary_neox_kvcache_prefill_f16)."""
    name = "rotary_embedding_neox_kvcache_prefill: "
    for t in (query, key, value, cos_sin_cache, key_cache, value_cache):
        if t.dtype != torch.float16:
            raise RuntimeError(name + "float16 tensors expected")
        if not t.is_cuda or t.device != query.device:
            raise RuntimeError(name + "all tensors must be on one CUDA device")
    if positions.dtype != torch.int64 or not positions.is_contiguous() or positions.device != query.device:
        raise RuntimeError(name + "positions must be contiguous int64 on the device")
    if query.dim() != 4 or key.dim() != 4 or value.dim() != 4 or key_cache.dim() != 4:
        raise RuntimeError(name + "shape mismatch")
    B, T, H, D = query.shape
    Hkv = key.shape[2]
    if (key.shape != (B, T, Hkv, D) or value.shape != key.shape or D != head_size or key_cache.shape[0] != B
            or key_cache.shape[1] != Hkv or key_cache.shape[3] != D or value_cache.shape != key_cache.shape
            or value_cache.stride() != key_cache.stride() or positions.numel() != B * T or first_row < 0
            or (first_row_dev is None and first_row + T > key_cache.shape[2])):
        raise RuntimeError(name + "shape mismatch")
    if first_row_dev is not None and (first_row_dev.dtype != torch.int64 or first_row_dev.numel() != 1
                                      or first_row_dev.device != query.device):
        raise RuntimeError(name + "first_row_dev must be one int64 on the device")
    for t in (query, key, value):
        if t.stride(-1) != 1 or t.stride(-2) != D or (B != 1 and T != 1 and t.stride(0) != T * t.stride(1)):
            raise RuntimeError(name + "[heads, head_size] must be dense and the tokens of all rows one stride apart")
    tdim = 0 if T == 1 else 1   # (a size-1 dimension's stride is arbitrary)
    if key_cache.stride(-1) != 1 or not cos_sin_cache.is_contiguous():
        raise RuntimeError(name + "cache rows must be dense")
    strides = (ctypes.c_long * 6)(query.stride(tdim), key.stride(tdim), value.stride(tdim), key_cache.stride(0),
                                  key_cache.stride(1), key_cache.stride(2))
    with torch.cuda.device(query.device):
        check(_lib.lib().eetq_rotary_neox_kvcache_prefill_f16(_ptr(positions), _ptr(query), _ptr(key), _ptr(value),
                                                              _ptr(cos_sin_cache), _ptr(key_cache), _ptr(value_cache), B, T,
                                                              _ptr(first_row_dev) if first_row_dev is not None else None,
                                                              int(first_row), H, Hkv, int(head_size), cos_sin_cache.shape[1],
                                                              strides, key_cache.shape[2], _stream_ptr()))
    return None


@_eager_only
def greedy_handover(logits, out_tokens, column, next_token, position):
    """Greedy decode hand-over in one launch: ``argmax`` of ``logits`` [B, V] (fp16; first index on ties, NaN = maximum, like
    ``torch.argmax``) goes to ``out_tokens[:, column]`` and ``next_token`` [B]; then ``position += 1`` and ``column += 1``
    (int64 device scalars).  (eetq_greedy_handover_f16)"""
    name = "greedy_handover: "
    if not logits.is_cuda or logits.dtype != torch.float16 or logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError(name + "logits must be a float16 CUDA tensor [B, V] with dense rows")
    B, V = logits.shape
    for t in (out_tokens, column, next_token, position):
        if t.dtype != torch.int64 or t.device != logits.device:
            raise RuntimeError(name + "int64 tensors on the logits' device expected")
    if (out_tokens.dim() != 2 or out_tokens.shape[0] != B or out_tokens.stride(1) != 1 or next_token.numel() != B
            or not next_token.is_contiguous() or column.numel() != 1 or position.numel() != 1 or V <= 0):
        raise RuntimeError(name + "shape mismatch")
    with torch.cuda.device(logits.device):
        check(_lib.lib().eetq_greedy_handover_f16(_ptr(logits), logits.stride(0), V, B, _ptr(out_tokens), out_tokens.stride(0),
                                                  out_tokens.shape[1], _ptr(column), _ptr(next_token), _ptr(position),
                                                  _stream_ptr()))
    return None


@_eager_only
def silu_mul(gate_up, glu8=False):
    """``silu(gate) * up`` on a fused gate|up projection output [..., 2*I] -> [..., I] in one launch (extension).
    ``glu8``: the columns come in groups of 16 = 8 gate + the 8 matching up columns instead of [all gate | all up]."""
    if gate_up.dtype != torch.float16 or not gate_up.is_cuda or not gate_up.is_contiguous():
        raise RuntimeError("silu_mul: expected a contiguous float16 CUDA tensor")
    inter = gate_up.shape[-1] // 2
    if gate_up.shape[-1] != 2 * inter or inter % 8:
        raise RuntimeError("silu_mul: last dimension must be 2*I with I a multiple of 8")
    out = torch.empty(tuple(gate_up.shape[:-1]) + (inter,), dtype=torch.float16, device=gate_up.device)
    rows = out.numel() // inter if inter else 0
    with torch.cuda.device(gate_up.device):
        fn = _lib.lib().eetq_silu_mul_glu8_f16 if glu8 else _lib.lib().eetq_silu_mul_f16
        check(fn(_ptr(gate_up), _ptr(out), rows, inter, _stream_ptr()))
    return out
