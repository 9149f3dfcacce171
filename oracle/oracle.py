"""ctypes/numpy front end of ``libeetq_oracle.so`` (see eetq_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libeetq_oracle.so")

__all__ = [
    "build", "lib", "quantize", "sm80_pack", "sm80_pack_closed_form", "sm80_unpack", "gfx950_pack",
    "gfx950_unpack", "sm80_reader_unpack", "ref_gemv_sm80", "quantize_i4", "i4_values", "i4_from_values", "sm80_pack_i4",
    "sm80_unpack_i4", "sm80_reader_unpack_i4", "gfx950_pack_i4", "gfx950_unpack_i4", "w8a16_gemm", "w8a16_gemm_bias_act", "w8a16_gemm_f32acc", "dequant", "rmsnorm_f16", "rotary_neox_f16", "rotary_neox",
    "f32_to_f16_bits", "f16_bits_to_f32",
]

_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "eetq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libeetq_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.oracle_quantize_f16.argtypes = [vp, sz, sz, vp, vp]
        L.oracle_quantize_f32.argtypes = [vp, sz, sz, vp, vp]
        for name in ("oracle_sm80_pack", "oracle_sm80_pack_closed_form", "oracle_sm80_unpack",
                     "oracle_gfx950_pack", "oracle_gfx950_unpack", "oracle_sm80_reader_unpack"):
            getattr(L, name).argtypes = [vp, sz, sz, vp]
            getattr(L, name).restype = i32
        L.oracle_w8a16_gemm.argtypes = [vp, vp, vp, vp, sz, sz, sz]
        L.oracle_quantize_i4_f16.argtypes = [vp, sz, sz, vp, vp]
        L.oracle_quantize_i4_f32.argtypes = [vp, sz, sz, vp, vp]
        L.oracle_i4_unpack_values.argtypes = [vp, sz, sz, vp]
        L.oracle_i4_pack_values.argtypes = [vp, sz, sz, vp]
        for name in ("oracle_sm80_pack_i4", "oracle_sm80_unpack_i4", "oracle_sm80_reader_unpack_i4", "oracle_gfx950_pack_i4",
                     "oracle_gfx950_unpack_i4"):
            getattr(L, name).argtypes = [vp, sz, sz, vp]
            getattr(L, name).restype = i32
        L.oracle_w8a16_gemm_bias_act.argtypes = [vp, vp, vp, vp, i32, vp, sz, sz, sz]
        L.oracle_ref_gemv_sm80.argtypes = [vp, vp, vp, vp, sz, sz, sz]
        L.oracle_ref_gemv_sm80.restype = i32
        L.oracle_w8a16_gemm_f32acc.argtypes = [vp, vp, vp, vp, sz, sz, sz]
        L.oracle_dequant.argtypes = [vp, vp, vp, sz, sz]
        L.oracle_rmsnorm_f16.argtypes = [vp, vp, vp, ctypes.c_float, sz, sz]
        L.oracle_rotary_neox_f16.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz]
        L.oracle_rotary_neox_f32.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz]
        L.oracle_rotary_neox_f64.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz]
        L.oracle_f32_to_f16.argtypes = [ctypes.c_float]
        L.oracle_f32_to_f16.restype = ctypes.c_uint16
        L.oracle_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.oracle_f16_to_f32.restype = ctypes.c_float
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    a = np.ascontiguousarray(a)
    assert a.dtype == dtype, (a.dtype, dtype)
    return a


def f32_to_f16_bits(x):
    return int(lib().oracle_f32_to_f16(float(x)))


def f16_bits_to_f32(h):
    return float(lib().oracle_f16_to_f32(int(h)))


def quantize(w):
    """w: [K, N] float16 or float32 -> (q_raw int8 [K, N], scales [N] same dtype as w)."""
    assert w.ndim == 2
    K, N = w.shape
    q = np.empty((K, N), np.int8)
    if w.dtype == np.float16:
        w = _c(w, np.float16)
        s = np.empty(N, np.float16)
        lib().oracle_quantize_f16(_p(w), K, N, _p(q), _p(s))
    elif w.dtype == np.float32:
        w = _c(w, np.float32)
        s = np.empty(N, np.float32)
        lib().oracle_quantize_f32(_p(w), K, N, _p(q), _p(s))
    else:
        raise TypeError(w.dtype)
    return q, s


def _layout_call(fn, src):
    src = _c(src, np.int8)
    K, N = src.shape
    out = np.empty((K, N), np.int8)
    rc = fn(_p(src), K, N, _p(out))
    if rc != 0:
        raise ValueError("unsupported shape K=%d N=%d" % (K, N))
    return out


def sm80_pack(q_raw):
    return _layout_call(lib().oracle_sm80_pack, q_raw)


def sm80_pack_closed_form(q_raw):
    return _layout_call(lib().oracle_sm80_pack_closed_form, q_raw)


def sm80_unpack(packed):
    return _layout_call(lib().oracle_sm80_unpack, packed)


def sm80_reader_unpack(packed):
    """Raw int8 recovered from sm80 bytes by the reference GEMV's reader (kernel.h:294-376 + the int8->fp16 converter):
    the second, independent reference source for the processed layout."""
    return _layout_call(lib().oracle_sm80_reader_unpack, packed)


def ref_gemv_sm80(x, packed_sm80, scales):
    """The reference batched GEMV's own arithmetic (fp16 per-thread accumulation) on sm80 bytes, M <= 4."""
    x = _c(x, np.float16)
    packed_sm80 = _c(packed_sm80, np.int8)
    scales = _c(scales, np.float16)
    M, K = x.shape
    K2, N = packed_sm80.shape
    assert K == K2 and scales.shape == (N,)
    y = np.empty((M, N), np.float16)
    rc = lib().oracle_ref_gemv_sm80(_p(x), _p(packed_sm80), _p(scales), _p(y), M, N, K)
    if rc != 0:
        raise ValueError("oracle_ref_gemv_sm80 failed: %d" % rc)
    return y


def quantize_i4(w):
    """w: [K, N] float16/float32 -> (packed raw int4 [K, N/2] int8, scales [N])."""
    assert w.ndim == 2 and w.shape[1] % 2 == 0
    K, N = w.shape
    q = np.empty((K, N // 2), np.int8)
    if w.dtype == np.float16:
        s = np.empty(N, np.float16)
        lib().oracle_quantize_i4_f16(_p(_c(w, np.float16)), K, N, _p(q), _p(s))
    else:
        s = np.empty(N, np.float32)
        lib().oracle_quantize_i4_f32(_p(_c(w, np.float32)), K, N, _p(q), _p(s))
    return q, s


def i4_values(q_packed):
    """packed raw int4 [K, N/2] -> int8 [K, N] holding the values -8..7."""
    q_packed = _c(q_packed, np.int8)
    K, half = q_packed.shape
    q = np.empty((K, half * 2), np.int8)
    lib().oracle_i4_unpack_values(_p(q_packed), K, half * 2, _p(q))
    return q


def i4_from_values(q):
    q = _c(q, np.int8)
    K, N = q.shape
    out = np.empty((K, N // 2), np.int8)
    lib().oracle_i4_pack_values(_p(q), K, N, _p(out))
    return out


def _layout_call_i4(fn, src):
    src = _c(src, np.int8)
    K, half = src.shape
    out = np.empty((K, half), np.int8)
    rc = fn(_p(src), K, half * 2, _p(out))
    if rc != 0:
        raise ValueError("unsupported int4 shape K=%d N=%d" % (K, half * 2))
    return out


def sm80_pack_i4(q_packed):
    return _layout_call_i4(lib().oracle_sm80_pack_i4, q_packed)


def sm80_unpack_i4(packed):
    return _layout_call_i4(lib().oracle_sm80_unpack_i4, packed)


def sm80_reader_unpack_i4(packed):
    return _layout_call_i4(lib().oracle_sm80_reader_unpack_i4, packed)


def gfx950_pack_i4(q_packed):
    return _layout_call_i4(lib().oracle_gfx950_pack_i4, q_packed)


def gfx950_unpack_i4(packed):
    return _layout_call_i4(lib().oracle_gfx950_unpack_i4, packed)


def gfx950_pack(q_raw):
    return _layout_call(lib().oracle_gfx950_pack, q_raw)


def gfx950_unpack(packed):
    return _layout_call(lib().oracle_gfx950_unpack, packed)


def _gemm(fn, x, q_raw, scales):
    x = _c(x, np.float16)
    q_raw = _c(q_raw, np.int8)
    scales = _c(scales, np.float16)
    M, K = x.shape
    K2, N = q_raw.shape
    assert K == K2 and scales.shape == (N,)
    y = np.empty((M, N), np.float16)
    fn(_p(x), _p(q_raw), _p(scales), _p(y), M, N, K)
    return y


def w8a16_gemm(x, q_raw, scales):
    """Contract y = fp16(sum_k fp32(x) * fp32(fp16(q*s))) with exact (double) accumulation."""
    return _gemm(lib().oracle_w8a16_gemm, x, q_raw, scales)


def w8a16_gemm_bias_act(x, q_raw, scales, bias, act):
    """FT bias + activation epilogue: fp16(act(acc + bias)); act in {"relu", "gelu", "silu"}; bias may be None."""
    x = _c(x, np.float16)
    q_raw = _c(q_raw, np.int8)
    scales = _c(scales, np.float16)
    M, K = x.shape
    K2, N = q_raw.shape
    assert K == K2 and scales.shape == (N,)
    b = _c(bias, np.float16) if bias is not None else None
    y = np.empty((M, N), np.float16)
    lib().oracle_w8a16_gemm_bias_act(_p(x), _p(q_raw), _p(scales), _p(b) if b is not None else None,
                                     {"relu": 1, "gelu": 2, "silu": 3}[act], _p(y), M, N, K)
    return y


def w8a16_gemm_f32acc(x, q_raw, scales):
    """Same contract, strict k-ordered fp32 accumulation (one legal order)."""
    return _gemm(lib().oracle_w8a16_gemm_f32acc, x, q_raw, scales)


def dequant(q_raw, scales):
    q_raw = _c(q_raw, np.int8)
    scales = _c(scales, np.float16)
    K, N = q_raw.shape
    w = np.empty((K, N), np.float16)
    lib().oracle_dequant(_p(q_raw), _p(scales), _p(w), K, N)
    return w


def rmsnorm_f16(x, gamma, eps):
    x = _c(x, np.float16)
    gamma = _c(gamma, np.float16)
    rows = int(np.prod(x.shape[:-1]))
    cols = x.shape[-1]
    out = np.empty_like(x)
    lib().oracle_rmsnorm_f16(_p(x), _p(gamma), _p(out), float(eps), rows, cols)
    return out


def rotary_neox_f16(positions, q, k, cache, head_size):
    """Returns rotated copies (q', k'); q, k: [tokens, heads, head_size] float16."""
    positions = _c(positions.reshape(-1), np.int64)
    q = _c(q, np.float16).copy()
    k = _c(k, np.float16).copy()
    cache = _c(cache, np.float16)
    tokens = positions.shape[0]
    heads = q.size // (tokens * head_size)
    lib().oracle_rotary_neox_f16(_p(positions), _p(q), _p(k), _p(cache), tokens, heads, head_size,
                                 cache.shape[1])
    return q, k


def rotary_neox(positions, q, k, cache, head_size):
    """The same for float16 / float32 / float64 operands (dtype taken from q): rotated copies (q', k')."""
    dt = np.dtype(q.dtype)
    if dt == np.float16:
        return rotary_neox_f16(positions, q, k, cache, head_size)
    fn = {np.dtype(np.float32): "oracle_rotary_neox_f32", np.dtype(np.float64): "oracle_rotary_neox_f64"}[dt]
    positions = _c(positions.reshape(-1), np.int64)
    q = _c(q, dt).copy()
    k = _c(k, dt).copy()
    cache = _c(cache, dt)
    tokens = positions.shape[0]
    heads = q.size // (tokens * head_size)
    getattr(lib(), fn)(_p(positions), _p(q), _p(k), _p(cache), tokens, heads, head_size, cache.shape[1])
    return q, k
