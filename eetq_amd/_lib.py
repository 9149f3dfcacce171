"""Loader for ``libeetq_amd.so`` (the C ABI declared in ``include/eetq_amd.h``).

The product path has no CPU fallback: if the shared library is missing and cannot be built with hipcc,
importing :mod:`eetq_amd.ops` raises.  The library is kept in-tree (``eetq_amd/libeetq_amd.so``).
"""
import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libeetq_amd.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

EETQ_OK = 0
ERR_UNSUPPORTED = -3  # EETQ_ERR_UNSUPPORTED (include/eetq_amd.h)
DTYPE_F16, DTYPE_F32, DTYPE_F64 = 0, 1, 2
LAYOUT_ROW_MAJOR, LAYOUT_GFX950, LAYOUT_SM80 = 0, 1, 2
PATH_AUTO, PATH_GEMV, PATH_MFMA, PATH_STREAM, PATH_MID, PATH_SPLITK, PATH_TILESPLIT = 0, 1, 2, 3, 4, 5, 6
ACT_IDENTITY, ACT_RELU, ACT_GELU, ACT_SILU = 0, 1, 2, 3

_lib = None


def _sources_newer_than_lib():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for name in os.listdir(CSRC_DIR):
        if name.endswith((".hip", ".hpp", "Makefile")) and os.path.getmtime(os.path.join(CSRC_DIR, name)) > t:
            return True
    hdr = os.path.join(_HERE, "..", "include", "eetq_amd.h")
    return os.path.exists(hdr) and os.path.getmtime(hdr) > t


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into ``eetq_amd/libeetq_amd.so`` (hipcc cross-compiles w/o a GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("eetq_amd: hipcc not found; cannot build libeetq_amd.so")
    if force or _sources_newer_than_lib():
        cmd = ["make", "-C", CSRC_DIR, "-j8", "HIPCC=" + hipcc] + (["-B"] if force else [])
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or res.returncode != 0:
            print(res.stdout)
        if res.returncode != 0:
            raise RuntimeError("eetq_amd: building libeetq_amd.so failed:\n" + res.stdout[-4000:])
    return LIB_PATH


def _declare(L):
    vp, sz, i32, f32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_float
    sigs = {
        "eetq_quantize_i8": [vp, i32, sz, sz, vp, vp, i32, vp, vp, vp],
        "eetq_quantize_i8_ws": [vp, i32, sz, sz, vp, vp, i32, vp, vp, sz, vp],
        "eetq_abi_version": [],
        "eetq_release_stream_workspace": [vp],
        "eetq_quantize_i8_host": [vp, i32, sz, sz, vp, vp, i32, vp],
        "eetq_pack_i8": [vp, sz, sz, vp, i32, vp],
        "eetq_unpack_i8": [vp, sz, sz, vp, i32, vp],
        "eetq_pack_i8_host": [vp, sz, sz, vp, i32],
        "eetq_unpack_i8_host": [vp, sz, sz, vp, i32],
        "eetq_w8a16_gemm": [vp, vp, vp, vp, i32, i32, i32, vp],
        "eetq_w8a16_gemm_ex": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "eetq_w8a16_gemm_bias": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "eetq_w8a16_gemm_fused": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "eetq_w8a16_gemm_act": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
        "eetq_quantize_i4": [vp, i32, sz, sz, vp, vp, i32, vp, vp, vp],
        "eetq_pack_i4": [vp, sz, sz, vp, i32, vp],
        "eetq_unpack_i4": [vp, sz, sz, vp, i32, vp],
        "eetq_w4a16_gemm": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
        "eetq_w4a16_gemm_ex": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "eetq_rmsnorm_f16": [vp, vp, vp, f32, i32, i32, vp],
        "eetq_rotary_neox_f16": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "eetq_rotary_neox": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
        "eetq_w8a16_gemv_grouped": [vp, i32, vp],
        "eetq_decode_dropped_steps": [vp, i32],
        "eetq_rotary_neox_strided_f16": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
        "eetq_w8a16_gemv_rmsnorm": [vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, vp],
        "eetq_w8a16_gemv_silu_gated": [vp, vp, vp, vp, vp, vp, i32, i32, vp],
        "eetq_silu_mul_f16": [vp, vp, i32, i32, vp],
        "eetq_silu_mul_glu8_f16": [vp, vp, i32, i32, vp],
        "eetq_w8a16_gemv_glu8": [vp, vp, f32, vp, vp, vp, vp, i32, i32, vp],
        "eetq_w8a16_gemm_glu8": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
        "eetq_rotary_neox_kvcache_f16": [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp],
        "eetq_greedy_handover_f16": [vp, ctypes.c_long, i32, i32, vp, ctypes.c_long, i32, vp, vp, vp, vp],
        "eetq_rotary_neox_kvcache_prefill_f16": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp],
        "eetq_decode_attention_f16": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, ctypes.c_float, vp, vp, i32, vp,
                                      vp],
        "eetq_rope_decode_attention_f16": [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                           ctypes.c_float, vp, vp, i32, vp, vp],
        "eetq_prefill_attention_f16": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, ctypes.c_float, vp, vp],
        "eetq_prefill_attention_supported": [i32],
        "eetq_decode_attention_splits": [i32, i32, i32],
        "eetq_prof_begin": [i32],
        "eetq_prof_end": [vp, i32, vp],
        "eetq_diag_stream_read": [vp, sz, vp, vp],
        "eetq_diag_empty": [vp, i32, i32, vp],
        "eetq_diag_clock_stamp": [vp, i32, vp],
        "eetq_diag_attn_stamps": [vp],
        "eetq_diag_stream_plan": [i32, i32, i32, i32, i32, vp, vp, vp],
        "eetq_diag_auto_path": [i32, i32, i32, i32, vp, vp],
        "eetq_diag_splitk_plan": [i32, i32, i32, vp, vp, vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    L.eetq_quantize_workspace_floats.restype = ctypes.c_size_t
    L.eetq_quantize_workspace_floats.argtypes = [sz, sz]
    L.eetq_release_workspace.restype = i32
    L.eetq_release_workspace.argtypes = [ctypes.POINTER(ctypes.c_size_t)]
    L.eetq_last_error.restype = ctypes.c_char_p
    L.eetq_last_error.argtypes = []
    L.eetq_version.restype = ctypes.c_char_p
    L.eetq_version.argtypes = []
    L.eetq_device_supported.restype = i32
    L.eetq_device_supported.argtypes = []
    return L


EXPORTED_SYMBOLS = (
    "eetq_quantize_i8", "eetq_quantize_i8_ws", "eetq_abi_version", "eetq_quantize_i8_host", "eetq_pack_i8", "eetq_unpack_i8", "eetq_pack_i8_host",
    "eetq_unpack_i8_host", "eetq_w8a16_gemm", "eetq_w8a16_gemm_ex", "eetq_w8a16_gemm_bias", "eetq_w8a16_gemm_fused", "eetq_w8a16_gemm_act", "eetq_quantize_i4", "eetq_pack_i4", "eetq_unpack_i4", "eetq_w4a16_gemm", "eetq_w4a16_gemm_ex", "eetq_rmsnorm_f16",
    "eetq_rotary_neox_f16", "eetq_rotary_neox_strided_f16", "eetq_w8a16_gemv_rmsnorm", "eetq_w8a16_gemv_silu_gated", "eetq_silu_mul_f16", "eetq_silu_mul_glu8_f16", "eetq_w8a16_gemv_glu8", "eetq_w8a16_gemm_glu8", "eetq_rotary_neox_kvcache_f16", "eetq_rotary_neox_kvcache_prefill_f16", "eetq_greedy_handover_f16", "eetq_decode_attention_f16", "eetq_decode_attention_splits", "eetq_rope_decode_attention_f16", "eetq_prefill_attention_f16", "eetq_prefill_attention_supported", "eetq_prof_begin", "eetq_prof_end", "eetq_diag_stream_read", "eetq_diag_empty", "eetq_diag_clock_stamp", "eetq_diag_attn_stamps", "eetq_diag_stream_plan", "eetq_diag_auto_path", "eetq_diag_splitk_plan", "eetq_last_error", "eetq_version", "eetq_device_supported",
    "eetq_quantize_workspace_floats", "eetq_release_workspace", "eetq_release_stream_workspace", "eetq_rotary_neox", "eetq_w8a16_gemv_grouped", "eetq_decode_dropped_steps",
)


def _share_hip_runtime_with_torch():
    """torch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  If it is already loaded, our NEEDED
    entry resolves to it and both sides share ONE HIP runtime (streams, events and graph capture are then the same
    objects).  Loaded in the other order the process would hold two runtimes -- torch's and /opt/rocm's -- and torch
    stream handles would be meaningless to our launches.  So: import torch first whenever it is installed."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib():
    """Return the loaded library, building it first when the sources are newer (dev machines only)."""
    global _lib
    if _lib is None:
        _share_hip_runtime_with_torch()
        def _needs_build():
            return not os.path.exists(LIB_PATH) or (os.path.isdir(CSRC_DIR) and shutil.which("hipcc")
                                                    and _sources_newer_than_lib())
        if _needs_build():
            # several ranks of one node may get here together (torchrun): one builds, the others wait and re-check
            import fcntl
            with open(LIB_PATH + ".lock", "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    if _needs_build():
                        build()
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
        try:
            _lib = _declare(ctypes.CDLL(LIB_PATH))
        except OSError as e:  # no fallback on purpose
            raise RuntimeError("eetq_amd: cannot load %s (%s). The HIP extension is required." % (LIB_PATH, e))
    return _lib


def check(status):
    """Non-zero status -> RuntimeError (what the reference's C++ exceptions become through pybind)."""
    if status != EETQ_OK:
        msg = lib().eetq_last_error()
        raise RuntimeError(msg.decode() if msg else "eetq_amd: error %d" % status)
