"""``import EETQ`` before the compiled module exists.

The operator module of the reference is a compiled extension called ``EETQ`` (csrc/eetpy.cpp:7-19).  Here it is built
in-tree to ``EETQ.cpython-*.so`` next to this file (eetq_amd/_ext.py); Python's import system prefers an extension module
over a ``.py`` of the same name, so this file is only ever executed when that ``.so`` is missing -- a fresh checkout, or a
machine without a C++ compiler.  It builds and loads the compiled module when it can, and otherwise re-exports the ctypes
binding of the same C ABI (eetq_amd/ops_ctypes.py), so the reference's import line works under both boundaries.  There is
still no CPU implementation behind either: the operators raise without libeetq_amd.so or without a GPU.
"""
import importlib
import importlib.util
import sys


def _compiled():
    from eetq_amd import _ext
    path = _ext.build()          # g++ against the installed torch headers; raises without a compiler
    import torch  # noqa: F401   (libtorch must be mapped before the extension)
    from eetq_amd import _lib
    _lib.lib()
    importlib.invalidate_caches()
    spec = importlib.util.spec_from_file_location("EETQ", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


try:
    sys.modules[__name__] = _compiled()   # `import EETQ` hands out whatever sys.modules holds once this file has run
except Exception as _e:  # noqa: BLE001  no compiler / headers: the pure-Python binding of the same C ABI
    _compiled_error = _e
    from eetq_amd.ops_ctypes import (layernorm_forward, preprocess_weights, quant_weights, rotary_embedding_neox,  # noqa: F401
                                     w8_a16_gemm, w8_a16_gemm_)
    __all__ = ["w8_a16_gemm", "w8_a16_gemm_", "preprocess_weights", "quant_weights", "rotary_embedding_neox",
               "layernorm_forward"]
