"""eetq_amd -- MI355X (gfx950) implementation of EETQ's W8A16 weight-only GEMM hot path.

Layout of the package (only what the path needs):
  csrc/            hand-written HIP kernels + the C ABI (include/eetq_amd.h) -> libeetq_amd.so
  _lib.py          ctypes loader (no CPU fallback)
  ops.py           the reference's native operator surface (module ``EETQ``: csrc/eetpy.cpp:7-19)
  modules/qlinear  W8A16Linear / EetqLinear / EetqLinearMMFunction (python/eetq/modules/qlinear.py)
  utils/quantizer  eet_quantize() (python/eetq/utils/quantizer.py:40-61)
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def __getattr__(name):
    # torch-dependent parts are imported lazily so that `import eetq_amd` stays cheap
    if name in ("ops", "modules", "utils"):
        import importlib
        return importlib.import_module("." + name, __name__)
    if name in ("quant_weights", "preprocess_weights", "w8_a16_gemm", "w8_a16_gemm_", "layernorm_forward",
                "rotary_embedding_neox"):
        from . import ops
        return getattr(ops, name)
    if name in ("W8A16Linear", "EetqLinear", "EetqLinearMMFunction", "quantize_and_preprocess_weights"):
        from .modules import qlinear
        return getattr(qlinear, name)
    if name in ("eet_quantize", "find_layers", "set_op_by_name"):
        from .utils import quantizer
        return getattr(quantizer, name)
    raise AttributeError(name)
