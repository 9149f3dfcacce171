import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as _oracle
    _oracle.build()
    return _oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def linear_default_weight(out_features, in_features, seed):
    """nn.Linear default init (U(+-1/sqrt(in))) as float16 numpy [out, in] + the torch module recipe used by
    the reference's examples/layers/test_qlinear.py:20-26."""
    import torch
    torch.manual_seed(seed)
    lin = torch.nn.Linear(in_features, out_features, bias=False, dtype=torch.float16)
    return lin


def np_quantize_reference(w):
    """Independent numpy restatement of cutlass_preprocessors.cc:581-678 (second implementation, used to
    cross-check the C oracle): fp32 math, IEEE division, round half away from zero, NaN -> 127."""
    w32 = w.astype(np.float32)
    a = np.abs(w32)
    a = np.where(np.isnan(a), np.float32(0), a)  # std::max(a, NaN) keeps a
    amax = a.max(axis=0).astype(np.float32)
    s32 = (amax * np.float32(1.0 / 128.0)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (w32 / s32[None, :]).astype(np.float32)
    rounded = np.sign(r) * np.floor(np.abs(r) + np.float32(0.5))
    rounded = np.where(np.isnan(r), np.float32(np.nan), rounded)
    hi = np.where(rounded < 127, rounded, np.float32(127))   # std::min(127, x): NaN -> 127
    lo = np.where(-128 < hi, hi, np.float32(-128))
    q = lo.astype(np.int32).astype(np.int8)
    scales = s32.astype(w.dtype)
    return q, scales
