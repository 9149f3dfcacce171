"""Library-owned scratch (include/eetq_amd.h, "library-owned scratch"): split-K regions are owned per launch stream and never
shared -- more concurrent streams than regions must still produce correct results -- and eetq_release_workspace frees it."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import eetq_amd.ops as _ops
    return _ops


def _close(a, b):
    a, b = a.float(), b.float()
    return bool(((a - b).abs() <= 1e-3 * b.abs().max() + 2e-3 * b.abs()).all())


def test_splitk_twenty_concurrent_streams_then_release(ops):
    """17 <= M <= 128 takes the split-K tile with an in-launch reduction through per-stream scratch.  20 streams launch
    concurrently (16 regions per device: four of them must fall back to the unsplit kernel, never to a shared region);
    every result is compared with the one the default stream produced alone, and repeated launches on one stream are
    bit-identical."""
    from eetq_amd import _lib
    torch.manual_seed(3)
    K, N = 4096, 4096
    w = (torch.rand(K, N, device=DEV) - 0.5).half() * 0.05
    qw, s = ops.quant_weights(w, torch.int8, False)
    xs = [torch.rand(M, K, dtype=torch.float16, device=DEV) for M in (64, 32, 128, 96)]
    refs = [ops.w8_a16_gemm(x, qw, s) for x in xs]          # default stream, alone
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(20)]
    outs = [[None] * 6 for _ in streams]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for rep in range(6):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i][rep] = ops.w8_a16_gemm(xs[(i + rep) % 4], qw, s)
    torch.cuda.synchronize()
    for i in range(len(streams)):
        for rep in range(6):
            assert _close(outs[i][rep], refs[(i + rep) % 4]), (i, rep)
        # same stream, same input (reps 0 and 4 use the same x): same bits
        assert torch.equal(outs[i][0], outs[i][4]) and torch.equal(outs[i][1], outs[i][5]), i
    # the first 16 streams own regions: their results equal the default stream's bit for bit as long as the plan splits
    # the same way; all of them must at least be close -- checked above.  Now free everything and run again.
    freed = ctypes.c_size_t(0)
    _lib.check(_lib.lib().eetq_release_workspace(ctypes.byref(freed)))
    assert freed.value >= 40 << 20, freed.value
    freed2 = ctypes.c_size_t(123)
    _lib.check(_lib.lib().eetq_release_workspace(ctypes.byref(freed2)))
    assert freed2.value == 0                                  # nothing left
    again = [ops.w8_a16_gemm(x, qw, s) for x in xs]
    torch.cuda.synchronize()
    for a, r in zip(again, refs):
        assert torch.equal(a, r)                              # scratch re-created on demand, same result


def test_quantize_null_workspace_and_release():
    """eetq_quantize_i8 with workspace = NULL uses (and eetq_release_workspace frees) the library's per-device buffer."""
    from eetq_amd import _lib
    L = _lib.lib()
    K, N = 256, 128
    w = (torch.rand(K, N, device=DEV) - 0.5).half()
    q1 = torch.empty(K, N, dtype=torch.int8, device=DEV)
    q2 = torch.empty_like(q1)
    s1 = torch.empty(N, dtype=torch.float16, device=DEV)
    s2 = torch.empty_like(s1)
    ws = torch.empty(L.eetq_quantize_workspace_floats(K, N), dtype=torch.float32, device=DEV)
    assert ws.numel() == N * 2
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.eetq_quantize_i8(p(w), _lib.DTYPE_F16, K, N, None, p(q1), _lib.LAYOUT_GFX950, p(s1), p(ws), stream))
    _lib.check(L.eetq_quantize_i8(p(w), _lib.DTYPE_F16, K, N, None, p(q2), _lib.LAYOUT_GFX950, p(s2), None, stream))
    torch.cuda.synchronize()
    assert torch.equal(q1, q2) and torch.equal(s1, s2)
    freed = ctypes.c_size_t(0)
    _lib.check(L.eetq_release_workspace(ctypes.byref(freed)))
    assert freed.value >= 65536 * 4
