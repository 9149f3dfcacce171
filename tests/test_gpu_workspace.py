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


@pytest.mark.parametrize("K,N", [(1024, 512), (4096, 4096)])
def test_quantize_workspace_contract_of_both_abi_revisions(oracle, K, N):
    """ABI revision 1 documented the quantiser's workspace as N floats and had no size argument; revision 2 wants
    N * ceil(K / 128) for the fast route.  A caller built against revision 1 must not get out-of-bounds writes: the size-less
    entry treats a caller workspace as N floats (atomicMax route), eetq_quantize_i8_ws validates the size it is told.
    Every route gives the oracle's bytes; a guard band behind the N floats stays untouched."""
    from eetq_amd import _lib
    L = _lib.lib()
    assert L.eetq_abi_version() >= 2
    torch.manual_seed(K)
    w = ((torch.rand(K, N) - 0.5) * 0.2).half()
    q_ref, s_ref = oracle.quantize(w.numpy())
    want = oracle.gfx950_pack(q_ref)
    wd = w.to(DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    full = L.eetq_quantize_workspace_floats(K, N)
    assert full == N * ((K + 127) // 128) and full > N
    guard = 7.25

    def run(entry, ws_floats):
        ws = torch.full((full + 64,), guard, dtype=torch.float32, device=DEV)
        q = torch.zeros(K, N, dtype=torch.int8, device=DEV)
        s = torch.zeros(N, dtype=torch.float16, device=DEV)
        if entry == "v1":
            st = L.eetq_quantize_i8(p(wd), _lib.DTYPE_F16, K, N, None, p(q), _lib.LAYOUT_GFX950, p(s), p(ws), stream)
        else:
            st = L.eetq_quantize_i8_ws(p(wd), _lib.DTYPE_F16, K, N, None, p(q), _lib.LAYOUT_GFX950, p(s), p(ws), ws_floats,
                                       stream)
        torch.cuda.synchronize()
        return st, q, s, ws

    for entry, floats in (("v1", N), ("v2", N), ("v2", full - 1), ("v2", full)):
        st, q, s, ws = run(entry, floats)
        assert st == 0, (entry, floats, L.eetq_last_error())
        assert np.array_equal(q.cpu().numpy(), want), (entry, floats)
        assert s.cpu().numpy().tobytes() == s_ref.tobytes()
        used = N if floats < full else full
        assert bool((ws[used:] == guard).all()), (entry, floats)          # nothing written behind what the caller owns
    st, q, s, ws = run("v2", N - 1)
    assert st != 0 and b"workspace" in L.eetq_last_error()
    assert bool((ws == guard).all()) and not q.any()                      # rejected before any launch


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_both_quantiser_routes_agree_on_special_values(oracle, dtype):
    """Round-4 ADVICE: revision 1's entry (atomicMax maxima, `colmax_kernel<..., false>`) and revision 2's sized entry (row-block
    maxima) feed the scales from DIFFERENT column-maximum kernels.  They must give byte-identical weights and scales -- also on
    columns holding NaN, +-Inf, zeros only, subnormals and the largest finite value (the reference's std::max / std::min NaN
    order, cutlass_preprocessors.cc:619-648) -- and both must equal the oracle."""
    from eetq_amd import _lib
    L = _lib.lib()
    K, N = 512, 256
    tdt = torch.float16 if dtype == "f16" else torch.float32
    torch.manual_seed(3)
    w = ((torch.rand(K, N) - 0.5) * 0.3).to(tdt)
    big = torch.finfo(tdt).max
    w[:, 0] = 0                                    # zero column: scale 0, q = 127 by the NaN order
    w[5, 1] = float("nan")
    w[:, 2] = float("nan")                         # all-NaN column
    w[7, 3] = float("inf")
    w[9, 4] = float("-inf")
    w[11, 5], w[12, 5] = float("inf"), float("nan")
    w[13, 6] = big
    w[14, 7] = -big
    w[:, 8] = torch.finfo(tdt).tiny / 4            # subnormal column
    w[300, 9] = float("nan")                       # NaN in a later 128-row block only
    w[200, 10], w[400, 10] = float("-inf"), float("inf")
    q_ref, s_ref = oracle.quantize(w.numpy())
    wd = w.to(DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    code = _lib.DTYPE_F16 if dtype == "f16" else _lib.DTYPE_F32
    full = L.eetq_quantize_workspace_floats(K, N)
    outs = []
    for entry, floats in (("v1", N), ("v2", N), ("v2", full), ("v2-null", 0)):
        ws = torch.zeros(full, dtype=torch.float32, device=DEV)
        raw = torch.zeros(K, N, dtype=torch.int8, device=DEV)
        q = torch.zeros(K, N, dtype=torch.int8, device=DEV)
        sc = torch.zeros(N, dtype=tdt, device=DEV)
        if entry == "v1":
            st = L.eetq_quantize_i8(p(wd), code, K, N, p(raw), p(q), _lib.LAYOUT_GFX950, p(sc), p(ws), stream)
        elif entry == "v2":
            st = L.eetq_quantize_i8_ws(p(wd), code, K, N, p(raw), p(q), _lib.LAYOUT_GFX950, p(sc), p(ws), floats, stream)
        else:
            st = L.eetq_quantize_i8_ws(p(wd), code, K, N, p(raw), p(q), _lib.LAYOUT_GFX950, p(sc), None, 0, stream)
        torch.cuda.synchronize()
        assert st == 0, (entry, L.eetq_last_error())
        outs.append((entry, floats, raw.cpu().numpy(), q.cpu().numpy(), sc.cpu().numpy()))
    for entry, floats, raw, q, sc in outs:
        assert np.array_equal(raw, q_ref), (entry, floats, np.argwhere(raw != q_ref)[:4])
        assert np.array_equal(q, oracle.gfx950_pack(q_ref)), (entry, floats)
        assert sc.tobytes() == s_ref.tobytes(), (entry, floats)          # NaN scales compare by their bytes


def test_release_stream_workspace_returns_a_region_to_the_pool(ops):
    """Regions are never reclaimed by guessing (a stream that is capturing elsewhere, or a destroyed stream whose graphs are
    still replayed, keeps its region): a stream that found none runs unsplit until some owner hands its region back with
    eetq_release_stream_workspace."""
    from eetq_amd import _lib
    L = _lib.lib()
    _lib.check(L.eetq_release_workspace(None))
    torch.manual_seed(4)
    K, N, M = 4096, 4096, 32                                  # (two K slices; from M = 33 on AUTO cuts rows here: no scratch)
    w = (torch.rand(K, N, device=DEV) - 0.5).half() * 0.05
    qw, s = ops.quant_weights(w, torch.int8, False)
    x = torch.rand(M, K, dtype=torch.float16, device=DEV)
    ref = ops.w8_a16_gemm(x, qw, s)                           # default stream: owns region 0, split plan
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(17)]
    outs = []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            outs.append(ops.w8_a16_gemm(x, qw, s))
    torch.cuda.synchronize()
    for o in outs[:15]:
        assert torch.equal(o, ref)                            # own region, same plan: same bits
    for o in outs[15:]:
        assert _close(o, ref)                                 # no region left: unsplit kernel
    sp = lambda st: ctypes.c_void_p(st.cuda_stream)
    _lib.check(L.eetq_release_stream_workspace(sp(streams[16])))      # owns nothing: fine
    _lib.check(L.eetq_release_stream_workspace(sp(streams[0])))
    with torch.cuda.stream(streams[16]):
        late = ops.w8_a16_gemm(x, qw, s)
    with torch.cuda.stream(streams[1]):
        keep = ops.w8_a16_gemm(x, qw, s)
    torch.cuda.synchronize()
    assert torch.equal(late, ref) and torch.equal(keep, ref)
    # the same through the Python operator surface (round-4 ADVICE: the release must be reachable from the bindings): stream 2 hands
    # its region back, a stream that had none takes it over and runs the split plan
    import eetq_amd.ops as top
    fresh = torch.cuda.Stream()
    fresh.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(fresh):
        before = ops.w8_a16_gemm(x, qw, s)
    torch.cuda.synchronize()
    top.release_stream_workspace(streams[2])
    top.release_stream_workspace()                         # current stream: owns region 0; released, re-acquired by the next call
    with torch.cuda.stream(fresh):
        after = ops.w8_a16_gemm(x, qw, s)
    torch.cuda.synchronize()
    assert _close(before, ref) and torch.equal(after, ref)
    assert top.release_workspace() >= 40 << 20
    _lib.check(L.eetq_release_workspace(None))
