"""W4A16 (int4 weight-only): the quint4x2 branch of quant_weights / preprocess_weights bit-exact against the oracle's
restatement of cutlass_preprocessors.cc (writer) pinned by the reference GEMV's Int4b reader, and the W4A16 GEMM
(extension) against the same numerics contract as W8A16: y = fp16(sum_k fp32(x) * fp32(fp16(q4 * s)))."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import eetq_amd.ops as _ops
    if _ops.BOUNDARY != "ext":
        pytest.skip("int4 goes through the compiled module (EETQ_AMD_BOUNDARY=ctypes selects the twin binding)")
    return _ops


def _tier_a(y, ref):
    y, ref = y.astype(np.float32), ref.astype(np.float32)
    return np.abs(y - ref) <= 1e-3 * np.abs(ref).max() + 2e-3 * np.abs(ref)


@pytest.mark.parametrize("K,N,dtype", [(128, 16, np.float16), (128, 64, np.float32), (256, 192, np.float16),
                                       (4096, 4096, np.float16), (1024, 11008, np.float16)])
def test_quant_weights_int4_bit_exact(ops, oracle, K, N, dtype):
    rng = np.random.default_rng(K + N)
    w = (rng.standard_normal((K, N)) * 0.05).astype(dtype)
    w[:, 5] = 0                                        # all-zero column: scale 0, q = -8
    w[3, 7] = np.abs(w[:, 7]).max() * 2                # +amax -> +8 -> clipped to 7
    raw, processed, scales = ops.quant_weights(torch.from_numpy(w), torch.quint4x2, True)
    assert raw.shape == (K, N // 2) and processed.shape == (K, N // 2) and raw.dtype == torch.int8 and raw.device.type == "cpu"
    q, s = oracle.quantize_i4(w)
    assert np.array_equal(raw.numpy(), q)
    assert scales.numpy().tobytes() == s.tobytes()
    assert np.array_equal(processed.numpy(), oracle.gfx950_pack_i4(q))
    two = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.quint4x2, False)
    assert len(two) == 2 and torch.equal(two[0].cpu(), processed) and two[0].is_cuda
    if N % 64 == 0:
        sm = ops.quant_weights(torch.from_numpy(w), torch.quint4x2, False, layout="sm80")[0]
        assert np.array_equal(sm.numpy(), oracle.sm80_pack_i4(q))     # the bytes the reference would have produced


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
@pytest.mark.parametrize("K,N", [(256, 128), (384, 80), (4224, 64)])
def test_quant_weights_int4_near_ties_native_layout_only(ops, oracle, dtype, K, N):
    """quant_weights(..., quint4x2, return_unprocessed=False) in the native layout is ONE fused quantise + pack launch whose
    quotient is fma(w, rcp(s), 8) with an exact sign test (quant.hip::quant_pack_kernel<T, 4>): bit-identical to the oracle on a
    matrix made of rounding ties (k + 0.5) * amax / 8 and their neighbours, on NaN / inf / zero columns, a ragged last strip and
    a 16-tile walk."""
    rng = np.random.default_rng(K * 3 + N)
    if dtype == np.float32:
        amax = (10.0 ** rng.uniform(-3, 1, N)).astype(np.float32)
        amax[1], amax[2] = 1e-36, 3e37
    else:
        amax = (2.0 ** rng.integers(-10, 4, N) * rng.choice([1.0, 1.5, 1.25], N)).astype(np.float16)
        amax[1] = 2.0 ** -20
    scale = amax.astype(np.float32) * np.float32(1.0 / 8.0)
    k = rng.integers(-8, 8, (K, N)).astype(np.float32) + np.float32(0.5)
    if dtype == np.float32:
        off = (rng.choice([-1.0, 1.0], (K, N)) * 10.0 ** rng.uniform(-5.5, -3.5, (K, N))).astype(np.float32)
        k[:, 1::2] += off[:, 1::2]
    w = (k * scale[None, :]).astype(dtype)
    step = rng.integers(-2, 3, (K, N))
    for _ in range(2):
        w = np.where(step > 0, np.nextafter(w, dtype(np.inf)), np.where(step < 0, np.nextafter(w, dtype(-np.inf)), w)).astype(dtype)
    w = np.clip(w, -amax[None, :], amax[None, :]).astype(dtype)
    w[rng.integers(0, K, N), np.arange(N)] = amax * rng.choice([-1, 1], N).astype(dtype)
    w[5, 9] = np.nan
    w[:, 6] = 0
    w[17, 40] = np.inf
    q, s = oracle.quantize_i4(w)
    processed, scales = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.quint4x2, False)
    assert scales.cpu().numpy().tobytes() == s.tobytes()
    got, ref = processed.cpu().numpy(), oracle.gfx950_pack_i4(q)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} bytes differ"
    raw = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.quint4x2, True)[0]      # the three-kernel route agrees
    assert np.array_equal(raw.cpu().numpy(), q)


@pytest.mark.parametrize("layout", ["gfx950", "sm80"])
@pytest.mark.parametrize("K,N", [(128, 64), (384, 256), (4096, 4096)])
def test_preprocess_unprocess_int4(ops, oracle, layout, K, N):
    rng = np.random.default_rng(K * 3 + N)
    q = oracle.i4_from_values(rng.integers(-8, 8, (K, N), dtype=np.int8))
    want = oracle.gfx950_pack_i4(q) if layout == "gfx950" else oracle.sm80_pack_i4(q)
    got = ops.preprocess_weights(torch.from_numpy(q), True, layout)
    assert np.array_equal(got.numpy(), want)
    assert np.array_equal(ops.unprocess_weights(got, layout, True).numpy(), q)
    d = ops.preprocess_weights(torch.from_numpy(q).to(DEV), True, layout)
    assert d.is_cuda and np.array_equal(d.cpu().numpy(), want)
    if layout == "sm80":   # checkpoint interop: reference bytes -> native
        assert np.array_equal(ops.convert_layout(torch.from_numpy(want), "sm80", "gfx950", is_int4=True).numpy(),
                              oracle.gfx950_pack_i4(q))


def test_int4_shape_errors(ops):
    with pytest.raises(RuntimeError, match="multiple of 128"):
        ops.preprocess_weights(torch.zeros(64, 32, dtype=torch.int8), True)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.preprocess_weights(torch.zeros(128, 16, dtype=torch.int8), True, "sm80")
    with pytest.raises(RuntimeError):
        ops.quant_weights(torch.zeros(128, 15, dtype=torch.float16), torch.quint4x2)


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("K,N", [(128, 16), (256, 48), (1024, 256), (4096, 512), (11008, 64), (2176, 32),
                                 # the Llama-2-13B widths: straight-line instantiations at M = 1 (8 waves x 5 tiles, 12 x 9)
                                 (5120, 272), (13824, 80), (8192, 48)])
def test_w4a16_gemv_vs_oracle(ops, oracle, M, K, N):
    rng = np.random.default_rng(M + K + N)
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float16)
    x = (rng.random((M, K)) - 0.3).astype(np.float16)
    qp, s = oracle.quantize_i4(w)
    processed = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    y = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), processed, torch.from_numpy(s).to(DEV)).cpu().numpy()
    ref = oracle.w8a16_gemm(x, oracle.i4_values(qp), s)      # same contract on the int4 values
    assert y.shape == (M, N) and _tier_a(y, ref).all(), np.abs(y.astype(np.float32) - ref.astype(np.float32)).max()


@pytest.mark.parametrize("K,N", [(2048, 8208), (2176, 16400), (4096, 24592), (5120, 13824), (5120, 27648), (2048, 8192),
                                 # N = 5120 on 256 CUs: 8-column units on int4 tiles (gemv_half_kernel<..., BITS = 4>)
                                 (5120, 5120), (13824, 5120), (4096, 1024)])
def test_w4a16_gemv_many_tile_rows_vs_oracle(ops, oracle, K, N):
    """M = 1 with more tile rows than fit the chip at once (N / 16 > 2 * CUs: 513, 1025, 1537 rows and the 13B fused shapes).
    Oracle on sampled columns from both ends, the expansion route on all of them; bias + residual epilogue; launch-to-launch bit
    identity.  (Written for the several-rows-per-workgroup kernel of round 4, which passed it and was shelved on its timings --
    tools/experiments/i4_rows_kernel.patch, profiles/r04_i4_rows.txt; kept as coverage of large-N decode shapes.)"""
    rng = np.random.default_rng(K + N)
    qp = rng.integers(-128, 128, (K, N // 2), dtype=np.int8)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    x = (rng.random((1, K)) - 0.5).astype(np.float16)
    pk = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    xd, sd = torch.from_numpy(x).to(DEV), torch.from_numpy(s).to(DEV)
    y = ops.w8_a16_gemm(xd, pk, sd)
    assert torch.equal(y, ops.w8_a16_gemm(xd, pk, sd))
    got = y.cpu().numpy()
    vals = oracle.i4_values(qp)
    for c0 in (0, 16, 32, 48, (N // 32) * 16, N - 64):
        cs = slice(c0, c0 + 64)
        ref = oracle.w8a16_gemm(x, np.ascontiguousarray(vals[:, cs]), s[cs])
        assert _tier_a(got[:, cs], ref).all(), (c0, np.abs(got[:, cs].astype(np.float32) - ref.astype(np.float32)).max())
    whole = ops.w8_a16_gemm(xd, pk, sd, path="mfma")            # the expansion route over all columns
    assert _tier_a(got, whole.cpu().numpy()).all()
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(1, N, dtype=torch.float16, device=DEV)
    assert torch.equal(ops.w8_a16_gemm(xd, pk, sd, bias=bias, residual=res), y + bias + res)


def test_w4a16_explicit_paths_and_their_limits(ops, oracle):
    """eetq_w4a16_gemm_ex: the explicit kernel paths agree with AUTO where both apply (tier A; bit-identical where AUTO takes
    that very kernel) and refuse what they cannot run instead of mis-computing."""
    K, N = 1024, 256
    rng = np.random.default_rng(5)
    qp = rng.integers(-128, 128, (K, N // 2), dtype=np.int8)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    pk = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    sd = torch.from_numpy(s).to(DEV)
    for M, same, others in ((1, "gemv", ("stream", "splitk", "mfma")), (4, "stream", ("gemv", "splitk", "mfma")),
                            (16, "stream", ("splitk", "mfma")), (40, "splitk", ("mfma",)), (200, "mfma", ())):
        x = torch.rand(M, K, dtype=torch.float16, device=DEV) - 0.5
        auto = ops.w8_a16_gemm(x, pk, sd)
        assert torch.equal(ops.w8_a16_gemm(x, pk, sd, path=same), auto), (M, same)
        for p in others:
            assert _tier_a(ops.w8_a16_gemm(x, pk, sd, path=p).cpu().numpy(), auto.float().cpu().numpy()).all(), (M, p)
    x = torch.rand(20, K, dtype=torch.float16, device=DEV)
    for bad in ("gemv", "stream", "mid", "tilesplit"):
        with pytest.raises(RuntimeError):
            ops.w8_a16_gemm(x, pk, sd, path=bad)
    with pytest.raises(RuntimeError):
        ops.w8_a16_gemm(torch.rand(200, K, dtype=torch.float16, device=DEV), pk, sd, path="splitk")   # M > 128


def test_w4a16_identity_is_exact_dequant(ops, oracle):
    """x = I selects single products: the GEMM must return fp16(q4 * s) exactly, for every nibble value and k position
    (GEMV path for the first rows, the expanded-int8 route for the whole identity)."""
    K, N = 256, 64
    rng = np.random.default_rng(1)
    q = rng.integers(-8, 8, (K, N), dtype=np.int8)
    q[:16, 0] = np.arange(-8, 8)
    qp = oracle.i4_from_values(q)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    processed = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    want = oracle.dequant(q, s)
    eye = torch.eye(K, dtype=torch.float16, device=DEV)
    full = ops.w8_a16_gemm(eye, processed, torch.from_numpy(s).to(DEV)).cpu().numpy()
    assert np.array_equal(full, want)
    for r in (0, 1, 7, 31, 32, 100, 255):
        one = ops.w8_a16_gemm(eye[r:r + 1], processed, torch.from_numpy(s).to(DEV)).cpu().numpy()
        assert np.array_equal(one[0], want[r]), r


@pytest.mark.parametrize("M", [5, 8, 16, 17, 33, 64, 65, 128, 129, 200, 1024])
def test_w4a16_larger_batches(ops, oracle, M):
    """2 <= M <= 16: the register-streaming MFMA kernel on int4 tiles; 17 <= M <= 128: the split-K MFMA tile on int4 tiles
    (round 4; it was the expansion route); above: nibbles expanded to int8 tiles + the W8A16 kernels.  All against the oracle
    on the exact integers; the expansion route (M > 128, or forced with path="mfma") is bit-identical to W8A16 on them."""
    K, N = 1024, 384
    rng = np.random.default_rng(M)
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float16)
    x = (rng.random((M, K)) - 0.3).astype(np.float16)
    qp, s = oracle.quantize_i4(w)
    processed = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(M, N, dtype=torch.float16, device=DEV)
    xd, sd = torch.from_numpy(x).to(DEV), torch.from_numpy(s).to(DEV)
    y = ops.w8_a16_gemm(xd, processed, sd)
    rows = sorted(set([0, M // 2, M - 1]))
    ref = oracle.w8a16_gemm(x[rows], oracle.i4_values(qp), s)
    assert _tier_a(y.cpu().numpy()[rows], ref).all()
    p8 = torch.from_numpy(oracle.gfx950_pack(oracle.i4_values(qp))).to(DEV)
    y8 = ops.w8_a16_gemm(xd, p8, sd)
    if M > 128:  # the expanded weight is exactly the int8 tile image of the same integers: bit-identical to W8A16 on them
        assert torch.equal(y, y8)
    else:        # same dequantised values, another summation order
        assert _tier_a(y.cpu().numpy(), y8.float().cpu().numpy()).all()
        if M > 16:
            assert torch.equal(ops.w8_a16_gemm(xd, processed, sd, path="mfma"), y8)    # the expansion route is still there
    assert torch.equal(ops.w8_a16_gemm(xd, processed, sd, bias=bias, residual=res), y + bias + res)
    assert torch.equal(ops.w8_a16_gemm(xd, processed, sd), y)                            # launch to launch: same bits


# 7B and 13B layer shapes, K with a 128-deep last step (384, 11008 + 128), one k tile in all (128), ragged N (not a multiple of
# the 32- / 64-column blocks), every row-tile count
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 11008), (11008, 4096), (5120, 5120), (5120, 13824), (13824, 5120),
                                 (128, 16), (384, 80), (11136, 48), (1024, 1040)])
@pytest.mark.parametrize("M", [17, 32, 40, 64, 96, 128])
def test_w4a16_splitk_tile_vs_oracle(ops, oracle, M, K, N):
    """gemm_splitk_kernel<..., BITS = 4>, AUTO's plan: oracle on sampled rows and columns (big shapes) / everything (small)."""
    rng = np.random.default_rng(K + N + M)
    qp = rng.integers(-128, 128, (K, N // 2), dtype=np.int8)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    x = (rng.random((M, K)) - 0.5).astype(np.float16)
    pk = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    xd, sd = torch.from_numpy(x).to(DEV), torch.from_numpy(s).to(DEV)
    y1 = ops.w8_a16_gemm(xd, pk, sd)
    y2 = ops.w8_a16_gemm(xd, pk, sd, path="splitk")
    assert torch.equal(y1, y2)                                   # AUTO takes this kernel at these M
    y = y1.cpu().numpy()
    rows = sorted(set([0, 1, M // 3, M - 1]))
    c0 = 0 if N <= 512 else int(rng.integers(0, N // 16 - 16)) * 16
    cols = slice(c0, min(N, c0 + 256))
    for cs in (cols, slice(max(0, N - 64), N)):
        ref = oracle.w8a16_gemm(x[rows], np.ascontiguousarray(oracle.i4_values(qp)[:, cs]), s[cs])
        assert _tier_a(y[rows][:, cs], ref).all(), (cs, np.abs(y[rows][:, cs].astype(np.float32) - ref.astype(np.float32)).max())


@pytest.mark.parametrize("K,N,M,plans", [
    (4096, 4096, 64, [(1, 1, 22), (1, 2, 22), (1, 4, 22), (2, 1, 22), (2, 2, 22), (2, 4, 22), (1, 2, 33), (2, 4, 33)]),
    (2048, 1024, 100, [(1, 1, 22), (2, 2, 22), (1, 4, 22)]),
    (1024, 2048, 20, [(1, 2, 33), (2, 1, 33), (2, 4, 22)]),
    (5120, 13824, 48, [(2, 2, 33), (1, 4, 22)]),
    # the geometry that exposed the store-data hazard in the int8 kernel: split launches, two workgroups per CU, more
    # workgroups than the chip holds
    (4096, 16384, 64, [(1, 2, 22)]),
])
def test_w4a16_splitk_every_plan(ops, oracle, K, N, M, plans):
    """Every (column blocks, slices, ring) instantiation of the int4 tile forced through EETQ_AMD_SPLITK_PLAN: tier A against
    the oracle on sampled rows, against the expansion route on the whole output, and launch-to-launch bit identity."""
    rng = np.random.default_rng(K + N + M)
    qp = rng.integers(-128, 128, (K, N // 2), dtype=np.int8)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    x = (rng.random((M, K)) - 0.5).astype(np.float16)
    pk = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    xd, sd = torch.from_numpy(x).to(DEV), torch.from_numpy(s).to(DEV)
    whole = ops.w8_a16_gemm(xd, pk, sd, path="mfma").cpu().numpy()
    rows = sorted(set([0, M // 2, M - 1]))
    cols = slice(0, 256)
    ref = oracle.w8a16_gemm(x[rows], np.ascontiguousarray(oracle.i4_values(qp)[:, cols]), s[cols])
    for nb, S, ring in plans:
        os.environ["EETQ_AMD_SPLITK_PLAN"] = "%d,%d,%d" % (nb, S, ring)
        try:
            y1 = ops.w8_a16_gemm(xd, pk, sd, path="splitk")
            y2 = ops.w8_a16_gemm(xd, pk, sd, path="splitk")
            torch.cuda.synchronize()
        finally:
            os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
        got = y1.cpu().numpy()
        assert torch.equal(y1, y2), (nb, S, ring)
        assert _tier_a(got[rows][:, cols], ref).all(), (nb, S, ring)
        assert _tier_a(got, whole).all(), (nb, S, ring, np.abs(got.astype(np.float32) - whole.astype(np.float32)).max())


@pytest.mark.parametrize("K,N,M,plans", [
    (4096, 4096, 128, [(2, 1, 33, 4), (1, 1, 33, 2), (2, 2, 22, 4), None]),       # None = AUTO (takes the row-group plan here)
    (4096, 6144, 100, [(2, 1, 33, 2), (1, 2, 22, 4), None]),
    (2176, 1040, 120, [(2, 1, 22, 2), (1, 4, 22, 4), None]),                     # K % 256 != 0 (last step 128 deep), ragged N
])
def test_w4a16_splitk_row_groups(ops, oracle, K, N, M, plans):
    """Row groups on int4 tiles (the int8 tile's geometry, gemm_splitk_kernel<..., BITS = 4>, blockIdx.y = row group): tier A against
    the oracle on sampled rows, against the expansion route on the whole output, launch-to-launch bit identity."""
    rng = np.random.default_rng(K + N + M)
    qp = rng.integers(-128, 128, (K, N // 2), dtype=np.int8)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    x = (rng.random((M, K)) - 0.5).astype(np.float16)
    pk = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    xd, sd = torch.from_numpy(x).to(DEV), torch.from_numpy(s).to(DEV)
    whole = ops.w8_a16_gemm(xd, pk, sd, path="mfma").cpu().numpy()
    rows = sorted(set([0, 33, M // 2, M - 1]))
    cols = slice(0, 256)
    ref = oracle.w8a16_gemm(x[rows], np.ascontiguousarray(oracle.i4_values(qp)[:, cols]), s[cols])
    for plan in plans:
        if plan is not None:
            os.environ["EETQ_AMD_SPLITK_PLAN"] = "%d,%d,%d,%d" % plan
        try:
            y1 = ops.w8_a16_gemm(xd, pk, sd, path="splitk" if plan is not None else "auto")
            y2 = ops.w8_a16_gemm(xd, pk, sd, path="splitk" if plan is not None else "auto")
            torch.cuda.synchronize()
        finally:
            os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
        got = y1.cpu().numpy()
        assert torch.equal(y1, y2), plan
        assert _tier_a(got[rows][:, cols], ref).all(), plan
        assert _tier_a(got, whole).all(), (plan, np.abs(got.astype(np.float32) - whole.astype(np.float32)).max())


def test_w4a16_medium_batch_is_graph_capturable(ops, oracle):
    """The expansion route needed a per-stream scratch that cannot be created during capture; the int4 tile does not."""
    K, N, M = 2048, 1024, 64
    rng = np.random.default_rng(3)
    qp = rng.integers(-128, 128, (K, N // 2), dtype=np.int8)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    pk = torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV)
    xd = torch.rand(M, K, dtype=torch.float16, device=DEV)
    sd = torch.from_numpy(s).to(DEV)
    eager = ops.w8_a16_gemm(xd, pk, sd)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.w8_a16_gemm(xd, pk, sd)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.w8_a16_gemm(xd, pk, sd)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("K,N", [(4096, 4096), (5120, 15360), (5120, 27648), (13824, 5120), (512, 64), (128, 16), (2048, 1024),
                                 (8192, 8192), (11008, 4096), (8192, 1024)])
@pytest.mark.parametrize("M", [1, 2, 5, 16])
def test_w4a16_stream_kernel_shapes(ops, oracle, M, K, N):
    """The int4 stream kernel over the decode shapes (every wave-count / depth instantiation of its launcher), against the
    oracle GEMM on the exact integers (rows sampled).  M = 1 through AUTO: the dot-product GEMV, or -- round 4, deep K and big
    weights -- this kernel with one row; whichever AUTO takes must also agree (tier A) with the other one."""
    rng = np.random.default_rng(K + N + M)
    qp = rng.integers(-128, 128, (K, N // 2), dtype=np.int8)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    x = (rng.random((M, K)) - 0.5).astype(np.float16)
    y = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV),
                        torch.from_numpy(s).to(DEV)).cpu().numpy()
    rows = sorted(set([0, M // 3, M - 1]))
    cols = slice(0, min(N, 256))
    ref = oracle.w8a16_gemm(x[rows], np.ascontiguousarray(oracle.i4_values(qp)[:, cols]), s[cols])
    assert _tier_a(y[rows][:, cols], ref).all()
    if M == 1 and K % 128 == 0:
        for path in ("gemv", "stream"):
            other = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV),
                                    torch.from_numpy(s).to(DEV), path=path).cpu().numpy()
            assert _tier_a(other[:, cols], ref).all(), path


def test_w4a16_linear_module(ops, oracle):
    from eetq_amd.modules.qlinear import W4A16Linear
    torch.manual_seed(2)
    lin = torch.nn.Linear(512, 256, bias=True, dtype=torch.float16, device=DEV)
    mod = W4A16Linear.from_torch(lin)
    assert mod.qweight.shape == (512, 128) and mod.weight_scales.shape == (256,)
    x = torch.rand(3, 512, dtype=torch.float16, device=DEV)
    with torch.no_grad():
        ref = lin(x)
    y = mod(x)
    assert y.shape == ref.shape
    # 4-bit per-channel quantisation of a U(+-1/sqrt(512)) layer: error ~ scale/2 * sqrt(K) * E|x|
    assert (y - ref).abs().max().item() < 0.12
    qp, s = oracle.quantize_i4(lin.weight.detach().t().contiguous().cpu().numpy())
    want = oracle.w8a16_gemm(x.cpu().numpy(), oracle.i4_values(qp), s).astype(np.float32) + lin.bias.detach().cpu().numpy().astype(np.float32)
    assert np.abs(y.cpu().numpy().astype(np.float32) - want).max() < 3e-3


def test_w4a16_expansion_route_on_concurrent_streams(ops, oracle):
    """M > 16 expands the nibbles into a library-owned buffer: launches on different streams must not share it.  Two streams,
    two different weights, interleaved many times: every result equals the one computed alone."""
    rng = np.random.default_rng(9)
    K, N, M = 2048, 1024, 96
    ws, ss, refs = [], [], []
    x = torch.from_numpy((rng.random((M, K)) - 0.5).astype(np.float16)).to(DEV)
    for i in range(2):
        qp = rng.integers(-128, 128, (K, N // 2), dtype=np.int8)
        ws.append(torch.from_numpy(oracle.gfx950_pack_i4(qp)).to(DEV))
        ss.append(torch.from_numpy((rng.random(N) * 0.02 + 0.001).astype(np.float16)).to(DEV))
        refs.append(ops.w8_a16_gemm(x, ws[i], ss[i]).clone())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for it in range(40):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs[i].append(ops.w8_a16_gemm(x, ws[i], ss[i]))
    torch.cuda.synchronize()
    for i in range(2):
        assert all(torch.equal(o, refs[i]) for o in outs[i]), i
