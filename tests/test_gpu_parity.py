"""Parity of the HIP path (through the C ABI, via eetq_amd.ops) against the oracle.  Needs an MI355X.

Bars: quantise / pack / unpack -> bit-exact.  GEMM/GEMV -> tier-A tolerance vs the oracle contract
(|err| <= 1e-3 * max|y| + 2e-3 * |y|) and the reference's own atol = 1e-2 vs CPU torch
float16 nn.Linear.  Full BASELINE sizes are covered by size-independent properties (identity -> exact dequant,
pack/unpack round trip, GEMV-vs-MFMA cross check, torch fp32 reference)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import eetq_amd.ops as _ops
    from eetq_amd import _lib
    assert _lib.lib().eetq_device_supported() == 1, "kernels are built for gfx950 only"
    return _ops


def _tier_a(y, ref):
    y = y.astype(np.float32)
    ref = ref.astype(np.float32)
    tol = 1e-3 * np.abs(ref).max() + 2e-3 * np.abs(ref)
    return np.abs(y - ref) <= tol


def _rand_case(K, N, M, seed, wscale=0.02):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((K, N)) * wscale).astype(np.float16)
    x = rng.random((M, K)).astype(np.float16)
    return w, x


# ---------------------------------------------------------------- quantise / pack: bit-exact

@pytest.mark.parametrize("name", ["quant_edge_f16_k128_n64", "quant_rand_f16_k192_n256", "quant_rand_f32_k64_n64"])
def test_quant_weights_golden(ops, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    w = torch.from_numpy(g["w"])
    raw, processed, scales = ops.quant_weights(w, torch.int8, True)
    assert raw.device.type == "cpu" and processed.shape == w.shape and scales.dtype == w.dtype
    assert np.array_equal(raw.numpy(), g["q"])
    assert scales.numpy().tobytes() == g["s"].tobytes()
    assert np.array_equal(processed.numpy(), g["gfx950"])
    p2, s2 = ops.quant_weights(w, torch.int8, False)
    assert np.array_equal(p2.numpy(), g["gfx950"]) and s2.numpy().tobytes() == g["s"].tobytes()
    p3, _ = ops.quant_weights(w, torch.int8, False, layout="sm80")
    assert np.array_equal(p3.numpy(), g["sm80"])
    # examples/layers/test_w8a16_gemm.py:33-41: preprocess(raw) == processed
    assert np.array_equal(ops.preprocess_weights(raw).numpy(), g["gfx950"])
    assert np.array_equal(ops.preprocess_weights(raw, False, "sm80").numpy(), g["sm80"])


@pytest.mark.parametrize("K,N,dtype", [(64, 16, np.float16), (64, 48, np.float32), (256, 128, np.float16),
                                       (1024, 4096, np.float16), (4096, 4096, np.float16), (4096, 11008, np.float16),
                                       (512, 1040, np.float32), (320, 80, np.float16), (576, 2000, np.float32)])
def test_quantize_bit_exact_vs_oracle(ops, oracle, K, N, dtype):
    rng = np.random.default_rng(K * 7 + N)
    w = (rng.standard_normal((K, N)) * 0.02).astype(dtype)
    w[:, N // 2] = 0
    w[3, 1] = 0.5
    raw, processed, scales = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.int8, True)
    assert raw.is_cuda and processed.is_cuda and scales.is_cuda
    q, s = oracle.quantize(w)
    assert np.array_equal(raw.cpu().numpy(), q)
    assert scales.cpu().numpy().tobytes() == s.tobytes()
    assert np.array_equal(processed.cpu().numpy(), oracle.gfx950_pack(q))


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_quantize_row_block_partials_edge_values(ops, oracle, dtype):
    """The two-launch quantiser reduces per-row-block maxima (3 blocks of 128 rows + a 5-tile strip here): NaN is ignored
    by the running maximum like std::max (cutlass_preprocessors.cc:619-628), inf makes the scale inf, an all-zero column
    quantises to 127, ties round away from zero -- wherever in K the special value sits; both target layouts."""
    K, N = 320, 128
    rng = np.random.default_rng(99)
    w = (rng.standard_normal((K, N)) * 0.05).astype(dtype)
    w[:, 5] = 0                                   # zero column -> scale 0 -> 0/0 -> 127
    w[7, 9] = np.nan                              # NaN in the first row block
    w[300, 10] = np.nan                           # NaN in the last (partial) row block
    w[200, 11] = np.inf                           # inf in the middle block
    w[130, 12] = -np.inf
    w[:, 13] = 0
    w[319, 13] = 1.0                              # the column maximum is the very last row
    w[0, 14] = 2.0                                # ... or the very first
    w[:, 15] = (np.arange(K) % 5 - 2) * 0.5 / 64  # exact ties: k * amax / 256
    w[1, 15] = 1.0
    for layout in ("gfx950", "sm80"):
        raw, processed, scales = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.int8, True, layout=layout)
        q, s = oracle.quantize(w)
        assert np.array_equal(raw.cpu().numpy(), q), layout
        assert scales.cpu().numpy().tobytes() == s.tobytes(), layout
        ref = oracle.gfx950_pack(q) if layout == "gfx950" else oracle.sm80_pack(q)
        assert np.array_equal(processed.cpu().numpy(), ref), layout
        only, scales2 = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.int8, False, layout=layout)  # no row-major copy
        assert np.array_equal(only.cpu().numpy(), ref), layout
        assert scales2.cpu().numpy().tobytes() == s.tobytes(), layout


def _near_tie_matrix(dtype, K, N, seed):
    """Every element sits on, or a few ulps beside, a rounding tie of w / scale: the inputs on which a reciprocal-based
    quotient and the reference's IEEE division (cutlass_preprocessors.cc:644-648) could round differently."""
    rng = np.random.default_rng(seed)
    if dtype == np.float32:
        amax = (10.0 ** rng.uniform(-3, 1, N)).astype(np.float32)
        amax[1], amax[2], amax[3] = 1e-36, 3e35, 1e-41          # scale below / above the trusted range, subnormal scale
    else:
        amax = (2.0 ** rng.integers(-10, 4, N) * rng.choice([1.0, 1.5, 1.25], N)).astype(np.float16)
        amax[1], amax[2] = 2.0 ** -20, 6e-8                       # subnormal fp16 columns
    scale = amax.astype(np.float32) * np.float32(1.0 / 128.0)
    k = rng.integers(-128, 128, (K, N)).astype(np.float32) + np.float32(0.5)
    if dtype == np.float32:  # odd columns: just OUTSIDE the band that falls back to the division (3e-5 .. 3e-4 from a tie)
        off = (rng.choice([-1.0, 1.0], (K, N)) * 10.0 ** rng.uniform(-4.5, -3.5, (K, N))).astype(np.float32)
        k[:, 1::2] += off[:, 1::2]
    w = (k * scale[None, :]).astype(dtype)
    for _ in range(3):                                            # -3 .. +3 ulps of the input type
        step = rng.integers(-1, 2, (K, N))
        w = np.where(step > 0, np.nextafter(w, dtype(np.inf)), np.where(step < 0, np.nextafter(w, dtype(-np.inf)), w)).astype(dtype)
    w = np.clip(w, -amax[None, :], amax[None, :]).astype(dtype)
    w[rng.integers(0, K, N), np.arange(N)] = amax * rng.choice([-1, 1], N).astype(dtype)  # the maximum is attained
    return w


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
@pytest.mark.parametrize("K,N", [(256, 128), (320, 80), (9216, 64)])
def test_quantize_near_ties_native_layout_only(ops, oracle, dtype, K, N):
    """quant_weights(..., return_unprocessed=False) in the native layout runs the column-major kernel whose quotient is
    fma(w, rcp(s), 128) with an exact fallback near ties: bit-identical to the oracle on a matrix made of ties and their
    neighbours, on special values, on a ragged last strip and on the 4-tile strip (K > 8192)."""
    w = _near_tie_matrix(dtype, K, N, K + N)
    w[5, 9] = np.nan
    w[:, 6] = 0
    if N > 64:
        w[17, 70] = np.inf
    processed, scales = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.int8, False)
    q, s = oracle.quantize(w)
    assert scales.cpu().numpy().tobytes() == s.tobytes()
    got = processed.cpu().numpy()
    ref = oracle.gfx950_pack(q)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} bytes differ"
    raw, processed2, _ = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.int8, True)  # the row-major kernel agrees
    assert np.array_equal(raw.cpu().numpy(), q)
    assert np.array_equal(processed2.cpu().numpy(), ref)


def test_quantize_default_init_4096(ops, oracle):
    """The distribution of the headline configs: nn.Linear default init, seed 1."""
    torch.manual_seed(1)
    lin = torch.nn.Linear(4096, 4096, bias=False, dtype=torch.float16)
    w = lin.weight.detach().t().contiguous()
    processed, scales = ops.quant_weights(w, torch.int8, False)
    q, s = oracle.quantize(w.numpy())
    assert scales.numpy().tobytes() == s.tobytes()
    assert np.array_equal(processed.numpy(), oracle.gfx950_pack(q))


@pytest.mark.parametrize("layout", ["gfx950", "sm80"])
@pytest.mark.parametrize("K,N", [(64, 64), (192, 256), (4096, 4096), (11008, 4096)])
def test_pack_unpack_roundtrip_and_oracle(ops, oracle, layout, K, N):
    rng = np.random.default_rng(K + N)
    q = rng.integers(-128, 128, (K, N), dtype=np.int8)
    packed = ops.preprocess_weights(torch.from_numpy(q).to(DEV), False, layout)
    ref = oracle.gfx950_pack(q) if layout == "gfx950" else oracle.sm80_pack(q)
    assert np.array_equal(packed.cpu().numpy(), ref)
    back = ops.unprocess_weights(packed, layout)
    assert np.array_equal(back.cpu().numpy(), q)
    if layout == "sm80":  # NVIDIA-written checkpoint bytes -> native layout
        native = ops.convert_layout(packed, "sm80", "gfx950")
        assert np.array_equal(native.cpu().numpy(), oracle.gfx950_pack(q))


def test_quant_weights_expert_stack(ops, oracle):
    """[E, K, N] input (reference fpA_intB_gemm_wrapper.cu:36, 45-66): outputs [E, K, N] / [E, N]; the reference quantises
    expert 0 only (it passes the 2-D shape on, :82 / :90), here every expert holds the oracle's bits."""
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((3, 128, 64)) * 0.05).astype(np.float16)
    raw, processed, scales = ops.quant_weights(torch.from_numpy(w), torch.int8, True)
    assert tuple(raw.shape) == (3, 128, 64) and tuple(processed.shape) == (3, 128, 64) and tuple(scales.shape) == (3, 64)
    assert scales.dtype == torch.float16 and not raw.is_cuda
    for e in range(3):
        q, s = oracle.quantize(w[e])
        assert np.array_equal(raw[e].numpy(), q)
        assert scales[e].numpy().tobytes() == s.tobytes()
        assert np.array_equal(processed[e].numpy(), oracle.gfx950_pack(q))
    two = ops.quant_weights(torch.from_numpy(w), torch.int8, False)
    assert len(two) == 2 and torch.equal(two[0], processed) and torch.equal(two[1], scales)


def test_shape_errors(ops):
    with pytest.raises(RuntimeError):
        ops.quant_weights(torch.zeros(32, 64, dtype=torch.float16), torch.int8)        # K % 64
    with pytest.raises(RuntimeError):
        ops.quant_weights(torch.zeros(64, 24, dtype=torch.float16), torch.int8)        # N % 16
    with pytest.raises(RuntimeError):
        ops.preprocess_weights(torch.zeros(64, 32, dtype=torch.int8), False, "sm80")   # N % 64 for sm80
    with pytest.raises(RuntimeError):
        ops.quant_weights(torch.zeros(2, 2, 64, 64, dtype=torch.float16), torch.int8)  # 4-D (reference :36)
    x = torch.zeros(1, 100, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.w8_a16_gemm(x, torch.zeros(100, 64, dtype=torch.int8, device=DEV), torch.ones(64, dtype=torch.float16, device=DEV))


# ---------------------------------------------------------------- GEMV / GEMM vs the oracle contract

def _run_gemm(ops, oracle, w, x, path="auto"):
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    y = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), processed, scales, path=path)
    torch.cuda.synchronize()
    return y.cpu().numpy(), q, s


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("K,N", [(64, 16), (128, 48), (1024, 256), (4096, 512), (11008, 64), (1088, 32)])
def test_gemv_vs_oracle(ops, oracle, M, K, N):
    w, x = _rand_case(K, N, M, seed=K + N + M)
    y, q, s = _run_gemm(ops, oracle, w, x, path="gemv")
    ref = oracle.w8a16_gemm(x, q, s)
    ok = _tier_a(y, ref)
    assert ok.all(), "max err %g at %s" % (np.abs(y.astype(np.float32) - ref.astype(np.float32)).max(),
                                          np.argwhere(~ok)[:4])


@pytest.mark.parametrize("M", [2, 8, 13, 16, 17, 24, 32])
@pytest.mark.parametrize("K,N", [(2048, 8192), (2112, 8256), (2048, 8208), (2048, 5120), (2048, 6144)])
def test_stream_wide_n_vs_oracle(ops, oracle, M, K, N):
    """Shapes where the launcher's cost model puts two 16-column tile rows into one workgroup (they share the activation
    fragments): N = 8192 / 8256, and N = 5120 / 6144 where one tile row per workgroup would leave a second, mostly empty
    round of workgroups; 8208 is the N % 32 != 0 fallback to one tile row.  M = 17 / 24 / 32 (round 6): the 32-row per-wave
    ring with two MFMA row tiles per weight tile (streamk_kernel<..., XM = 5>), two tile rows per workgroup beyond one
    workgroup per CU."""
    w, x = _rand_case(K, N, M, seed=7 * K + N + M)
    x[:, 1::2] *= -1
    y, q, s = _run_gemm(ops, oracle, w, x, path="stream")
    ref = oracle.w8a16_gemm(x, q, s)
    ok = _tier_a(y, ref)
    assert ok.all(), "max err %g at %s" % (np.abs(y.astype(np.float32) - ref.astype(np.float32)).max(),
                                          np.argwhere(~ok)[:4])


@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 33, 48, 64])
@pytest.mark.parametrize("K,N", [(64, 16), (128, 48), (1024, 256), (4096, 512), (11008, 64), (2048, 32), (2112, 16)])
def test_stream_mfma_vs_oracle(ops, oracle, M, K, N):
    w, x = _rand_case(K, N, M, seed=5 * K + N + M)
    x[:, ::3] *= -1
    y, q, s = _run_gemm(ops, oracle, w, x, path="stream")
    ref = oracle.w8a16_gemm(x, q, s)
    ok = _tier_a(y, ref)
    assert ok.all(), "max err %g at %s" % (np.abs(y.astype(np.float32) - ref.astype(np.float32)).max(),
                                          np.argwhere(~ok)[:4])


@pytest.mark.parametrize("M", [1, 7, 32, 33, 48, 64, 65, 100, 128])
@pytest.mark.parametrize("K,N", [(64, 16), (128, 80), (1024, 256), (4096, 512), (11008, 64), (2048, 48), (2112, 144),
                                 (320, 32)])
def test_mid_tile_vs_oracle(ops, oracle, M, K, N):
    w, x = _rand_case(K, N, M, seed=11 * K + N + M)
    x[:, ::3] *= -1
    y, q, s = _run_gemm(ops, oracle, w, x, path="mid")
    ref = oracle.w8a16_gemm(x, q, s)
    ok = _tier_a(y, ref)
    assert ok.all(), "max err %g at %s" % (np.abs(y.astype(np.float32) - ref.astype(np.float32)).max(),
                                          np.argwhere(~ok)[:4])


@pytest.mark.parametrize("M", [1, 17, 32, 33, 48, 64, 65, 100, 128])
@pytest.mark.parametrize("K,N", [(256, 32), (512, 80), (1024, 256), (4096, 512), (4096, 4096), (11008, 64), (2048, 48),
                                 (2112, 144), (5120, 5120), (1280, 11008)])
def test_splitk_tile_vs_oracle(ops, oracle, M, K, N):
    """Split-K medium-batch kernel (planned column-block width / slice count per shape): ragged N and K, every row-tile
    count, shapes that plan to 1, 2 and 4 slices; against the oracle (sampled columns for the big shapes)."""
    w, x = _rand_case(K, N, M, seed=M * 13 + K + N)
    cols = np.arange(N) if N <= 512 else np.unique(np.concatenate([np.arange(48), np.arange(N - 48, N),
                                                                    np.random.default_rng(N).integers(0, N, 160)]))
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    y = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), processed, torch.from_numpy(s).to(DEV), path="splitk").cpu().numpy()
    ref = oracle.w8a16_gemm(x, np.ascontiguousarray(q[:, cols]), np.ascontiguousarray(s[cols]))
    assert y.shape == (M, N)
    assert _tier_a(y[:, cols], ref).all(), np.abs(y[:, cols].astype(np.float32) - ref.astype(np.float32)).max()


def test_splitk_is_deterministic_and_back_to_back_safe(ops, oracle):
    """The in-launch reduction adds the slices' partial tiles in slice order whatever the arrival order: 200 back-to-back
    launches (tickets are monotonic and never reset) on two shapes give bit-identical results every time, also when
    interleaved with other kernels and replayed from a HIP graph."""
    w, x = _rand_case(4096, 4096, 64, seed=7)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    first = ops.w8_a16_gemm(xd, processed, scales, path="splitk")
    ref = oracle.w8a16_gemm(x[:4], q, s)
    assert _tier_a(first[:4].cpu().numpy(), ref).all()
    x2 = xd[:40, :2048].contiguous()
    p2 = ops.preprocess_weights(torch.from_numpy(np.ascontiguousarray(q[:2048, :1024])).to(DEV))
    first2 = ops.w8_a16_gemm(x2, p2, scales[:1024].contiguous(), path="splitk")
    for i in range(200):
        assert torch.equal(ops.w8_a16_gemm(xd, processed, scales, path="splitk"), first), i
        if i % 3 == 0:
            assert torch.equal(ops.w8_a16_gemm(x2, p2, scales[:1024].contiguous(), path="splitk"), first2), i
    y = torch.empty_like(first)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.w8_a16_gemm_(xd, processed, scales, y, 64, 4096, 4096)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            ops.w8_a16_gemm_(xd, processed, scales, y, 64, 4096, 4096)
    for _ in range(20):
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, ops.w8_a16_gemm(xd, processed, scales))


@pytest.mark.parametrize("M,K,N", [(128, 8192, 4096),     # four slices of 32 K steps
                                   (100, 5120, 5120),     # two slices of 40; ragged last rows of the single row tile
                                   (256, 11008, 4096),    # two row tiles, two slices of 86
                                   (128, 11008, 4096),    # four slices of 43 (an odd step count per slice)
                                   (130, 4160, 4112),     # does not split (130 tiles x 2 > CUs): the unsplit tiled kernel
                                   (128, 1088, 4096),     # K too shallow to split
                                   (512, 11008, 4096)])   # the 128 x 128 tile, two slices of 86 (64 KiB partial tiles)
def test_tile_splitk_vs_oracle_and_unsplit(ops, oracle, M, K, N):
    """path="tilesplit": K slices of the tiled kernel's 128 x 64 tile, partial tiles handed over through the split-K scratch
    and added in slice order.  Tier A against the oracle on sampled rows and against the unsplit tiled kernel on the whole
    output; bias + residual epilogue bit-identical to separate adds; two launches give the same bits; AUTO for M > 96 agrees."""
    w, x = _rand_case(K, N, M, seed=M + K + N)
    x[:, ::5] *= -1
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    rows = sorted(set([0, 31, M // 2, M - 1]))
    ref = oracle.w8a16_gemm(x[rows], q, s)
    y1 = ops.w8_a16_gemm(xd, processed, scales, path="tilesplit")
    y2 = ops.w8_a16_gemm(xd, processed, scales, path="tilesplit")
    whole = ops.w8_a16_gemm(xd, processed, scales, path="mfma")
    assert torch.equal(y1, y2)
    got = y1.cpu().numpy()
    assert _tier_a(got[rows], ref).all(), np.abs(got[rows].astype(np.float32) - ref.astype(np.float32)).max()
    assert _tier_a(got, whole.cpu().numpy()).all()
    assert _tier_a(ops.w8_a16_gemm(xd, processed, scales).cpu().numpy(), whole.cpu().numpy()).all()
    g = torch.Generator(device=DEV); g.manual_seed(1)
    bias = torch.rand(N, device=DEV, generator=g).half()
    res = torch.rand(M, N, device=DEV, generator=g).half()
    fused = ops.w8_a16_gemm(xd, processed, scales, path="tilesplit", bias=bias, residual=res)
    assert torch.equal(fused, (y1 + bias) + res)


@pytest.mark.parametrize("M,K,N", [(1024, 5120, 5120), (1000, 13824, 5120), (512, 5120, 10240 + 128)])
def test_mfma_ragged_last_round_runs_in_two_k_slices(ops, oracle, M, K, N):
    """M = 1024 at N = 5120 is 320 wide tiles on 256 CUs: whole rounds run unsplit, the ragged round's columns as narrow tiles
    in two K slices (gemm.hip).  Tier A against the oracle on sampled rows and against torch's fp32 matmul over the kernel's
    fp16 dequantised weights everywhere; fused bias + residual bit-identical to separate adds; repeated launches identical."""
    w, x = _rand_case(K, N, M, seed=M + K + N)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    rows = sorted(set([0, 127, 128, M // 2, M - 1]))
    y1 = ops.w8_a16_gemm(xd, processed, scales, path="mfma")
    y2 = ops.w8_a16_gemm(xd, processed, scales, path="mfma")
    assert torch.equal(y1, y2)
    got = y1.cpu().numpy()
    assert _tier_a(got[rows], oracle.w8a16_gemm(x[rows], q, s)).all()
    wdq = (torch.from_numpy(q).to(DEV).half() * scales[None, :]).float()
    ref = xd.float() @ wdq
    assert bool(((y1.float() - ref).abs() <= 1e-3 * ref.abs().max() + 2e-3 * ref.abs()).all())
    g = torch.Generator(device=DEV); g.manual_seed(2)
    bias = torch.rand(N, device=DEV, generator=g).half()
    res = torch.rand(M, N, device=DEV, generator=g).half()
    assert torch.equal(ops.w8_a16_gemm(xd, processed, scales, path="mfma", bias=bias, residual=res), (y1 + bias) + res)


def test_tile_splitk_is_deterministic_across_launches_streams_and_graphs(ops, oracle):
    """The split form of the tiled kernel shares the split-K tickets (one array per slice count, monotonic): interleaved with
    split-K launches of the same slice counts, on two streams, and replayed from a HIP graph, every result is bit-identical
    to the first."""
    w, x = _rand_case(11008, 4096, 128, seed=11)   # four slices, like the split-K launches below: the SAME ticket array (K > 8192:
                                                   # AUTO, which the graph below launches, takes the K-sliced tiled kernel too)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    first = ops.w8_a16_gemm(xd, processed, scales, path="tilesplit")
    x64 = xd[:64].contiguous()
    first64 = ops.w8_a16_gemm(x64, processed, scales, path="splitk")
    for i in range(100):
        assert torch.equal(ops.w8_a16_gemm(xd, processed, scales, path="tilesplit"), first), i
        if i % 2 == 0:
            assert torch.equal(ops.w8_a16_gemm(x64, processed, scales, path="splitk"), first64), i
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        on_side = [ops.w8_a16_gemm(xd, processed, scales, path="tilesplit") for _ in range(20)]
    on_main = [ops.w8_a16_gemm(xd, processed, scales, path="tilesplit") for _ in range(20)]
    torch.cuda.synchronize()
    assert all(torch.equal(t, first) for t in on_side + on_main)
    y = torch.empty_like(first)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            ops.w8_a16_gemm_(xd, processed, scales, y, 128, 4096, 11008)
    for _ in range(10):
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, first)


@pytest.mark.parametrize("K,N,M,plans", [
    # plans = (column blocks, K slices, ring): every instantiation of the library, including the ones AUTO does not pick.  The
    # first three rows are among the (ring 22, S > 1) plans that returned wrong elements when two split workgroups shared a CU
    # (the slab stores' data-register hazard, fixed in round 3: gemm_splitk_kernel.hpp).
    (4096, 4096, 64, [(1, 4, 22), (1, 2, 22), (2, 4, 22), (1, 2, 33), (2, 4, 33), (2, 1, 33)]),
    (4096, 11008, 32, [(2, 2, 22), (2, 4, 22), (1, 2, 22), (1, 1, 22), (2, 1, 33)]),
    (4096, 11008, 64, [(1, 2, 22), (1, 4, 22), (2, 1, 33), (1, 1, 22)]),
    (4160, 4112, 50, [(1, 4, 22), (2, 2, 22), (1, 2, 33)]),            # K % 256 != 0, N % 32 != 0, M % 32 != 0
    (2048, 1024, 128, [(1, 2, 22), (2, 4, 22), (1, 1, 22)]),
    (2048, 1024, 9, [(1, 2, 33), (2, 4, 22)]),
    # split launches with two workgroups per CU AND more workgroups than the chip holds (1024 / 688 of them, rings of 80 KiB):
    # the geometry in which the slab stores' data registers were overwritten before round 3's fix -- three launches out of
    # three returned wrong elements (gemm_splitk_kernel.hpp, tools/experiments/sk_debug.py)
    (4096, 16384, 64, [(1, 2, 22)]),
    (2048, 22016, 50, [(1, 2, 22), (1, 4, 22)]),
])
def test_splitk_every_plan(ops, oracle, K, N, M, plans):
    """Every (column blocks, slices, ring) instantiation forced through EETQ_AMD_SPLITK_PLAN: tier A against the oracle on
    sampled rows, tier A against the tiled kernel on the whole output, and two launches give the same bits."""
    w, x = _rand_case(K, N, M, seed=K + N + M)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    rows = sorted(set([0, M // 2, M - 1]))
    ref = oracle.w8a16_gemm(x[rows], q, s)
    whole = ops.w8_a16_gemm(xd, processed, scales, path="mfma").cpu().numpy()
    for nb, S, ring in plans:
        os.environ["EETQ_AMD_SPLITK_PLAN"] = "%d,%d,%d" % (nb, S, ring)
        try:
            y1 = ops.w8_a16_gemm(xd, processed, scales, path="splitk")
            y2 = ops.w8_a16_gemm(xd, processed, scales, path="splitk")
            torch.cuda.synchronize()
        finally:
            os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
        got = y1.cpu().numpy()
        assert torch.equal(y1, y2), (nb, S, ring)
        assert _tier_a(got[rows], ref).all(), (nb, S, ring)
        assert _tier_a(got, whole).all(), (nb, S, ring, np.abs(got.astype(np.float32) - whole.astype(np.float32)).max())


@pytest.mark.parametrize("K,N,M,plans", [
    # plans = (column blocks, K slices, ring, ROW GROUPS): the batch cut along M (round 5, gemm_splitk_kernel.hpp "R"): r groups of
    # 32 * ceil(M / (32 r)) rows, each with its own slab / ticket unit when the plan also slices K
    (4096, 4096, 128, [(2, 1, 33, 4), (1, 1, 33, 2), (2, 2, 33, 2), (2, 2, 22, 4), (1, 4, 22, 4)]),
    (4096, 4096, 256, [(2, 1, 33, 4), (2, 1, 22, 8), (2, 2, 22, 2), (1, 2, 22, 3)]),
    (4096, 4096, 100, [(2, 1, 33, 4), (2, 1, 33, 2), (1, 2, 22, 2)]),            # last group ragged: rows 96..99
    (4096, 6144, 130, [(2, 1, 22, 2), (2, 1, 33, 3), (2, 2, 22, 5)]),            # last group: 2 rows / 34 rows
    (4160, 4112, 200, [(1, 1, 22, 2), (2, 4, 22, 4), (1, 2, 33, 4)]),            # K % 256 != 0, N % 32 != 0
    (2048, 1024, 512, [(2, 1, 22, 4), (2, 2, 22, 8), (1, 1, 33, 16)]),
    (1024, 2048, 1000, [(2, 1, 22, 8), (2, 2, 33, 16), (1, 4, 22, 32)]),         # up to 32 groups (M <= 1024)
    # M <= 96 (the planner's search cuts rows from M = 33 on): 32-row groups, with and without K slices, a ragged last group
    (4096, 4096, 64, [(1, 1, 33, 2), (2, 1, 33, 2), (2, 2, 22, 2), (1, 4, 22, 2)]),
    (5120, 5120, 96, [(2, 1, 33, 3), (2, 2, 22, 3), (1, 1, 22, 3), (2, 4, 22, 2)]),
    (8192, 2048, 70, [(2, 4, 22, 3), (1, 2, 33, 3), (2, 2, 33, 2)]),             # last group: 6 rows / 22 rows
    (4096, 2048, 112, [(1, 1, 33, 4), (2, 2, 33, 4), (1, 2, 33, 2)]),
])
def test_splitk_row_groups(ops, oracle, K, N, M, plans):
    """Row groups of the split-K tile, every combination with column blocks / K slices forced through
    EETQ_AMD_SPLITK_PLAN="nb,s,ring,r": tier A against the oracle on rows of the first, a middle and the last group, tier A against
    the tiled kernel on the whole output, the same bits from two launches, and the fused bias + residual epilogue bit-identical to
    the separate adds (the residual pointer moves with the row group)."""
    w, x = _rand_case(K, N, M, seed=K + N + M)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    rows = sorted(set([0, 31, 32, M // 2, M - 33 if M > 40 else 0, M - 1]))
    ref = oracle.w8a16_gemm(x[rows], q, s)
    whole = ops.w8_a16_gemm(xd, processed, scales, path="mfma").cpu().numpy()
    torch.manual_seed(M)
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(M, N, dtype=torch.float16, device=DEV)
    for nb, S, ring, r in plans:
        os.environ["EETQ_AMD_SPLITK_PLAN"] = "%d,%d,%d,%d" % (nb, S, ring, r)
        try:
            y1 = ops.w8_a16_gemm(xd, processed, scales, path="splitk")
            y2 = ops.w8_a16_gemm(xd, processed, scales, path="splitk")
            y3 = ops.w8_a16_gemm(xd, processed, scales, path="splitk", bias=bias, residual=res)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("EETQ_AMD_SPLITK_PLAN", None)
        got = y1.cpu().numpy()
        assert torch.equal(y1, y2), (nb, S, ring, r)
        assert torch.equal(y3, (y1 + bias) + res), (nb, S, ring, r)
        assert _tier_a(got[rows], ref).all(), (nb, S, ring, r)
        assert _tier_a(got, whole).all(), (nb, S, ring, r, np.abs(got.astype(np.float32) - whole.astype(np.float32)).max())


@pytest.mark.parametrize("K,N,M", [(4096, 4096, 100), (4096, 4096, 128), (4096, 4096, 256), (4096, 6144, 128), (5120, 5120, 192),
                                   (4096, 4096, 192),
                                   # M <= 128 through the planner's search (gemm_splitk.hip::splitk_plan): 32- and 64-row groups
                                   (4096, 4096, 48), (4096, 4096, 64), (4096, 4096, 96), (5120, 5120, 96), (5120, 5120, 128),
                                   (8192, 8192, 128), (4096, 2048, 112), (8192, 2048, 128)])
def test_auto_takes_the_row_group_plan_and_matches_the_oracle(ops, oracle, K, N, M):
    """AUTO where its split-K plan cuts the batch into row groups (eetq_diag_auto_path: SPLITK, detail = groups): oracle on sampled
    rows, the tiled kernel on everything, graph capture and bit-identical replays."""
    import ctypes
    from eetq_amd import _lib
    p, d = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.lib().eetq_diag_auto_path(8, M, N, K, ctypes.byref(p), ctypes.byref(d)))
    assert (p.value, d.value > 0) == (5, True), (p.value, d.value)
    w, x = _rand_case(K, N, M, seed=K + N + M)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    rows = sorted(set([0, M // 3, M - 1]))
    y = ops.w8_a16_gemm(xd, processed, scales)
    assert _tier_a(y.cpu().numpy()[rows], oracle.w8a16_gemm(x[rows], q, s)).all()
    assert _tier_a(y.cpu().numpy(), ops.w8_a16_gemm(xd, processed, scales, path="mfma").cpu().numpy()).all()
    out = torch.empty_like(y)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.w8_a16_gemm_(xd, processed, scales, out, M, N, K)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        ops.w8_a16_gemm_(xd, processed, scales, out, M, N, K)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, y)


@pytest.mark.parametrize("M,K,N", [(5, 64, 16), (8, 256, 128), (17, 128, 144), (64, 1024, 256), (128, 512, 128),
                                   (130, 192, 272), (300, 2048, 384), (8, 4096, 4096), (64, 4096, 1024),
                                   # narrow (128 x 64) and wide tiles with ragged edges: N below / not a multiple of the
                                   # tile width, M one past a tile, odd and even K step counts
                                   (200, 512, 16), (200, 576, 48), (129, 640, 80), (600, 704, 208), (257, 384, 8192),
                                   (1100, 320, 4160)])
def test_mfma_gemm_vs_oracle(ops, oracle, M, K, N):
    w, x = _rand_case(K, N, M, seed=K + N + M)
    x[:, ::7] *= -1  # signed activations
    y, q, s = _run_gemm(ops, oracle, w, x, path="mfma")
    ref = oracle.w8a16_gemm(x, q, s)
    ok = _tier_a(y, ref)
    assert ok.all(), "max err %g at %s" % (np.abs(y.astype(np.float32) - ref.astype(np.float32)).max(),
                                          np.argwhere(~ok)[:4])


def test_mfma_transpose_detecting(ops, oracle):
    """Asymmetric, structured inputs: catches swapped rows/cols or a wrong k mapping (guide rule 16)."""
    K, N, M = 128, 128, 128
    q = np.zeros((K, N), np.int8)
    for k in range(K):
        for n in range(N):
            q[k, n] = ((3 * k + 5 * n) % 255) - 127
    s = (np.arange(N) % 7 + 1).astype(np.float16) / 256
    x = ((np.arange(M)[:, None] * 2 + np.arange(K)[None, :]) % 13 - 6).astype(np.float16) / 8
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    for path in ("mfma",):
        y = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), processed, torch.from_numpy(s).to(DEV), path=path).cpu().numpy()
        ref = oracle.w8a16_gemm(x, q, s)
        assert _tier_a(y, ref).all()
    y = ops.w8_a16_gemm(torch.from_numpy(x[:3]).to(DEV), processed, torch.from_numpy(s).to(DEV), path="gemv").cpu().numpy()
    assert _tier_a(y, oracle.w8a16_gemm(x[:3], q, s)).all()


@pytest.mark.parametrize("K,N", [(128, 64), (4096, 4096)])
def test_identity_gemm_is_exact_dequant(ops, oracle, K, N):
    """Size-independent property at full size (python/eetq/modules/qlinear.py:83-86): eye(K) @ W == fp16(q*s),
    bit-exact, for every byte value that occurs -- exercises M = K (4096) through the MFMA path."""
    rng = np.random.default_rng(K)
    w = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    q, s = oracle.quantize(w)
    if K >= 256:
        q[:256, 0] = np.arange(-128, 128, dtype=np.int8)  # every byte value
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    eye = torch.eye(K, dtype=torch.float16, device=DEV)
    y = ops.w8_a16_gemm(eye, processed, torch.from_numpy(s).to(DEV)).cpu().numpy()
    assert np.array_equal(y, oracle.dequant(q, s))
    # and one row at a time through the GEMV path
    y1 = ops.w8_a16_gemm(eye[5:6].contiguous(), processed, torch.from_numpy(s).to(DEV)).cpu().numpy()
    assert np.array_equal(y1[0], oracle.dequant(q, s)[5])


@pytest.mark.parametrize("M", [1, 4, 8, 64, 1024])
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 11008), (11008, 4096)])
def test_llama7b_shapes_vs_torch_fp32(ops, oracle, M, K, N):
    """BASELINE.json configs[3]: Llama-2-7B shapes.  The oracle is too slow at M=1024, so the comparator is a
    torch fp32 matmul on the GPU over the oracle-dequantised weight (same contract, fp32 accumulate), plus an
    oracle check of 4 sampled rows."""
    torch.manual_seed(1)
    lin = torch.nn.Linear(K, N, bias=False, dtype=torch.float16)
    w = lin.weight.detach().t().contiguous()
    x = torch.rand(M, K, dtype=torch.float16)
    processed, scales = ops.quant_weights(w, torch.int8, False)
    y = ops.w8_a16_gemm(x.to(DEV), processed.to(DEV), scales.to(DEV))
    q, s = oracle.quantize(w.numpy())
    wdq = torch.from_numpy(oracle.dequant(q, s)).to(DEV)
    ref = (x.to(DEV).float() @ wdq.float()).cpu().numpy()
    assert _tier_a(y.cpu().numpy(), ref).all()
    rows = sorted(set([0, M // 3, M // 2, M - 1]))
    ref_rows = oracle.w8a16_gemm(x.numpy()[rows], q, s)
    assert _tier_a(y.cpu().numpy()[rows], ref_rows).all()
    # reference's own bar vs fp16 nn.Linear on the original weights (atol 1e-2 at K=1024, scaled by sqrt(K/1024))
    with torch.no_grad():
        y_lin = lin(x[rows]).numpy().astype(np.float32)
    assert np.abs(y.cpu().numpy()[rows].astype(np.float32) - y_lin).max() <= 1e-2 * np.sqrt(K / 1024.0)


def test_gemv_and_mfma_agree(ops, oracle):
    w, x = _rand_case(4096, 4096, 4, seed=99)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    a = ops.w8_a16_gemm(xd, processed, scales, path="gemv").float()
    b = ops.w8_a16_gemm(xd, processed, scales, path="mfma").float()
    assert (a - b).abs().max().item() <= 2e-3 * a.abs().max().item()


def test_reference_test_qlinear_recipe_on_gpu(ops, golden_dir):
    """examples/layers/test_qlinear.py:19-36 end to end: W8A16Linear.from_torch + forward vs fp16 nn.Linear, atol 1e-2."""
    from eetq_amd.modules.qlinear import W8A16Linear
    man = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["linear"][1]
    torch.manual_seed(1)
    lin = torch.nn.Linear(1024, 4096, bias=False, dtype=torch.float16)
    qlin = W8A16Linear.from_torch(lin.to(DEV))
    x = torch.rand(128, 1024, dtype=torch.float16)
    out = qlin(x.to(DEV)).cpu()
    gold = torch.from_numpy(np.load(os.path.join(golden_dir, man["file"]))["y"])
    assert torch.allclose(out[:: man["row_step"]], gold, atol=1e-2)


def test_3d_input_and_inplace_variant(ops, oracle):
    w, x = _rand_case(256, 128, 12, seed=5)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    x3 = torch.from_numpy(x).to(DEV).reshape(3, 4, 256)
    y3 = ops.w8_a16_gemm(x3, processed, scales)
    assert y3.shape == (3, 4, 128)
    out = torch.empty(12, 128, dtype=torch.float16, device=DEV)
    ret = ops.w8_a16_gemm_(x3, processed, scales, out, 12, 128, 256)
    assert ret.data_ptr() == out.data_ptr()
    assert torch.equal(out, y3.reshape(12, 128))
    assert _tier_a(out.cpu().numpy(), oracle.w8a16_gemm(x, q, s)).all()


def test_non_default_stream(ops, oracle):
    w, x = _rand_case(1024, 256, 2, seed=8)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        y = ops.w8_a16_gemm(xd, processed, scales)
    st.synchronize()
    assert _tier_a(y.cpu().numpy(), oracle.w8a16_gemm(x, q, s)).all()


# ---------------------------------------------------------------- side ops

@pytest.mark.parametrize("shape", [(2, 3, 4096), (1, 1, 5120), (4, 7, 520), (1, 5, 100)])
def test_layernorm_forward(ops, oracle, shape):
    torch.manual_seed(shape[-1])
    x = (torch.randn(*shape) * 3).half()
    g = (torch.rand(shape[-1]) + 0.5).half()
    out = torch.empty_like(x, device=DEV)
    assert ops.layernorm_forward(x.to(DEV), g.to(DEV), out, 1e-6) is None
    ref = oracle.rmsnorm_f16(x.numpy(), g.numpy(), 1e-6).astype(np.float32)
    got = out.cpu().numpy().astype(np.float32)
    assert np.allclose(got, ref, rtol=2e-3, atol=2e-3)     # fp32 reduction order differs; <= 1-2 fp16 ulp
    # ... in units in the last place: the fp32 sum of squares is the only order-dependent quantity, so no output may be more
    # than one fp16 step away from the oracle's, and almost all are identical
    def ordered(a):  # fp16 bit patterns (sign-magnitude) -> integers in value order
        i = np.ascontiguousarray(a).view(np.uint16).astype(np.int32)
        return np.where(i & 0x8000, -(i & 0x7FFF), i)
    gi = ordered(out.cpu().numpy())
    ri = ordered(oracle.rmsnorm_f16(x.numpy(), g.numpy(), 1e-6))
    assert np.abs(gi - ri).max() <= 1 and (gi != ri).mean() < 0.02
    ones = torch.full((1, 1, 64), 3.0, dtype=torch.float16, device=DEV)
    gam = torch.tensor([65000.0, -65000.0] * 32, dtype=torch.float16, device=DEV)
    out = torch.empty_like(ones)
    ops.layernorm_forward(ones, gam, out, 0.0)
    assert np.array_equal(out.cpu().numpy(), oracle.rmsnorm_f16(ones.cpu().numpy(), gam.cpu().numpy(), 0.0))


@pytest.mark.parametrize("heads,hs,rot", [(4, 64, 64), (40, 128, 128), (8, 64, 32)])
def test_rotary_embedding_neox(ops, oracle, heads, hs, rot):
    torch.manual_seed(heads)
    b, t = 2, 5
    q = torch.randn(b, t, 1, heads, hs).half()
    k = torch.randn(b, t, 1, heads, hs).half()
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    fr = torch.einsum("i,j->ij", torch.arange(64).float(), inv)
    cache = torch.cat([fr.cos(), fr.sin()], -1).half()
    pos = torch.randint(0, 64, (b, t))
    qd, kd = q.to(DEV), k.to(DEV)
    assert ops.rotary_embedding_neox(pos.to(DEV), qd, kd, hs, cache.to(DEV)) is None
    qo, ko = oracle.rotary_neox_f16(pos.numpy(), q.numpy().reshape(b * t, heads, hs), k.numpy().reshape(b * t, heads, hs),
                                    cache.numpy(), hs)
    assert np.array_equal(qd.cpu().numpy().reshape(b * t, heads, hs), qo)   # fp16 op-for-op: bit-exact
    assert np.array_equal(kd.cpu().numpy().reshape(b * t, heads, hs), ko)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("heads,hs,rot", [(8, 64, 64), (5, 128, 64)])
def test_rotary_embedding_neox_float_and_double(ops, oracle, dtype, heads, hs, rot):
    """The reference dispatches float / double / half / bf16 (pos_encoding_kernels.cu:73-86); float and double are plain
    IEEE operations of that type here and in the oracle (no contraction): bit-exact.  bf16 stays rejected (out of scope)."""
    torch.manual_seed(heads + rot)
    b, t = 2, 5
    q = torch.randn(b, t, 1, heads, hs, dtype=dtype)
    k = torch.randn(b, t, 1, heads, hs, dtype=dtype)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).double() / rot))
    fr = torch.einsum("i,j->ij", torch.arange(64).double(), inv)
    cache = torch.cat([fr.cos(), fr.sin()], -1).to(dtype)
    pos = torch.randint(0, 64, (b, t))
    qd, kd = q.to(DEV), k.to(DEV)
    assert ops.rotary_embedding_neox(pos.to(DEV), qd, kd, hs, cache.to(DEV)) is None
    qo, ko = oracle.rotary_neox(pos.numpy(), q.numpy().reshape(b * t, heads, hs), k.numpy().reshape(b * t, heads, hs),
                                cache.numpy(), hs)
    assert np.array_equal(qd.cpu().numpy().reshape(b * t, heads, hs), qo)
    assert np.array_equal(kd.cpu().numpy().reshape(b * t, heads, hs), ko)
    # against the textbook rotation in float64: only rounding differences
    x, y = q[..., : rot // 2].double(), q[..., rot // 2: rot].double()
    c = cache[pos][..., : rot // 2].double()[:, :, None, None, :]
    sn = cache[pos][..., rot // 2:].double()[:, :, None, None, :]
    assert torch.allclose(qd.cpu()[..., : rot // 2].double(), x * c - y * sn, atol=1e-5 if dtype == torch.float32 else 1e-12)
    with pytest.raises(RuntimeError):
        ops.rotary_embedding_neox(pos.to(DEV), qd.bfloat16(), kd.bfloat16(), hs, cache.to(DEV).bfloat16())
    with pytest.raises(RuntimeError):   # mixed dtypes
        ops.rotary_embedding_neox(pos.to(DEV), qd, kd.half(), hs, cache.to(DEV))


@pytest.mark.parametrize("hq,hk,hs", [(8, 8, 64), (8, 2, 128), (5, 1, 64)])
def test_rotary_strided_in_fused_qkv(ops, oracle, hq, hk, hs):
    """q and k slices of a fused QKV projection output are rotated in place (token stride = the fused row length,
    k_heads <= q_heads); bit-exact against the oracle run on dense copies, v and nothing else touched."""
    torch.manual_seed(hq * 10 + hk)
    b, t = 2, 7
    row = (hq + 2 * hk) * hs
    qkv = torch.randn(b, t, row).half()
    inv = 1.0 / (10000 ** (torch.arange(0, hs, 2).float() / hs))
    fr = torch.einsum("i,j->ij", torch.arange(96).float(), inv)
    cache = torch.cat([fr.cos(), fr.sin()], -1).half()
    pos = torch.randint(0, 96, (b, t))
    d = qkv.to(DEV)
    q = d[..., : hq * hs].unflatten(-1, (hq, hs))
    k = d[..., hq * hs: (hq + hk) * hs].unflatten(-1, (hk, hs))
    assert ops.rotary_embedding_neox_strided(pos.to(DEV), q, k, hs, cache.to(DEV)) is None
    q0 = qkv[..., : hq * hs].reshape(b * t, hq, hs).numpy()
    k0 = qkv[..., hq * hs: (hq + hk) * hs].reshape(b * t, hk, hs).numpy()
    qo, _ = oracle.rotary_neox_f16(pos.numpy(), q0, q0.copy(), cache.numpy(), hs)
    ko, _ = oracle.rotary_neox_f16(pos.numpy(), k0, k0.copy(), cache.numpy(), hs)
    got = d.cpu()
    assert np.array_equal(got[..., : hq * hs].reshape(b * t, hq, hs).numpy(), qo)
    assert np.array_equal(got[..., hq * hs: (hq + hk) * hs].reshape(b * t, hk, hs).numpy(), ko)
    assert torch.equal(got[..., (hq + hk) * hs:], qkv[..., (hq + hk) * hs:])
    with pytest.raises(RuntimeError):   # transposed views do not collapse to one token stride
        ops.rotary_embedding_neox_strided(pos.to(DEV), q.transpose(1, 2), k.transpose(1, 2), hs, cache.to(DEV))


# ---------------------------------------------------------------- modules / eet_quantize

def test_eetq_linear_forward_backward(ops, oracle):
    from eetq_amd.modules.qlinear import EetqLinear
    K, N = 256, 128
    w, x = _rand_case(K, N, 6, seed=21)
    q, s = oracle.quantize(w)
    mod = EetqLinear(K, N, bias=True, device=DEV)
    mod.register_scale(DEV)
    mod.weight.copy_(torch.from_numpy(oracle.gfx950_pack(q)))
    mod.weight_scales.copy_(torch.from_numpy(s))
    mod.bias.copy_(torch.linspace(-1, 1, N).half())
    xin = torch.from_numpy(x).to(DEV).reshape(1, 6, K).requires_grad_(True)
    mod.train()
    y = mod(xin)
    ref = oracle.w8a16_gemm(x, q, s).astype(np.float32) + mod.bias.float().cpu().numpy()
    assert np.allclose(y.detach().cpu().numpy().reshape(6, N).astype(np.float32), ref, atol=4e-3, rtol=4e-3)
    gy = torch.ones_like(y)
    y.backward(gy)
    wdq = torch.from_numpy(oracle.dequant(q, s)).float()
    gref = (torch.ones(6, N) @ wdq.t()).numpy()
    assert np.allclose(xin.grad.cpu().numpy().reshape(6, K).astype(np.float32), gref, atol=2e-2, rtol=1e-2)
    mod.eval()
    assert torch.equal(mod(xin.detach()), y.detach())


@pytest.mark.parametrize("M", [1, 3, 8, 40, 200])
def test_fused_bias_is_bit_identical_to_separate_add(ops, oracle, M):
    """the kernel-epilogue bias must give the same bits as the reference's `output + bias`."""
    K, N = 512, 272
    w, x = _rand_case(K, N, M, seed=77 + M)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    bias = (torch.randn(N) * 0.5).half().to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    fused = ops.w8_a16_gemm(xd, processed, scales, bias=bias)
    separate = ops.w8_a16_gemm(xd, processed, scales) + bias
    assert torch.equal(fused, separate)
    ref = oracle.w8a16_gemm(x, q, s).astype(np.float32) + bias.float().cpu().numpy()
    assert np.allclose(fused.float().cpu().numpy(), ref, atol=3e-3, rtol=3e-3)


@pytest.mark.parametrize("M,path", [(1, "auto"), (3, "gemv"), (8, "auto"), (8, "stream"), (48, "auto"), (64, "stream"),
                                    (100, "mid"), (200, "auto"), (130, "mfma")])
def test_fused_residual_is_bit_identical_to_separate_adds(ops, oracle, M, path):
    """y = fp16(acc) + bias + residual inside the epilogue of every kernel == the three separate fp16 ops of a decoder
    block (`residual + (proj(x) + bias)`), bit for bit; also with residual aliasing the output (in-place accumulate)."""
    K, N = 512, 272
    w, x = _rand_case(K, N, M, seed=900 + M)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    torch.manual_seed(M)
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(M, N, dtype=torch.float16, device=DEV)
    plain = ops.w8_a16_gemm(xd, processed, scales, path=path)
    for b in (None, bias):
        fused = ops.w8_a16_gemm(xd, processed, scales, path=path, bias=b, residual=res)
        separate = res + (plain + b if b is not None else plain)
        assert torch.equal(fused, separate)
    out = res.clone()
    from eetq_amd import _lib
    import ctypes
    _lib.check(_lib.lib().eetq_w8a16_gemm_fused(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(processed.data_ptr()),
                                                ctypes.c_void_p(scales.data_ptr()), None, ctypes.c_void_p(out.data_ptr()),
                                                ctypes.c_void_p(out.data_ptr()), M, N, K, 0,
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    # (the C entry runs EETQ_PATH_AUTO, which need not be the kernel `path` names: M = 130 takes the row-group plan of the split-K tile)
    assert torch.equal(out, res + ops.w8_a16_gemm(xd, processed, scales))
    with pytest.raises(RuntimeError):
        ops.w8_a16_gemm(xd, processed, scales, residual=res[:, :-16].contiguous())


@pytest.mark.parametrize("M,path", [(1, "auto"), (3, "gemv"), (8, "auto"), (33, "stream"), (64, "mid"), (100, "mid"),
                                    (200, "mfma"), (1024, "auto")])
@pytest.mark.parametrize("act", ["relu", "gelu", "silu"])
def test_activation_epilogues_vs_oracle(ops, oracle, M, path, act):
    """Bias + activation epilogue (FT family: fpA_intB_gemm.cu:35-62, epilogue_helpers.h:20-71) in every kernel:
    fp16(act(acc + bias)), sum and activation in fp32, against the oracle's restatement; plus no-bias and + residual."""
    K, N = 1024, 320
    w, x = _rand_case(K, N, M, seed=M + len(act), wscale=0.05)
    x = (x - 0.5).astype(np.float16)                 # both signs, so every activation sees its whole domain
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    bias = (np.random.default_rng(5).standard_normal(N) * 0.5).astype(np.float16)
    xd, bd = torch.from_numpy(x).to(DEV), torch.from_numpy(bias).to(DEV)
    y = ops.w8_a16_gemm(xd, processed, scales, path=path, bias=bd, activation=act).cpu().numpy()
    ref = oracle.w8a16_gemm_bias_act(x, q, s, bias, act)
    assert _tier_a(y, ref).all(), np.abs(y.astype(np.float32) - ref.astype(np.float32)).max()
    y0 = ops.w8_a16_gemm(xd, processed, scales, path=path, activation=act).cpu().numpy()
    assert _tier_a(y0, oracle.w8a16_gemm_bias_act(x, q, s, None, act)).all()
    if act == "relu":
        assert (y >= 0).all() and (y == 0).any()
    res = torch.from_numpy((np.random.default_rng(6).standard_normal((M, N))).astype(np.float16)).to(DEV)
    yr = ops.w8_a16_gemm(xd, processed, scales, path=path, bias=bd, residual=res, activation=act)
    assert torch.equal(yr, torch.from_numpy(y).to(DEV) + res)     # residual: an fp16 add after the activation


def test_fuse_w8a16_linears_qkv(ops):
    """one launch over concatenated output channels == the three separate launches, bit for bit."""
    from eetq_amd.modules.qlinear import W8A16Linear
    from eetq_amd.utils.fuse import fuse_w8a16_linears
    torch.manual_seed(3)
    K = 256
    lins = [torch.nn.Linear(K, n, bias=True).half().to(DEV) for n in (128, 64, 48)]
    parts = [W8A16Linear.from_torch(l) for l in lins]
    fused = fuse_w8a16_linears(parts)
    for M in (1, 5, 70):
        x = torch.rand(M, K, dtype=torch.float16, device=DEV) - 0.5
        outs = fused(x)
        for o, p in zip(outs, parts):
            assert torch.equal(o, p(x))


def test_eet_quantize_hf_llama_tiny(ops):
    """Config 5 plumbing at toy scale: a transformers LlamaForCausalLM goes through eet_quantize and still generates
    the same greedy tokens as its fp16 self (lm_head stays fp16, every other nn.Linear becomes W8A16Linear)."""
    transformers = pytest.importorskip("transformers")
    from eetq_amd.modules.qlinear import W8A16Linear
    from eetq_amd.utils.quantizer import eet_quantize
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=4, vocab_size=512, max_position_embeddings=128)
    torch.manual_seed(0)
    model = transformers.LlamaForCausalLM(cfg).half().to(DEV).eval()
    prompt = torch.randint(0, 512, (2, 16), device=DEV)
    with torch.no_grad():
        ref_logits = model(prompt).logits.float()
    eet_quantize(model)
    n_q = sum(isinstance(m, W8A16Linear) for m in model.modules())
    assert n_q == 2 * 7 and isinstance(model.lm_head, torch.nn.Linear)
    with torch.no_grad():
        logits = model(prompt).logits.float()
    # int8 per-channel quantisation error on a random-init toy model: logits stay close, relative to their spread
    assert (logits - ref_logits).abs().max().item() < 0.05 * ref_logits.abs().max().item() + 0.05
    out = model.generate(prompt, max_new_tokens=4, do_sample=False, pad_token_id=0)
    assert out.shape == (2, 20)


def test_eet_quantize_tiny_model(ops):
    from eetq_amd.modules.qlinear import W8A16Linear
    from eetq_amd.utils.quantizer import eet_quantize
    torch.manual_seed(0)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([torch.nn.Linear(128, 256, bias=True), torch.nn.Linear(256, 128, bias=False)])
            self.lm_head = torch.nn.Linear(128, 64, bias=False)

        def forward(self, x):
            return self.lm_head(self.layers[1](torch.relu(self.layers[0](x))))

    model = Tiny().half().to(DEV)
    x = torch.rand(3, 128, dtype=torch.float16, device=DEV)
    with torch.no_grad():
        ref = model(x)
    eet_quantize(model)
    assert isinstance(model.layers[0], W8A16Linear) and isinstance(model.layers[1], W8A16Linear)
    assert isinstance(model.lm_head, torch.nn.Linear)
    assert set(model.layers[0].state_dict().keys()) == {"qweight", "weight_scales", "bias"}
    out = model(x)
    assert torch.allclose(out, ref, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("kv_heads", [4, 2])
def test_eet_accelerator_hf_llama_tiny(ops, kv_heads):
    """eet_accelerator (reference python/eetq/utils/accelerator.py:15-19) on a toy transformers Llama: the fused-QKV /
    in-place-rotary attention must reproduce the stock fp16 block, and the quantised + fused model must agree with
    plain eet_quantize of the same weights (fusion of per-channel W8A16 projections is exact)."""
    transformers = pytest.importorskip("transformers")
    import copy
    from eetq_amd.modules.llama_modules import EETLlamaAttention, EETLlamaMLP
    from eetq_amd.modules.qlinear import W8A16Linear
    from eetq_amd.utils import eet_accelerator, eet_quantize
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=kv_heads, vocab_size=512, max_position_embeddings=128)
    torch.manual_seed(0)
    stock = transformers.LlamaForCausalLM(cfg).half().to(DEV).eval()
    prompt = torch.randint(0, 512, (2, 16), device=DEV)
    with torch.no_grad():
        ref = stock(prompt).logits.float()
        ref_tokens = stock.generate(prompt, max_new_tokens=6, do_sample=False, pad_token_id=0)
    spread = ref.abs().max().item()

    fp16 = eet_accelerator(copy.deepcopy(stock), quantize=False, fused_attn=True)
    assert sum(isinstance(m, EETLlamaAttention) for m in fp16.modules()) == 2
    with torch.no_grad():
        got = fp16(prompt).logits.float()
        tokens = fp16.generate(prompt, max_new_tokens=6, do_sample=False, pad_token_id=0)
    assert (got - ref).abs().max().item() < 4e-3 * spread + 4e-3   # same math, fp16 GEMM rounding differs with N
    assert tokens.shape == ref_tokens.shape

    plain = eet_quantize(copy.deepcopy(stock))
    fused = eet_accelerator(copy.deepcopy(stock), quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True,
                            fused_residual=True)
    assert sum(isinstance(m, EETLlamaMLP) for m in fused.modules()) == 2
    assert isinstance(fused.model.layers[0].self_attn.qkv_proj, W8A16Linear) and isinstance(fused.lm_head, torch.nn.Linear)
    with torch.no_grad():
        a = plain(prompt).logits.float()
        b = fused(prompt).logits.float()
        # decode through the KV cache, token by token, must follow the prefill path
        out = fused.generate(prompt, max_new_tokens=6, do_sample=False, pad_token_id=0)
    assert (a - b).abs().max().item() < 4e-3 * spread + 4e-3
    assert out.shape == (2, 22)
    with torch.no_grad():
        full = fused(out[:, :-1]).logits[:, -1].float()
        assert torch.equal(full.argmax(-1), out[:, -1]) or (full.topk(2).values[:, 0] - full.topk(2).values[:, 1]).min() < 1e-2
        # the single-token attention used under HIP-graph capture (two batched matrix-vector products + softmax) must
        # agree with the library attention of the eager path
        for layer in fused.model.layers:
            layer.self_attn.decode_math_attention = "always"
        out_math = fused.generate(prompt, max_new_tokens=6, do_sample=False, pad_token_id=0)
        lg_a = fused(out[:, :-1]).logits[:, -1].float()
        assert out_math.shape == out.shape
        agree = (out_math == out).float().mean().item()
        assert agree > 0.9 or (lg_a.topk(2).values[:, 0] - lg_a.topk(2).values[:, 1]).min() < 1e-2


def test_auto_dispatch_fuzz_vs_oracle(ops, oracle):
    """Seeded random shapes through the AUTO dispatch (every kernel and every ragged edge the launchers can pick), with
    bias and residual on a third of them; tier-A against the oracle."""
    rng = np.random.default_rng(20260926)
    cases = []
    for _ in range(70):
        M = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 31, 32, 33, 47, 64, 65, 96, 127, 128, 129, 160, 255, 256, 300]))
        K = 64 * int(rng.integers(1, 18))
        N = 16 * int(rng.integers(1, 40))
        cases.append((M, K, N))
    cases += [(40, 512, 10240), (100, 384, 10304), (129, 320, 8208), (8, 2048, 8192), (16, 2112, 8256)]  # wide-N rules
    for idx, (M, K, N) in enumerate(cases):
        w, x = _rand_case(K, N, M, seed=1000 + idx)
        x[:, ::5] *= -1
        q, s = oracle.quantize(w)
        processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
        scales = torch.from_numpy(s).to(DEV)
        xd = torch.from_numpy(x).to(DEV)
        ref = oracle.w8a16_gemm(x, q, s).astype(np.float32)
        if idx % 3 == 0:
            torch.manual_seed(idx)
            bias = torch.randn(N, dtype=torch.float16, device=DEV)
            res = torch.randn(M, N, dtype=torch.float16, device=DEV)
            y = ops.w8_a16_gemm(xd, processed, scales, bias=bias, residual=res)
            plain = ops.w8_a16_gemm(xd, processed, scales)
            assert torch.equal(y, res + (plain + bias)), (M, K, N)
            y = plain
        else:
            y = ops.w8_a16_gemm(xd, processed, scales)
        got = y.cpu().numpy().astype(np.float32)
        ok = _tier_a(got, ref)
        assert ok.all(), "M=%d K=%d N=%d: max err %g at %s" % (M, K, N, np.abs(got - ref).max(), np.argwhere(~ok)[:4])


@pytest.mark.parametrize("B,H,Hkv,S,D", [(1, 40, 40, 1232, 128), (2, 8, 2, 77, 64), (3, 4, 4, 1, 128), (1, 5, 1, 300, 64)])
def test_decode_attention_vs_torch(ops, B, H, Hkv, S, D):
    """Split-KV single-query attention kernel against a plain PyTorch fp32 reference of the same op (softmax(scale q k^T
    + mask) v), strided query (a slice of a fused QKV row), grouped-query heads, additive -inf mask incl. a fully masked
    row; tolerance 2e-3 absolute on O(1) outputs (fp16 output rounding 5e-4, fp32 accumulation)."""
    torch.manual_seed(B * 1000 + S)
    row = (H + 2 * Hkv) * D
    qkv = torch.randn(B, 1, row, dtype=torch.float16, device=DEV)
    q = qkv[..., : H * D].unflatten(-1, (H, D))[:, 0]                     # [B, H, D], batch stride = fused row
    k = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    v = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    mask = torch.zeros(B, 1, 1, S, dtype=torch.float16, device=DEV)
    valid = torch.randint(1, S + 1, (B,))
    for b in range(B):
        mask[b, ..., int(valid[b]):] = float("-inf")
    scale = D ** -0.5
    for m in (None, mask):
        for splits in (None, 1, min(S, 7)):
            out = ops.decode_attention(q, k, v, mask=m, scaling=scale, splits=splits)
            kk = k.float().repeat_interleave(H // Hkv, dim=1)
            vv = v.float().repeat_interleave(H // Hkv, dim=1)
            s = torch.einsum("bhd,bhsd->bhs", q.float(), kk) * scale
            if m is not None:
                s = s + m[:, 0].float()
            ref = torch.einsum("bhs,bhsd->bhd", torch.softmax(s, dim=-1), vv)
            assert out.shape == (B, H, D) and out.dtype == torch.float16
            assert (out.float() - ref).abs().max().item() < 2e-3, (m is not None, splits)
    dead = torch.full((B, 1, 1, S), float("-inf"), dtype=torch.float16, device=DEV)
    assert torch.count_nonzero(ops.decode_attention(q, k, v, mask=dead, scaling=scale)) == 0
    with pytest.raises(RuntimeError):
        ops.decode_attention(q.float(), k, v)


def test_rotary_kvcache_write(ops, oracle):
    """Decode-step rotary + cache write: q rotated in place (bit-exact vs the oracle), rotated k and v land in the cache
    rows at positions[b], every other cache element untouched, an out-of-range position writes nothing."""
    torch.manual_seed(3)
    B, H, Hkv, D, S = 3, 8, 2, 64, 40
    row = (H + 2 * Hkv) * D
    qkv = torch.randn(B, 1, row).half()
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.einsum("i,j->ij", torch.arange(64).float(), inv)
    cache = torch.cat([fr.cos(), fr.sin()], -1).half()
    pos = torch.tensor([5, 39, 17])
    d = qkv.to(DEV)
    q = d[..., : H * D].unflatten(-1, (H, D))[:, 0]
    k = d[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))[:, 0]
    v = d[..., (H + Hkv) * D:].unflatten(-1, (Hkv, D))[:, 0]
    kc = torch.full((B, Hkv, S, D), 7.0, dtype=torch.float16, device=DEV)
    vc = torch.full((B, Hkv, S, D), -3.0, dtype=torch.float16, device=DEV)
    ops.rotary_embedding_neox_kvcache(pos.to(DEV), q, k, v, D, cache.to(DEV), kc, vc)
    q0 = qkv[:, 0, : H * D].reshape(B, H, D).numpy()
    k0 = qkv[:, 0, H * D: (H + Hkv) * D].reshape(B, Hkv, D).numpy()
    qo, _ = oracle.rotary_neox_f16(pos.numpy(), q0, q0.copy(), cache.numpy(), D)
    ko, _ = oracle.rotary_neox_f16(pos.numpy(), k0, k0.copy(), cache.numpy(), D)
    got = d.cpu()
    assert np.array_equal(got[:, 0, : H * D].reshape(B, H, D).numpy(), qo)
    assert torch.equal(got[:, 0, H * D:], qkv[:, 0, H * D:])                       # k and v rows of the projection: unchanged
    kc, vc = kc.cpu(), vc.cpu()
    for b in range(B):
        assert np.array_equal(kc[b, :, int(pos[b])].numpy(), ko[b])
        assert torch.equal(vc[b, :, int(pos[b])], qkv[b, 0, (H + Hkv) * D:].reshape(Hkv, D))
        others = [j for j in range(S) if j != int(pos[b])]
        assert (kc[b][:, others] == 7.0).all() and (vc[b][:, others] == -3.0).all()
    kc2 = torch.zeros(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    assert ops.decode_dropped_steps(reset=True) == 0          # nothing dropped so far (and reset)
    ops.rotary_embedding_neox_kvcache(torch.tensor([S, S + 3, -1]).to(DEV), q, k, v, D, cache.to(DEV), kc2, kc2.clone())
    assert torch.count_nonzero(kc2) == 0
    # ... and the skipped writes are counted: a full cache is not silent (stock StaticLayer.update raises there)
    assert ops.decode_dropped_steps(reset=True) == 3
    assert ops.decode_dropped_steps() == 0


@pytest.mark.parametrize("B,T,H,Hkv,D", [(1, 33, 8, 8, 128), (2, 17, 8, 2, 64), (3, 1, 4, 4, 64), (1, 200, 4, 1, 128)])
def test_rotary_kvcache_prefill(ops, oracle, B, T, H, Hkv, D):
    """Prompt form of the rotary + cache write: q [B, T, H, D] rotated in place (bit-exact vs the oracle), rotated k and v in cache
    rows base .. base + T - 1 of their batch row -- base a host integer or a device counter (read, not advanced) -- every other
    cache element and the k / v parts of the projection untouched; rows beyond the cache are refused (host base) or skipped
    and counted (device base).  The cache must hold what rotary_embedding_neox_strided + index_copy_ put there."""
    torch.manual_seed(B * 100 + T)
    S = T + 11
    row = (H + 2 * Hkv) * D
    qkv = torch.randn(B, T, row).half()
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.einsum("i,j->ij", torch.arange(S + 40).float(), inv)
    table = torch.cat([fr.cos(), fr.sin()], -1).half()
    pos = (torch.arange(T)[None, :] + torch.arange(B)[:, None] * 3 + 2).contiguous()     # per-row offsets: positions != cache rows
    q0 = qkv[..., : H * D].reshape(B * T, H, D).numpy()
    k0 = qkv[..., H * D: (H + Hkv) * D].reshape(B * T, Hkv, D).numpy()
    qo, _ = oracle.rotary_neox_f16(pos.reshape(-1).numpy(), q0, q0.copy(), table.numpy(), D)
    ko, _ = oracle.rotary_neox_f16(pos.reshape(-1).numpy(), k0, k0.copy(), table.numpy(), D)
    for base, dev_base in ((0, False), (7, False), (5, True)):
        d = qkv.to(DEV)
        q = d[..., : H * D].unflatten(-1, (H, D))
        k = d[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))
        v = d[..., (H + Hkv) * D:].unflatten(-1, (Hkv, D))
        kc = torch.full((B, Hkv, S, D), 7.0, dtype=torch.float16, device=DEV)
        vc = torch.full((B, Hkv, S, D), -3.0, dtype=torch.float16, device=DEV)
        counter = torch.tensor(base, dtype=torch.int64, device=DEV)
        if dev_base:
            ops.rotary_embedding_neox_kvcache_prefill(pos.to(DEV), q, k, v, D, table.to(DEV), kc, vc, first_row_dev=counter)
            assert int(counter) == base
        else:
            ops.rotary_embedding_neox_kvcache_prefill(pos.to(DEV), q, k, v, D, table.to(DEV), kc, vc, first_row=base)
        got = d.cpu()
        assert np.array_equal(got[..., : H * D].reshape(B * T, H, D).numpy(), qo)
        assert torch.equal(got[..., H * D:], qkv[..., H * D:])
        kc, vc = kc.cpu(), vc.cpu()
        assert np.array_equal(kc[:, :, base: base + T].transpose(1, 2).reshape(B * T, Hkv, D).numpy(), ko)
        assert torch.equal(vc[:, :, base: base + T].transpose(1, 2), qkv[..., (H + Hkv) * D:].reshape(B, T, Hkv, D))
        assert (kc[:, :, :base] == 7.0).all() and (kc[:, :, base + T:] == 7.0).all()
        assert (vc[:, :, :base] == -3.0).all() and (vc[:, :, base + T:] == -3.0).all()
    # ... the same cache contents as the two-step stock sequence
    d = qkv.to(DEV)
    q = d[..., : H * D].unflatten(-1, (H, D))
    k = d[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))
    ops.rotary_embedding_neox_strided(pos.to(DEV), q, k, D, table.to(DEV))
    kc2 = torch.full((B, Hkv, S, D), 7.0, dtype=torch.float16, device=DEV)
    kc2.index_copy_(2, torch.arange(T, device=DEV) + 5, k.transpose(1, 2))
    assert torch.equal(kc2.cpu(), kc)
    # rows beyond the cache: a host base is refused, a device base skips the tokens that do not fit and counts them
    d = qkv.to(DEV)
    q = d[..., : H * D].unflatten(-1, (H, D))
    k = d[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))
    v = d[..., (H + Hkv) * D:].unflatten(-1, (Hkv, D))
    kz = torch.zeros(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.rotary_embedding_neox_kvcache_prefill(pos.to(DEV), q, k, v, D, table.to(DEV), kz, kz.clone(), first_row=12)
    assert ops.decode_dropped_steps(reset=True) == 0
    ops.rotary_embedding_neox_kvcache_prefill(pos.to(DEV), q, k, v, D, table.to(DEV), kz, kz.clone(),
                                              first_row_dev=torch.tensor(S - 1, dtype=torch.int64, device=DEV))
    assert torch.count_nonzero(kz[:, :, : S - 1]) == 0 and torch.count_nonzero(kz[:, :, S - 1]) > 0
    assert ops.decode_dropped_steps(reset=True) == B * (T - 1)


def test_eet_attention_static_cache_prompt_matches_stock_path(ops):
    """A prompt on an INITIALISED transformers StaticCache: the one-launch rotary + cache write (+ counter add) must leave the
    cache and the counter exactly as the stock rotary + StaticLayer.update leave them, and give the same logits (same attention
    call on the same cache and mask); under fresh_static_prefill (cache empty, prompt unpadded) the causal attention over the
    prompt's own rows must agree within fp16 attention tolerance; a second chunk of prompt appends behind the first."""
    transformers = pytest.importorskip("transformers")
    import copy
    from eetq_amd.utils import eet_accelerator
    from eetq_amd.modules.llama_modules import fresh_static_prefill
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=512, max_position_embeddings=128)
    torch.manual_seed(0)
    stock = transformers.LlamaForCausalLM(cfg).half().to(DEV).eval()
    model = eet_accelerator(copy.deepcopy(stock), quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True,
                            fused_residual=True)
    prompt = torch.randint(0, 512, (2, 19), device=DEV)
    more = torch.randint(0, 512, (2, 5), device=DEV)

    def run(fast, fresh=False):
        for layer in model.model.layers:
            layer.self_attn.static_prefill = fast
        cache = transformers.StaticCache(config=cfg, max_cache_len=40)
        with torch.no_grad():
            model(prompt[:, :1], past_key_values=cache, use_cache=True)     # allocates the cache tensors
            if fresh:      # counters only: the model derives positions from them; the rows keep the stale token
                for l in cache.layers:
                    l.cumulative_length.zero_()
                with fresh_static_prefill():
                    a = model(prompt, past_key_values=cache, use_cache=True).logits.float()
            else:
                cache.reset()
                a = model(prompt, past_key_values=cache, use_cache=True).logits.float()
            b = model(more, past_key_values=cache, use_cache=True).logits.float()    # appended behind the prompt, never "fresh"
        return a, b, [(l.keys.clone(), l.values.clone(), int(l.cumulative_length)) for l in cache.layers]

    ra, rb, rc = run(False)
    ga, gb, gc = run(True)
    assert torch.equal(ga, ra) and torch.equal(gb, rb)
    for (k0, v0, n0), (k1, v1, n1) in zip(rc, gc):
        assert n0 == n1 == 24 and torch.equal(k0, k1) and torch.equal(v0, v1)
    fa, fb, fc = run(True, fresh=True)
    spread = ra.abs().max().item()
    assert (fa - ra).abs().max().item() < 4e-3 * spread + 4e-3 and (fb - rb).abs().max().item() < 4e-3 * spread + 4e-3
    assert fc[0][2] == 24 and torch.equal(fc[0][0], rc[0][0])             # layer 0's cache does not depend on any attention


def test_eet_attention_static_cache_decode_matches_stock_path(ops):
    """Token-by-token decode on a transformers StaticCache: the fused path (rotary + cache write in one launch, split-KV
    attention) must track the same model running its stock cache update and attention call."""
    transformers = pytest.importorskip("transformers")
    import copy
    from eetq_amd.utils import eet_accelerator
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=512, max_position_embeddings=128)
    torch.manual_seed(0)
    stock = transformers.LlamaForCausalLM(cfg).half().to(DEV).eval()
    model = eet_accelerator(copy.deepcopy(stock), quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True,
                            fused_residual=True)
    prompt = torch.randint(0, 512, (2, 9), device=DEV)

    def run(mode):
        for layer in model.model.layers:
            layer.self_attn.decode_math_attention = mode
        cache = transformers.StaticCache(config=cfg, max_cache_len=32)
        logits = []
        with torch.no_grad():
            out = model(prompt, past_key_values=cache, use_cache=True)
            tok = out.logits[:, -1].argmax(-1, keepdim=True)
            for _ in range(6):
                out = model(tok, past_key_values=cache, use_cache=True)
                logits.append(out.logits[:, -1].float())
                tok = out.logits[:, -1].argmax(-1, keepdim=True)
        return torch.stack(logits)

    ref = run(False)      # stock cache update + stock attention call
    got = run(True)       # fused decode path
    spread = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 4e-3 * spread + 4e-3


def test_rotary_kvcache_slot_differs_from_position(ops, oracle):
    """Left-padded batches: the rotary index (real tokens so far) and the cache row (the cache's token counter) differ.
    The rotation must use positions[b], the write must go to the slot -- one shared device scalar or one per row."""
    torch.manual_seed(4)
    B, H, Hkv, D, S = 3, 4, 2, 64, 24
    row = (H + 2 * Hkv) * D
    qkv = torch.randn(B, 1, row).half()
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.einsum("i,j->ij", torch.arange(64).float(), inv)
    cache = torch.cat([fr.cos(), fr.sin()], -1).half()
    pos = torch.tensor([3, 9, 6])                      # rotary positions (padding not counted)
    k0 = qkv[:, 0, H * D: (H + Hkv) * D].reshape(B, Hkv, D).numpy()
    ko, _ = oracle.rotary_neox_f16(pos.numpy(), k0, k0.copy(), cache.numpy(), D)
    for slots in (torch.tensor(9), torch.tensor([9]), torch.tensor([11, 2, 20])):
        d = qkv.to(DEV)
        q = d[..., : H * D].unflatten(-1, (H, D))[:, 0]
        k = d[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))[:, 0]
        v = d[..., (H + Hkv) * D:].unflatten(-1, (Hkv, D))[:, 0]
        kc = torch.zeros(B, Hkv, S, D, dtype=torch.float16, device=DEV)
        vc = torch.zeros(B, Hkv, S, D, dtype=torch.float16, device=DEV)
        ops.rotary_embedding_neox_kvcache(pos.to(DEV), q, k, v, D, cache.to(DEV), kc, vc, slots=slots.to(DEV))
        kc, vc = kc.cpu(), vc.cpu()
        for b in range(B):
            slot = int(slots.reshape(-1)[b if slots.numel() == B else 0])
            assert np.array_equal(kc[b, :, slot].numpy(), ko[b])
            assert torch.equal(vc[b, :, slot], qkv[b, 0, (H + Hkv) * D:].reshape(Hkv, D))
            others = [j for j in range(S) if j != slot]
            assert torch.count_nonzero(kc[b][:, others]) == 0 and torch.count_nonzero(vc[b][:, others]) == 0
    with pytest.raises(RuntimeError):
        ops.rotary_embedding_neox_kvcache(pos.to(DEV), q, k, v, D, cache.to(DEV), kc.to(DEV), vc.to(DEV),
                                          slots=torch.tensor([1, 2]).to(DEV))


def test_decode_attention_valid_length_and_counter(ops):
    """kv_len bounds the attended rows of a pre-allocated cache without any mask (rows beyond it hold garbage here), a
    shared [1, S] mask row serves the whole batch, a mask with the wrong batch is refused, `advance` bumps the counter once."""
    torch.manual_seed(11)
    B, H, Hkv, S, D = 2, 8, 4, 96, 64
    q = torch.randn(B, H, D, dtype=torch.float16, device=DEV)
    k = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    v = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    k[:, :, 41:] = 300.0      # "stale" rows: attending any of them would dominate the softmax
    v[:, :, 41:] = -500.0
    counter = torch.tensor(40, dtype=torch.int64, device=DEV)

    def ref(n, mask=None):
        kk = k[:, :, :n].float().repeat_interleave(H // Hkv, dim=1)
        vv = v[:, :, :n].float().repeat_interleave(H // Hkv, dim=1)
        s = torch.einsum("bhd,bhsd->bhs", q.float(), kk) * D ** -0.5
        if mask is not None:
            s = s + mask[..., :n].float()[:, None, :]
        return torch.einsum("bhs,bhsd->bhd", torch.softmax(s, -1), vv)

    out = ops.decode_attention(q, k, v, kv_len=counter, kv_len_bias=1, advance=counter)
    assert (out.float() - ref(41)).abs().max().item() < 2e-3
    assert int(counter.item()) == 41
    out = ops.decode_attention(q, k, v, kv_len=counter, splits=5)          # bias 0: rows < 41
    assert (out.float() - ref(41)).abs().max().item() < 2e-3 and int(counter.item()) == 41
    shared = torch.zeros(1, S, dtype=torch.float16, device=DEV)
    shared[0, 7:19] = float("-inf")
    out = ops.decode_attention(q, k, v, mask=shared, kv_len=counter)
    assert (out.float() - ref(41, shared.expand(B, S))).abs().max().item() < 2e-3
    with pytest.raises(RuntimeError):
        ops.decode_attention(q, k, v, mask=torch.zeros(3, S, dtype=torch.float16, device=DEV))
    with pytest.raises(RuntimeError):
        ops.decode_attention(q, k, v, kv_len=torch.tensor([1, 2], device=DEV))
    zero = torch.tensor(0, dtype=torch.int64, device=DEV)
    assert torch.count_nonzero(ops.decode_attention(q, k, v, kv_len=zero)) == 0   # nothing valid -> zeros, not NaN


@pytest.mark.parametrize("with_mask", [True, False])
def test_eet_attention_static_cache_left_padded_batch(ops, with_mask):
    """A left-padded batch on a transformers StaticCache: position_ids (= cumsum(mask) - 1) are smaller than the cache's
    token counter for the padded row.  The fused decode path must write the new token at the counter's row (not at the
    position) and give the same logits as the stock cache update + stock attention.  Without a mask on the decode steps
    (stand-alone use) rows beyond the counter must still not be attended."""
    transformers = pytest.importorskip("transformers")
    import copy
    from eetq_amd.utils import eet_accelerator
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=512, max_position_embeddings=128)
    torch.manual_seed(0)
    stock = transformers.LlamaForCausalLM(cfg).half().to(DEV).eval()
    model = eet_accelerator(copy.deepcopy(stock), quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True,
                            fused_residual=True)
    T, NEW, L = 9, 5, 32
    prompt = torch.randint(1, 512, (2, T), device=DEV)
    keep = torch.ones(2, T, dtype=torch.int64, device=DEV)
    if with_mask:
        keep[1, :4] = 0                                  # row 1 is left-padded by 4 tokens
        prompt[1, :4] = 0

    def run(mode):
        for layer in model.model.layers:
            layer.self_attn.decode_math_attention = mode
        cache = transformers.StaticCache(config=cfg, max_cache_len=L)
        mask = keep.clone()
        pos = (mask.cumsum(-1) - 1).clamp(min=0)
        logits = []
        with torch.no_grad():
            out = model(prompt, attention_mask=mask, position_ids=pos, past_key_values=cache, use_cache=True)
            tok = out.logits[:, -1].argmax(-1, keepdim=True)
            for _ in range(NEW):
                mask = torch.cat([mask, torch.ones(2, 1, dtype=mask.dtype, device=DEV)], -1)
                pos = pos[:, -1:] + 1
                out = model(tok, attention_mask=mask if with_mask else None, position_ids=pos, past_key_values=cache,
                            use_cache=True)
                logits.append(out.logits[:, -1].float())
                tok = out.logits[:, -1].argmax(-1, keepdim=True)
        counters = [int(l.cumulative_length.item()) for l in cache.layers]
        return torch.stack(logits), counters

    ref, c_ref = run(False)     # stock cache update + stock attention
    got, c_got = run(True)      # fused decode path
    assert c_ref == c_got == [T + NEW] * cfg.num_hidden_layers
    spread = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 4e-3 * spread + 4e-3


def test_eet_attention_mask_updated_in_place_is_not_stale(ops):
    """One boolean mask buffer updated in place between steps (custom static decode loops do this): the additive form must
    be rebuilt, not served from the previous step."""
    from eetq_amd.modules.llama_modules import _EETAttentionBase
    m = torch.ones(1, 1, 1, 16, dtype=torch.bool, device=DEV)
    a = _EETAttentionBase._decode_mask_rows(m, 1, 16, torch.float16, torch.device(DEV))
    assert torch.count_nonzero(a) == 0
    m[..., 5] = False
    b = _EETAttentionBase._decode_mask_rows(m, 1, 16, torch.float16, torch.device(DEV))
    assert b[0, 5].item() == float("-inf") and torch.count_nonzero(b) == 1
    f = torch.zeros(1, 1, 1, 16, dtype=torch.float16, device=DEV)
    assert _EETAttentionBase._decode_mask_rows(f, 1, 16, torch.float16, torch.device(DEV)).data_ptr() == f.data_ptr()
    assert not hasattr(f, "_eet_additive")
    assert _EETAttentionBase._decode_mask_rows(torch.zeros(1, 2, 3, 16, device=DEV), 1, 16, torch.float16,
                                               torch.device(DEV)) is False


def test_silu_mul_matches_torch(ops):
    torch.manual_seed(5)
    gu = (torch.randn(3, 5, 2 * 704, device=DEV) * 3).half()
    ref = torch.nn.functional.silu(gu[..., :704]) * gu[..., 704:]
    out = ops.silu_mul(gu)
    assert out.shape == ref.shape
    # same roundings as the two torch ops (fp32 silu -> fp16, fp16 multiply); allow one fp16 ulp for expf differences
    diff = (out.float() - ref.float()).abs()
    assert (diff <= 1e-3 * ref.float().abs() + 1e-6).all()
    assert (out == ref).float().mean().item() > 0.98
    with pytest.raises(RuntimeError):
        ops.silu_mul(gu[..., :-2].contiguous())


def test_generate_with_static_cache_under_torch_compile(ops):
    """transformers compiles the decode forward on its own for generate(cache_implementation="static"); the ctypes-backed
    operators must break the graph instead of being traced, and the tokens must equal the dynamic-cache run."""
    transformers = pytest.importorskip("transformers")
    import copy
    from eetq_amd.utils import eet_accelerator
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=512, max_position_embeddings=128)
    torch.manual_seed(0)
    stock = transformers.LlamaForCausalLM(cfg).half().to(DEV).eval()
    model = eet_accelerator(copy.deepcopy(stock), quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True,
                            fused_residual=True)
    prompt = torch.randint(0, 512, (2, 9), device=DEV)
    a = model.generate(prompt, max_new_tokens=6, do_sample=False, pad_token_id=0)
    try:
        b = model.generate(prompt, max_new_tokens=6, do_sample=False, pad_token_id=0, cache_implementation="static")
    except Exception as e:  # a missing compiler backend on the box is not this library's concern
        if "ctypes" in str(e) or "cuda_stream" in str(e) or "data_ptr" in str(e):
            raise
        pytest.skip("torch.compile unavailable here: %s" % type(e).__name__)
    assert (a == b).float().mean().item() > 0.9


@pytest.mark.parametrize("M,K,N", [(640, 320, 6784), (1024, 320, 5120), (300, 384, 11136)])
def test_tiled_gemm_column_split_launches(ops, oracle, M, K, N):
    """Shapes whose wide tiles end in a mostly empty last round: the launcher runs the complete rounds with 128 x 128 tiles
    and the remaining columns with 128 x 64 tiles in a second launch (weights, scales, bias, residual and y offset by the
    column split).  Parity against the oracle, and the fused epilogue against separate adds, across the split."""
    w, x = _rand_case(K, N, M, seed=3 * M + N)
    x[:, ::3] *= -1
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    y = ops.w8_a16_gemm(xd, processed, scales, path="mfma")
    ref = oracle.w8a16_gemm(x, q, s)
    ok = _tier_a(y.cpu().numpy(), ref)
    assert ok.all(), "max err %g at %s" % (np.abs(y.cpu().numpy().astype(np.float32) - ref.astype(np.float32)).max(),
                                          np.argwhere(~ok)[:4])
    torch.manual_seed(M)
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(M, N, dtype=torch.float16, device=DEV)
    assert torch.equal(ops.w8_a16_gemm(xd, processed, scales, path="mfma", bias=bias, residual=res), res + (y + bias))


@pytest.mark.parametrize("K,N", [(2048, 5120), (2176, 6144), (4096, 5120), (2048, 13824), (2048, 24), (4224, 40),
                                 # 8 + 8 + 4 columns per CU (gemv_mixed_kernel, round 4): N / 256 = 20, 12, 28 and K / 256 >= 16
                                 (5120, 5120), (13824, 5120), (4096, 3072), (8192, 7168)])
def test_gemv_half_tile_row_units_vs_oracle(ops, oracle, K, N):
    """M = 1 shapes for which the launcher picks 8-column units (gemv_half_kernel) or the 8 + 8 + 4 mix (gemv_mixed_kernel):
    N/16 a little above a multiple of the CU count, and small N; N = 24 / 40 end in half a tile row... which the native layout
    does not allow (N % 16), so those must be rejected, not mis-computed."""
    if N % 16:
        w = np.zeros((K, 32), np.float16)
        q, s = oracle.quantize(w)
        with pytest.raises(RuntimeError):
            ops.w8_a16_gemm(torch.zeros(1, K, dtype=torch.float16, device=DEV),
                            torch.zeros(K, N, dtype=torch.int8, device=DEV), torch.zeros(N, dtype=torch.float16, device=DEV))
        return
    w, x = _rand_case(K, N, 1, seed=K + 3 * N)
    x[:, ::2] *= -1
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    y = ops.w8_a16_gemm(xd, processed, scales)
    ref = oracle.w8a16_gemm(x, q, s)
    ok = _tier_a(y.cpu().numpy(), ref)
    assert ok.all(), "max err %g at %s" % (np.abs(y.cpu().numpy().astype(np.float32) - ref.astype(np.float32)).max(),
                                          np.argwhere(~ok)[:4])
    torch.manual_seed(N)
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(1, N, dtype=torch.float16, device=DEV)
    assert torch.equal(ops.w8_a16_gemm(xd, processed, scales, bias=bias, residual=res), res + (y + bias))


def test_eet_attention_grows_its_rotary_table(ops):
    """A sequence longer than config.max_position_embeddings must not index past the cos|sin table."""
    transformers = pytest.importorskip("transformers")
    from eetq_amd.utils import eet_accelerator
    cfg = transformers.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                                   num_key_value_heads=2, vocab_size=64, max_position_embeddings=16)
    torch.manual_seed(0)
    model = eet_accelerator(transformers.LlamaForCausalLM(cfg).half().to(DEV).eval(), quantize=True, fused_attn=True)
    attn = model.model.layers[0].self_attn
    assert attn.rotary_emb.max_seq_len_cached == 16
    with torch.no_grad():
        out = model(torch.randint(0, 64, (1, 40), device=DEV)).logits
    assert attn.rotary_emb.max_seq_len_cached >= 40 and torch.isfinite(out).all()


@pytest.mark.parametrize("K,N", [(4096, 512), (5120, 5120), (2048, 8192), (13824, 64), (1024, 48)])
def test_gemv_rmsnorm_prologue(ops, oracle, K, N):
    """norm -> projection as one launch (M = 1) against the two separate operators: the normalised vector may differ by an
    fp16 ulp in rare elements (the sum of squares is added in another order), so outputs are compared within
    2e-3 * max|y|; with residual and bias the fused epilogue must still match separate adds of the fused result's base."""
    w, x = _rand_case(K, N, 1, seed=K + N)
    x *= 3.0
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    torch.manual_seed(K)
    gamma = (torch.rand(K, device=DEV) + 0.5).half()
    eps = 1e-5
    normed = torch.empty_like(xd)
    ops.layernorm_forward(xd, gamma, normed, eps)
    sep = ops.w8_a16_gemm(normed, processed, scales)
    fused = ops.w8_a16_gemm(xd, processed, scales, norm=(gamma, eps))
    assert (fused.float() - sep.float()).abs().max().item() <= 2e-3 * sep.float().abs().max().item() + 1e-4
    # the oracle's RMS-norm + GEMM as the independent reference
    xn = oracle.rmsnorm_f16(x, gamma.cpu().numpy(), eps)
    ref = oracle.w8a16_gemm(xn, q, s).astype(np.float32)
    assert np.all(np.abs(fused.cpu().numpy().astype(np.float32) - ref) <= 2e-3 * np.abs(ref).max() + 2e-3 * np.abs(ref))
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(1, N, dtype=torch.float16, device=DEV)
    assert torch.equal(ops.w8_a16_gemm(xd, processed, scales, norm=(gamma, eps), bias=bias, residual=res),
                       res + (fused + bias))
    # more than one row: the operator falls back to a separate norm launch, same results as the separate ops
    x2 = torch.cat([xd, xd * 0.5])
    n2 = torch.empty_like(x2)
    ops.layernorm_forward(x2, gamma, n2, eps)
    assert torch.equal(ops.w8_a16_gemm(x2, processed, scales, norm=(gamma, eps)), ops.w8_a16_gemm(n2, processed, scales))


@pytest.mark.parametrize("K,N", [(4096, 512), (13824, 5120), (2048, 8192), (1024, 48)])
def test_gemv_gated_activation_prologue(ops, oracle, K, N):
    """silu(gate) * up -> projection as one launch (M = 1) against silu_mul + GEMV: the same roundings, so bit-identical;
    with residual/bias; multi-row inputs take the two-launch route."""
    w, _ = _rand_case(K, N, 1, seed=K + N)
    q, s = oracle.quantize(w)
    processed = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    scales = torch.from_numpy(s).to(DEV)
    torch.manual_seed(K)
    gu = (torch.randn(1, 2 * K, device=DEV) * 2).half()
    sep = ops.w8_a16_gemm(ops.silu_mul(gu), processed, scales)
    fused = ops.w8_a16_gemm(gu, processed, scales, gated=True)
    assert torch.equal(fused, sep)
    ref = oracle.w8a16_gemm(ops.silu_mul(gu).cpu().numpy(), q, s).astype(np.float32)
    assert np.all(np.abs(fused.cpu().numpy().astype(np.float32) - ref) <= 1e-3 * np.abs(ref).max() + 2e-3 * np.abs(ref))
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(1, N, dtype=torch.float16, device=DEV)
    assert torch.equal(ops.w8_a16_gemm(gu, processed, scales, gated=True, bias=bias, residual=res), res + (sep + bias))
    gu3 = (torch.randn(3, 2 * K, device=DEV)).half()
    assert torch.equal(ops.w8_a16_gemm(gu3, processed, scales, gated=True),
                       ops.w8_a16_gemm(ops.silu_mul(gu3), processed, scales))
    with pytest.raises(RuntimeError):
        ops.w8_a16_gemm(gu[:, : K], processed, scales, gated=True)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("B,T,H,Hkv,S,base", [(1, 128, 2, 2, 160, 0), (1, 200, 4, 2, 256, 0), (2, 130, 4, 4, 300, 37), (1, 64, 2, 1, 64, 0),
                                               (1, 1, 2, 2, 16, 5), (2, 515, 8, 2, 600, 0)])
def test_prefill_attention_vs_torch(ops, B, T, H, Hkv, S, base, D):
    """The prompt's causal attention on the matrix cores (eetq_prefill_attention_f16) against a float32 softmax(q k^T) v of the same
    fp16 inputs: strided query view of a fused QKV row, grouped-query heads, ragged token counts, a prompt appended behind `base`
    cached rows.  fp16 flash tolerance: 3e-3 absolute on outputs of order 1 (probabilities are rounded to fp16 for the second
    product, as in flash-attn, which the reference's block calls here: python/eetq/modules/llama_modules.py:131-143)."""
    if ops.BOUNDARY != "ext":
        pytest.skip("prefill_attention lives in the compiled module")
    torch.manual_seed(B * 1000 + T)
    qkv = torch.randn(B, T, (H + 2 * Hkv) * D, dtype=torch.float16, device=DEV)
    q = qkv[..., :H * D].unflatten(-1, (H, D))
    kc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    vc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    keys = base + T
    out = ops.prefill_attention(q, kc, vc, keys)
    assert out.shape == (B, T, H, D) and out.is_contiguous()
    qq = q.transpose(1, 2).float()
    k, v = kc[:, :, :keys].float(), vc[:, :, :keys].float()
    if Hkv != H:
        k, v = k.repeat_interleave(H // Hkv, 1), v.repeat_interleave(H // Hkv, 1)
    sc = torch.matmul(qq, k.transpose(2, 3)) / D ** 0.5
    mask = torch.arange(keys, device=DEV)[None, :] > (torch.arange(T, device=DEV)[:, None] + base)
    ref = torch.matmul(torch.softmax(sc.masked_fill(mask, float("-inf")), -1), v).transpose(1, 2)
    assert not out.isnan().any()
    assert (out.float() - ref).abs().max().item() < 3e-3
    # rows beyond `keys` must not matter: poison them
    kc[:, :, keys:] = float("nan")
    vc[:, :, keys:] = float("nan")
    assert torch.equal(ops.prefill_attention(q, kc, vc, keys), out)
    with pytest.raises(RuntimeError):
        ops.prefill_attention(q[..., :32], kc[..., :32], vc[..., :32], keys)   # head_dim 32: unsupported, loudly
