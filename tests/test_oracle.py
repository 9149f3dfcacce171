"""CPU tests of the oracle: it is pinned against the reference's own assertions and against CPU torch
nn.Linear (BASELINE.json configs[0]), cross-checked by an independent numpy restatement, and frozen by the
committed golden fixtures.  No GPU, no product code."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import np_quantize_reference


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


# ---------------------------------------------------------------- fp16 conversion helpers

def test_f16_conversions_match_numpy(oracle):
    for h in list(range(0, 65536, 97)) + [0, 1, 0x3FF, 0x400, 0x7BFF, 0x7C00, 0x8000, 0xFBFF, 0xFC00]:
        v = oracle.f16_bits_to_f32(h)
        ref = np.array([h], np.uint16).view(np.float16).astype(np.float32)[0]
        assert (np.isnan(v) and np.isnan(ref)) or v == ref
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(300).astype(np.float32) * s for s in (1e-8, 6e-8, 1e-5, 1.0, 300.0, 7e4)])
    xs = np.concatenate([xs, np.array([65504, 65519.99, 65520, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25, 0.0, -0.0],
                                      np.float32)])
    with np.errstate(over="ignore"):
        ref = xs.astype(np.float16).view(np.uint16)
    for x, r in zip(xs, ref):
        assert oracle.f32_to_f16_bits(x) == int(r), x


# ---------------------------------------------------------------- quantiser

@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_quantize_matches_independent_numpy(oracle, dtype):
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((256, 128)) * 0.03).astype(dtype)
    q, s = oracle.quantize(w)
    q2, s2 = np_quantize_reference(w)
    assert np.array_equal(q, q2)
    assert np.array_equal(s.view(np.uint16 if dtype == np.float16 else np.uint32),
                          s2.view(np.uint16 if dtype == np.float16 else np.uint32))


def test_quantize_edge_cases(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "quant_edge_f16_k128_n64.npz"))
    w = g["w"]
    q, s = oracle.quantize(w)
    # frozen outputs
    assert np.array_equal(q, g["q"]) and np.array_equal(s.view(np.uint16), g["s"].view(np.uint16))
    # facts stated in SURVEY.md section 7 / cutlass_preprocessors.cc:638-648
    assert s[0] == 0 and np.all(q[:, 0] == 127)                    # 0/0 = NaN -> min(127, NaN) = 127
    assert q[5, 1] == -128 and np.count_nonzero(q[:, 1]) == 1       # -amax -> -128
    assert q[7, 2] == 127                                           # +amax -> +128 clipped
    ties = q[1:11, 3].tolist()                                      # C round(): half away from zero
    assert ties == [1, -1, 2, -2, 3, -3, 127, -127, 127, -128]
    assert np.all(q[:, 4] == 127)                                   # equal subnormals -> +128 clipped
    assert q[3, 5] == 127 and q[4, 5] == -128
    assert np.isinf(s[6]) and q[2, 6] == 127 and q[9, 6] == 0       # inf/inf = NaN -> 127 ; 1/inf = 0
    assert q[11, 7] == 127                                          # NaN element -> 127, ignored by the max
    q2, s2 = np_quantize_reference(w)
    assert np.array_equal(q, q2)
    assert np.array_equal(s.view(np.uint16), s2.view(np.uint16))


def test_round_half_even_would_differ(oracle):
    """Known answer from the compiled reference: SURVEY.md (section 7 and Appendix A) ran the reference's own
    cutlass_preprocessors.cc on a default-init fp16 Linear(256 -> 128), seed 1, and recorded that exactly 208 of its
    32 768 int8 values differ from a round-half-to-even quantiser.  The oracle must reproduce that count."""
    torch.manual_seed(1)
    w = torch.nn.Linear(256, 128, bias=False, dtype=torch.float16).weight.detach().t().contiguous().numpy()
    q, s = oracle.quantize(w)
    s32 = np.abs(w.astype(np.float32)).max(axis=0) * np.float32(1 / 128)
    q_even = np.clip(np.rint(w.astype(np.float32) / s32), -128, 127).astype(np.int8)
    assert np.count_nonzero(q != q_even) == 208


@pytest.mark.parametrize("name", ["quant_rand_f16_k192_n256", "quant_rand_f32_k64_n64", "quant_edge_f16_k128_n64"])
def test_golden_quant_fixtures(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    q, s = oracle.quantize(g["w"])
    assert np.array_equal(q, g["q"])
    assert s.tobytes() == g["s"].tobytes()
    assert np.array_equal(oracle.gfx950_pack(q), g["gfx950"])
    assert np.array_equal(oracle.sm80_pack(q), g["sm80"])


# ---------------------------------------------------------------- layouts

def test_sm80_four_step_equals_closed_form(oracle):
    """examples/layers/test_w8a16_gemm.py:33-41 pins preprocess(raw) == processed; here the 4-step restatement
    (P1..P4) must equal the closed form verified in SURVEY.md section 8a, and unpack must invert it."""
    rng = np.random.default_rng(5)
    for K, N in [(64, 64), (192, 256), (128, 64), (256, 320)]:
        q = rng.integers(-128, 128, (K, N), dtype=np.int8)
        a = oracle.sm80_pack(q)
        assert np.array_equal(a, oracle.sm80_pack_closed_form(q))
        assert np.array_equal(oracle.sm80_unpack(a), q)


def test_sm80_reader_recovers_writer_input(oracle):
    """Second-source pin of the processed layout.  The writer (cutlass_preprocessors.cc:137-195, 201-335, 337-358,
    432-495, restated as oracle.sm80_pack) and the reader (the reference GEMV's addressing, converter and un-shuffle:
    weightOnlyBatchedGemv/kernel.h:118-214, 233-292, 294-376; interleaved_numeric_conversion.h:53-85, restated as
    oracle.sm80_reader_unpack) are separate reference files written by different authors: the reader recovering the
    writer's input for every byte value and position is what the reference itself relies on."""
    rng = np.random.default_rng(17)
    for K, N in [(64, 64), (128, 64), (192, 256), (4096, 64), (2048 + 64, 128)]:
        q = rng.integers(-128, 128, (K, N), dtype=np.int8)
        q[:256 if K >= 256 else K, 0] = np.arange(-128, 128, dtype=np.int16)[:min(K, 256)].astype(np.int8)  # every byte value
        packed = oracle.sm80_pack(q)
        assert np.array_equal(oracle.sm80_reader_unpack(packed), q), (K, N)
        assert np.array_equal(oracle.sm80_reader_unpack(packed), oracle.sm80_unpack(packed))
    # one-hot probes: every (k mod 128, n mod 4) position of a 128 x 64 weight lands where the reader looks for it
    K, N = 128, 64
    for k in range(0, K, 7):
        for n in range(0, N, 5):
            q = np.zeros((K, N), np.int8)
            q[k, n] = -77
            assert np.array_equal(oracle.sm80_reader_unpack(oracle.sm80_pack(q)), q)


def test_reference_gemv_numerics_vs_contract(oracle):
    """The reference GEMV's own arithmetic (fp16 hfma2 accumulation per thread, fp32 across threads, kernel.h:325-329,
    411-467) restated on sm80 bytes: it must agree with the contract oracle (fp32 accumulate of the same fp16(q*s)
    weights) to fp16-accumulation accuracy, which also proves the reader consumes scales / activations at the right
    indices.  The contract is the tighter of the two: the HIP kernels are held to it, not to the fp16 chain."""
    torch.manual_seed(1)
    K, N = 4096, 64
    w = torch.nn.Linear(K, N, bias=False, dtype=torch.float16).weight.detach().t().contiguous().numpy()
    q, s = oracle.quantize(w)
    sm80 = oracle.sm80_pack(q)
    for M in (1, 3):
        x = torch.rand(M, K, dtype=torch.float16).numpy()
        y_ref = oracle.ref_gemv_sm80(x, sm80, s).astype(np.float64)
        y = oracle.w8a16_gemm(x, q, s).astype(np.float64)
        # each thread adds 32 products (|p| <= ~0.016) in fp16: error per thread ~ 32 * 2^-11 * 0.25, 256 threads in fp32
        assert np.abs(y_ref - y).max() <= 2e-2 * np.abs(y).max()
        assert np.corrcoef(y_ref.ravel(), y.ravel())[0, 1] > 0.9999
    # permuting the activations must change the reader-based result: it really indexes k
    x = torch.rand(1, K, dtype=torch.float16).numpy()
    y1 = oracle.ref_gemv_sm80(x, sm80, s)
    y2 = oracle.ref_gemv_sm80(np.ascontiguousarray(x[:, ::-1]), sm80, s)
    assert not np.array_equal(y1, y2)


def test_sm80_layout_spot_values(oracle):
    """Hand-derived positions: out[swz((n>>1)*2K + (k>>6)*128 + (n&1)*64 + (k&63))] = q[perm16 row][n] + 128."""
    K, N = 64, 64
    q = np.zeros((K, N), np.int8)
    q[8, 3] = 5     # source row 8 is written at row 2 of its 16-group (perm16[2] = 8)
    out = oracle.sm80_pack(q).reshape(-1).view(np.uint8)
    p = (3 >> 1) * 2 * K + 0 * 128 + (3 & 1) * 64 + 2
    p = (p & ~3) + [0, 2, 1, 3][p & 3]
    assert out[p] == 128 + 5
    assert np.count_nonzero(out != 128) == 1


def test_gfx950_layout_definition(oracle):
    K, N = 128, 32
    rng = np.random.default_rng(9)
    q = rng.integers(-128, 128, (K, N), dtype=np.int8)
    packed = oracle.gfx950_pack(q).reshape(-1).view(np.uint8)
    swz = [0, 2, 1, 3]
    for k, n in [(0, 0), (1, 0), (2, 0), (17, 5), (63, 15), (64, 16), (127, 31), (70, 3)]:
        tile = (n >> 4) * (K >> 6) + (k >> 6)
        lane = ((k >> 4) & 3) * 16 + (n & 15)
        j = k & 15
        off = tile * 1024 + lane * 16 + (j & ~3) + swz[j & 3]
        assert packed[off] == (int(q[k, n]) + 128) & 0xFF
    assert np.array_equal(oracle.gfx950_unpack(oracle.gfx950_pack(q)), q)
    with pytest.raises(ValueError):
        oracle.gfx950_pack(np.zeros((32, 16), np.int8))
    with pytest.raises(ValueError):
        oracle.sm80_pack(np.zeros((64, 32), np.int8))


# ---------------------------------------------------------------- GEMM contract vs CPU torch nn.Linear

def _linear_recipe(M, K, N):
    torch.manual_seed(1)
    lin = torch.nn.Linear(K, N, bias=False, dtype=torch.float16)
    x = torch.rand(M, K, dtype=torch.float16)
    return lin, x


def test_pin_reference_test_qlinear_recipe(oracle, golden_dir):
    """examples/layers/test_qlinear.py:20-36: seed 1, Linear(1024->4096), x = rand(128, 1024), atol = 1e-2 --
    the only numeric tolerance in the reference.  Oracle(quantise + contract GEMM) must meet it against CPU
    torch fp16 nn.Linear, both recomputed now and as committed in the fixture."""
    man = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["linear"][1]
    lin, x = _linear_recipe(128, 1024, 4096)
    w = lin.weight.detach().numpy()
    assert _crc(w) == man["w_crc32"] and _crc(x.numpy()) == man["x_crc32"], "torch RNG drifted: regenerate fixtures"
    q, s = oracle.quantize(np.ascontiguousarray(w.T))
    rows = slice(None, None, man["row_step"])
    y = oracle.w8a16_gemm(x.numpy()[rows], q, s)
    y_gold = np.load(os.path.join(golden_dir, man["file"]))["y"]
    with torch.no_grad():
        y_now = lin(x).numpy()[rows]
    assert np.array_equal(y_now, y_gold)
    assert np.allclose(y.astype(np.float32), y_gold.astype(np.float32), atol=1e-2, rtol=0)
    assert torch.allclose(torch.from_numpy(y), torch.from_numpy(y_gold), atol=1e-2)  # the reference's literal check


def test_pin_config0_cpu_linear(oracle, golden_dir):
    """BASELINE.json configs[0]: EetqLinear in=4096 out=4096, M=1, CPU float16 reference forward."""
    man = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["linear"][2]
    lin, x = _linear_recipe(1, 4096, 4096)
    w = lin.weight.detach().numpy()
    assert _crc(w) == man["w_crc32"] and _crc(x.numpy()) == man["x_crc32"]
    q, s = oracle.quantize(np.ascontiguousarray(w.T))
    y = oracle.w8a16_gemm(x.numpy(), q, s).astype(np.float32)
    y_gold = np.load(os.path.join(golden_dir, man["file"]))["y"].astype(np.float32)
    # tier-B (vs original weights): quantisation error dominates; 1e-2 scaled by sqrt(K/1024)
    assert np.abs(y - y_gold).max() <= 1e-2 * 2.0
    # tier-A (vs Linear on the dequantised weights): only fp16/fp32 rounding differences remain
    wdq = torch.from_numpy(oracle.dequant(q, s))             # [K, N] fp16
    with torch.no_grad():
        y_a = torch.nn.functional.linear(x, wdq.t().contiguous()).numpy().astype(np.float32)
    assert np.abs(y - y_a).max() <= 1e-3 * np.abs(y_a).max() + 2e-3 * 0  # atol = 1e-3 * max|y|


def test_small_fixture_full_inputs(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "linear_small_m4_k256_n128.npz"))
    q, s = oracle.quantize(np.ascontiguousarray(g["w"].T))
    y = oracle.w8a16_gemm(g["x"], q, s)
    assert np.allclose(y.astype(np.float32), g["y"].astype(np.float32), atol=1e-2, rtol=0)


def test_fp32_order_band_is_tiny(oracle):
    """'summation order free': exact (double) accumulation vs strict fp32 left-to-right differ by <= 1 fp16 ulp."""
    rng = np.random.default_rng(11)
    K, N, M = 1024, 256, 8
    w = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    x = rng.random((M, K)).astype(np.float16)
    q, s = oracle.quantize(w)
    a = oracle.w8a16_gemm(x, q, s).astype(np.float32)
    b = oracle.w8a16_gemm_f32acc(x, q, s).astype(np.float32)
    ulp = np.spacing(np.abs(a).astype(np.float16)).astype(np.float32)
    assert np.all(np.abs(a - b) <= ulp)


def test_identity_gemm_is_dequant(oracle):
    """python/eetq/modules/qlinear.py:83-86: multiplying an identity dequantises the weight exactly."""
    rng = np.random.default_rng(13)
    K, N = 128, 64
    w = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    q, s = oracle.quantize(w)
    eye = np.eye(K, dtype=np.float16)
    assert np.array_equal(oracle.w8a16_gemm(eye, q, s), oracle.dequant(q, s))


# ---------------------------------------------------------------- side ops

def test_rmsnorm_against_torch(oracle):
    torch.manual_seed(0)
    x = (torch.randn(3, 5, 512) * 2).half()
    g = (torch.rand(512) + 0.5).half()
    ref = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6)) * g.float()
    out = oracle.rmsnorm_f16(x.numpy(), g.numpy(), 1e-6).astype(np.float32)
    assert np.allclose(out, ref.numpy(), rtol=2e-3, atol=1e-3)
    ones = np.full((1, 8), 3.0, np.float16)              # x * rsqrt(mean x^2) == 1 -> out = gamma, clamped
    out = oracle.rmsnorm_f16(ones, np.array([65000, -65000] * 4, np.float16), 0.0)
    lim = np.float16(65504.0 - 1000.0)                    # clamp_inf_for_half (reduction.cuh:78-82)
    assert np.array_equal(out[0], np.array([lim, -lim] * 4, np.float16))


def test_rotary_against_torch(oracle):
    torch.manual_seed(0)
    tokens, heads, hs = 6, 4, 64
    q = torch.randn(tokens, heads, hs).half()
    k = torch.randn(tokens, heads, hs).half()
    inv = 1.0 / (10000 ** (torch.arange(0, hs, 2).float() / hs))
    t = torch.arange(32).float()
    fr = torch.einsum("i,j->ij", t, inv)
    cache = torch.cat([fr.cos(), fr.sin()], -1).half()
    pos = torch.tensor([3, 0, 7, 31, 2, 9])
    qo, ko = oracle.rotary_neox_f16(pos.numpy(), q.numpy(), k.numpy(), cache.numpy(), hs)
    c = cache[pos][:, None, : hs // 2].float()
    s = cache[pos][:, None, hs // 2:].float()
    def rot(v):
        v = v.float()
        x, y = v[..., : hs // 2], v[..., hs // 2:]
        return torch.cat([x * c - y * s, y * c + x * s], -1)
    assert np.allclose(qo.astype(np.float32), rot(q).numpy(), atol=4e-3, rtol=4e-3)
    assert np.allclose(ko.astype(np.float32), rot(k).numpy(), atol=4e-3, rtol=4e-3)


def test_activation_epilogue_restatement(oracle):
    """fp16(act(acc + bias)) against numpy: ReLU exact, GELU (tanh form) and SiLU to fp16 rounding; with act applied to a
    value the fp16 identity epilogue would have rounded first, the two differ -- that is the point of the fp32 epilogue."""
    rng = np.random.default_rng(21)
    K, N, M = 256, 64, 5
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float16)
    q, s = oracle.quantize(w)
    x = rng.standard_normal((M, K)).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float16)
    acc = x.astype(np.float64) @ oracle.dequant(q, s).astype(np.float64)
    z = acc.astype(np.float32) + bias.astype(np.float32)[None, :]
    z64 = z.astype(np.float64)
    want = {"relu": np.maximum(z64, 0.0),
            "gelu": 0.5 * z64 * (1.0 + np.tanh(0.7978845608028654 * z64 * (1.0 + 0.044715 * z64 * z64))),
            "silu": z64 / (1.0 + np.exp(-z64))}
    for act, ref in want.items():
        got = oracle.w8a16_gemm_bias_act(x, q, s, bias, act)
        assert np.array_equal(got, ref.astype(np.float32).astype(np.float16)), act
    nob = oracle.w8a16_gemm_bias_act(x, q, s, None, "relu")
    assert np.array_equal(nob, np.maximum(acc.astype(np.float32), 0).astype(np.float16))


# ---------------------------------------------------------------- int4 (W4A16)

def test_int4_quantise_restatement(oracle):
    """cutlass_preprocessors.cc:605-674 with PACKED_INT4_WEIGHT_ONLY: scale = amax / 8, q = clamp(int(round(w / s)), -8, 7),
    two values per byte along N (even column in the low nibble); +amax -> 8 -> clipped to 7, -amax -> -8; an all-zero
    column has scale 0 and q = -8 (int(NaN) is INT_MIN on x86, clamped)."""
    rng = np.random.default_rng(3)
    K, N = 64, 8
    w = rng.standard_normal((K, N)).astype(np.float16)
    w[:, 3] = 0
    w[0, 0], w[1, 0] = 7.0, -7.0                # column 0: amax 7 -> scale 0.875
    w[2:, 0] = (np.arange(K - 2) % 15 - 7) * np.float16(0.4375)   # exact ties at half-integers of w / s
    qp, s = oracle.quantize_i4(w)
    q = oracle.i4_values(qp)
    assert qp.shape == (K, N // 2) and q.min() >= -8 and q.max() <= 7
    assert s[0] == np.float16(0.875) and q[0, 0] == 7 and q[1, 0] == -8
    ties = [int(v) for v in q[2:17, 0]]          # w/s = -3.5, -3, -2.5, ... : half away from zero
    assert ties == [-4, -3, -3, -2, -2, -1, -1, 0, 1, 1, 2, 2, 3, 3, 4]
    assert s[3] == 0 and np.all(q[:, 3] == -8)
    s32 = np.abs(w.astype(np.float32)).max(axis=0) * np.float32(0.125)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = w.astype(np.float32) / s32
    ref = np.sign(r) * np.floor(np.abs(r) + np.float32(0.5))
    ref = np.where(np.isnan(ref), -8, np.clip(ref, -8, 7)).astype(np.int8)
    assert np.array_equal(q, ref)
    assert np.array_equal(oracle.i4_from_values(q), qp)
    assert (qp.view(np.uint8)[5, 1] & 0xF) == (int(q[5, 2]) & 0xF) and (qp.view(np.uint8)[5, 1] >> 4) == (int(q[5, 3]) & 0xF)
    q32, s32b = oracle.quantize_i4(w.astype(np.float32))
    assert np.array_equal(q32, qp) and s32b.dtype == np.float32


def test_int4_sm80_layout_writer_reader_and_inverse(oracle):
    """The int4 processed layout from both reference sides: the writer (cutlass_preprocessors.cc P1..P4 for int4) and the
    reader (the GEMV's Int4b addressing + the int4 converter); reader(writer(q)) == q for every nibble value and position,
    and the closed-form inverse agrees."""
    rng = np.random.default_rng(8)
    for K, N in [(64, 64), (128, 64), (192, 256), (2048 + 64, 128)]:
        q = rng.integers(-8, 8, (K, N), dtype=np.int8)
        q[:16, 1] = np.arange(-8, 8, dtype=np.int8)
        qp = oracle.i4_from_values(q)
        packed = oracle.sm80_pack_i4(qp)
        assert np.array_equal(oracle.sm80_reader_unpack_i4(packed), qp), (K, N)
        assert np.array_equal(oracle.sm80_unpack_i4(packed), qp), (K, N)
    K, N = 128, 64
    for k in range(0, K, 5):
        for n in range(0, N, 7):
            q = np.zeros((K, N), np.int8)
            q[k, n] = -5
            qp = oracle.i4_from_values(q)
            assert np.array_equal(oracle.sm80_reader_unpack_i4(oracle.sm80_pack_i4(qp)), qp)
    with pytest.raises(ValueError):
        oracle.sm80_pack_i4(np.zeros((64, 16), np.int8))       # N = 32


def test_int4_native_layout_definition(oracle):
    K, N = 256, 32
    rng = np.random.default_rng(2)
    q = rng.integers(-8, 8, (K, N), dtype=np.int8)
    qp = oracle.i4_from_values(q)
    packed = oracle.gfx950_pack_i4(qp).reshape(-1).view(np.uint8)
    for k, n in [(0, 0), (1, 0), (2, 0), (7, 3), (8, 3), (31, 15), (32, 0), (127, 15), (128, 16), (255, 31), (77, 9)]:
        tile = (n >> 4) * (K >> 7) + (k >> 7)
        lane = ((k >> 5) & 3) * 16 + (n & 15)
        d, j = (k >> 3) & 3, k & 7
        nib = (tile * 1024 + lane * 16 + d * 4) * 2 + (j >> 1) + 4 * (j & 1)
        assert (packed[nib >> 1] >> (4 * (nib & 1))) & 0xF == int(q[k, n]) + 8
    assert np.array_equal(oracle.gfx950_unpack_i4(oracle.gfx950_pack_i4(qp)), qp)
    with pytest.raises(ValueError):
        oracle.gfx950_pack_i4(np.zeros((64, 16), np.int8))     # K % 128
