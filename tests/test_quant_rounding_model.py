"""The arithmetic of eetq_amd/csrc/quant.hip::quant_pack_kernel, restated in numpy and checked against the oracle on CPU.

The kernel does not divide: b = fma(w, rcp(s), 128) names the two integers the result can be (floor(b), floor(b) + 1) and
the sign of fma(m, s, -w), m the tie between them, picks one; a zero is an exact tie and goes away from zero like C round()
(cutlass_preprocessors.cc:644-648).  This test is the proof obligation of that shortcut in executable form: with the
reciprocal off by up to one ulp either way, every element outside the kernel's own fallback set must equal the oracle's
round(w / s).  (The GPU kernel itself is compared with the oracle in tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

from oracle import oracle


def _fma(a, b, c):  # fp32 fma: the product of two fp32 is exact in fp64
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _model(w, s, fp32_input, ulp, bias=128):
    w = w.astype(np.float32)
    s = np.broadcast_to(s.astype(np.float32), w.shape)
    with np.errstate(all="ignore"):
        r = (np.float32(1) / s).astype(np.float32)
    r = np.where(ulp > 0, np.nextafter(r, np.float32(np.inf)), np.where(ulp < 0, np.nextafter(r, np.float32(-np.inf)), r))
    r = r.astype(np.float32)
    b = _fma(w, r, np.full_like(w, bias))
    f = np.floor(b)
    m = (f - np.float32(bias - 0.5)).astype(np.float32)
    z = _fma(m, s, -w)
    up = z.view(np.int32) <= np.where(w < 0, -1, 0)
    q = np.clip(f + up, 0, 2 * bias - 1) - bias
    fallback = ~((s > 1e-30) & (s < 1e30)) | np.isnan(b)
    if fp32_input:
        hulp = ((m.view(np.int32) & 0x7F800000) - (24 << 23)).astype(np.int32).view(np.float32)
        with np.errstate(all="ignore"):
            fallback |= (z != 0) & (np.abs(z) <= s * hulp)
    return q.astype(np.int8), fallback


def _matrices():
    rng = np.random.default_rng(7)
    K, N = 512, 256
    yield "fp16 uniform (nn.Linear-like)", ((rng.random((K, N)) * 2 - 1) / 64).astype(np.float16)
    yield "fp16 normal", (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    yield "fp16 tiny columns", (rng.standard_normal((K, N)) * 10.0 ** rng.uniform(-7, -3, N)[None, :]).astype(np.float16)
    yield "fp32 normal", (rng.standard_normal((K, N)) * 0.02).astype(np.float32)
    bf = (rng.standard_normal((K, N)) * 0.02).astype(np.float32)
    yield "fp32 on the bf16 grid", (bf.view(np.uint32) & 0xFFFF0000).view(np.float32)
    amax = (10.0 ** rng.uniform(-3, 1, N)).astype(np.float32)
    w = ((rng.integers(-128, 128, (K, N)) + 0.5) * (amax * np.float32(1 / 128))[None, :]).astype(np.float32)
    for _ in range(2):
        st = rng.integers(-1, 2, (K, N))
        w = np.where(st > 0, np.nextafter(w, np.float32(np.inf)), np.where(st < 0, np.nextafter(w, np.float32(-np.inf)), w))
    w = np.clip(w, -amax[None, :], amax[None, :]).astype(np.float32)
    w[0, :] = amax
    yield "fp32 ties and their neighbours", w


@pytest.mark.parametrize("name,w", list(_matrices()), ids=lambda v: v if isinstance(v, str) else "")
def test_division_free_rounding_equals_the_oracle(name, w):
    q_ref, _ = oracle.quantize(w)
    amax = np.abs(w.astype(np.float32)).max(axis=0)
    s = amax * np.float32(1 / 128)
    rng = np.random.default_rng(len(name))
    q, fallback = _model(w, s[None, :], w.dtype == np.float32, rng.integers(-1, 2, w.shape))
    keep = ~fallback
    assert keep.mean() > 0.5, "the fallback set must stay the exception"
    assert np.array_equal(q[keep], q_ref[keep]), f"{name}: {int((q[keep] != q_ref[keep]).sum())} elements differ"
    if w.dtype == np.float16:
        assert keep.all() or not np.isfinite(s).all() or (s <= 1e-30).any()


@pytest.mark.parametrize("name,w", [m for m in _matrices() if "tiny" not in m[0]], ids=lambda v: v if isinstance(v, str) else "")
def test_division_free_rounding_equals_the_oracle_int4(name, w):
    """The int4 instantiation (quant_pack_kernel<T, 4>): scale = amax / 8, bias 8, nibble = q + 8 in [0, 15]."""
    w = (w.astype(np.float32) * np.float32(1.0 / 16.0)).astype(w.dtype) if "ties" in name else w   # ties of amax / 8 as well
    if w.shape[1] % 2:
        w = w[:, :-1]
    packed, _ = oracle.quantize_i4(w)                       # [K, N/2], even column in the low nibble, two's complement
    lo = (packed.astype(np.uint8) & 0xF).astype(np.int8)
    hi = (packed.astype(np.uint8) >> 4).astype(np.int8)
    q_ref = np.empty(w.shape, np.int8)
    q_ref[:, 0::2] = np.where(lo > 7, lo - 16, lo)
    q_ref[:, 1::2] = np.where(hi > 7, hi - 16, hi)
    amax = np.abs(w.astype(np.float32)).max(axis=0)
    s = amax * np.float32(1 / 8)
    rng = np.random.default_rng(len(name) + 4)
    q, fallback = _model(w, s[None, :], w.dtype == np.float32, rng.integers(-1, 2, w.shape), bias=8)
    keep = ~fallback
    assert keep.mean() > 0.5
    assert np.array_equal(q[keep], q_ref[keep]), f"{name}: {int((q[keep] != q_ref[keep]).sum())} elements differ"
