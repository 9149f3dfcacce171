"""Gated-MLP activation in the gate|up projection's epilogue ("glu8" column order).

The reference computes act_fn(gate_proj(x)) * up_proj(x) with separate launches (transformers' LlamaMLP over W8A16Linear);
here the fused gate|up weight is column-interleaved so the M = 1 GEMV writes silu(gate) * up directly.  Everything must be
bit-identical to the plain-order path (whose pieces are checked against the oracle in test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import eetq_amd.ops as o
    return o


def _mlp_parts(K, I, bias, seed):
    from eetq_amd.modules.qlinear import W8A16Linear
    torch.manual_seed(seed)
    mk = lambda n_in, n_out: W8A16Linear.from_torch(torch.nn.Linear(n_in, n_out, bias=bias, dtype=torch.float16, device=DEV))
    return mk(K, I), mk(K, I), mk(I, K)


@pytest.mark.parametrize("K,I,bias", [(4096, 11008, False), (5120, 13824, False), (1024, 2816, True), (512, 192, True),
                                      (4096, 4096, False), (2048, 5120, False)])
def test_glu8_gemv_equals_projection_then_silu_mul(ops, K, I, bias):
    """M = 1 (one launch, with and without the RMS-norm prologue), M = 2..4, 8 (small-batch kernel's epilogue), 64, 300 (the other
    kernels + the glu8 silu_mul launch, or the tiled MFMA kernel's gated write-out where AUTO runs the shape on it), 1024 and
    1100 (prompts: the gated write-out, whole and ragged row tiles, column-split launches at I = 11008): all equal to gate|up in
    plain order followed by silu_mul."""
    from eetq_amd.utils.fuse import fuse_w8a16_linears
    gate, up, _ = _mlp_parts(K, I, bias, seed=K + I)
    plain = fuse_w8a16_linears([gate, up]).fused
    glu = fuse_w8a16_linears([gate, up], glu8=True).fused
    assert glu.glu8 and not plain.glu8 and glu.qweight.shape == plain.qweight.shape
    gamma = (torch.rand(K, dtype=torch.float16, device=DEV) + 0.5)
    for M in (1, 2, 4, 8, 64, 300, 1024, 1100):
        x = torch.randn(M, K, dtype=torch.float16, device=DEV)
        for norm in (None, (gamma, 1e-5)):
            ref = ops.silu_mul(plain(x, norm=norm))
            got = glu(x, norm=norm, activation="silu_glu8")
            assert got.shape == (M, I) and torch.equal(got, ref), (M, norm is not None)
    x3 = torch.randn(1, 1, K, dtype=torch.float16, device=DEV)      # [B, T, K] single token
    assert torch.equal(glu(x3, activation="silu_glu8"), ops.silu_mul(plain(x3)))
    with pytest.raises(RuntimeError):
        glu(x3, residual=torch.zeros(1, 1, I, dtype=torch.float16, device=DEV), activation="silu_glu8")


def test_glu8_prompt_write_out_is_the_path_taken(ops):
    """eetq_w8a16_gemm_glu8 at M > 16: EETQ_OK exactly where eetq_diag_auto_path names the unsplit tiled MFMA kernel for the plain
    projection (the gated write-out ran: the result equals projection + silu_mul), EETQ_ERR_UNSUPPORTED with nothing written
    everywhere else -- so the operator's fallback to two launches is a documented branch, not a silent one."""
    import ctypes
    from eetq_amd import _lib
    from eetq_amd.utils.fuse import fuse_w8a16_linears
    L = _lib.lib()
    gate, up, _ = _mlp_parts(4096, 11008, True, seed=5)
    glu = fuse_w8a16_linears([gate, up], glu8=True).fused
    N, K = 2 * 11008, 4096
    for M, fused in ((1024, True), (513, True), (64, True), (24, False), (32, False)):
        path, aux = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(L.eetq_diag_auto_path(8, M, N, K, ctypes.byref(path), ctypes.byref(aux)))
        assert (path.value == _lib.PATH_MFMA or (path.value == _lib.PATH_TILESPLIT and aux.value == 1)) == fused, (M, path.value)
        x = torch.randn(M, K, dtype=torch.float16, device=DEV)
        y = torch.full((M, N // 2), 7.0, dtype=torch.float16, device=DEV)
        st = L.eetq_w8a16_gemm_glu8(x.data_ptr(), glu.qweight.data_ptr(), glu.weight_scales.data_ptr(), glu.bias.data_ptr(),
                                    y.data_ptr(), M, N, K, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if fused:
            assert st == 0
            assert torch.equal(y, ops.silu_mul(ops.w8_a16_gemm(x, glu.qweight, glu.weight_scales, bias=glu.bias), glu8=True))
        else:
            assert st == _lib.ERR_UNSUPPORTED and (y == 7.0).all()


def test_glu8_silu_mul_kernel_vs_torch(ops):
    torch.manual_seed(1)
    gu = torch.randn(5, 2 * 176, dtype=torch.float16, device=DEV) * 3
    g = gu.unflatten(-1, (-1, 2, 8))[..., 0, :].flatten(-2)
    u = gu.unflatten(-1, (-1, 2, 8))[..., 1, :].flatten(-2)
    assert torch.equal(ops.silu_mul(gu, glu8=True), torch.nn.functional.silu(g.float()).half() * u)
    assert torch.equal(ops.silu_mul(gu), torch.nn.functional.silu(gu[:, :176].float()).half() * gu[:, 176:])


def test_eet_mlp_glu8_equals_plain_order(ops):
    from eetq_amd.modules.llama_modules import EETLlamaMLP
    gate, up, down = _mlp_parts(1024, 2816, False, seed=3)
    a, b = EETLlamaMLP(gate, up, down, glu8=True), EETLlamaMLP(gate, up, down, glu8=False)
    assert a.glu8 and not b.glu8
    gamma = torch.rand(1024, dtype=torch.float16, device=DEV) + 0.5
    for shape in ((1, 1, 1024), (2, 1, 1024), (1, 37, 1024)):
        x = torch.randn(*shape, dtype=torch.float16, device=DEV)
        assert torch.equal(a(x), b(x))
        assert torch.equal(a(x, residual=x, norm=(gamma, 1e-6)), b(x, residual=x, norm=(gamma, 1e-6)))
