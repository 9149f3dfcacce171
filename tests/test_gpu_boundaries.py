"""The two bindings of the C ABI -- the compiled module ``EETQ`` (product boundary) and the ctypes twin -- must give
bit-identical results and the same error behaviour: they are two doors to the same kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def both():
    from eetq_amd import _ext, ops_ctypes
    return _ext.load(), ops_ctypes


def test_product_ops_are_the_compiled_module(both):
    import os
    if os.environ.get("EETQ_AMD_BOUNDARY", "").lower() == "ctypes":
        pytest.skip("this run selects the ctypes twin on purpose")
    ext, _ = both
    import EETQ
    from eetq_amd import ops
    from eetq.modules.qlinear import W8A16Linear  # noqa: F401  (the reference's import path keeps working)
    assert EETQ is ext and ops.BOUNDARY == "ext" and ops.w8_a16_gemm is ext.w8_a16_gemm


@pytest.mark.parametrize("M", [1, 5, 40, 300])
def test_same_bits_through_both_boundaries(both, M):
    ext, ct = both
    torch.manual_seed(M)
    K, N = 512, 384
    w = (torch.randn(K, N) * 0.03).half()
    a = ext.quant_weights(w, torch.int8, True)
    b = ct.quant_weights(w, torch.int8, True)
    assert len(a) == len(b) == 3 and all(torch.equal(p, q) and p.device.type == "cpu" for p, q in zip(a, b))
    assert torch.equal(ext.preprocess_weights(a[0]), ct.preprocess_weights(a[0])) and torch.equal(ext.preprocess_weights(a[0]), a[1])
    assert torch.equal(ext.unprocess_weights(a[1]), a[0])
    assert torch.equal(ext.preprocess_weights(a[0], False, "sm80"), ct.preprocess_weights(a[0], False, "sm80"))
    qw, s = a[1].to(DEV), a[2].to(DEV)
    x = torch.rand(2, M, K, dtype=torch.float16, device=DEV)[0]
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(M, N, dtype=torch.float16, device=DEV)
    assert torch.equal(ext.w8_a16_gemm(x, qw, s), ct.w8_a16_gemm(x, qw, s))
    assert torch.equal(ext.w8_a16_gemm(x, qw, s, bias=bias, residual=res), ct.w8_a16_gemm(x, qw, s, bias=bias, residual=res))
    for act in ("relu", "gelu", "silu"):
        assert torch.equal(ext.w8_a16_gemm(x, qw, s, bias=bias, activation=act), ct.w8_a16_gemm(x, qw, s, bias=bias, activation=act))
    y1 = torch.empty(M, N, dtype=torch.float16, device=DEV)
    y2 = torch.empty_like(y1)
    assert ext.w8_a16_gemm_(x, qw, s, y1, M, N, K) is y1
    ct.w8_a16_gemm_(x, qw, s, y2, M, N, K)
    assert torch.equal(y1, y2)
    gamma = torch.rand(K, dtype=torch.float16, device=DEV)
    assert torch.equal(ext.w8_a16_gemm(x, qw, s, norm=(gamma, 1e-5)), ct.w8_a16_gemm(x, qw, s, norm=(gamma, 1e-5)))
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    assert ext.layernorm_forward(x, gamma, o1, 1e-5) is None
    ct.layernorm_forward(x, gamma, o2, 1e-5)
    assert torch.equal(o1, o2)


def test_same_errors_through_both_boundaries(both):
    ext, ct = both
    x = torch.rand(1, 64, dtype=torch.float16, device=DEV)
    qw = torch.zeros(64, 64, dtype=torch.int8, device=DEV)
    s = torch.ones(64, dtype=torch.float16, device=DEV)
    for mod in (ext, ct):
        with pytest.raises(RuntimeError, match="multiple of 64"):
            mod.w8_a16_gemm(x[:, :32], qw[:32], s)                       # K % 64, from the C ABI
        with pytest.raises(RuntimeError, match="float16"):
            mod.w8_a16_gemm(x.float(), qw, s)
        with pytest.raises(RuntimeError, match="same device|CUDA"):
            mod.w8_a16_gemm(x, qw.cpu(), s)
        with pytest.raises(RuntimeError, match="unknown weight layout"):
            mod.preprocess_weights(qw, False, "sm90")
        with pytest.raises(RuntimeError, match="activation"):
            mod.w8_a16_gemm(x, qw, s, activation="tanh")
        with pytest.raises(RuntimeError):
            mod.w8_a16_gemm(x, qw, s, path="gemv", bias=torch.zeros(3, dtype=torch.float16, device=DEV))


def test_compiled_module_uses_the_current_stream_and_device_guard(both):
    ext, _ = both
    torch.manual_seed(0)
    K, N = 1024, 256
    qw, s = ext.quant_weights((torch.randn(K, N, device=DEV) * 0.02).half(), torch.int8, False)
    x = torch.rand(3, K, dtype=torch.float16, device=DEV)
    ref = ext.w8_a16_gemm(x, qw, s)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        big = torch.rand(4096, 4096, device=DEV) @ torch.rand(4096, 4096, device=DEV)   # keeps `side` busy first
        out = ext.w8_a16_gemm(x, qw, s)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(out, ref)
    # capturable: no allocation-free requirement beyond torch's own caching allocator, no sync inside
    g = torch.cuda.CUDAGraph()
    y = torch.empty(3, N, dtype=torch.float16, device=DEV)
    with torch.cuda.stream(side):
        ext.w8_a16_gemm_(x, qw, s, y, 3, N, K)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        ext.w8_a16_gemm_(x, qw, s, y, 3, N, K)
    y.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    del big


def test_splitk_opt_out_env_var_uses_the_unsplit_tile():
    """EETQ_AMD_SPLITK=0 (read once per process) makes AUTO run 17 <= M <= 128 on the unsplit medium tile: same results as the
    explicit 'mid' path bit for bit, and within tolerance of the default split-K result."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_ROOT"])
import eetq_amd.ops as ops
torch.manual_seed(0)
K, N, M = 2048, 1024, 48
w = torch.randn(K, N, dtype=torch.float16)
_, wq, s = ops.quant_weights(w, torch.int8, True)
x = torch.randn(M, K, dtype=torch.float16, device="cuda:0")
wq, s = wq.cuda(), s.cuda()
auto = ops.w8_a16_gemm(x, wq, s)
mid = ops.w8_a16_gemm(x, wq, s, path="mid")
split = ops.w8_a16_gemm(x, wq, s, path="splitk")
print("AUTO_IS_MID", int(torch.equal(auto, mid)), "AUTO_IS_SPLIT", int(torch.equal(auto, split)),
      "CLOSE", int(torch.allclose(auto.float(), split.float(), atol=2e-2, rtol=2e-3)))
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for flag in ("0", "1"):
        env = dict(os.environ, EETQ_AMD_SPLITK=flag, GRAFT_ROOT=root)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs[flag] = res.stdout.strip().splitlines()[-1].split()
    assert outs["0"][1] == "1" and outs["0"][5] == "1", outs        # opted out: AUTO == mid
    assert outs["1"][3] == "1", outs                                # default: AUTO == split-K


_TALL_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
import eetq_amd.ops as ops
out = {{}}
for K, N, M in {cases!r}:
    g = torch.Generator(device="cuda:0").manual_seed(K + N + M)
    w = torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0", generator=g)
    s = torch.rand(N, dtype=torch.float16, device="cuda:0", generator=g) * 0.01
    x = torch.randn(M, K, dtype=torch.float16, device="cuda:0", generator=g)
    b = torch.randn(N, dtype=torch.float16, device="cuda:0", generator=g)
    r = torch.randn(M, N, dtype=torch.float16, device="cuda:0", generator=g)
    out["%d_%d_%d" % (K, N, M)] = ops.w8_a16_gemm(x, w, s, path="mfma", bias=b, residual=r).cpu().numpy()
np.savez({dst!r}, **out)
"""


def test_tall_tile_equals_the_128_row_tile(tmp_path):
    """(and the deep tile, RB = 2: 256 x 128 on FOUR waves with 256 accumulators per lane, hook value 2)  The 256 x 128 tile on eight waves (gemm_tile_kernel<..., RH = 2>: both row halves read one weight stage, activation fragments
    through a four-deep window, two-phase epilogue; measured 5 - 23 % behind the 128 x 128 tile and therefore only reachable through
    the A/B hook EETQ_AMD_TILE_TALL) must give the 128-row tile's BITS: whole and ragged row tiles, ragged column edge, bias and
    residual, an odd and an even number of K steps."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [(4096, 4096, 2048), (1024, 640, 300), (704, 4096, 1000), (2048, 1008, 513)]
    got = {}
    for name, hook in (("tile128", None), ("tall256", "1"), ("deep256", "2")):
        dst = str(tmp_path / (name + ".npz"))
        env = dict(os.environ)
        if hook:
            env.update(EETQ_AMD_TUNING="1", EETQ_AMD_TILE_TALL=hook)
        r = subprocess.run([sys.executable, "-c", _TALL_CHILD.format(root=root, cases=cases, dst=dst)], env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        got[name] = np.load(dst)
    for key in got["tile128"].files:
        assert np.array_equal(got["tile128"][key], got["tall256"][key]), key
        assert np.array_equal(got["tile128"][key], got["deep256"][key]), key
