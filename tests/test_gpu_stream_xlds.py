"""Small batches (2 <= M <= 16) with the activations staged in LDS (eetq_amd/csrc/streamk_kernel.hpp, XM = 1 / 2 / 3; round 4):
either the M rows are copied once per workgroup into LDS by LDS-DMA ("block"), or every wave DMAs the rows of its next k tile into
its own ring ("ring", M <= 8); the MFMA A fragments are LDS reads, instead of 16 clamped rows from L2 per weight tile.  The
fragments, the MFMAs and their order are those of the register form at the same workgroup size, so the results must be
BIT-IDENTICAL to it (checked against a second process that runs with EETQ_AMD_I8_STREAM_PLAN=regs / EETQ_AMD_I4_STREAM_PLAN=regs)
and tier A against the oracle.  Shapes: the reference's batched-GEMV range and the small-M end of its CUTLASS range on Llama-2-7B / 13B projections
(weightOnlyBatchedGemv/kernelLauncher.cu:165-192, fpA_intB_gemm_template.h dispatch)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (bits, K, N, M): every case is one the dispatcher stages in LDS by default on a 256-CU MI355X (streamk.hip::pick_plan /
# pick_plan_i4) -- block copies, 8- and 16-row rings with one and two tile rows per workgroup, 8 and 16 waves, the int4 4- and 16-row rings
CASES = [(8, 4096, 11008, 2), (8, 4096, 11008, 5), (8, 4096, 11008, 8), (8, 4096, 12288, 3), (8, 5120, 13824, 4), (8, 4096, 4096, 4),
         (8, 4096, 4096, 7), (8, 11008, 4096, 2), (8, 11008, 4096, 8), (8, 8192, 1024, 4), (8, 2048, 8256, 7), (8, 3072, 9216, 8),
         (8, 3072, 9216, 2), (8, 4096, 22016, 3), (8, 4096, 22016, 6), (8, 8192, 8192, 5), (8, 4096, 14336, 2),
         (8, 4096, 4096, 16), (8, 4096, 4096, 9), (8, 5120, 5120, 2), (8, 4096, 11008, 12), (8, 11008, 4096, 11), (8, 5120, 5120, 13), (8, 2048, 8256, 16),
         (4, 4096, 11008, 2), (4, 4096, 11008, 7), (4, 4096, 12288, 4), (4, 8192, 1024, 6), (4, 4096, 4096, 3), (4, 4096, 4096, 8),
         (4, 4096, 4096, 13), (4, 4096, 4096, 16), (4, 11008, 4096, 6), (4, 13824, 5120, 5), (4, 5120, 27648, 4),
         (4, 11008, 4096, 12), (4, 5120, 5120, 14), (4, 8192, 1024, 9)]


def _inputs(bits, K, N, M):
    rng = np.random.default_rng(bits * 1000003 + K * 31 + N * 7 + M)
    q = rng.integers(-128, 128, (K, N if bits == 8 else N // 2), dtype=np.int8)
    s = (rng.random(N) * 0.02 + 0.001).astype(np.float16)
    x = (rng.random((M, K)) - 0.5).astype(np.float16)
    return q, s, x


_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
sys.path.insert(0, {tests!r})
import oracle
from test_gpu_stream_xlds import CASES, _inputs
import eetq_amd.ops as ops
out = {{}}
for bits, K, N, M in CASES:
    q, s, x = _inputs(bits, K, N, M)
    pk = oracle.gfx950_pack(q) if bits == 8 else oracle.gfx950_pack_i4(q)
    y = ops.w8_a16_gemm(torch.from_numpy(x).cuda(), torch.from_numpy(pk).cuda(), torch.from_numpy(s).cuda())
    out["%d_%d_%d_%d" % (bits, K, N, M)] = y.cpu().numpy()
np.savez({dst!r}, **out)
"""


@pytest.fixture(scope="module")
def register_form(tmp_path_factory, oracle):
    """The same cases in a process that takes the activation fragments straight from L2 into registers (the form of rounds 1-3),
    with the tile rows per workgroup and the workgroup size the rule picks for the LDS forms."""
    dst = str(tmp_path_factory.mktemp("xlds") / "regs.npz")
    env = dict(os.environ, EETQ_AMD_TUNING="1", EETQ_AMD_I8_STREAM_PLAN="regs,0,0", EETQ_AMD_I4_STREAM_PLAN="regs,0,0")
    code = _CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), dst=dst)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(dst)


@pytest.mark.parametrize("bits,K,N,M", CASES)
def test_lds_staged_rows_equal_register_form_and_oracle(oracle, register_form, bits, K, N, M):
    import eetq_amd.ops as ops
    q, s, x = _inputs(bits, K, N, M)
    pk = oracle.gfx950_pack(q) if bits == 8 else oracle.gfx950_pack_i4(q)
    y = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), torch.from_numpy(pk).to(DEV), torch.from_numpy(s).to(DEV)).cpu().numpy()
    assert np.array_equal(y, register_form["%d_%d_%d_%d" % (bits, K, N, M)])          # same fragments, same MFMAs, same order
    cols = slice(N - 512, N)                                                           # the last workgroups' columns
    vals = q if bits == 8 else oracle.i4_values(q)
    ref = oracle.w8a16_gemm(x, np.ascontiguousarray(vals[:, cols]), s[cols]).astype(np.float32)
    got = y[:, cols].astype(np.float32)
    assert np.all(np.abs(got - ref) <= 1e-3 * np.abs(ref).max() + 2e-3 * np.abs(ref))   # tier A


_CHILD_ONE = """
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
sys.path.insert(0, {tests!r})
import oracle
from test_gpu_stream_xlds import _inputs
import eetq_amd.ops as ops
q, s, x = _inputs(8, 4096, 4096, 4)
y = ops.w8_a16_gemm(torch.from_numpy(x).cuda(), torch.from_numpy(oracle.gfx950_pack(q)).cuda(), torch.from_numpy(s).cuda())
np.save({dst!r}, y.cpu().numpy())
"""


def test_stray_tuning_variables_do_not_reach_production_launches(oracle, tmp_path):
    """The A/B hooks change the kernel (and, through the wave count, the summation order): they answer only under
    EETQ_AMD_TUNING=1.  A process that merely has EETQ_AMD_I8_STREAM_PLAN / _WAVES in its environment must produce the bits of a
    clean process; with the switch the forced 8-wave register form runs (tier A, another summation order)."""
    import eetq_amd.ops as ops
    q, s, x = _inputs(8, 4096, 4096, 4)
    clean = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), torch.from_numpy(oracle.gfx950_pack(q)).to(DEV),
                            torch.from_numpy(s).to(DEV)).cpu().numpy()
    outs = {}
    for name, extra in (("stray", {}), ("tuning", {"EETQ_AMD_TUNING": "1"})):
        dst = str(tmp_path / (name + ".npy"))
        env = dict(os.environ, EETQ_AMD_I8_STREAM_PLAN="regs,1,8", EETQ_AMD_I8_STREAM_WAVES="8", **extra)
        env.pop("EETQ_AMD_TUNING", None) if name == "stray" else None
        code = _CHILD_ONE.format(root=ROOT, tests=os.path.join(ROOT, "tests"), dst=dst)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = np.load(dst)
    assert np.array_equal(outs["stray"], clean)                     # the variables were never read
    ref = oracle.w8a16_gemm(x, q, s).astype(np.float32)
    got = outs["tuning"].astype(np.float32)
    assert np.all(np.abs(got - ref) <= 1e-3 * np.abs(ref).max() + 2e-3 * np.abs(ref))
    assert not np.array_equal(outs["tuning"], clean)                # 8 waves x registers: a different summation order


def test_lds_staged_rows_rows_do_not_leak(oracle):
    """Rows >= M of the 16-row MFMA tile read row M-1 (never stored); a NaN / Inf in one row must stay in that row, and the
    output must not depend on what follows the activations in memory (the copy is bounded by M*K*2 bytes)."""
    import eetq_amd.ops as ops
    K, N, M = 4096, 11008, 3
    q, s, x = _inputs(8, K, N, M)
    pk = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV)
    sd = torch.from_numpy(s).to(DEV)
    big = torch.full((M + 13, K), float("nan"), dtype=torch.float16, device=DEV)     # NaNs right behind the M rows
    big[:M] = torch.from_numpy(x).to(DEV)
    clean = ops.w8_a16_gemm(torch.from_numpy(x).to(DEV), pk, sd)
    assert torch.equal(ops.w8_a16_gemm(big[:M], pk, sd), clean) and torch.isfinite(clean).all()
    big[1, 77] = float("inf")
    y = ops.w8_a16_gemm(big[:M], pk, sd)
    assert torch.equal(y[0], clean[0]) and torch.equal(y[2], clean[2]) and not torch.isfinite(y[1]).all()


def test_lds_staged_rows_bias_residual_glu_and_graph(oracle):
    """The epilogues ride on the same kernel; > 64 KiB of dynamic LDS (M = 8, K = 4096) inside a captured graph."""
    import eetq_amd.ops as ops
    K, N, M = 4096, 11008, 8
    q, s, x = _inputs(8, K, N, M)
    pk, sd, xd = torch.from_numpy(oracle.gfx950_pack(q)).to(DEV), torch.from_numpy(s).to(DEV), torch.from_numpy(x).to(DEV)
    torch.manual_seed(1)
    bias = torch.randn(N, dtype=torch.float16, device=DEV)
    res = torch.randn(M, N, dtype=torch.float16, device=DEV)
    plain = ops.w8_a16_gemm(xd, pk, sd)
    fused = ops.w8_a16_gemm(xd, pk, sd, bias=bias, residual=res)
    assert torch.equal(fused, (plain + bias) + res)           # fp16 add after the fp16 rounding, like the reference's separate ops
    act = ops.w8_a16_gemm(xd[:4], pk, sd, activation="silu_glu8")
    assert act.shape == (4, N // 2) and torch.equal(act, ops.silu_mul(ops.w8_a16_gemm(xd[:4], pk, sd), glu8=True))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.w8_a16_gemm(xd, pk, sd)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.w8_a16_gemm(xd, pk, sd)
    xd.mul_(0.5)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ops.w8_a16_gemm(xd, pk, sd))


def test_plan_query_uses_the_device_cu_count():
    """eetq_diag_stream_plan with cus <= 0 asks the device: an MI355X has 256 CUs, so both calls agree (tests/test_abi.py pins the
    256-CU table on the host)."""
    import ctypes
    from eetq_amd import _lib
    lib = _lib.lib()
    for bits, K, N, M in CASES:
        got = []
        for cus in (0, 256):
            f, t, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            assert lib.eetq_diag_stream_plan(bits, M, N, K, cus, ctypes.byref(f), ctypes.byref(t), ctypes.byref(w)) == 0
            got.append((f.value, t.value, w.value))
        assert got[0] == got[1], (bits, M, N, K, got)
        assert got[0][0] in (1, 2), (bits, M, N, K, got)          # every case of this file is one the rule stages in LDS


def test_cold_graph_capture_of_the_large_lds_forms():
    """The first launch of a kernel that needs > 64 KiB of dynamic LDS opts in (hipFuncSetAttribute) -- also when that first launch
    happens INSIDE a graph capture, in a fresh process (9 <= M <= 11 on one tile row per CU: 16 waves x 3 slots x 2 KiB + reduction)."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch, eetq_amd.ops as ops\n"
        "torch.manual_seed(0)\n"
        "w = torch.randint(-128, 127, (4096, 4096), dtype=torch.int8, device='cuda:0')\n"
        "s = torch.rand(4096, dtype=torch.float16, device='cuda:0') * 0.01\n"
        "for M in (10, 12, 4):\n"
        "    x = torch.randn(M, 4096, dtype=torch.float16, device='cuda:0')\n"
        "    torch.cuda.synchronize()\n"
        "    g = torch.cuda.CUDAGraph()\n"
        "    with torch.cuda.graph(g):\n"
        "        y = ops.w8_a16_gemm(x, w, s)\n"
        "    g.replay(); torch.cuda.synchronize()\n"
        "    assert torch.equal(y, ops.w8_a16_gemm(x, w, s)) and float(y.float().abs().max()) > 0, M\n"
        "print('cold capture ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cold capture ok" in r.stdout, r.stderr[-2000:]
