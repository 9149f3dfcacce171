"""Grouped decode GEMV (eetq_w8a16_gemv_grouped / ops.w8_a16_gemv_grouped): independent M = 1 problems in one dispatch per
group of equal K.  Tier A against the oracle and against the separate launches everywhere; bit-identical to a separate launch
wherever both take the same kernel body (round 4: dispatches with more than two tile rows per CU run the 8-wave body, like single
launches with that many rows; small dispatches the 16-wave bodies).  Reference numerics: weightOnlyBatchedGemv/kernel.h:294-468,
argument checks: kernelLauncher.cu:122-232."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import eetq_amd.ops as _ops
    return _ops


def _tier_a(y, ref):
    y = y.astype(np.float32)
    ref = ref.astype(np.float32)
    return np.abs(y - ref) <= 1e-3 * np.abs(ref).max() + 2e-3 * np.abs(ref)


def _problem(ops, oracle, K, N, seed, with_oracle=True):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    x = rng.random((1, K)).astype(np.float16)
    qw, s = ops.quant_weights(torch.from_numpy(w).to(DEV), torch.int8, False)
    ref = None
    if with_oracle:
        q, sc = oracle.quantize(w)
        ref = oracle.w8a16_gemm(x, q, sc)
    return torch.from_numpy(x).to(DEV), qw, s, ref


def test_grouped_equals_separate_launches_7b_shapes(ops, oracle):
    """q / k / v / o (4096 x 4096) and gate / up (4096 x 11008) of a Llama-2-7B layer: two groups by K -- here all K = 4096,
    one dispatch.  Bit-identical to six separate calls, with bias and residual on some of them; tier A vs the oracle."""
    shapes = [(4096, 4096)] * 4 + [(4096, 11008)] * 2
    probs = [_problem(ops, oracle, K, N, 100 + i) for i, (K, N) in enumerate(shapes)]
    xs, ws, ss = [p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs]
    torch.manual_seed(0)
    biases = [None, torch.randn(4096, dtype=torch.float16, device=DEV), None, None, None,
              torch.randn(11008, dtype=torch.float16, device=DEV)]
    residuals = [None, None, torch.randn(1, 4096, dtype=torch.float16, device=DEV), None,
                 torch.randn(1, 11008, dtype=torch.float16, device=DEV), None]
    outs = ops.w8_a16_gemv_grouped(xs, ws, ss, biases, residuals)
    assert len(outs) == 6
    for i, (x, w, s, ref) in enumerate(probs):
        single = ops.w8_a16_gemm(x, w, s, bias=biases[i], residual=residuals[i])
        assert outs[i].shape == single.shape and _tier_a(outs[i].cpu().numpy(), single.cpu().numpy()).all(), i
        if shapes[i] == (4096, 11008):      # 688 tile rows: the separate launch runs the 8-wave body too (2400 rows in the dispatch)
            assert torch.equal(outs[i], single), i
        if biases[i] is None and residuals[i] is None:
            assert _tier_a(outs[i].cpu().numpy(), ref).all(), i
    assert all(torch.equal(a, b) for a, b in zip(outs, ops.w8_a16_gemv_grouped(xs, ws, ss, biases, residuals)))   # call to call
    # a SMALL dispatch (two 4096 x 4096 problems = 512 tile rows = two per CU) keeps the 16-wave straight-line body of the
    # separate 4096 x 4096 launch: bit-identical
    two = ops.w8_a16_gemv_grouped(xs[:2], ws[:2], ss[:2], biases[:2], residuals[:2])
    for i in range(2):
        assert torch.equal(two[i], ops.w8_a16_gemm(xs[i], ws[i], ss[i], bias=biases[i], residual=residuals[i])), i
    # without the optional lists
    plain = ops.w8_a16_gemv_grouped(xs, ws, ss)
    for i, (x, w, s, ref) in enumerate(probs):
        assert _tier_a(plain[i].cpu().numpy(), ops.w8_a16_gemm(x, w, s).cpu().numpy()).all() and _tier_a(plain[i].cpu().numpy(), ref).all()


def test_grouped_mixed_k_13b_shapes_and_fallback(ops, oracle):
    """Llama-2-13B shapes (K = 5120 and K = 13824: two groups), plus a K the grouped kernel does not take (1024: launched
    alone through the ordinary dispatcher), in one call and in scrambled order."""
    shapes = [(5120, 5120), (13824, 5120), (5120, 13824), (1024, 512), (5120, 5120), (13824, 5120)]
    probs = [_problem(ops, oracle, K, N, 200 + i) for i, (K, N) in enumerate(shapes)]
    outs = ops.w8_a16_gemv_grouped([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
    for i, (x, w, s, ref) in enumerate(probs):
        assert _tier_a(outs[i].cpu().numpy(), ref).all(), (i, shapes[i])
        single = ops.w8_a16_gemm(x, w, s)
        assert _tier_a(outs[i].cpu().numpy(), single.cpu().numpy()).all()
        if shapes[i] == (1024, 512):      # the fallback inside the grouped call IS the single launch
            assert torch.equal(outs[i], single), shapes[i]
        # (5120 x 13824: since round 4 a single launch with more than two tile rows per CU runs 8-wave workgroups, the grouped
        # kernel 16-wave ones -- same products, another summation order: tier A above, no bit identity)


def test_grouped_more_than_one_dispatch_and_graph_replay(ops, oracle):
    """40 problems of one K (32 per dispatch -> two dispatches), captured in a HIP graph and replayed on new inputs."""
    K, N = 2048, 4096     # 256 tile rows per problem, 8192 / 2048 per dispatch: the 8-wave body (separate launches: 16 waves)
    base = [_problem(ops, oracle, K, N, 300 + i, with_oracle=(i % 8 == 0)) for i in range(40)]
    xs, ws, ss = [p[0].clone() for p in base], [p[1] for p in base], [p[2] for p in base]
    eager = ops.w8_a16_gemv_grouped(xs, ws, ss)
    for i, p in enumerate(base):
        assert _tier_a(eager[i].cpu().numpy(), ops.w8_a16_gemm(p[0], p[1], p[2]).cpu().numpy()).all(), i
        if p[3] is not None:
            assert _tier_a(eager[i].cpu().numpy(), p[3]).all()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.w8_a16_gemv_grouped(xs, ws, ss)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        captured = ops.w8_a16_gemv_grouped(xs, ws, ss)
    for x in xs:
        x.mul_(0.5)                                           # new activations in the same buffers
    g.replay()
    torch.cuda.synchronize()
    fresh = ops.w8_a16_gemv_grouped(xs, ws, ss)              # the same launches, eagerly, on the new activations
    for i in range(40):
        assert torch.equal(captured[i], fresh[i]), i
        assert _tier_a(captured[i].cpu().numpy(), ops.w8_a16_gemm(xs[i], ws[i], ss[i]).cpu().numpy()).all(), i


def test_grouped_argument_checks_and_ctypes_twin(ops):
    from eetq_amd import ops_ctypes as ct
    torch.manual_seed(4)
    w = (torch.rand(2048, 64, device=DEV) - 0.5).half()
    qw, s = ops.quant_weights(w, torch.int8, False)
    x = torch.rand(1, 2048, dtype=torch.float16, device=DEV)
    assert ops.w8_a16_gemv_grouped([], [], []) == []
    a = ops.w8_a16_gemv_grouped([x, x[0]], [qw, qw], [s, s])
    b = ct.w8_a16_gemv_grouped([x, x[0]], [qw, qw], [s, s])
    assert a[0].shape == (1, 64) and a[1].shape == (64,)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    for fn in (ops.w8_a16_gemv_grouped, ct.w8_a16_gemv_grouped):
        with pytest.raises(RuntimeError):
            fn([x], [qw, qw], [s])                               # list lengths
        with pytest.raises(RuntimeError):
            fn([x.float()], [qw], [s])                            # dtype
        with pytest.raises(RuntimeError):
            fn([torch.rand(2, 2048, dtype=torch.float16, device=DEV)], [qw], [s])   # more than one row
        with pytest.raises(RuntimeError):
            fn([x], [qw], [s], [torch.zeros(8, dtype=torch.float16, device=DEV)])   # bias length
        with pytest.raises(RuntimeError):
            fn([x.cpu()], [qw], [s])                              # host tensor
