"""One-launch decode step (eetq_rope_decode_attention_f16): rotary + KV-cache write + split-KV attention + chunk merge.

The two-launch pair it replaces (eetq_rotary_neox_kvcache_f16, eetq_decode_attention_f16) is checked against the oracle
(rotation: bit-exact) and a PyTorch fp32 reference (attention) in test_gpu_parity.py; here the one-launch form must give
the SAME BITS as that pair -- cache rows, output, token counter -- and is checked against the fp32 reference as well."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import eetq_amd.ops as o
    return o


def _table(D, rows=4096):
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.einsum("i,j->ij", torch.arange(rows).float(), inv)
    return torch.cat([fr.cos(), fr.sin()], -1).half().to(DEV)


def _views(qkv, H, Hkv, D):
    q = qkv[..., : H * D].unflatten(-1, (H, D))[:, 0]
    k = qkv[..., H * D: (H + Hkv) * D].unflatten(-1, (Hkv, D))[:, 0]
    v = qkv[..., (H + Hkv) * D:].unflatten(-1, (Hkv, D))[:, 0]
    return q, k, v


def _reference(q_rot, kc, vc, n, scale, mask=None):
    """fp32 softmax(scale q k^T + mask) v over the first n cache rows."""
    H, Hkv = q_rot.shape[1], kc.shape[1]
    kk = kc[:, :, :n].float().repeat_interleave(H // Hkv, dim=1)
    vv = vc[:, :, :n].float().repeat_interleave(H // Hkv, dim=1)
    s = torch.einsum("bhd,bhsd->bhs", q_rot.float(), kk) * scale
    if mask is not None:
        s = s + mask[..., :n].float()[:, None, :]
    return torch.einsum("bhs,bhsd->bhd", torch.softmax(s, -1), vv)


@pytest.mark.parametrize("B,H,Hkv,S,D,filled", [(1, 40, 40, 1100, 128, 1024), (2, 8, 2, 90, 64, 37), (3, 4, 4, 64, 128, 0),
                                               (1, 32, 8, 300, 128, 299), (2, 6, 3, 17, 64, 5)])
def test_one_launch_equals_two_launches(ops, B, H, Hkv, S, D, filled):
    torch.manual_seed(B * 100 + S)
    table = _table(D)
    scale = D ** -0.5
    row = (H + 2 * Hkv) * D
    kc0 = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    vc0 = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    kc0[:, :, filled:] = 250.0     # stale tail: attending it, or reading the new row's OLD contents, would show
    vc0[:, :, filled:] = -400.0
    tickets = torch.zeros(B * H + 1, dtype=torch.int32, device=DEV)
    mask = torch.zeros(B, S, dtype=torch.float16, device=DEV)
    mask[:, 1:filled:3] = float("-inf")
    for m in (None, mask):
        for splits in (None, 1, min(S, 5)):
            kc_a, vc_a, kc_b, vc_b = kc0.clone(), vc0.clone(), kc0.clone(), vc0.clone()
            cnt_a = torch.tensor(filled, dtype=torch.int64, device=DEV)
            cnt_b = cnt_a.clone()
            steps = min(4, S - filled)
            for step in range(steps):
                qkv = torch.randn(B, 1, row, dtype=torch.float16, device=DEV)
                pos = torch.full((B,), filled + step, dtype=torch.int64, device=DEV) - torch.arange(B, device=DEV)
                pos = pos.clamp_(min=0)   # rows of a left-padded batch: position < slot
                # two launches (q rotated in place: work on a copy)
                qkv_a = qkv.clone()
                q, k, v = _views(qkv_a, H, Hkv, D)
                ops.rotary_embedding_neox_kvcache(pos, q, k, v, D, table, kc_a, vc_a, slots=cnt_a)
                out_a = ops.decode_attention(q, kc_a, vc_a, mask=m, scaling=scale, splits=splits, kv_len=cnt_a, kv_len_bias=1,
                                             advance=cnt_a)
                # one launch
                q2, k2, v2 = _views(qkv, H, Hkv, D)
                keep = qkv.clone()
                out_b = ops.rope_decode_attention(pos, q2, k2, v2, table, kc_b, vc_b, tickets, slots=cnt_b, mask=m,
                                                  scaling=scale, splits=splits, kv_len=cnt_b, kv_len_bias=1, advance=cnt_b)
                assert torch.equal(qkv, keep), "the one-launch form must not write q back"
                assert torch.equal(out_a, out_b), (m is not None, splits, step)
                assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)
                assert int(cnt_a.item()) == int(cnt_b.item()) == filled + step + 1
                assert torch.count_nonzero(tickets) == 0, "tickets must be zero between launches"
                ref = _reference(q, kc_a, vc_a, filled + step + 1, scale, m)
                assert (out_b.float() - ref).abs().max().item() < 2e-3
                assert torch.isfinite(out_b).all()


def test_per_row_slots_full_cache_and_errors(ops):
    """Per-row slots (no shared counter), a slot outside the cache (nothing written, nothing rotated in: the valid rows are
    still attended), refusal of partial rotation tables and of undersized ticket buffers."""
    torch.manual_seed(5)
    B, H, Hkv, S, D = 3, 8, 4, 48, 64
    table = _table(D, 128)
    qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, dtype=torch.float16, device=DEV)
    kc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    vc = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    tickets = torch.zeros(B * H + 1, dtype=torch.int32, device=DEV)
    pos = torch.tensor([5, 9, 2], device=DEV)
    slots = torch.tensor([11, 40, S + 3], device=DEV)     # row 2: outside the cache
    n = torch.tensor(S, dtype=torch.int64, device=DEV)
    kc_a, vc_a, kc_b, vc_b = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    qkv_a = qkv.clone()
    q, k, v = _views(qkv_a, H, Hkv, D)
    ops.decode_dropped_steps(reset=True)
    ops.rotary_embedding_neox_kvcache(pos, q, k, v, D, table, kc_a, vc_a, slots=slots)
    assert ops.decode_dropped_steps(reset=True) == 1          # batch row 2's token had no cache row: counted, not silent
    out_a = ops.decode_attention(q, kc_a, vc_a, kv_len=n)
    q2, k2, v2 = _views(qkv, H, Hkv, D)
    out_b = ops.rope_decode_attention(pos, q2, k2, v2, table, kc_b, vc_b, tickets, slots=slots, kv_len=n)
    assert ops.decode_dropped_steps(reset=True) == 1          # the one-launch form counts the same event once
    assert torch.equal(out_a, out_b) and torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)
    assert torch.equal(kc_b[2], kc[2]) and not torch.equal(kc_b[0], kc[0])
    assert int(n.item()) == S and torch.count_nonzero(tickets) == 0
    with pytest.raises(RuntimeError):   # partial rotation is the two-launch form's job
        ops.rope_decode_attention(pos, q2, k2, v2, table[:, : D // 2].contiguous(), kc_b, vc_b, tickets, slots=slots)
    with pytest.raises(RuntimeError):
        ops.rope_decode_attention(pos, q2, k2, v2, table, kc_b, vc_b, tickets[: B * H], slots=slots)
    with pytest.raises(RuntimeError):
        ops.rope_decode_attention(pos, q2, k2, v2, table, kc_b, vc_b, tickets.long(), slots=slots)


def test_many_steps_under_graph_replay(ops):
    """The launch is capturable (no host synchronisation, tickets self-resetting): 40 replays of one captured step walk the
    counter and keep matching the eager two-launch pair run on a twin cache."""
    torch.manual_seed(9)
    B, H, Hkv, S, D = 2, 16, 4, 160, 128
    table = _table(D, 256)
    scale = D ** -0.5
    row = (H + 2 * Hkv) * D
    kc_a = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    vc_a = torch.randn(B, Hkv, S, D, dtype=torch.float16, device=DEV)
    kc_b, vc_b = kc_a.clone(), vc_a.clone()
    cnt_a = torch.tensor(100, dtype=torch.int64, device=DEV)
    cnt_b = cnt_a.clone()
    tickets = torch.zeros(B * H + 1, dtype=torch.int32, device=DEV)
    qkv = torch.randn(B, 1, row, dtype=torch.float16, device=DEV)
    pos = torch.zeros(B, dtype=torch.int64, device=DEV)
    q2, k2, v2 = _views(qkv, H, Hkv, D)
    out_b = torch.empty(B, H, D, dtype=torch.float16, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out_b.copy_(ops.rope_decode_attention(pos, q2, k2, v2, table, kc_b, vc_b, tickets, slots=cnt_b, scaling=scale,
                                                  kv_len=cnt_b, kv_len_bias=1, advance=cnt_b))
    torch.cuda.current_stream().wait_stream(side)
    for step in range(40):
        qkv.copy_(torch.randn(B, 1, row, dtype=torch.float16, device=DEV))
        pos.fill_(100 + step)
        qkv_a = qkv.clone()
        q, k, v = _views(qkv_a, H, Hkv, D)
        ops.rotary_embedding_neox_kvcache(pos, q, k, v, D, table, kc_a, vc_a, slots=cnt_a)
        out_a = ops.decode_attention(q, kc_a, vc_a, scaling=scale, kv_len=cnt_a, kv_len_bias=1, advance=cnt_a)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_a, out_b), step
        assert int(cnt_b.item()) == 101 + step
    assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b) and torch.count_nonzero(tickets) == 0


def test_greedy_handover_is_torch_argmax_plus_bookkeeping():
    """eetq_greedy_handover_f16: argmax (first index on ties, NaN = maximum, -0 == +0, all -inf rows, unaligned and odd-length rows)
    written to the output column and the next-token tensor, position and column advanced -- against torch.argmax + the four
    torch launches it replaces; a column outside the output buffer writes nothing there but still hands the token on."""
    import eetq_amd.ops as ops
    torch.manual_seed(5)
    for B, V in ((1, 32000), (4, 32000), (3, 1001), (2, 7), (1, 128256)):
        lg = (torch.randn(B, V, device=DEV) * 3).half()
        if V > 100:
            lg[0, 17] = lg[0, 90] = lg[0].max() + 1          # a tie: the first index wins
            if B > 1:
                lg[1, 5] = float("nan")                       # NaN is the maximum ...
                lg[1, 3] = float("inf")                       # ... even next to +inf
            if B > 2:
                lg[2] = float("-inf")                         # nothing but -inf: index 0
        else:
            lg[0] = 0.0
            lg[0, 2] = -0.0                                   # -0 == +0: index 0
        ref = lg.argmax(-1)
        out = torch.full((B, 6), -7, dtype=torch.int64, device=DEV)
        col = torch.tensor([[2]], dtype=torch.int64, device=DEV)
        tok = torch.full((B, 1), -1, dtype=torch.int64, device=DEV)
        pos = torch.tensor([40], dtype=torch.int64, device=DEV)
        ops.greedy_handover(lg, out, col, tok, pos)
        assert torch.equal(tok[:, 0], ref) and torch.equal(out[:, 2], ref)
        assert (out[:, [0, 1, 3, 4, 5]] == -7).all() and int(col) == 3 and int(pos) == 41
        view = lg[:, 1:]                                       # rows that are not 16-byte aligned
        ops.greedy_handover(view, out, col, tok, pos)
        assert torch.equal(tok[:, 0], view.argmax(-1)) and torch.equal(out[:, 3], view.argmax(-1)) and int(col) == 4
        col.fill_(6)                                           # beyond the buffer: no write, the token is still handed on
        before = out.clone()
        ops.greedy_handover(lg, out, col, tok, pos)
        assert torch.equal(out, before) and torch.equal(tok[:, 0], ref) and int(col) == 7 and int(pos) == 43
    with pytest.raises(RuntimeError):
        ops.greedy_handover(lg.float(), out, col, tok, pos)
