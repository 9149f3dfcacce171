"""AUTO's regret on held-out shapes (round-4 verdict, item 4): on twelve (K, N, M) points of shapes the dispatch rules were NOT
read off -- Llama-3-8B / 70B, Qwen2-7B projections, 7168^2 -- and of the few-tile shapes this round's row-group plan serves,
EETQ_PATH_AUTO must be within 5 % of the best explicitly forced kernel path, timed in this process on the same rotating weights
(tools/auto_regret.py: graph-replayed chains; the whole table is profiles/r05_auto_regret_*.jsonl).  A point over the limit is
measured a second time before it fails (two chains of the same kernel differ by ~1 %, a busy box by more)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# (K, N, M, what AUTO is expected to take) -- every point had a margin of >= 5 % between AUTO's path and the runner-up
POINTS = [
    (4096, 28672, 64, "mfma"),           # wide N from M = 33: tiled kernel 35 us, split-K 43
    (3584, 18944, 48, "mfma"),           # 24 vs 34
    (4096, 6144, 24, "splitk"),          # 32-column blocks, all of K: 9.0 = the round-1 tile on the same decomposition (9.1)
    (14336, 4096, 128, "tilesplit"),     # deep K, few tiles: four K slices, 27.6 vs 30.0 (split-K) / 68 (unsplit)
    (8192, 10240, 32, "splitk"),         # 23.5 vs 28.8 (mid) / 31.5 (tiled)
    (8192, 10240, 96, "splitk"),         # three 32-row groups, two workgroups per CU: 34 vs 39 (tiled kernel)
    (7168, 7168, 128, "splitk"),         # two 64-row groups 25.2 vs 26.4 (K-sliced tiled kernel; it keeps K > 8192: point 4)
    (4096, 4096, 128, "splitk"),         # row groups: 12.5 vs 20.5 (tiled, K-sliced tiled)
    (4096, 4096, 256, "splitk"),         # row groups: 17.2 vs 20.7
    (4096, 6144, 8, "stream"),           # 6.6 vs 9.9
    (14336, 4096, 2, "stream"),          # 11.4 vs 12.2 (GEMV) / 14.1
    (28672, 8192, 8, "stream"),          # 37.1 vs 43.0
]


@pytest.mark.parametrize("K,N,M,want", POINTS)
def test_auto_is_within_5_percent_of_the_best_forced_path(K, N, M, want):
    import auto_regret
    L = max(2, int(400e6 // (K * N)))
    g = torch.Generator(device="cuda:0")
    g.manual_seed(K + N + M)
    ws = [torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda:0", generator=g) for _ in range(L)]
    s = torch.rand(N, dtype=torch.float16, device="cuda:0", generator=g) * 0.01
    row = auto_regret.measure(K, N, M, ws, s, 0.012)
    assert row["auto_path"].split("/")[0] == want, row
    if row["regret"] > 0.05:
        again = auto_regret.measure(K, N, M, ws, s, 0.03)
        row = again if again["regret"] < row["regret"] else row
    assert row["regret"] <= 0.05, row
