"""world_size-2 gloo test of the replica fan-out used by bench.py --gpus N (runs on CPU)."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import sys, time, torch
    sys.path.insert(0, %r)
    from eetq_amd.utils.replicas import ReplicaGroup
    g = ReplicaGroup(backend="gloo", device="cpu")
    assert g.world_size == 2
    x = torch.full((4, 8), float(g.rank + 1))
    g.fan_out(x)                                  # identical prompts on every replica
    assert torch.equal(x, torch.ones(4, 8))
    secs = g.timed(lambda: time.sleep(0.05 * (g.rank + 1)))
    assert 0.09 < secs < 1.0, secs               # MAX over ranks (rank 1 sleeps 0.1 s)
    y = (x * 3).half()
    crcs = g.gather_checksums(y)
    assert len(crcs) == 2 and crcs[0] == crcs[1]
    crcs = g.gather_checksums(y + g.rank)
    assert crcs[0] != crcs[1]
    g.close()
    print("rank", g.rank, "ok")
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_replicas_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


def test_single_replica_is_a_noop():
    import torch
    from eetq_amd.utils.replicas import ReplicaGroup
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        g = ReplicaGroup(backend="gloo", device="cpu")
        t = torch.arange(4.0)
        assert g.fan_out(t) is t and g.world_size == 1
        assert g.timed(lambda: None) >= 0
        assert len(g.gather_checksums(t)) == 1
    finally:
        os.environ.update(env)
