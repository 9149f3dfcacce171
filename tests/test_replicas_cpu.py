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


# ---- the 8-GPU line the driver launches, as far as it can run without GPUs ------------------------------------------------
# `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...`
# Eight ranks here, with torch.cuda's device calls stubbed (there is no GPU) and gloo carrying the collectives: device
# selection by LOCAL_RANK, the fan-out, bench.py's own timing helpers (graph_length / timed_replays: every rank must time the
# same number of replays; the result is the MAX over ranks), the checksum agreement, ONE JSON line from rank 0, and eight
# ranks arriving together at the build lock of libeetq_amd.so.
WORKER8 = textwrap.dedent("""
    import json, os, sys, time, torch
    sys.path.insert(0, %(root)r)
    picked = []
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 8
    torch.cuda.set_device = lambda d: picked.append(torch.device(d).index)
    torch.cuda.synchronize = lambda *a, **k: None
    from eetq_amd.utils.replicas import ReplicaGroup
    import bench
    g = ReplicaGroup()                        # device from LOCAL_RANK, backend from EETQ_REPLICA_BACKEND=gloo
    assert g.world_size == 8 and g.device == torch.device("cuda", g.local_rank) and picked == [g.local_rank], (g.device, picked)
    assert g.backend == "gloo" and g.collective_device == torch.device("cpu")
    x = torch.full((1, 64), float(g.rank))    # rank 0 draws the activations, everyone receives them
    g.fan_out(x)
    assert torch.equal(x, torch.zeros(1, 64))

    class FakeGraph:                          # stands for a captured HIP graph of `graph_len` launches; slower on high ranks
        replays = 0
        def replay(self):
            FakeGraph.replays += 1
            time.sleep(0.002 * (1 + g.rank %% 3))
    steps, nbuf = 20, 40
    graph_len = bench.graph_length(steps, nbuf)
    assert graph_len == 1000
    seconds, replays = bench.timed_replays(g, [FakeGraph()], steps, 0.05)
    everyone = [None] * 8
    torch.distributed.all_gather_object(everyone, (seconds, replays, FakeGraph.replays))
    assert len(set(e[0] for e in everyone)) == 1, everyone      # the same MAX-over-ranks time on every rank
    assert len(set(e[1] for e in everyone)) == 1, everyone      # ... for the same number of replays
    assert seconds >= 0.05 and seconds >= replays * 0.006 * 0.9  # the slowest rank (6 ms per replay) sets the time
    crcs = g.gather_checksums(x.half())
    assert len(crcs) == 8 and len(set(crcs)) == 1
    assert len(set(g.gather_checksums(x.half() + (g.rank == 5)))) == 2
    if g.rank == 0:
        print(json.dumps({"n_gpus": g.world_size, "timed_steps": replays * graph_len, "ms_per_step": seconds * 1e3 / (replays * graph_len)}))
    g.close()
""")


def test_eight_replicas_gloo_run_the_bench_helpers(tmp_path):
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % {"root": ROOT})
    port = str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", port, str(script)]
    env = dict(os.environ, EETQ_REPLICA_BACKEND="gloo", OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                            # rank 0 prints ONE line
    import json
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 8 and doc["timed_steps"] % 1000 == 0


LOCK_WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %(root)r)
    from eetq_amd import _lib
    marker, log = %(marker)r, %(log)r
    real_path = _lib.LIB_PATH
    _lib._sources_newer_than_lib = lambda: not os.path.exists(marker)     # "stale" until somebody has built
    def fake_build(force=False, verbose=False):
        with open(log, "a") as f:
            f.write("%%d\\n" %% os.getpid())
        time.sleep(0.5)                                                    # a build takes a while: the others must wait
        open(marker, "w").close()
    _lib.build = fake_build
    L = _lib.lib()
    assert L.eetq_abi_version() >= 2
    print("loaded", os.getpid())
""")


def test_eight_ranks_racing_on_the_build_lock(tmp_path):
    """torchrun starts the ranks of a node together; with a stale library every one of them finds `_needs_build()` true.  The
    flock in _lib.lib() lets ONE build while the others wait and re-check; nobody loads a half-written .so."""
    marker, log = str(tmp_path / "built"), str(tmp_path / "builds.log")
    script = tmp_path / "lock_worker.py"
    script.write_text(LOCK_WORKER % {"root": ROOT, "marker": marker, "log": log})
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for _ in range(8)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0 and "loaded" in out, out[-2000:]
    assert len(open(log).read().split()) == 1                     # exactly one rank built


def test_collectives_use_device_tensors_under_rccl():
    """RCCL moves device memory only: under backend "nccl" the checksum all-gather and the MAX all-reduce must be handed
    tensors on the replica's GPU (a CPU tensor raises inside RCCL on the 8-GPU node, where this cannot be tried first)."""
    import torch
    from eetq_amd.utils.replicas import ReplicaGroup
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "EETQ_REPLICA_BACKEND") if k in os.environ}
    try:
        g = ReplicaGroup(backend="nccl", device="cpu")            # world_size 1: no process group is created
        g.device = torch.device("cuda", 5)                        # what rank 5 of an 8-GPU node holds
        assert g.collective_device == torch.device("cuda", 5)
        g.backend = "gloo"
        assert g.collective_device == torch.device("cpu")
    finally:
        os.environ.update(env)
    src = open(os.path.join(ROOT, "eetq_amd", "utils", "replicas.py")).read()
    assert src.count("device=self.collective_device") == 2        # both collective inputs are built there


# RCCL that cannot come up (a broken IPC / xGMI setup on the node) must not cost the run its number: the group falls back to
# gloo on every rank, says so, and the three collectives still work.  "nccl" does not exist on this CPU box, which is exactly
# the failure: init_process_group("nccl", device_id=...) raises on both ranks.
WORKER_FALLBACK = textwrap.dedent("""
    import sys, torch
    sys.path.insert(0, %r)
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a, **k: None
    from eetq_amd.utils.replicas import ReplicaGroup
    g = ReplicaGroup(backend="nccl", device="cuda:0")
    assert g.backend == "gloo" and g.backend_note.startswith("gloo (RCCL init failed"), (g.backend, g.backend_note)
    assert g.collective_device.type == "cpu"
    x = torch.full((3,), float(g.rank + 5))
    g.fan_out(x)
    assert torch.equal(x, torch.full((3,), 5.0))
    assert g.max_over_ranks(float(g.rank)) == 1.0
    g.close()
    print("rank", g.rank, "ok")
""")


def test_rccl_failure_falls_back_to_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER_FALLBACK % ROOT)
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
        assert "falls back to gloo" in out
