"""Replica fan-out for multi-GPU runs: one process per GPU, model replicated, no data-path collective.

The reference has no communication layer at all (zero NCCL call sites); BASELINE.json's
north star asks for "RCCL over xGMI used only to fan out identical prompts".  So the only collectives are
  * one ``broadcast`` of the input batch (prompts / activations) from rank 0,
  * one ``all_gather`` of a per-rank checksum + one ``all_reduce(MAX)`` of the timed-region duration,
all latency-bound and outside the kernels.  Backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU (tests).
"""
import os
import time
import zlib

import torch
import torch.distributed as dist

__all__ = ["ReplicaGroup"]


class ReplicaGroup:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        use_cuda = torch.cuda.is_available() if device is None else torch.device(device).type == "cuda"
        if device is None:
            # one replica per GPU; with fewer visible GPUs than local ranks (a test box) the ranks wrap around and share
            device = torch.device("cuda", self.local_rank % max(1, torch.cuda.device_count())) if use_cuda else torch.device("cpu")
        self.device = torch.device(device)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        # EETQ_REPLICA_BACKEND=gloo: a test hook -- two replicas on ONE GPU (RCCL refuses two ranks per device) still run the
        # whole N > 1 path of bench.py, with gloo carrying the three latency-bound collectives
        self.backend = backend or os.environ.get("EETQ_REPLICA_BACKEND") or ("nccl" if self.device.type == "cuda" else "gloo")
        self._own_pg = False
        self.backend_note = None  # set when RCCL could not be brought up and gloo carries the collectives instead
        if self.world_size > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            self._init_group()
            self._own_pg = True

    def _init_group(self):
        """RCCL first (device_id makes the communicator come up inside init_process_group, on every rank at once, so a
        failure shows here and on all of them); if that raises, the three collectives -- all of them latency-bound and outside
        the kernels -- are carried by gloo through host memory and `backend_note` says so: N replicas still produce a number.
        EETQ_REPLICA_STRICT=1 keeps the exception."""
        if self.backend != "nccl":
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world_size)
            return
        try:
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world_size, device_id=self.device)
            probe = torch.ones(1, device=self.device)
            dist.all_reduce(probe)  # first collective on the communicator: a broken xGMI / IPC setup fails here, not mid-bench
            torch.cuda.synchronize(self.device)
            if int(probe.item()) != self.world_size:
                raise RuntimeError("RCCL all_reduce probe returned %r for %d ranks" % (probe.item(), self.world_size))
        except Exception as e:  # noqa: BLE001 -- whatever RCCL raised, the replicas themselves do not need it
            if os.environ.get("EETQ_REPLICA_STRICT") == "1":
                raise
            import sys
            print("[eetq_amd] rank %d: RCCL process group failed (%s: %s); fan-out falls back to gloo" %
                  (self.rank, type(e).__name__, str(e).splitlines()[0] if str(e) else ""), file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            self.backend = "gloo"
            self.backend_note = "gloo (RCCL init failed: %s)" % type(e).__name__
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world_size)

    @property
    def collective_device(self):
        """Where the tensors handed to the collectives live: RCCL ("nccl") only moves device memory, so they sit on this
        replica's GPU; gloo moves host memory."""
        return self.device if self.backend == "nccl" else torch.device("cpu")

    def synchronize(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    # ---- collectives (no-ops for a single replica) ----
    def fan_out(self, tensor, src=0):
        """Every replica receives rank ``src``'s tensor (identical prompts / activations)."""
        if self.world_size > 1:
            dist.broadcast(tensor, src=src)
        return tensor

    def barrier(self):
        if self.world_size > 1:
            dist.barrier()
        self.synchronize()

    def max_over_ranks(self, seconds):
        if self.world_size == 1:
            return float(seconds)
        t = torch.tensor([seconds], dtype=torch.float64, device=self.collective_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn):
        """barrier + sync, run fn(), sync + barrier; returns the MAX wall time over all replicas (seconds)."""
        self.barrier()
        t0 = time.perf_counter()
        fn()
        self.synchronize()
        t1 = time.perf_counter()
        local = t1 - t0
        self.barrier()
        return self.max_over_ranks(local)

    def gather_checksums(self, tensor):
        """CRC32 of every replica's result bytes (list of world_size ints): replicas must agree bit for bit."""
        crc = zlib.crc32(tensor.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()) & 0xFFFFFFFF
        if self.world_size == 1:
            return [crc]
        t = torch.tensor([crc], dtype=torch.int64, device=self.collective_device)
        out = [torch.zeros_like(t) for _ in range(self.world_size)]
        dist.all_gather(out, t)
        return [int(o.item()) for o in out]

    def close(self):
        if self._own_pg and dist.is_initialized():
            dist.destroy_process_group()
            self._own_pg = False
