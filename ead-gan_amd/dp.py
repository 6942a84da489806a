"""Plain data parallelism for the train step: one process per GPU, RCCL (torch.distributed backend "nccl") all-reduce
of the flat fp32 gradient arenas after each of the three backward passes.  Samples are independent except through
gradient averaging; BatchNorm statistics stay per-rank (documented mode), spectral-norm u/v need no communication
(functions of the replicated weights only)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: EG_DIST_BACKEND=gloo EG_SHARE_GPU=1 runs N ranks on device 0 over gloo (RCCL refuses
    # two ranks on one device); the driver's N-GPU runs never set these
    backend = backend or os.environ.get("EG_DIST_BACKEND") or None
    if os.environ.get("EG_SHARE_GPU"):
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradAllReduce:
    """callable(flat_grad): in-place mean over ranks.  One collective per arena per backward pass (the arenas are
    58 MB / 45 MB fp32 on CelebA: large, few messages -- what per-link-bound xGMI rings want)."""

    def __init__(self, world: int, group=None, force: bool = False):
        self.world, self.group, self.force = world, group, force
        self.backend = dist.get_backend(group) if dist.is_initialized() else None

    def __call__(self, flat: torch.Tensor):
        if self.world <= 1 and not self.force:
            return
        if self.backend == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.mul_(1.0 / self.world)


    # asynchronous form: the collective runs on RCCL's own stream; compute enqueued after start() overlaps with it and
    # finish() makes the compute stream wait (works eagerly and under hipGraph capture: the dependency is an event edge)
    def start(self, flat: torch.Tensor):
        if self.world <= 1 and not self.force:
            return None
        if self.backend == "nccl":
            return (dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None)
        return (dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat)

    def finish(self, handle):
        if handle is None:
            return
        work, flat = handle
        work.wait()
        if flat is not None:
            flat.mul_(1.0 / self.world)


class SyncBN:
    """Collectives of synchronised BatchNorm (statistics over the global batch, so that N ranks x B/N images reproduce one rank
    x B images): per BatchNorm layer one exchange of 3*C floats in the forward and 2*C floats in the backward.  The gather is a
    slot-wise all-reduce (every rank writes its block into a zeroed [world][3*C] buffer): works on RCCL and gloo alike and is
    capturable into the step's hipGraph."""

    def __init__(self, world: int, rank: int, group=None):
        self.world, self.rank, self.group = world, rank, group
        self._buf = {}

    def gather_stats(self, stats: torch.Tensor) -> torch.Tensor:
        """stats: [3*C] of this rank -> [world, 3*C] of all ranks"""
        n = stats.numel()
        buf = self._buf.get((n, stats.device))
        if buf is None:
            buf = self._buf[(n, stats.device)] = torch.zeros(self.world, n, device=stats.device, dtype=torch.float32)
        buf.zero_()
        buf[self.rank].copy_(stats)
        if self.world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return buf

    def reduce_sums(self, sums: torch.Tensor) -> torch.Tensor:
        """in-place sum over ranks of the [2*C] backward sums"""
        if self.world > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        return sums


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
