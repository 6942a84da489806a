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
    """callable(flat_grad): in-place mean over ranks.  The CelebA trainer hands it one bucket per layer group as soon as the group's
    weight gradients are complete (celeba.py BUCKETS: 5 buckets of 0.2-33 MB per network), the small-network trainers one arena per
    backward pass -- few, large messages, what per-link-bound xGMI rings want.

    ``wire="bf16"``: the bucket crosses the links as bf16 (half the bytes; the mean is taken on the bf16 values, the result is
    widened back into the fp32 arena).  Changes the gradients by bf16 rounding (rel. 2^-9 per element): off by default, meant
    for the 16-bit compute modes whose gradients carry that rounding already."""

    def __init__(self, world: int, group=None, force: bool = False, wire: str = "f32"):
        assert wire in ("f32", "bf16"), wire
        self.world, self.group, self.force, self.wire = world, group, force, wire
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self._stage = {}

    def _staged(self, flat):
        key = (flat.data_ptr(), flat.numel())
        st = self._stage.get(key)
        if st is None:
            st = self._stage[key] = torch.empty(flat.numel(), device=flat.device, dtype=torch.bfloat16)
        st.copy_(flat)
        return st

    def _launch(self, buf, async_op):
        if self.backend == "nccl":
            return dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _settle(self, flat, buf):
        """after the collective: scale the SUM of the backends without AVG, widen the bf16 wire buffer back"""
        if buf is not flat:
            flat.copy_(buf)
        if self.backend != "nccl":
            flat.mul_(1.0 / self.world)

    def __call__(self, flat: torch.Tensor):
        if self.world <= 1 and not self.force:
            return
        buf = self._staged(flat) if self.wire == "bf16" else flat
        self._launch(buf, False)
        self._settle(flat, buf)

    # asynchronous form: the collective runs on RCCL's own stream; compute enqueued after start() overlaps with it and
    # finish() makes the compute stream wait (works eagerly and under hipGraph capture: the dependency is an event edge)
    def start(self, flat: torch.Tensor):
        if self.world <= 1 and not self.force:
            return None
        buf = self._staged(flat) if self.wire == "bf16" else flat
        return (self._launch(buf, True), flat, buf)

    def finish(self, handle):
        if handle is None:
            return
        work, flat, buf = handle
        work.wait()
        self._settle(flat, buf)


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
