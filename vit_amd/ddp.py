"""Data-parallel gradient reduction for the encoder: one process per GPU, clips sharded by rank.

The reference has no multi-GPU path; clips are independent through the encoder, so ranks exchange
nothing but parameter gradients.  The encoder backward hands over one flat f32 gradient buffer per
transformer block as soon as that block's gradient kernels are enqueued (reverse layer order); each
buffer (~50 MB for a 1024-wide block) is all-reduced on a side stream through torch.distributed
(backend "nccl" = RCCL over xGMI) while the next block's backward runs on the compute stream.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of a global batch owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class BucketReducer:
    """Averages flat gradient buckets across ranks, overlapping with whatever the caller enqueues next."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, always_reduce: bool = False):
        """always_reduce: issue the collectives even in a world of one (exercises the stream hand-over on a
        single GPU; the result is unchanged)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.always_reduce = always_reduce
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._pending: List[Tuple[object, torch.Tensor]] = []
        self.buckets_reduced = 0
        self.bytes_reduced = 0

    def submit(self, flat: torch.Tensor) -> None:
        """Start the all-reduce of one bucket; its producer kernels are already on the current stream."""
        if self.world == 1 and not self.always_reduce:
            return
        native_avg = self.backend == "nccl"
        op = dist.ReduceOp.AVG if native_avg else dist.ReduceOp.SUM
        if flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=flat.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
            flat.record_stream(self._comm_stream)
        else:
            work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self._pending.append((work, flat, native_avg))
        self.buckets_reduced += 1
        self.bytes_reduced += flat.numel() * flat.element_size()

    def finish(self) -> None:
        """Make the current stream wait for every outstanding bucket (and finish the averaging)."""
        for work, flat, native_avg in self._pending:
            work.wait()
            if not native_avg:
                flat.mul_(1.0 / self.world)
        self._pending.clear()


def all_gather_with_grad(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Concatenate `x` (B_local, ...) over ranks along dim 0 with a gradient: the local slice of the
    incoming gradient flows back.  Used for the in-batch-negative InfoNCE term of the stage-1 objective
    (the projections are (B,128): tiny), so that N ranks contrast over the same global batch one rank would."""
    return _AllGather.apply(x, group)


class _AllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        world = dist.get_world_size(group)
        ctx.rank = dist.get_rank(group)
        ctx.n = x.shape[0]
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous(), group=group)
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None
