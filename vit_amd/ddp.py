"""Data-parallel gradient reduction for the encoder: one process per GPU, clips sharded by rank.

The reference has no multi-GPU path; clips are independent through the encoder, so ranks exchange
nothing but parameter gradients.  The encoder backward hands over one flat f32 gradient buffer per
transformer block as soon as that block's gradient kernels are enqueued (reverse layer order); each
buffer (~50 MB for a 1024-wide block) is all-reduced on a side stream through torch.distributed
(backend "nccl" = RCCL over xGMI) while the next block's backward runs on the compute stream.

Backends other than RCCL (gloo: CPU tests, and rehearsals with several ranks sharing one GPU) do not get
device buffers handed to them: a bucket on the device is copied to host memory, reduced there and copied
back, one bucket at a time, when `finish()` joins.  That path has no overlap and needs none -- it exists to
check semantics, not to be fast.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of a global batch owned by `rank`; sizes differ by at most one.
    With unequal shards the per-rank mean losses must be weighted by `shard_weight` before backward, because the
    reducer averages gradients over ranks, not over clips."""
    base, rem = divmod(global_batch, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_weight(local_batch: int, global_batch: int, world: int) -> float:
    """Factor for a rank's mean-over-local-clips loss so that the rank-AVERAGED gradient equals the gradient of the
    mean over the global batch: (B_local / B_global) * world; 1.0 for equal shards."""
    return local_batch * world / float(global_batch)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Make every rank start from rank `src`'s parameters and buffers (replicas must be identical before the first
    step).  RCCL broadcasts device tensors in place; other backends go through host memory."""
    native = dist.get_backend(group) == "nccl"
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if native or not t.is_cuda:
                dist.broadcast(t.data, src=src, group=group)
            else:
                host = t.data.cpu()
                dist.broadcast(host, src=src, group=group)
                t.data.copy_(host)
    invalidate_operand_caches(module)


def invalidate_operand_caches(module: torch.nn.Module) -> None:
    """Drop every cached bf16 operand copy of weights below `module` (the encoder engine's, and those of predictor
    blocks and stand-alone modules).  Needed after parameters were written through `.data` (a broadcast, a raw copy):
    such writes do not bump the version counters the caches key on, so a forward that ran before the write would
    otherwise leave stale copies in use."""
    for sub in module.modules():
        cache = getattr(getattr(sub, "_engine_obj", None), "weights", None)
        if cache is not None:
            cache.invalidate()
        per_module = sub.__dict__.get("_operand_cache")
        if per_module is not None:
            per_module.invalidate()


class BucketReducer:
    """Averages flat gradient buckets across ranks, overlapping with whatever the caller enqueues next."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, always_reduce: bool = False, log=None):
        """always_reduce: issue the collectives even in a world of one (exercises the stream hand-over on a
        single GPU; the result is unchanged).  log: optional callable(str) told about every bucket (debugging)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.always_reduce = always_reduce
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.log = log
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._pending: List[Tuple[object, torch.Tensor, object]] = []
        self.buckets_reduced = 0
        self.bytes_reduced = 0
        # measure_exposed: bracket every join with device events on the compute stream.  The first event completes
        # when the last kernel enqueued before the join (the end of backward) is done, the second when the last
        # bucket's reduction is: the span is the communication time that backward did NOT hide.
        self.measure_exposed = False
        self._exposed: List[Tuple[object, object]] = []

    def _say(self, msg: str) -> None:
        if self.log is not None:
            self.log(msg)

    def submit(self, flat: torch.Tensor) -> None:
        """Start the all-reduce of one bucket; its producer kernels are already on the current stream."""
        if self.world == 1 and not self.always_reduce:
            return
        idx = self.buckets_reduced
        self.buckets_reduced += 1
        self.bytes_reduced += flat.numel() * flat.element_size()
        if self.backend == "nccl" and flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=flat.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            flat.record_stream(self._comm_stream)
            self._pending.append((work, flat, None))
            self._say(f"bucket {idx}: {flat.numel() * 4 >> 20} MiB all-reduce enqueued on the side stream")
        elif flat.is_cuda:
            # reduced through host memory in finish(): only the "producers done" point is recorded now
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(flat.device))
            self._pending.append((None, flat, ready))
            self._say(f"bucket {idx}: {flat.numel() * 4 >> 20} MiB deferred (host-staged reduce at the join)")
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, flat, "cpu"))

    def finish(self) -> None:
        """Make the current stream wait for every outstanding bucket (and finish the averaging)."""
        timed = self.measure_exposed and self._pending and self._pending[0][1].is_cuda
        if timed:
            before = torch.cuda.Event(enable_timing=True)
            before.record()
        self._finish_pending()
        if timed:
            after = torch.cuda.Event(enable_timing=True)
            after.record()
            self._exposed.append((before, after))

    def exposed_ms(self) -> List[float]:
        """Per join since the last call: milliseconds the compute stream spent waiting for reductions after its own
        work was done (needs `measure_exposed = True`; synchronises the device)."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed.clear()
        return out

    def _finish_pending(self) -> None:
        for n, (work, flat, tag) in enumerate(self._pending):
            if work is not None:
                work.wait()
                if tag == "cpu":
                    flat.mul_(1.0 / self.world)
            else:
                tag.synchronize()
                host = flat.detach().cpu()
                self._say(f"join {n}: on the host, reducing")
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(host.mul_(1.0 / self.world))
                self._say(f"join {n}: back on the device")
        self._pending.clear()

    def reduce_gradients(self, params: Iterable[torch.nn.Parameter]) -> None:
        """Average the gradients of parameters that are NOT covered by the encoder's buckets (the small heads on top
        of it) as one flat bucket, and join.  Call after backward and before clipping / the optimizer step: the clip
        norm is only the same on every rank once every gradient has been averaged."""
        if self.world == 1 and not self.always_reduce:
            return
        # the bucket covers the FIXED list of trainable parameters, zeros standing in for a missing gradient: its size
        # and the fact that a collective is issued must not depend on which parameters got a gradient on this rank (an
        # empty shard, a head one rank did not use), or ranks would disagree about the collective and hang
        params = [p for p in params if p.requires_grad]
        if params:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32)
                              for p in params])
            self.submit(flat)
            self.finish()
            off = 0
            for p in params:
                n = p.numel()
                piece = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = piece.to(p.dtype).clone()
                else:
                    p.grad.copy_(piece)
                off += n
        else:
            self.finish()


def all_gather_with_grad(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Concatenate `x` (B_local, ...) over ranks along dim 0 with a gradient: the local slice of the
    incoming gradient flows back.  Used for the in-batch-negative InfoNCE term of the stage-1 objective
    (the projections are (B,128): tiny), so that N ranks contrast over the same global batch one rank would.
    Shards may differ in size."""
    return _AllGather.apply(x, group)


class _AllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        # RCCL gathers device tensors; any other backend (gloo) is given host tensors
        via = x.device if (dist.get_backend(group) == "nccl" or not x.is_cuda) else torch.device("cpu")
        n = torch.tensor([x.shape[0]], dtype=torch.int64, device=via)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        sizes = [int(s) for s in sizes]
        biggest = max(sizes)
        mine = x.detach().contiguous().to(via)
        if mine.shape[0] < biggest:
            pad = mine.new_zeros((biggest - mine.shape[0],) + tuple(mine.shape[1:]))
            mine = torch.cat([mine, pad], dim=0)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        ctx.begin, ctx.n = sum(sizes[:rank]), sizes[rank]
        return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0).to(x.device)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.begin:ctx.begin + ctx.n].contiguous(), None
