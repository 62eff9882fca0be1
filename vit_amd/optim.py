"""AdamW + global-norm gradient clipping as two multi-tensor launches.

Restates the optimizer side of the reference's training step (stage1_surgvu_pretraining.py:390
`torch.optim.AdamW`, :469 `clip_grad_norm_(params, max_norm=1.0)`, :470 `optimizer.step()`): same update
arithmetic as torch.optim.AdamW, with the clip coefficient applied inside the update kernel (gradients are
read once, never rewritten).  State (exp_avg, exp_avg_sq) lives in two flat f32 buffers.
"""
from __future__ import annotations

import math
import struct
from typing import Iterable, List, Optional

import torch

from . import _lib
from ._lib import check

CHUNK = 65536     # VA_OPT_CHUNK in include/vit_amd.h


class FusedAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FusedAdamW: no trainable parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda or not p.is_contiguous():
                raise ValueError("FusedAdamW needs contiguous f32 parameters on one HIP device")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.steps = [0] * len(self.params)          # torch.optim counts steps per parameter
        self.device = dev
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4            # keep every slice 16-byte aligned
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._offs = offs
        self.total_sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._table_key = None
        self._table = self._partial = self._desc = None

    # learning-rate schedulers of torch expect param_groups; one group is enough here
    @property
    def param_groups(self):
        return [{"params": self.params, "lr": self.lr}]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _build(self, live):
        key = tuple(i for i, _ in live)
        if key != self._table_key:                        # which parameters have gradients changed: new chunk table
            pairs = []
            for slot, (i, p) in enumerate(live):
                pairs += [(slot, c) for c in range((p.numel() + CHUNK - 1) // CHUNK)]
            self._table = torch.tensor(pairs, dtype=torch.int32).reshape(-1).to(self.device)
            self._partial = torch.empty(len(pairs), dtype=torch.float32, device=self.device)
            self._table_key = key
        b1, b2 = self.betas
        raw = b"".join(struct.pack("<QQQQqff", p.data_ptr(), p.grad.data_ptr(),
                                   self.exp_avg.data_ptr() + 4 * self._offs[i],
                                   self.exp_avg_sq.data_ptr() + 4 * self._offs[i], p.numel(),
                                   1.0 / (1.0 - b1 ** self.steps[i]), 1.0 / math.sqrt(1.0 - b2 ** self.steps[i]))
                       for i, p in live)
        self._desc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device, non_blocking=False)
        return self._table.numel() // 2

    @torch.no_grad()
    def step(self) -> None:
        live = []
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != self.device:
                raise ValueError("FusedAdamW: gradients must be contiguous f32 on the parameters' device")
            live.append((i, p))
        if not live:
            return
        for i, _ in live:
            self.steps[i] += 1
        n_chunks = self._build(live)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            check(lib.va_multi_sumsq(self._desc.data_ptr(), self._table.data_ptr(), n_chunks, self._partial.data_ptr(),
                                     self.total_sumsq.data_ptr(), stream), "va_multi_sumsq")
        check(lib.va_multi_adamw(self._desc.data_ptr(), self._table.data_ptr(), n_chunks, self.lr, self.betas[0],
                                 self.betas[1], self.eps, self.weight_decay,
                                 self.total_sumsq.data_ptr() if clip else None,
                                 float(self.max_grad_norm) if clip else 0.0, stream), "va_multi_adamw")
        # the kernels write the parameters behind autograd's back: bump the version counters so that cached
        # bf16 operand copies are refreshed
        for _, p in live:
            p.add_(0)

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the last step's gradients (device scalar), when clipping is enabled."""
        return self.total_sumsq.sqrt()
