"""AdamW + global-norm gradient clipping as multi-tensor launches.

Restates the optimizer side of the reference's training step (stage1_surgvu_pretraining.py:390
`torch.optim.AdamW`, :469 `clip_grad_norm_(params, max_norm=1.0)`, :470 `optimizer.step()`; two parameter groups
with the encoder at 0.1x the learning rate in stage1_vjepa2_finetuning.py:216-232): same update arithmetic as
torch.optim.AdamW, with the clip coefficient applied inside the update kernel (gradients are read once, never
rewritten).

`FusedAdamW` is a `torch.optim.Optimizer`: `param_groups` is the persistent list schedulers write `lr` into
(CosineAnnealingLR s1:391, LambdaLR ft:235-239), and `state_dict()` / `load_state_dict()` use torch.optim.AdamW's
layout (per parameter `step`, `exp_avg`, `exp_avg_sq`), so the `optimizer` entry of a checkpoint (s1:504) written by
either optimizer loads into the other.  Parameters with `requires_grad=False` stay in their groups, as they do in
torch.optim.AdamW (the frozen encoder of stage1_vjepa2_finetuning.py:166-170 when a caller passes all parameters), so
the parameter indices and per-group counts of the two state dicts agree; they get no moments and no state entry, and
are skipped by `step()`.  The moments live in two flat f32 buffers; the per-parameter state tensors are views into them.
"""
from __future__ import annotations

import math
import struct
from typing import List, Optional

import torch

from . import _lib
from ._lib import check

CHUNK = 65536     # VA_OPT_CHUNK in include/vit_amd.h


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        # every parameter of every group, frozen ones included: position = its index in the state dict
        self.params: List[torch.nn.Parameter] = [p for g in self.param_groups for p in g["params"]]
        trainable = [p for p in self.params if p.requires_grad]
        if not trainable:
            raise ValueError("FusedAdamW: no trainable parameters")
        dev = trainable[0].device
        for p in trainable:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda or not p.is_contiguous():
                raise ValueError("FusedAdamW needs contiguous f32 parameters on one HIP device")
        self.max_grad_norm = max_grad_norm
        self.device = dev
        offs, total = [], 0
        for p in self.params:                            # frozen parameters get no slice of the moment buffers
            offs.append(total if p.requires_grad else None)
            if p.requires_grad:
                total += (p.numel() + 3) // 4 * 4        # keep every slice 16-byte aligned
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._offs = offs
        self._slot = {id(p): i for i, p in enumerate(self.params)}
        for i, p in enumerate(self.params):
            if offs[i] is not None:
                self.state[p] = self._fresh_state(i, p, 0.0)
        self.total_sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._table_key = None
        self._table = self._partial = self._desc = None
        self._group_chunks: List[int] = []

    # ------------------------------------------------------------------ state
    def _fresh_state(self, i: int, p: torch.Tensor, step: float) -> dict:
        o, n = self._offs[i], p.numel()
        return {"step": torch.tensor(float(step)),                       # torch.optim.AdamW counts steps per parameter
                "exp_avg": self.exp_avg[o:o + n].view(p.shape),
                "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape)}

    def state_dict(self):
        """torch.optim.AdamW's layout.  torch creates a parameter's state at its first step with a gradient, so a
        parameter that has not been stepped yet (the predictor under the stage-1 objective, ft:196-232) has no entry:
        the same here, although its slice of the flat buffers exists from the start."""
        sd = super().state_dict()
        never = {i for i, p in enumerate(self.params) if p in self.state and float(self.state[p]["step"]) == 0.0}
        sd["state"] = {k: v for k, v in sd["state"].items() if k not in never}
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict) -> None:
        """Accepts what `state_dict()` of this class or of torch.optim.AdamW (same parameter order) wrote; the loaded
        moments are copied into the flat buffers the kernels update."""
        super().load_state_dict(state_dict)
        for i, p in enumerate(self.params):
            st = self.state.get(p)
            if self._offs[i] is None:                    # frozen here: nothing to keep for it
                if st:
                    del self.state[p]
                continue
            if not st:                                   # a parameter the saved optimizer had not stepped yet
                o, n = self._offs[i], p.numel()
                self.exp_avg[o:o + n].zero_()
                self.exp_avg_sq[o:o + n].zero_()
                self.state[p] = self._fresh_state(i, p, 0.0)
                continue
            fresh = self._fresh_state(i, p, float(st["step"]))
            fresh["exp_avg"].copy_(st["exp_avg"])
            fresh["exp_avg_sq"].copy_(st["exp_avg_sq"])
            self.state[p] = fresh
        for g in self.param_groups:
            g["betas"] = tuple(g["betas"])

    # ------------------------------------------------------------------ step
    def _build(self, live_by_group):
        key = tuple(tuple(i for i, _ in grp) for grp in live_by_group)
        live = [ip for grp in live_by_group for ip in grp]
        if key != self._table_key:                        # which parameters have gradients changed: new chunk table
            pairs, per_group, slot = [], [], 0
            for grp in live_by_group:
                before = len(pairs)
                for _, p in grp:
                    pairs += [(slot, c) for c in range((p.numel() + CHUNK - 1) // CHUNK)]
                    slot += 1
                per_group.append(len(pairs) - before)
            self._table = torch.tensor(pairs, dtype=torch.int32).reshape(-1).to(self.device)
            self._partial = torch.empty(len(pairs), dtype=torch.float32, device=self.device)
            self._group_chunks = per_group
            self._table_key = key
        raw = []
        for g, grp in zip(self.param_groups, live_by_group):
            b1, b2 = g["betas"]
            for i, p in grp:
                step = float(self.state[p]["step"])
                raw.append(struct.pack("<QQQQqff", p.data_ptr(), p.grad.data_ptr(),
                                       self.exp_avg.data_ptr() + 4 * self._offs[i],
                                       self.exp_avg_sq.data_ptr() + 4 * self._offs[i], p.numel(),
                                       1.0 / (1.0 - b1 ** step), 1.0 / math.sqrt(1.0 - b2 ** step)))
        self._desc = torch.frombuffer(bytearray(b"".join(raw)), dtype=torch.uint8).to(self.device, non_blocking=False)
        return live

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        live_by_group = []
        for g in self.param_groups:
            grp = []
            for p in g["params"]:
                if p.grad is None:
                    continue
                gr = p.grad
                if gr.dtype != torch.float32 or not gr.is_contiguous() or gr.device != self.device:
                    raise ValueError("FusedAdamW: gradients must be contiguous f32 on the parameters' device")
                if self._offs[self._slot[id(p)]] is None:
                    raise ValueError("FusedAdamW: a parameter that was frozen (requires_grad=False) when the optimizer "
                                     "was built has a gradient now; build the optimizer after unfreezing it")
                grp.append((self._slot[id(p)], p))
            live_by_group.append(grp)
        if not any(live_by_group):
            return loss
        for grp in live_by_group:
            for _, p in grp:
                self.state[p]["step"] += 1
        live = self._build(live_by_group)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        n_chunks = self._table.numel() // 2
        if clip:      # ONE norm over every group: clip_grad_norm_ is called on all parameters together (s1:469, ft:338)
            check(lib.va_multi_sumsq(self._desc.data_ptr(), self._table.data_ptr(), n_chunks, self._partial.data_ptr(),
                                     self.total_sumsq.data_ptr(), stream), "va_multi_sumsq")
        first = 0
        for g, count in zip(self.param_groups, self._group_chunks):
            if count:
                check(lib.va_multi_adamw(self._desc.data_ptr(), self._table.data_ptr() + 8 * first, count, float(g["lr"]),
                                         g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                                         self.total_sumsq.data_ptr() if clip else None,
                                         float(self.max_grad_norm) if clip else 0.0, stream), "va_multi_adamw")
            first += count
        # the kernels wrote the parameters behind autograd's back: bump the version counters (no launch) so that
        # cached bf16 operand copies of the weights are refreshed
        torch.autograd.graph.increment_version([p for _, p in live])
        return loss

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the last step's gradients (device scalar), when clipping is enabled."""
        return self.total_sumsq.sqrt()
