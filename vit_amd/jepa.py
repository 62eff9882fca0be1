"""Pieces of a JEPA-style latent-prediction objective on top of the encoder: EMA target encoder update,
L1 latent loss with its gradient, and context/target token selection.  None of these exist in the
reference (its objective is order-classification + InfoNCE + masked-mean MSE); they are the HBM-bound
companions of the encoder and run as single kernels over flat buffers."""
from __future__ import annotations

from typing import Iterable

import torch

from . import ops


class FlatParameters:
    """Re-homes a module's f32 parameters into ONE flat buffer (parameters become views), so that whole-model
    updates such as the EMA are a single launch over 300 M elements instead of hundreds of small ones."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        total = sum((p.numel() + 3) // 4 * 4 for p in self.params)      # keep every view 16-byte aligned
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                if p.dtype != torch.float32 or p.device != dev:
                    raise ValueError("FlatParameters needs f32 parameters on one device")
                n = p.numel()
                view = self.flat[off:off + n].view_as(p)
                view.copy_(p)
                p.data = view
                off += (n + 3) // 4 * 4


def ema_update_(target: FlatParameters, online: FlatParameters, momentum: float, target_encoder=None) -> None:
    """target <- m*target + (1-m)*online, one launch.  The kernel writes the parameters behind autograd's
    back (no version bump), so pass the target `VideoViTEncoder` to have its bf16 operand copies dropped."""
    if target.flat.numel() != online.flat.numel():
        raise ValueError("target and online parameter sets differ")
    ops.ema_update_(target.flat, online.flat, momentum)
    if target_encoder is not None and getattr(target_encoder, "_engine_obj", None) is not None:
        target_encoder._engine_obj.weights.invalidate()


class L1LatentLoss(torch.autograd.Function):
    """mean |pred - target| with the gradient produced in the same pass over the data."""

    @staticmethod
    def forward(ctx, pred: torch.Tensor, target: torch.Tensor):
        loss, grad = ops.l1_loss(pred.contiguous(), target.detach().contiguous(), want_grad=True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def l1_latent_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return L1LatentLoss.apply(pred, target)
