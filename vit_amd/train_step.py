"""Training-step harnesses around the encoder.

`stage1_losses` / `stage1_step` restate the step of the reference's stage-1 loop
(stage1_surgvu_pretraining.py:418-470; frozen-encoder variant stage1_vjepa2_finetuning.py:281-339):
order classification on a shuffled clip, symmetric InfoNCE between a clip and its augmentation, and the
masked-mean denoising term on a no-grad pass, `loss = L_t + L_c + lambda * L_m`, global-norm clip 1.0, AdamW.
The heads' layers are tiny (B x 1024 MLPs) and stay ordinary torch modules; the three loss formulas on top of them
(value and gradient) are one kernel call each, and the encoder passes are the HIP path.

`jepa_latent_step` is the composite that exercises the token-gather, L1 and EMA kernels at full size; the
reference has no such objective (its "JEPA" is the three terms above).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .ddp import all_gather_with_grad, shard_weight
from .masking import StructuredTemporalMasking


class Stage1Heads(nn.Module):
    """The three heads the stage-1 loop builds on top of the encoder (s1:362-383), same layer layout."""

    def __init__(self, dim: int = 1024, use_mask_aux: bool = True):
        super().__init__()
        self.temporal_classifier = nn.Sequential(nn.Linear(dim, dim // 2), nn.ReLU(), nn.Dropout(0.1), nn.Linear(dim // 2, 2))
        self.contrastive_proj = nn.Sequential(nn.Linear(dim, dim), nn.ReLU(), nn.Linear(dim, 128))
        self.denoise_head = (nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim), nn.GELU(), nn.Linear(dim, dim))
                             if use_mask_aux else None)


    def ordered(self):
        """The heads in the order the reference's loops list them (s1:386-390, ft:205-210)."""
        return [m for m in (self.temporal_classifier, self.contrastive_proj, self.denoise_head) if m is not None]


def finetune_param_groups(model, heads: Stage1Heads, learning_rate: float = 4e-4, encoder_frozen: bool = True):
    """The optimizer groups of the fine-tuning loop (stage1_vjepa2_finetuning.py:166-170, 196-232): one group per
    sub-module in the reference's order -- the encoder first, at a tenth of the rate, when it trains; not at all (and
    with `requires_grad` off, in eval mode as ft:263-264 puts it every epoch) when frozen.  Pass the result to
    `FusedAdamW(groups, weight_decay=0.1, max_grad_norm=1.0)` or `torch.optim.AdamW(groups, weight_decay=0.1)`."""
    groups = []
    if encoder_frozen:
        for p in model.encoder.parameters():
            p.requires_grad = False
        model.encoder.eval()
    else:
        groups.append({"params": list(model.encoder.parameters()), "lr": learning_rate * 0.1})
    for part in (model.action_state_embed, model.video_proj, model.predictor_blocks, model.predictor_norm,
                 model.predictor_to_encoder, model.sterility_classifier, model.autoregressive_predictor, *heads.ordered()):
        groups.append({"params": list(part.parameters()), "lr": learning_rate})
    return groups


def warmup_cosine(warmup_epochs: int, num_epochs: int):
    """The fine-tuning schedule's multiplier for `torch.optim.lr_scheduler.LambdaLR` (ft:235-241): linear warm-up
    (epoch + 1) / warmup, then half a cosine over the remaining epochs."""
    import math

    def lr_lambda(epoch: int) -> float:
        if epoch < warmup_epochs:
            return (epoch + 1) / warmup_epochs
        return 0.5 * (1 + math.cos(math.pi * (epoch - warmup_epochs) / (num_epochs - warmup_epochs)))
    return lr_lambda


class _TokenPool(torch.autograd.Function):
    """Mean over the token axis, (B, N, D) -> (B, D): one reduction kernel forward, one broadcast kernel backward."""

    @staticmethod
    def forward(ctx, x):
        ctx.n = x.shape[1]
        return ops.token_pool(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.token_pool_bwd(g.contiguous(), ctx.n)


def _pool_tokens(x: torch.Tensor) -> torch.Tensor:
    """`x.mean(dim=1)` (s1:426, 431-432) on the pooling kernel."""
    return _TokenPool.apply(x)


def _pool_unmasked(tokens: torch.Tensor, num_temporal_patches: int, mask_ratio: float = 0.5):
    """(mean over the tokens `future_masking` leaves unmasked, mean over all tokens) of a no-grad pass (s1:449-460).
    The mask zeroes a suffix of the token axis, so the masked mean is a pooling over the first cutoff * S tokens:
    neither the zero-filled copy nor the boolean mask is materialised."""
    per_t = tokens.shape[1] // num_temporal_patches
    n_first = StructuredTemporalMasking.first_masked_row(tokens.shape[1], num_temporal_patches, mask_ratio)
    t = tokens.detach().contiguous()
    return ops.token_pool(t, n_first), ops.token_pool(t)


class _InfoNCE(torch.autograd.Function):
    """0.5 * (CE(z1 z2^T / tau) + CE(z2 z1^T / tau)) with z = normalised rows (s1:434-446): value and both gradients from
    one kernel call in the forward; the backward scales them."""

    @staticmethod
    def forward(ctx, p1, p2, temperature, normalize):
        loss, g1, g2 = ops.infonce_sym(p1.detach().contiguous(), p2.detach().contiguous(), temperature, normalize)
        ctx.save_for_backward(g1, g2)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        g1, g2 = ctx.saved_tensors
        return g1 * g, g2 * g, None, None


class _NormalizeRows(torch.autograd.Function):
    """F.normalize(p, dim=1) on the kernels (the arithmetic of the fused InfoNCE's own normalisation)."""

    @staticmethod
    def forward(ctx, p):
        z, inv = ops.normalize_rows(p.detach().contiguous())
        ctx.save_for_backward(z, inv)
        return z

    @staticmethod
    def backward(ctx, dz):
        z, inv = ctx.saved_tensors
        return ops.normalize_rows_bwd(dz.contiguous(), z, inv)


def normalize_rows(p):
    return _NormalizeRows.apply(p) if _on_kernels(p) and p.dim() == 2 else F.normalize(p, dim=1)


class _CrossEntropy(torch.autograd.Function):
    """The kernel leaves (softmax - onehot) unscaled; the backward applies (upstream / B) as ONE factor, so that a mean
    over a shard times the shard's weight rounds like the mean over the whole batch."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, d = ops.cross_entropy(logits.detach().contiguous(), labels.contiguous(), grad_scale=1.0)
        ctx.save_for_backward(d)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        # (a float tensor divided by a Python number is multiplied by the rounded reciprocal: the division is done in f64)
        return d * (g.double() / d.shape[0]).to(d.dtype), None


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        loss, d = ops.mse_loss(x.detach().contiguous(), y.detach().contiguous(), grad_scale=1.0)
        ctx.save_for_backward(d)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        return d * (g.double() * 2.0 / d.numel()).to(d.dtype), None


def _on_kernels(*tensors) -> bool:
    """Device f32 tensors take the loss kernels; anything else (the CPU rehearsals of the data-parallel step) the same
    formulas in torch."""
    return all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def infonce(p1, p2, temperature: float, normalize: bool = True):
    """Symmetric InfoNCE of two [B, D] projection sets; on the device one fused kernel call (value + gradients), else
    the formula in torch."""
    if _on_kernels(p1, p2) and p1.shape[0] <= 8192 and p1.shape[1] <= 2048:
        return _InfoNCE.apply(p1, p2, temperature, normalize)
    z1, z2 = (F.normalize(p1, dim=1), F.normalize(p2, dim=1)) if normalize else (p1, p2)
    labels = torch.arange(z1.shape[0], device=z1.device)
    return 0.5 * (F.cross_entropy(z1 @ z2.t() / temperature, labels) + F.cross_entropy(z2 @ z1.t() / temperature, labels))


def cross_entropy(logits, labels):
    if _on_kernels(logits) and labels.is_cuda and labels.dtype == torch.int64:
        return _CrossEntropy.apply(logits, labels)
    return F.cross_entropy(logits, labels)


def mse(x, y):
    if _on_kernels(x, y) and not y.requires_grad and x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:
        return _Mse.apply(x, y)
    return F.mse_loss(x, y)


def stage1_head_losses(heads: Stage1Heads, f_shuf, f_orig, f_aug, pooled_unmasked, pooled_all, order_label,
                       lambda_mask: float = 0.1, temperature: float = 0.07,
                       group: Optional[dist.ProcessGroup] = None):
    """The step's loss arithmetic on pooled encoder features (s1:426-465): f_* are (B, D) mean-pooled features of
    the shuffled / original / augmented views; pooled_unmasked / pooled_all the (B, D) means of the no-grad pass over
    the unmasked tokens and over all tokens (None without the denoise head).

    Returns (objective, (l_t, l_c, l_m)).  In a world of one the objective is l_t + l_c + lambda*l_m.  With several
    ranks the InfoNCE negatives span the global batch (projections are all-gathered), and the objective is scaled
    so that the rank-AVERAGED gradient equals the one-rank gradient on the concatenated batch: the gathered term
    -- identical on every rank -- times world, the per-rank means times (B_local / B_global) * world."""
    l_t = cross_entropy(heads.temporal_classifier(f_shuf), order_label)
    p1, p2 = heads.contrastive_proj(f_orig), heads.contrastive_proj(f_aug)
    world, b_local, b_global = 1, p1.shape[0], p1.shape[0]
    # rows are normalised locally; with several ranks the negatives span the global batch (the normalised rows are
    # gathered); the loss kernel takes normalised rows as they are -- the same arithmetic for any number of ranks
    z1, z2 = normalize_rows(p1), normalize_rows(p2)
    if group is not None or (dist.is_initialized() and dist.get_world_size() > 1):
        world = dist.get_world_size(group)
        z1, z2 = all_gather_with_grad(z1, group), all_gather_with_grad(z2, group)
        b_global = z1.shape[0]
    l_c = infonce(z1, z2, temperature, normalize=False)

    l_m = torch.zeros((), device=f_orig.device)
    if heads.denoise_head is not None and pooled_unmasked is not None:
        l_m = mse(heads.denoise_head(pooled_unmasked), pooled_all)
    if world > 1:
        w = shard_weight(b_local, b_global, world)
        objective = w * (l_t + lambda_mask * l_m) + world * l_c
    else:
        objective = l_t + l_c + lambda_mask * l_m
    return objective, (l_t.detach(), l_c.detach(), l_m.detach())


def stage1_losses(model, heads: Stage1Heads, video, video_aug, shuffled_video, order_label,
                  lambda_mask: float = 0.1, temperature: float = 0.07, fuse_views: bool = True,
                  encoder_grad: bool = True, group: Optional[dist.ProcessGroup] = None,
                  reuse_tokens: Optional[bool] = None):
    """Returns (objective, (temporal, contrastive, mask)); see `stage1_head_losses` for the objective.  `fuse_views`
    runs the three gradient-carrying views as ONE encoder pass over 3B clips (clips are independent, so the result is
    identical and every weight is read once instead of three times).  The token pooling on either side of the heads
    (the only part of them that moves more than a few megabytes) runs on the pooling kernels; the heads' small
    (B x 1024) layers are library GEMMs through torch."""
    B = video.shape[0]
    # The denoising term's no-grad pass encodes `video` once more (s1:449-451).  When no dropout is active in the encoder
    # (eval mode -- the frozen fine-tuning loop puts it there, ft:263-264 -- or p = 0) that pass repeats the `video`
    # view of the passes above bit for bit, so its tokens are taken from there instead of being computed again.
    enc = getattr(model, "encoder", None)
    deterministic = enc is not None and not any(m.training and m.p > 0 for m in enc.modules() if isinstance(m, nn.Dropout))
    if reuse_tokens is not None:          # False forces the separate pass (tests compare the two)
        deterministic = deterministic and reuse_tokens
    tokens = None
    with torch.set_grad_enabled(encoder_grad and torch.is_grad_enabled()):
        if fuse_views:
            all_tokens = model.encode_video(torch.cat([shuffled_video, video, video_aug], dim=0))
            feats = _pool_tokens(all_tokens)
            f_shuf, f_orig, f_aug = feats[:B], feats[B:2 * B], feats[2 * B:]
            if deterministic:
                tokens = all_tokens[B:2 * B].detach()
        else:
            f_shuf = _pool_tokens(model.encode_video(shuffled_video))
            orig_tokens = model.encode_video(video)
            f_orig = _pool_tokens(orig_tokens)
            f_aug = _pool_tokens(model.encode_video(video_aug))
            if deterministic:
                tokens = orig_tokens.detach()
    pooled_unmasked = pooled_all = None
    if heads.denoise_head is not None:
        with torch.no_grad():
            if tokens is None:
                tokens = model.encode_video(video)
            pooled_unmasked, pooled_all = _pool_unmasked(tokens, model.encoder.tubelet_embed.num_temporal_patches)
    return stage1_head_losses(heads, f_shuf, f_orig, f_aug, pooled_unmasked, pooled_all, order_label, lambda_mask,
                              temperature, group)


def stage1_step(model, heads: Stage1Heads, optimizer, batch: Dict[str, torch.Tensor],
                max_norm: Optional[float] = 1.0, reducer=None, lambda_mask: float = 0.1,
                **kw) -> Tuple[float, Tuple[torch.Tensor, ...]]:
    """One optimisation step (s1:418-470).  `optimizer` is a torch optimizer (then the gradients are clipped here
    with clip_grad_norm_) or a `vit_amd.optim.FusedAdamW` built with `max_grad_norm` (clipping happens inside its
    update kernel: pass max_norm=None).

    Data parallel: pass the `BucketReducer`.  The encoder's gradients are averaged bucket by bucket inside backward
    when the reducer is attached to it (`encoder.set_grad_reducer`); every other trained parameter (the heads; the
    encoder too if the reducer is not attached) is averaged here as one more bucket, and only then -- when every
    rank holds the same gradients -- is the global norm taken and the step made, so replicas stay identical.
    Returns the unscaled loss l_t + l_c + lambda*l_m of this rank's shard for logging."""
    optimizer.zero_grad(set_to_none=True)
    objective, parts = stage1_losses(model, heads, batch["video"], batch["video_augmented"], batch["shuffled_video"],
                                     batch["order_label"].long(), lambda_mask=lambda_mask, **kw)
    objective.backward()
    trained = [p for g in optimizer.param_groups for p in g["params"]]
    if reducer is not None:
        enc = getattr(model, "encoder", None)
        covered = set()
        if enc is not None and getattr(enc, "_grad_reducer", None) is reducer:
            covered = {id(p) for p in enc.parameters()}
        reducer.reduce_gradients([p for p in trained if id(p) not in covered])
    if max_norm is not None:
        torch.nn.utils.clip_grad_norm_(trained, max_norm=max_norm)
    optimizer.step()
    l_t, l_c, l_m = parts
    return float(l_t + l_c + lambda_mask * l_m), parts


def jepa_latent_step(encoder, target_encoder, video: torch.Tensor, target_idx: torch.Tensor,
                     momentum: float = 0.996, target_flat=None, online_flat=None):
    """loss = mean | gather(encoder(video)) - gather(target_encoder(video)) | over the target tokens, backward
    into `encoder`, then (if the flat parameter sets are given) target <- m*target + (1-m)*online."""
    from .jepa import ema_update_, l1_latent_loss
    with torch.no_grad():
        tgt = ops.token_gather(target_encoder(video).contiguous(), target_idx)
    z = encoder(video)
    pred = _TokenGather.apply(z, target_idx)
    loss = l1_latent_loss(pred, tgt)
    loss.backward()
    if target_flat is not None and online_flat is not None:
        ema_update_(target_flat, online_flat, momentum, target_encoder)
    return loss.detach()


class _TokenGather(torch.autograd.Function):
    """Row gather with its scatter as the backward (both single kernels)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n = x.shape[1]
        return ops.token_gather(x.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ops.token_scatter(g.contiguous(), idx, ctx.n), None
