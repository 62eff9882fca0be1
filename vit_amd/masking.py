"""Token masking for the pre-training objectives.

`StructuredTemporalMasking.future_masking` keeps the reference's static-method signature
(stage1_surgvu_pretraining.py:300-333); the zero-fill and the boolean mask are produced by one HIP kernel.
`sample_block_masks` draws seeded V-JEPA-style context/target index sets on the host (index generation is
cheap integer work; the token moves they drive are the device kernels in `vit_amd.ops`).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch

from . import ops


class StructuredTemporalMasking:
    @staticmethod
    def first_masked_row(n_tokens: int, num_temporal_patches: int, mask_ratio: float = 0.5) -> int:
        """Index of the first masked token: cutoff * tokens-per-timestep with cutoff = int(T * (1 - ratio)) (s1:318)."""
        per_t = n_tokens // num_temporal_patches
        if num_temporal_patches * per_t != n_tokens:
            raise ValueError("N must be a whole number of timesteps")
        return int(num_temporal_patches * (1 - mask_ratio)) * per_t

    @staticmethod
    def future_masking(encoded_features: torch.Tensor, num_temporal_patches: int, mask_ratio: float = 0.5):
        """(B,N,D) -> (masked features, bool mask (B,N), True = masked).  All tokens of timesteps
        >= int(T*(1-ratio)) are zeroed; the cut-off is deterministic and equal for every batch row."""
        B, N, D = encoded_features.shape
        per_t = N // num_temporal_patches
        cutoff = int(num_temporal_patches * (1 - mask_ratio))
        # tokens past num_temporal_patches*per_t (N not a multiple of T) stay unmasked, as in the reference loop
        if num_temporal_patches * per_t != N:
            raise ValueError("N must be a whole number of timesteps")
        return ops.future_mask(encoded_features.contiguous(), cutoff * per_t)


def sample_block_masks(seed: int, batch: int, grid: Tuple[int, int, int], n_target_blocks: int = 4,
                       target_scale=(0.15, 0.2), target_aspect=(0.75, 1.5), context_keep: float = 1.0):
    """Seeded multi-block masks: per clip, `n_target_blocks` spatial rectangles extended over all timesteps
    are the target tokens, the complement is the context.  Rows are truncated to the batch minimum so the
    index tensors are rectangular.  Returns (context_idx, target_idx), int32, sorted ascending per row.
    The draw order from numpy's PCG64 stream is: per clip, per block: area, aspect, top, left."""
    Tp, Hp, Wp = grid
    rng = np.random.default_rng(seed)
    ctx_rows, tgt_rows = [], []
    for _ in range(batch):
        plane = np.zeros((Hp, Wp), dtype=bool)
        for _ in range(n_target_blocks):
            area = rng.uniform(*target_scale) * Hp * Wp
            asp = rng.uniform(*target_aspect)
            bh = int(min(Hp, max(1, round(math.sqrt(area * asp)))))
            bw = int(min(Wp, max(1, round(math.sqrt(area / asp)))))
            top = int(rng.integers(0, Hp - bh + 1))
            left = int(rng.integers(0, Wp - bw + 1))
            plane[top:top + bh, left:left + bw] = True
        vol = np.broadcast_to(plane[None], (Tp, Hp, Wp)).reshape(-1)
        tgt, ctx = np.nonzero(vol)[0], np.nonzero(~vol)[0]
        if context_keep < 1.0:
            keep = max(1, int(len(ctx) * context_keep))
            ctx = np.sort(rng.permutation(ctx)[:keep])
        ctx_rows.append(ctx)
        tgt_rows.append(tgt)
    nc, nt = min(map(len, ctx_rows)), min(map(len, tgt_rows))
    to_t = lambda rows, n: torch.from_numpy(np.stack([r[:n] for r in rows]).astype(np.int32))
    return to_t(ctx_rows, nc), to_t(tgt_rows, nt)
