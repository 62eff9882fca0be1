"""CPU oracle for the video-ViT encoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (state-dict driven) fp32 restatement of
the arithmetic of the reference encoder path.  It exists so that the HIP path
can be checked on a machine where the reference sources are absent.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; the product package `vit_amd` never does.

Pinning: every function below is checked against the imported reference
modules by `tests/golden/make_golden.py` (run where /root/reference exists);
the resulting input/output vectors are committed under `tests/golden/` and
re-checked by `tests/test_oracle_golden.py`.  The reference itself ships no
tests or golden vectors, so those generated fixtures are the only pin.
Functions with no reference counterpart (JEPA block masks, token gather /
scatter, EMA, L1 latent loss) are "parity unpinned": this file is then the
definition, and says so in the docstring.

Reference citations use: swm = surgical_world_model.py,
s1 = stage1_surgvu_pretraining.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

ROPE_SPLIT = (256, 384, 384)      # swm:731-735 (t, h, w) column segments of the 1024-wide Q/K
ROPE_BASE = 10000.0               # swm:172
ROPE_TEMPORAL_SCALE = 0.6         # swm:742, swm:203
LN_EPS = 1e-6                     # swm:686, 695, 757


# --------------------------------------------------------------------------
# deterministic closed-form parameter fill (weights are too large to commit)
# --------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _name_seed(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def hashed_uniform(name: str, shape, scale: float, offset: float = 0.0) -> torch.Tensor:
    """Uniform(-scale, scale) + offset values that depend only on (name, flat index)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(_name_seed(name))
        bits = _splitmix64(idx)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / float(1 << 53))  # [0,1)
    v = (2.0 * u - 1.0) * scale + offset
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def fill_state_dict_(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Overwrite every floating tensor of a state dict with the closed-form fill.

    Scales: matrices U(+-1/sqrt(fan_in)), LayerNorm weight 1 +- 0.1, biases +-0.05.
    `inv_freq*` buffers are left alone (they are derived constants).
    """
    out = {}
    for k, v in sd.items():
        if "inv_freq" in k or not v.is_floating_point():
            out[k] = v.clone()
            continue
        if v.dim() >= 2:
            fan_in = int(np.prod(v.shape[1:]))
            out[k] = hashed_uniform(k, v.shape, 1.0 / math.sqrt(fan_in))
        elif k.endswith("weight"):
            out[k] = hashed_uniform(k, v.shape, 0.1, 1.0)
        else:
            out[k] = hashed_uniform(k, v.shape, 0.05)
    return out


def encoder_param_shapes(depth: int, dim: int = 1024, in_ch: int = 3,
                         tubelet=(2, 16, 16)) -> Dict[str, Tuple[int, ...]]:
    """State-dict keys and shapes of the encoder (verified against the reference in make_golden)."""
    s: Dict[str, Tuple[int, ...]] = {
        "tubelet_embed.projection.weight": (dim, in_ch) + tuple(tubelet),
        "tubelet_embed.projection.bias": (dim,),
    }
    for i in range(depth):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (dim,)
        s[p + "norm1.bias"] = (dim,)
        s[p + "attn.qkv_proj.weight"] = (3 * dim, dim)
        s[p + "attn.out_proj.weight"] = (dim, dim)
        s[p + "norm2.weight"] = (dim,)
        s[p + "norm2.bias"] = (dim,)
        s[p + "mlp.0.weight"] = (4 * dim, dim)
        s[p + "mlp.0.bias"] = (4 * dim,)
        s[p + "mlp.3.weight"] = (dim, 4 * dim)
        s[p + "mlp.3.bias"] = (dim,)
    s["norm.weight"] = (dim,)
    s["norm.bias"] = (dim,)
    return s


def make_encoder_params(depth: int, dim: int = 1024, requires_grad: bool = False) -> Dict[str, torch.Tensor]:
    shapes = encoder_param_shapes(depth, dim)
    sd = fill_state_dict_({k: torch.empty(v) for k, v in shapes.items()})
    if requires_grad:
        for v in sd.values():
            v.requires_grad_(True)
    return sd


# --------------------------------------------------------------------------
# RoPE tables and token coordinates
# --------------------------------------------------------------------------
def rope_inv_freq(d: int) -> torch.Tensor:
    """swm:184-186: base^(-arange(0,d,2)/d)."""
    return ROPE_BASE ** (-torch.arange(0, d, 2).float() / d)


def rope_tables(Tp: int, Hp: int, Wp: int):
    """swm:192-222.  Returns ((cos_t, sin_t), (cos_h, sin_h), (cos_w, sin_w)); each (len, d_seg)
    with the half-table repeated twice along the last axis; temporal positions scaled by 0.6."""
    out = []
    for n, d, sc in ((Tp, ROPE_SPLIT[0], ROPE_TEMPORAL_SCALE), (Hp, ROPE_SPLIT[1], 1.0), (Wp, ROPE_SPLIT[2], 1.0)):
        pos = torch.arange(n).float() * sc
        ang = pos[:, None] * rope_inv_freq(d)[None, :]
        ang = torch.cat([ang, ang], dim=-1)
        out.append((ang.cos(), ang.sin()))
    return tuple(out)


def token_coords(Tp: int, Hp: int, Wp: int):
    """swm:760-770.  Token n = (t*Hp + h)*Wp + w  (t slowest, w fastest)."""
    n = torch.arange(Tp * Hp * Wp)
    return n // (Hp * Wp), (n // Wp) % Hp, n % Wp


def rope_rotate(x: torch.Tensor, Tp: int, Hp: int, Wp: int, inverse: bool = False) -> torch.Tensor:
    """swm:33-51 + swm:265-281 on a (..., N, 1024) tensor: per column segment
    y[i] = x[i]*cos - x[i+half]*sin ; y[i+half] = x[i+half]*cos + x[i]*sin.
    `inverse=True` rotates by the negative angle (the transpose, used for gradients)."""
    tabs = rope_tables(Tp, Hp, Wp)
    idx = token_coords(Tp, Hp, Wp)
    outs, c0 = [], 0
    for (cos, sin), ix, d in zip(tabs, idx, ROPE_SPLIT):
        seg = x[..., c0:c0 + d]
        c, s = cos[ix], sin[ix]
        if inverse:
            s = -s
        a, b = seg[..., : d // 2], seg[..., d // 2:]
        rot = torch.cat([-b, a], dim=-1)
        outs.append(seg * c + rot * s)
        c0 += d
    return torch.cat(outs, dim=-1)


# --------------------------------------------------------------------------
# encoder pieces
# --------------------------------------------------------------------------
def tubelet_patches(video: torch.Tensor, tubelet=(2, 16, 16)) -> torch.Tensor:
    """(B,C,T,H,W) -> (B, N, C*tt*th*tw): row n = the patch of token n, columns ordered
    (c, dt, dh, dw) exactly like a flattened Conv3d filter (swm:78-84, 99-108)."""
    B, C, T, H, W = video.shape
    tt, th, tw = tubelet
    Tp, Hp, Wp = T // tt, H // th, W // tw
    v = video[:, :, : Tp * tt, : Hp * th, : Wp * tw]
    v = v.reshape(B, C, Tp, tt, Hp, th, Wp, tw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return v.reshape(B, Tp * Hp * Wp, C * tt * th * tw)


def tubelet_embed(video, weight, bias, tubelet=(2, 16, 16)):
    """swm:92-109 restated as patches @ W^T + b."""
    p = tubelet_patches(video, tubelet)
    return p @ weight.reshape(weight.shape[0], -1).t() + bias


def layer_norm(x, w, b, eps=LN_EPS):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x):
    """nn.GELU() default (swm:699): exact erf form."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# --------------------------------------------------------------------------
# training-mode dropout.  The reference draws its masks from torch's generator (nn.Dropout, swm:263, 700-702);
# the library under test draws them from a counter-based hash documented in its header.  This is an independent
# numpy restatement of that documented hash, so that a training-mode forward/backward can be compared on
# IDENTICAL masks; the rest of the arithmetic (mask * x / (1 - p)) is the reference's.
# --------------------------------------------------------------------------
def drop_thresh16(p: float) -> int:
    return int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))


def drop_scale(p: float) -> float:
    return float(np.float32(65536.0) / np.float32(65536 - drop_thresh16(p)))


def _drop_word(pair: np.ndarray, key: np.ndarray) -> np.ndarray:
    """h = mix(pair * 0x9E3779B1 + key), mix(w): w ^= w >> 16; w *= 0x21F0AAAD; w ^= w >> 15 (all mod 2^32)."""
    M = np.uint64(0xFFFFFFFF)
    w = (pair.astype(np.uint64) * np.uint64(0x9E3779B1) + key.astype(np.uint64)) & M
    w ^= w >> np.uint64(16)
    w = (w * np.uint64(0x21F0AAAD)) & M
    w ^= w >> np.uint64(15)
    return w


def drop_keep_pairs(key, pair: np.ndarray, odd: np.ndarray, p: float) -> np.ndarray:
    """Keep decision of the elements (pair index mod 2^32, odd flag): the high 16-bit field of the pair's word for
    an odd element, the low one for an even element, kept when the field >= round(p * 65536)."""
    M = np.uint64(0xFFFFFFFF)
    h = _drop_word(np.asarray(pair, dtype=np.uint64) & M, np.asarray(key, dtype=np.uint64) & M)
    field = np.where(np.asarray(odd) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return field >= np.uint64(drop_thresh16(p))


def drop_mask(key: int, shape, p: float) -> torch.Tensor:
    """Multiplier tensor (0 or 1/(1-p)) of a contiguous tensor of `shape`: flat element e is the (e & 1) field of
    pair e >> 1."""
    e = np.arange(int(np.prod(shape)), dtype=np.uint64)
    keep = drop_keep_pairs(key, e >> np.uint64(1), e & np.uint64(1), p)
    return torch.from_numpy(keep.astype(np.float32) * np.float32(drop_scale(p))).reshape(tuple(shape))


def attn_drop_mask(key: int, B: int, H: int, N: int, p: float) -> torch.Tensor:
    """Multiplier tensor [B, H, N, N] of the attention probabilities: head-batch bh = b*H + h uses the key
    key + bh * 0x85EBCA6B; element (q, k) is the (k & 1) field of pair q * ceil(N / 2) + (k >> 1)."""
    half = np.uint64((N + 1) // 2)
    bh = np.arange(B * H, dtype=np.uint64)[:, None, None]
    keys = (np.uint64(key) + bh * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    q = np.arange(N, dtype=np.uint64)[None, :, None]
    k = np.arange(N, dtype=np.uint64)[None, None, :]
    shape = (B * H, N, N)
    keep = drop_keep_pairs(np.broadcast_to(keys, shape), np.broadcast_to(q * half + (k >> np.uint64(1)), shape),
                           np.broadcast_to(k & np.uint64(1), shape), p)
    return torch.from_numpy(keep.astype(np.float32) * np.float32(drop_scale(p))).reshape(B, H, N, N)


def rope_attention(x, w_qkv, w_out, heads: int, grid: Optional[Tuple[int, int, int]],
                   attn_mask: Optional[torch.Tensor] = None, drop: Optional[torch.Tensor] = None):
    """swm:283-336.  w_qkv rows [0:D]=Q, [D:2D]=K, [2D:3D]=V.  drop: multiplier [B,H,N,N] standing for
    self.dropout(attn) in training mode (swm:326), None in eval mode."""
    B, N, D = x.shape
    hd = D // heads
    qkv = x @ w_qkv.t()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    if grid is not None:
        q = rope_rotate(q, *grid)
        k = rope_rotate(k, *grid)

    def split(t):
        return t.reshape(B, N, heads, hd).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    if attn_mask is not None:                       # True = masked (swm:322)
        m = attn_mask
        if m.dim() == 2:
            m = m[None, None]
        elif m.dim() == 3:
            m = m[:, None]
        s = s.masked_fill(m, float("-inf"))
    p = s.softmax(dim=-1)
    if drop is not None:
        p = p * drop
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, N, D)
    return o @ w_out.t()


def vit_block(x, P: Dict[str, torch.Tensor], prefix: str, heads: int, grid, attn_mask=None, drop=None):
    """swm:705-714.  drop: None (eval mode) or ((p_attn, p_fc1, p_fc2), (key_attn, key_fc1, key_fc2))."""
    g = lambda k: P[prefix + k]
    B, N, D = x.shape
    (pa, p1, p2), (ka, k1, k2) = drop if drop is not None else ((0, 0, 0), (0, 0, 0))
    x = x + rope_attention(layer_norm(x, g("norm1.weight"), g("norm1.bias")),
                           g("attn.qkv_proj.weight"), g("attn.out_proj.weight"), heads, grid, attn_mask,
                           attn_drop_mask(ka, B, heads, N, pa) if pa > 0 else None)
    h = layer_norm(x, g("norm2.weight"), g("norm2.bias"))
    h = gelu_erf(h @ g("mlp.0.weight").t() + g("mlp.0.bias"))
    if p1 > 0:
        h = h * drop_mask(k1, h.shape, p1)
    h = h @ g("mlp.3.weight").t() + g("mlp.3.bias")
    if p2 > 0:
        h = h * drop_mask(k2, h.shape, p2)
    return x + h


def encoder_forward(P: Dict[str, torch.Tensor], video: torch.Tensor, depth: int, heads: int,
                    tubelet=(2, 16, 16), drop_p=None, drop_keys=None) -> torch.Tensor:
    """swm:772-817.  (B,C,T,H,W) -> (B, N, D).  drop_p / drop_keys: per block three probabilities / keys
    (training mode); None for eval mode."""
    B, C, T, H, W = video.shape
    grid = (T // tubelet[0], H // tubelet[1], W // tubelet[2])
    x = tubelet_embed(video, P["tubelet_embed.projection.weight"], P["tubelet_embed.projection.bias"], tubelet)
    for i in range(depth):
        x = vit_block(x, P, f"blocks.{i}.", heads, grid,
                      drop=(drop_p[i], drop_keys[i]) if drop_p is not None else None)
    return layer_norm(x, P["norm.weight"], P["norm.bias"])


# --------------------------------------------------------------------------
# masking
# --------------------------------------------------------------------------
def future_masking(feats: torch.Tensor, num_temporal_patches: int, mask_ratio: float = 0.5):
    """s1:300-333.  cutoff = int(T*(1-ratio)); tokens of timesteps >= cutoff are zeroed.
    Deterministic, identical for every batch row.  Returns (masked, bool mask (B,N))."""
    B, N, D = feats.shape
    per_t = N // num_temporal_patches
    cutoff = int(num_temporal_patches * (1 - mask_ratio))
    mask = torch.zeros(B, N, dtype=torch.bool)
    mask[:, cutoff * per_t: num_temporal_patches * per_t] = True
    out = feats.clone()
    out[mask] = 0
    return out, mask


def jepa_block_masks(seed: int, batch: int, grid: Tuple[int, int, int], n_target_blocks: int = 4,
                     target_scale=(0.15, 0.2), target_aspect=(0.75, 1.5),
                     context_keep: float = 1.0):
    """PARITY UNPINNED (no reference symbol).  Seeded V-JEPA-style multi-block masks.

    For each batch row: `n_target_blocks` spatial rectangles (area fraction in target_scale,
    aspect in target_aspect) extended over all timesteps form the target set; the context set
    is the complement.  Every row is truncated to the batch-minimum count so index tensors are
    rectangular.  Returns (ctx_idx (B,Nc) int32 sorted, tgt_idx (B,Nt) int32 sorted).
    Uses numpy's PCG64 (`np.random.default_rng(seed)`), which is bit-stable across platforms.
    """
    Tp, Hp, Wp = grid
    rng = np.random.default_rng(seed)
    ctx_rows, tgt_rows = [], []
    for _ in range(batch):
        m = np.zeros((Hp, Wp), dtype=bool)
        for _ in range(n_target_blocks):
            area = rng.uniform(*target_scale) * Hp * Wp
            asp = rng.uniform(*target_aspect)
            bh = int(min(Hp, max(1, round(math.sqrt(area * asp)))))
            bw = int(min(Wp, max(1, round(math.sqrt(area / asp)))))
            top = int(rng.integers(0, Hp - bh + 1))
            left = int(rng.integers(0, Wp - bw + 1))
            m[top:top + bh, left:left + bw] = True
        full = np.broadcast_to(m[None], (Tp, Hp, Wp)).reshape(-1)
        tgt = np.nonzero(full)[0]
        ctx = np.nonzero(~full)[0]
        if context_keep < 1.0:
            keep = max(1, int(len(ctx) * context_keep))
            ctx = np.sort(rng.permutation(ctx)[:keep])
        ctx_rows.append(ctx)
        tgt_rows.append(tgt)
    nc = min(len(r) for r in ctx_rows)
    nt = min(len(r) for r in tgt_rows)
    ctx_idx = np.stack([r[:nc] for r in ctx_rows]).astype(np.int32)
    tgt_idx = np.stack([r[:nt] for r in tgt_rows]).astype(np.int32)
    return torch.from_numpy(ctx_idx), torch.from_numpy(tgt_idx)


def token_gather(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """PARITY UNPINNED.  out[b, j, :] = x[b, idx[b, j], :]."""
    return torch.gather(x, 1, idx.long()[..., None].expand(-1, -1, x.shape[-1]))


def token_scatter(src: torch.Tensor, idx: torch.Tensor, n_tokens: int, fill: float = 0.0) -> torch.Tensor:
    """PARITY UNPINNED.  out[b, idx[b, j], :] = src[b, j, :]; other rows = fill."""
    out = torch.full((src.shape[0], n_tokens, src.shape[-1]), fill, dtype=src.dtype)
    out.scatter_(1, idx.long()[..., None].expand(-1, -1, src.shape[-1]), src)
    return out


def ema_update(target: torch.Tensor, online: torch.Tensor, momentum: float) -> torch.Tensor:
    """PARITY UNPINNED.  target <- m*target + (1-m)*online, computed in fp32."""
    return momentum * target + (1.0 - momentum) * online


def l1_latent_loss(pred: torch.Tensor, target: torch.Tensor):
    """PARITY UNPINNED.  mean|pred - target| and d/dpred = sign(pred-target)/numel."""
    d = pred - target
    return d.abs().mean(), torch.sign(d) / d.numel()


# --------------------------------------------------------------------------
# stage-1 step math (s1:426-465), encoder features in, scalar losses out
# --------------------------------------------------------------------------
def stage1_losses(f_shuf, f_orig, f_aug, enc_tokens, order_label, heads: Dict[str, torch.Tensor],
                  Tp: int, temperature: float = 0.07, lambda_mask: float = 0.1):
    """f_* are (B,N,D) encoder outputs; enc_tokens the no-grad pass; `heads` holds the
    head parameters under the reference's nn.Sequential key names with prefixes
    temporal_classifier. / contrastive_proj. / denoise_head.  Dropout is the identity."""
    H = heads
    pooled = f_shuf.mean(dim=1)
    h = torch.relu(pooled @ H["temporal_classifier.0.weight"].t() + H["temporal_classifier.0.bias"])
    logits = h @ H["temporal_classifier.3.weight"].t() + H["temporal_classifier.3.bias"]
    l_t = F.cross_entropy(logits, order_label)

    def proj(f):
        z = torch.relu(f.mean(dim=1) @ H["contrastive_proj.0.weight"].t() + H["contrastive_proj.0.bias"])
        z = z @ H["contrastive_proj.2.weight"].t() + H["contrastive_proj.2.bias"]
        return F.normalize(z, dim=1)

    z1, z2 = proj(f_orig), proj(f_aug)
    lab = torch.arange(z1.shape[0])
    l_c = 0.5 * (F.cross_entropy(z1 @ z2.t() / temperature, lab) + F.cross_entropy(z2 @ z1.t() / temperature, lab))

    masked, mask = future_masking(enc_tokens.detach(), Tp, 0.5)
    keep = ~mask
    denom = keep.sum(dim=1, keepdim=True).clamp_min(1)
    pooled_m = (masked * keep[..., None]).sum(dim=1) / denom
    tgt = enc_tokens.detach().mean(dim=1)
    d = F.layer_norm(pooled_m, (pooled_m.shape[-1],), H["denoise_head.0.weight"], H["denoise_head.0.bias"], 1e-5)
    d = gelu_erf(d @ H["denoise_head.1.weight"].t() + H["denoise_head.1.bias"])
    d = d @ H["denoise_head.3.weight"].t() + H["denoise_head.3.bias"]
    l_m = F.mse_loss(d, tgt)
    return l_t + l_c + lambda_mask * l_m, (l_t, l_c, l_m)


def infonce_sym(p1, p2, temperature: float = 0.07, normalize: bool = True):
    """Symmetric InfoNCE of two (B, D) projection sets (s1:437-446): rows normalised (F.normalize) unless they come
    normalised, logits = z1 z2^T / tau, 0.5 * (CE(logits, diag) + CE(logits^T, diag))."""
    z1, z2 = (F.normalize(p1, dim=1), F.normalize(p2, dim=1)) if normalize else (p1, p2)
    lab = torch.arange(z1.shape[0])
    return 0.5 * (F.cross_entropy(z1 @ z2.t() / temperature, lab) + F.cross_entropy(z2 @ z1.t() / temperature, lab))


def encoder_flops_per_clip(depth: int, n_tokens: int, dim: int = 1024, patch_k: int = 1536) -> float:
    """Forward FLOPs of one clip: per token per layer 2*D*3D + 2*D*D + 2*2*D*4D + 4*N*D, plus patch embed."""
    per_tok = 2 * dim * 3 * dim + 2 * dim * dim + 4 * dim * 4 * dim + 4 * n_tokens * dim
    return depth * n_tokens * per_tok + 2 * patch_k * dim * n_tokens


# ------------------------------------------------------------------------------------------------
# predictor pieces (swm:431-555, 952-1028)
# ------------------------------------------------------------------------------------------------
def gated_delta_sequential(q, k, v, alpha, beta, state=None):
    """swm:523-555, literally: per step  S <- S @ (alpha*(I - beta*k k^T)) + beta * v k^T ;  o = S q.
    q,k,v [B,L,H,D]; alpha,beta [B,L,H]; state [B,H,D,D] or None -> (out [B,L,H,D], state)."""
    B, L, H, D = q.shape
    S = torch.zeros(B, H, D, D, dtype=q.dtype) if state is None else state.clone()
    eye = torch.eye(D, dtype=q.dtype)
    outs = []
    for t in range(L):
        kt, vt, qt = k[:, t], v[:, t], q[:, t]
        a = alpha[:, t, :, None, None]
        b = beta[:, t, :, None, None]
        decay = a * (eye - b * (kt[..., :, None] * kt[..., None, :]))
        S = S @ decay + b * (vt[..., :, None] * kt[..., None, :])
        outs.append((S @ qt[..., None])[..., 0])
    return torch.stack(outs, dim=1), S


def dwconv1d_causal_silu(x, weight, bias, normalize: bool):
    """swm:484-491: Conv1d(C, C, 4, padding=3, groups=C) cut to the first L outputs, SiLU, optional
    F.normalize over the channels.  x [B,L,C]; weight [C,1,4] or [C,4]."""
    B, L, C = x.shape
    w = weight.reshape(C, 4)
    xp = F.pad(x, (0, 0, 3, 0))                       # three zero rows in front
    y = bias + sum(w[:, i] * xp[:, i:i + L] for i in range(4))
    y = y * torch.sigmoid(y)
    return F.normalize(y, p=2, dim=-1) if normalize else y


def block_causal_mask(seq_len: int, tokens_per_timestep: int) -> torch.Tensor:
    """swm:1008-1028: bool [L,L], True = masked; a token sees its own and earlier timesteps.  (Rows of a
    trailing partial timestep are fully masked by the reference; the model never builds such a length.)"""
    step = torch.arange(seq_len) // tokens_per_timestep
    mask = step[None, :] > step[:, None]
    whole = (seq_len // tokens_per_timestep) * tokens_per_timestep
    mask[whole:, :] = True
    return mask


def mha_forward(x, P: Dict[str, torch.Tensor], prefix: str, heads: int, attn_mask=None):
    """nn.MultiheadAttention(batch_first=True) self-attention with in/out biases (swm:959-961, 986-988)."""
    B, L, D = x.shape
    hd = D // heads
    qkv = x @ P[prefix + "in_proj_weight"].t() + P[prefix + "in_proj_bias"]
    q, k, v = (t.reshape(B, L, heads, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if attn_mask is not None:
        s = s.masked_fill(attn_mask, float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, D)
    return o @ P[prefix + "out_proj.weight"].t() + P[prefix + "out_proj.bias"]


def gated_delta_layer(x, P: Dict[str, torch.Tensor], prefix: str, heads: int, state=None):
    """GatedDeltaLayer.forward in its sequential form (swm:468-521)."""
    B, L, D = x.shape
    q = dwconv1d_causal_silu(x @ P[prefix + "q_proj.weight"].t(), P[prefix + "q_conv1d.weight"], P[prefix + "q_conv1d.bias"], True)
    k = dwconv1d_causal_silu(x @ P[prefix + "k_proj.weight"].t(), P[prefix + "k_conv1d.weight"], P[prefix + "k_conv1d.bias"], True)
    v = dwconv1d_causal_silu(x @ P[prefix + "v_proj.weight"].t(), P[prefix + "v_conv1d.weight"], P[prefix + "v_conv1d.bias"], False)
    alpha = torch.sigmoid(x @ P[prefix + "a_proj.weight"].t())
    beta = torch.sigmoid(x @ P[prefix + "b_proj.weight"].t())
    inner = q.shape[-1]
    hd = inner // heads
    v, gate = v[..., :inner], v[..., inner:]
    out, state = gated_delta_sequential(q.reshape(B, L, heads, hd), k.reshape(B, L, heads, hd), v.reshape(B, L, heads, hd),
                                        alpha, beta, state)
    out = out.reshape(B, L, inner) * gate
    g = x @ P[prefix + "g_proj.weight"].t()
    out = out * (g * torch.sigmoid(g))
    out = F.layer_norm(out, (inner,), P[prefix + "o_norm.weight"], P[prefix + "o_norm.bias"], 1e-5)
    return out @ P[prefix + "o_proj.weight"].t(), state


def hybrid_predictor_block(x, P: Dict[str, torch.Tensor], prefix: str, heads: int, use_deltanet: bool,
                           attn_mask=None, state=None):
    """HybridPredictorBlock.forward (swm:975-997)."""
    h = layer_norm(x, P[prefix + "norm1.weight"], P[prefix + "norm1.bias"])
    x = x + mha_forward(h, P, prefix + "spatial_attn.", heads, attn_mask)
    if use_deltanet:
        d, state = gated_delta_layer(x, P, prefix + "temporal_layer.", heads, state)
        x = x + d
    h = layer_norm(x, P[prefix + "norm2.weight"], P[prefix + "norm2.bias"])
    h = gelu_erf(h @ P[prefix + "mlp.0.weight"].t() + P[prefix + "mlp.0.bias"])
    return x + h @ P[prefix + "mlp.2.weight"].t() + P[prefix + "mlp.2.bias"], state


def normalize_u8(frames: torch.Tensor, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> torch.Tensor:
    """ToTensor + Normalize of the reference's clip transform (stage1_surgvu_pretraining.py:241-244, 279-286) on a
    stack of frames: uint8 (B,T,H,W,3) -> f32 (B,3,T,H,W) = (u/255 - mean)/std."""
    x = frames.permute(0, 4, 1, 2, 3).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1, 1)
    return (x - m) / s
