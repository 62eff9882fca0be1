"""Tensor-level wrappers over the C ABI: shape/dtype checks on the host, raw pointers and the current
HIP stream across the boundary.  Nothing here computes; torch is used for memory and streams only."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check

EPI_BF16, EPI_BIAS_GELU, EPI_F32, EPI_GELU_BWD, EPI_GELU_ONLY, EPI_BIAS_GELU_GRAD, EPI_MUL_AUX = 0, 1, 2, 3, 4, 5, 6
BF16, F32 = torch.bfloat16, torch.float32


_current_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)     # the binding itself: this runs per tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str, inner_contig: bool = True):
    if not t.is_cuda:
        raise _lib.VitAmdError(f"{name}: tensor must live on a HIP device (vit_amd has no CPU path)")
    if t.device.index != _current_device():
        # kernels are launched on the CURRENT device's stream: a tensor of another device would be a wild pointer there
        raise _lib.VitAmdError(f"{name}: tensor is on {t.device} but the current device is cuda:{_current_device()}; "
                               "wrap the call in torch.cuda.device(tensor.device) (VideoViTEncoder.forward does)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if inner_contig and t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def _ld(t: torch.Tensor) -> int:
    """Leading dimension of a 2-D row-major view (rows may be strided)."""
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ------------------------------------------------------------------ GEMMs
def gemm_nt(x: torch.Tensor, w: torch.Tensor, epilogue: int, out: torch.Tensor, out2: Optional[torch.Tensor] = None,
            bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
            aux: Optional[torch.Tensor] = None, tile: int = 0, drop_key: int = 0, drop_p: float = 0.0) -> torch.Tensor:
    """out[M,N] = x[M,K] @ w[N,K]^T with a fused epilogue (see include/vit_amd.h).  drop_p > 0 applies the dropout
    of `dropout(…, drop_key, drop_p)` inside the epilogue: to the GELU output (EPI_BIAS_GELU, EPI_GELU_ONLY), to acc + bias before
    the residual (EPI_F32), to the product with the GELU derivative (EPI_GELU_BWD)."""
    _need(x, BF16, "x"); _need(w, BF16, "w")
    M, K = x.shape
    N, K2 = w.shape
    if K2 != K:
        raise ValueError(f"gemm_nt: K mismatch {K} vs {K2}")
    _need(out, F32 if epilogue == EPI_F32 else BF16, "out")
    if tuple(out.shape) != (M, N):
        raise ValueError(f"gemm_nt: out shape {tuple(out.shape)} != {(M, N)}")
    ldc = _ld(out)
    for t, nm, dt in ((out2, "out2", BF16), (aux, "aux", BF16), (resid, "resid", F32)):
        if t is not None:
            _need(t, dt, nm)
            if tuple(t.shape) != (M, N) or _ld(t) != ldc:
                raise ValueError(f"gemm_nt: {nm} must match out's shape and leading dimension")
    if bias is not None:
        _need(bias, F32, "bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise ValueError("gemm_nt: bias must be a contiguous [N] vector")
    lib = _lib.load()
    if drop_p > 0:
        check(lib.va_gemm_nt_dropout(_p(x), _p(w), M, N, K, _ld(x), _ld(w), epilogue, _p(out), _p(out2), _p(bias),
                                     _p(resid), _p(aux), ldc, tile, drop_key & 0xFFFFFFFF, float(drop_p), _stream()),
              "va_gemm_nt_dropout")
        return out
    check(lib.va_gemm_nt(_p(x), _p(w), M, N, K, _ld(x), _ld(w), epilogue, _p(out), _p(out2), _p(bias), _p(resid),
                         _p(aux), ldc, tile, _stream()), "va_gemm_nt")
    return out


# Weight gradients are summed over slices of the token dimension with f32 atomics, so their last bits depend on the
# order in which workgroups finish.  In deterministic mode every output tile is produced by ONE workgroup (one slice:
# a single add per element onto the incoming value) and bias gradients come from the same kernel run against a column
# of ones, so two runs give bit-identical gradients -- at several times the cost of the weight-gradient GEMMs.  A debug
# switch (run-twice comparisons, bisecting), not a training mode.
_deterministic = os.environ.get("VIT_AMD_DETERMINISTIC", "0") == "1"
_ones_cache = {}


def set_deterministic(on: bool) -> None:
    global _deterministic
    _deterministic = bool(on)


def is_deterministic() -> bool:
    return _deterministic


def _ones_column(M: int, device) -> torch.Tensor:
    key = (M, str(device))
    t = _ones_cache.get(key)
    if t is None:
        _ones_cache.clear()                      # one size at a time: the buffer is M x 128 bf16
        t = _ones_cache[key] = torch.ones(M, 128, dtype=BF16, device=device)
    return t


def gemm_tn_accum(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, dbias: Optional[torch.Tensor] = None,
                  split_k: int = 0) -> torch.Tensor:
    """dw[N,K] (f32) += dy[M,N]^T @ x[M,K]; optionally dbias[N] (f32) += dy.sum(0) in the same pass."""
    if _deterministic:
        split_k = 1
        if dbias is not None:
            sums = torch.zeros(dy.shape[1], 128, dtype=F32, device=dy.device)
            gemm_tn_accum(dy, _ones_column(dy.shape[0], dy.device), sums, None)
            dbias.add_(sums[:, 0])
            dbias = None
    _need(dy, BF16, "dy"); _need(x, BF16, "x"); _need(dw, F32, "dw")
    M, N = dy.shape
    M2, K = x.shape
    if M2 != M or tuple(dw.shape) != (N, K):
        raise ValueError(f"gemm_tn_accum: shapes dy{tuple(dy.shape)} x{tuple(x.shape)} dw{tuple(dw.shape)}")
    if dbias is not None:
        _need(dbias, F32, "dbias")
        if dbias.numel() != N or not dbias.is_contiguous():
            raise ValueError("gemm_tn_accum: dbias must be a contiguous [N] vector")
    lib = _lib.load()
    check(lib.va_gemm_tn_accum(_p(dy), _p(x), M, N, K, _ld(dy), _ld(x), _p(dw), _ld(dw), _p(dbias), split_k,
                               _stream()), "va_gemm_tn_accum")
    return dw


def colsum_accum(y: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _need(y, BF16, "y"); _need(out, F32, "out")
    M, N = y.shape
    if out.numel() != N or not out.is_contiguous():
        raise ValueError("colsum_accum: out must be a contiguous [N] vector")
    check(_lib.load().va_colsum_bf16_accum(_p(y), M, N, _ld(y), _p(out), _stream()), "va_colsum_bf16_accum")
    return out


# ------------------------------------------------------------------ patch gather
def tubelet_gather(video: torch.Tensor, tubelet: Tuple[int, int, int], patches: Optional[torch.Tensor] = None):
    _need(video, F32, "video")
    if video.dim() != 5 or not video.is_contiguous():
        raise ValueError("tubelet_gather: video must be a contiguous (B,C,T,H,W) tensor")
    B, C, T, H, W = video.shape
    tt, th, tw = tubelet
    M = B * (T // tt) * (H // th) * (W // tw)
    Kd = C * tt * th * tw
    if patches is None:
        patches = torch.empty(M, Kd, dtype=BF16, device=video.device)
    _need(patches, BF16, "patches")
    if tuple(patches.shape) != (M, Kd) or not patches.is_contiguous():
        raise ValueError("tubelet_gather: bad patches buffer")
    check(_lib.load().va_tubelet_gather(_p(video), _p(patches), B, C, T, H, W, tt, th, tw, _stream()), "va_tubelet_gather")
    return patches


def tubelet_scatter(patches: torch.Tensor, video_shape, tubelet: Tuple[int, int, int]) -> torch.Tensor:
    """Gradient of the patch matrix (f32 [tokens, C*tt*th*tw]) -> gradient of the video (B,C,T,H,W) f32."""
    _need(patches, F32, "patches")
    B, C, T, H, W = (int(v) for v in video_shape)
    tt, th, tw = tubelet
    M = B * (T // tt) * (H // th) * (W // tw)
    Kd = C * tt * th * tw
    if tuple(patches.shape) != (M, Kd) or not patches.is_contiguous():
        raise ValueError("tubelet_scatter: bad patches buffer")
    exact = T % tt == 0 and H % th == 0 and W % tw == 0
    video = (torch.empty if exact else torch.zeros)(B, C, T, H, W, dtype=F32, device=patches.device)
    check(_lib.load().va_tubelet_scatter(_p(patches), _p(video), B, C, T, H, W, tt, th, tw, _stream()), "va_tubelet_scatter")
    return video


# ------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, out, mean=None, rstd=None, eps: float = 1e-6):
    _need(x, F32, "x"); _need(gamma, F32, "gamma"); _need(beta, F32, "beta")
    M, D = x.shape
    if not (x.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (M, D)):
        raise ValueError("layernorm_fwd: x/out must be contiguous [M,D]")
    if out.dtype not in (BF16, F32):
        raise TypeError("layernorm_fwd: out must be bf16 or f32")
    if (mean is None) != (rstd is None):
        raise ValueError("layernorm_fwd: mean and rstd go together (both None for a forward that keeps no statistics)")
    if mean is not None:
        _need(mean, F32, "mean"); _need(rstd, F32, "rstd")
        if mean.numel() != M or rstd.numel() != M:
            raise ValueError("layernorm_fwd: mean/rstd must have M elements")
    check(_lib.load().va_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(out), int(out.dtype == BF16), _p(mean), _p(rstd),
                                       M, D, eps, _stream()), "va_layernorm_fwd")
    return out


def layernorm_bwd(dy, x, gamma, mean, rstd, resid_grad, dx_out, dx_bf16, dgamma, dbeta, drop_key: int = 0,
                  drop_p: float = 0.0, copy_colsum: Optional[torch.Tensor] = None):
    """drop_p > 0: dx_bf16 receives dropout(dx_out; drop_key, drop_p) instead of a plain bf16 copy.
    copy_colsum [D] f32: receives the column sums of dx_bf16 (overwritten)."""
    _need(x, F32, "x"); _need(dx_out, F32, "dx_out"); _need(dgamma, F32, "dgamma"); _need(dbeta, F32, "dbeta")
    M, D = x.shape
    if dy.dtype not in (BF16, F32) or tuple(dy.shape) != (M, D) or not dy.is_contiguous():
        raise ValueError("layernorm_bwd: dy must be contiguous [M,D] bf16/f32")
    for t in (resid_grad, dx_bf16):
        if t is not None and (tuple(t.shape) != (M, D) or not t.is_contiguous()):
            raise ValueError("layernorm_bwd: optional tensors must be contiguous [M,D]")
    if resid_grad is not None:
        _need(resid_grad, F32, "resid_grad")
    if dx_bf16 is not None:
        _need(dx_bf16, BF16, "dx_bf16")
    lib = _lib.load()
    rows = lib.va_layernorm_bwd_partial_rows(M)
    if copy_colsum is not None:
        _need(copy_colsum, F32, "copy_colsum")
        if copy_colsum.numel() != D or not copy_colsum.is_contiguous() or dx_bf16 is None:
            raise ValueError("layernorm_bwd: copy_colsum must be a contiguous [D] vector and needs dx_bf16")
    partial = torch.empty(rows * (3 if copy_colsum is not None else 2) * D, dtype=F32, device=x.device)
    if drop_p > 0 or copy_colsum is not None:
        check(lib.va_layernorm_bwd_dropout(_p(dy), int(dy.dtype == BF16), _p(x), _p(gamma), _p(mean), _p(rstd),
                                           _p(resid_grad), _p(dx_out), _p(dx_bf16), _p(partial), _p(dgamma), _p(dbeta),
                                           M, D, drop_key & 0xFFFFFFFF, float(drop_p), _p(copy_colsum), _stream()),
              "va_layernorm_bwd_dropout")
        return dx_out
    check(lib.va_layernorm_bwd(_p(dy), int(dy.dtype == BF16), _p(x), _p(gamma), _p(mean), _p(rstd), _p(resid_grad),
                               _p(dx_out), _p(dx_bf16), _p(partial), _p(dgamma), _p(dbeta), M, D, _stream()),
          "va_layernorm_bwd")
    return dx_out


# ------------------------------------------------------------------ RoPE
def rope_tables_device(inv_freq_t, inv_freq_h, inv_freq_w, grid: Tuple[int, int, int], temporal_scale: float,
                       device) -> torch.Tensor:
    """Half-width cos|sin tables in the layout va_rope3d_qk_inplace expects.  The angle arithmetic is
    done in f32 on the host exactly as the reference's VideoRoPE3D.forward does (swm:192-222):
    position (scaled for t) times inverse frequency, then cos / sin."""
    Tp, Hp, Wp = grid
    cos_parts, sin_parts = [], []
    for n, inv, sc in ((Tp, inv_freq_t, temporal_scale), (Hp, inv_freq_h, 1.0), (Wp, inv_freq_w, 1.0)):
        pos = torch.arange(n).float() * sc
        ang = torch.outer(pos, inv.detach().float().cpu())
        cos_parts.append(ang.cos().reshape(-1))
        sin_parts.append(ang.sin().reshape(-1))
    return torch.cat(cos_parts + sin_parts).to(device)


def rope3d_qk_(qkv: torch.Tensor, tables: torch.Tensor, grid: Tuple[int, int, int], inverse: bool = False):
    _need(qkv, BF16, "qkv"); _need(tables, F32, "tables")
    M, W3 = qkv.shape
    Tp, Hp, Wp = grid
    if W3 != 3072 or not qkv.is_contiguous():
        raise ValueError("rope3d_qk_: qkv must be a contiguous [M, 3072] buffer")
    if tables.numel() != 2 * (Tp * 128 + Hp * 192 + Wp * 192):
        raise ValueError("rope3d_qk_: table size does not match the grid")
    check(_lib.load().va_rope3d_qk_inplace(_p(qkv), _p(tables), M, Tp, Hp, Wp, int(inverse), _stream()),
          "va_rope3d_qk_inplace")
    return qkv


def rope_pos_code(batch: int, grid: Tuple[int, int, int], device) -> torch.Tensor:
    """int32 [batch * Tp*Hp*Wp]: t | h << 10 | w << 20 of every token row (token order t, h, w; swm:225-240)."""
    Tp, Hp, Wp = grid
    if max(grid) >= 1024:
        raise ValueError("rope_pos_code: grid coordinates must stay below 1024")
    n = torch.arange(Tp * Hp * Wp, dtype=torch.int64)
    code = (n // (Hp * Wp)) | (((n // Wp) % Hp) << 10) | ((n % Wp) << 20)
    return code.to(torch.int32).repeat(batch).to(device)


# VIT_AMD_FUSED_ROPE=0 makes the encoder take the two-pass route (projection GEMM, then the rotation pass): for A/B timing
_fused_rope = os.environ.get("VIT_AMD_FUSED_ROPE", "1") != "0"


def gemm_nt_qkv_rope(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, tables: torch.Tensor, pos_code: torch.Tensor,
                     grid: Tuple[int, int, int]) -> bool:
    """out[M, 3072] (bf16) = rotary(x[M, K] @ w[3072, K]^T): the QKV projection with the rotation of `rope3d_qk_` applied
    in the GEMM's epilogue.  Returns False when the fused kernel cannot take the call (small M, or no persistent-kernel
    ticket set left for the stream): the caller then runs gemm_nt + rope3d_qk_."""
    _need(x, BF16, "x"); _need(w, BF16, "w"); _need(out, BF16, "out"); _need(tables, F32, "tables")
    _need(pos_code, torch.int32, "pos_code")
    M, K = x.shape
    Tp, Hp, Wp = grid
    if tuple(w.shape) != (3072, K) or tuple(out.shape) != (M, 3072) or pos_code.numel() != M:
        raise ValueError("gemm_nt_qkv_rope: inconsistent shapes")
    if tables.numel() != 2 * (Tp * 128 + Hp * 192 + Wp * 192):
        raise ValueError("gemm_nt_qkv_rope: table size does not match the grid")
    if not _fused_rope or M < 2048 or K < 192 or K % 64:
        return False
    lib = _lib.load()
    rc = lib.va_gemm_nt_qkv_rope(_p(x), _p(w), M, K, _ld(x), _ld(w), _p(out), _ld(out), _p(tables), _p(pos_code), Tp, Hp, Wp,
                                 _stream())
    if rc == -3:          # VA_ERR_UNSUPPORTED_SHAPE: no ticket set left for this stream
        return False
    check(rc, "va_gemm_nt_qkv_rope")
    return True


def rope3d_qk_indexed_(qkv: torch.Tensor, cache, pos: torch.Tensor, n_tokens: int, transpose: bool = False):
    """General rotation of the Q and K thirds of a packed [M, 3072] bf16 buffer (swm:265-281): `cache` =
    ((cos_t, sin_t), (cos_h, sin_h), (cos_w, sin_w)) full-width f32 device tables as VideoRoPE3D.forward returns
    them, `pos` int32 [3, n_tokens] table rows of every token (validated by the caller)."""
    _need(qkv, BF16, "qkv"); _need(pos, torch.int32, "pos")
    M, W3 = qkv.shape
    if W3 != 3072 or not qkv.is_contiguous() or M % n_tokens:
        raise ValueError("rope3d_qk_indexed_: qkv must be a contiguous [M, 3072] buffer of whole sequences")
    if tuple(pos.shape) != (3, n_tokens) or not pos.is_contiguous():
        raise ValueError("rope3d_qk_indexed_: pos must be a contiguous int32 [3, N] tensor")
    flat = []
    for (c, s_), width in zip(cache, (256, 384, 384)):
        for t in (c, s_):
            _need(t, F32, "rope table")
            if t.dim() != 2 or t.shape[1] != width or not t.is_contiguous():
                raise ValueError(f"rope3d_qk_indexed_: tables must be contiguous [len, {width}] f32")
            flat.append(t)
    check(_lib.load().va_rope3d_qk_indexed(_p(qkv), *(_p(t) for t in flat), _p(pos), M, n_tokens, int(transpose),
                                           _stream()), "va_rope3d_qk_indexed")
    return qkv


# ------------------------------------------------------------------ attention
class PackedMask:
    """An explicit boolean attention mask (True = do not attend, swm:316-323) in the bit-packed form the kernels read:
    `by_query` / `by_key` int32 [batches, heads, N, W] with stride_b / stride_h = 0 where the mask is shared."""

    def __init__(self, by_query, by_key, stride_b: int, stride_h: int, n: int):
        self.by_query, self.by_key, self.stride_b, self.stride_h, self.n = by_query, by_key, stride_b, stride_h, n


def attn_mask_pack(mask: torch.Tensor, B: int, H: int) -> PackedMask:
    """mask: bool [N, N], [B, N, N], or 4-D [B or 1, H or 1, N, N] (what `scores.masked_fill(attn_mask, -inf)` takes
    after the reference's unsqueezing), on the device.  One kernel packs it by query and by key."""
    if mask.dtype != torch.bool or not mask.is_cuda:
        raise ValueError("attn_mask_pack: a boolean device tensor is required")
    if mask.dim() == 2:
        mask = mask[None, None]
    elif mask.dim() == 3:
        mask = mask[:, None]
    if mask.dim() != 4 or mask.shape[-1] != mask.shape[-2] or mask.shape[0] not in (1, B) or mask.shape[1] not in (1, H):
        raise ValueError(f"attn_mask_pack: mask of shape {tuple(mask.shape)} does not broadcast to ({B}, {H}, N, N)")
    mb, mh, N = mask.shape[0], mask.shape[1], mask.shape[-1]
    lib = _lib.load()
    words = int(lib.va_attn_mask_words(N))
    bytes_ = mask.contiguous().view(torch.uint8)
    by_q = torch.empty(mb * mh * words, dtype=torch.int32, device=mask.device)
    by_k = torch.empty_like(by_q)
    check(lib.va_attn_mask_pack(_p(bytes_), mb * mh, N, _p(by_q), _p(by_k), _stream()), "va_attn_mask_pack")
    return PackedMask(by_q, by_k, mh * words if mb > 1 else 0, words if mh > 1 else 0, N)


def attn_fwd(qkv: torch.Tensor, B: int, N: int, H: int, out: torch.Tensor, lse2: torch.Tensor,
             tokens_per_timestep: int = 0, drop_key: int = 0, drop_p: float = 0.0, mask: Optional[PackedMask] = None):
    """qkv: packed [B*N, 3*D] bf16 (q | k | v thirds); out [B*N, D] bf16; lse2 [B*H, N] f32.
    tokens_per_timestep > 0 selects the predictor's block-causal mask (forward only).
    drop_p > 0: training-mode dropout on the attention probabilities, masks derived from drop_key.
    mask: an explicit boolean mask packed by `attn_mask_pack` (no dropout with it)."""
    _need(qkv, BF16, "qkv"); _need(out, BF16, "out"); _need(lse2, F32, "lse2")
    M, W3 = qkv.shape
    D = W3 // 3
    if M != B * N or tuple(out.shape) != (M, D) or lse2.numel() != B * H * N or D % H:
        raise ValueError("attn_fwd: inconsistent shapes")
    hd = D // H
    es = qkv.element_size()
    ptrs = (qkv.data_ptr(), qkv.data_ptr() + D * es, qkv.data_ptr() + 2 * D * es, _p(out), _p(lse2))
    if mask is not None:
        if drop_p > 0 or tokens_per_timestep > 0 or mask.n != N:
            raise ValueError("attn_fwd: an explicit mask excludes dropout and the block-causal mode, and must be N x N")
        check(_lib.load().va_attn_fwd_masked(*ptrs, B, N, H, hd, _ld(qkv), _ld(out), _p(mask.by_query), mask.stride_b,
                                             mask.stride_h, _stream()), "va_attn_fwd_masked")
    elif tokens_per_timestep > 0:
        if drop_p > 0:
            raise ValueError("attn_fwd: dropout is not available with the block-causal mask")
        check(_lib.load().va_attn_fwd_block_causal(*ptrs, B, N, H, hd, _ld(qkv), _ld(out), tokens_per_timestep,
                                                   _stream()), "va_attn_fwd_block_causal")
    elif drop_p > 0:
        check(_lib.load().va_attn_fwd_dropout(*ptrs, B, N, H, hd, _ld(qkv), _ld(out), drop_key & 0xffffffff,
                                              float(drop_p), _stream()), "va_attn_fwd_dropout")
    else:
        check(_lib.load().va_attn_fwd(*ptrs, B, N, H, hd, _ld(qkv), _ld(out), _stream()), "va_attn_fwd")
    return out


def attn_bwd(qkv, out, d_out, lse2, dqkv, B: int, N: int, H: int, drop_key: int = 0, drop_p: float = 0.0,
             mask: Optional[PackedMask] = None):
    """drop_key / drop_p: the values the forward ran with (the masks are regenerated, not stored); mask: the forward's."""
    _need(qkv, BF16, "qkv"); _need(out, BF16, "out"); _need(d_out, BF16, "d_out"); _need(dqkv, BF16, "dqkv")
    M, W3 = qkv.shape
    D = W3 // 3
    if tuple(dqkv.shape) != (M, W3) or tuple(d_out.shape) != (M, D) or _ld(d_out) != _ld(out):
        raise ValueError("attn_bwd: inconsistent shapes")
    hd = D // H
    es = 2
    lib = _lib.load()
    delta = torch.empty(lib.va_attn_bwd_workspace_bytes(B, N, H) // 4, dtype=F32, device=qkv.device)   # workspace
    args = (qkv.data_ptr(), qkv.data_ptr() + D * es, qkv.data_ptr() + 2 * D * es, _p(out), _p(d_out), _p(lse2),
            _p(delta), dqkv.data_ptr(), dqkv.data_ptr() + D * es, dqkv.data_ptr() + 2 * D * es, B, N, H, hd, _ld(qkv),
            _ld(out), _ld(dqkv))
    if mask is not None:
        if drop_p > 0 or mask.n != N:
            raise ValueError("attn_bwd: an explicit mask excludes dropout and must be N x N")
        check(lib.va_attn_bwd_masked(*args, _p(mask.by_query), _p(mask.by_key), mask.stride_b, mask.stride_h, _stream()),
              "va_attn_bwd_masked")
    elif drop_p > 0:
        check(lib.va_attn_bwd_dropout(*args, drop_key & 0xffffffff, float(drop_p), _stream()), "va_attn_bwd_dropout")
    else:
        check(lib.va_attn_bwd(*args, _stream()), "va_attn_bwd")
    return dqkv


def dropout(x: torch.Tensor, key: int, p: float, out: Optional[torch.Tensor] = None,
            resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = [resid +] keep(x) / (1 - p) with the counter-based masks of the library (training-mode nn.Dropout,
    swm:700-702).  x / out: contiguous f32 or bf16 of one shape (out defaults to x, in place); resid f32."""
    out = x if out is None else out
    for t, name in ((x, "x"), (out, "out")):
        if t.dtype not in (BF16, F32) or not t.is_contiguous() or not t.is_cuda:
            raise ValueError(f"dropout: {name} must be a contiguous f32/bf16 device tensor")
    if out.shape != x.shape or (resid is not None and (resid.shape != x.shape or resid.dtype != F32
                                                       or not resid.is_contiguous())):
        raise ValueError("dropout: shape / dtype mismatch")
    check(_lib.load().va_dropout(_p(x), int(x.dtype == BF16), _p(out), int(out.dtype == BF16),
                                 _p(resid) if resid is not None else None, x.numel(), key & 0xffffffff, float(p),
                                 _stream()), "va_dropout")
    return out


# ------------------------------------------------------------------ predictor (inference)
def gated_delta_seq(q, k, v, alpha, beta, state=None):
    """q,k,v f32 [B,L,H,hd]; alpha,beta f32 [B,L,H]; state f32 [B,H,hd,hd] or None -> (out [B,L,H,hd], new state)."""
    for name, x in (("q", q), ("k", k), ("v", v), ("alpha", alpha), ("beta", beta)):
        _need(x, F32, name)
        if not x.is_contiguous():
            raise ValueError(f"gated_delta_seq: {name} must be contiguous")
    B, L, H, hd = q.shape
    if k.shape != q.shape or v.shape != q.shape or tuple(alpha.shape) != (B, L, H) or tuple(beta.shape) != (B, L, H):
        raise ValueError("gated_delta_seq: inconsistent shapes")
    if state is not None:
        _need(state, F32, "state")
        if tuple(state.shape) != (B, H, hd, hd) or not state.is_contiguous():
            raise ValueError("gated_delta_seq: state must be contiguous [B,H,hd,hd]")
    out = torch.empty_like(q)
    new_state = torch.empty(B, H, hd, hd, dtype=F32, device=q.device)
    check(_lib.load().va_gated_delta_seq(_p(q), _p(k), _p(v), _p(alpha), _p(beta), None if state is None else _p(state),
                                         _p(out), _p(new_state), B, L, H, hd, _stream()), "va_gated_delta_seq")
    return out, new_state


def dwconv1d_causal_silu(x, weight, bias, normalize: bool):
    """x f32 [B,L,C]; weight f32 [C,1,4] (nn.Conv1d layout) or [C,4]; bias [C] -> silu(causal conv) [, row-L2-normalised]."""
    _need(x, F32, "x"); _need(weight, F32, "weight"); _need(bias, F32, "bias")
    B, L, C = x.shape
    if weight.numel() != 4 * C or bias.numel() != C or not (x.is_contiguous() and weight.is_contiguous()):
        raise ValueError("dwconv1d_causal_silu: inconsistent shapes")
    out = torch.empty_like(x)
    check(_lib.load().va_dwconv1d_causal_silu(_p(x), _p(weight), _p(bias), _p(out), B, L, C, int(bool(normalize)),
                                              _stream()), "va_dwconv1d_causal_silu")
    return out


def gate_layernorm_fwd(o, gate, g, gamma, beta, out, eps: float = 1e-5):
    """out (bf16 [M,D]) = LayerNorm((o * gate) * silu(g)); o, gate f32 [M,D] contiguous; g f32 [M,D] contiguous, or bf16
    [M,D] with any row stride (a column block of the merged projection output) (swm:511-519)."""
    M, D = o.shape
    for name, x in (("o", o), ("gate", gate)):
        _need(x, F32, name)
        if tuple(x.shape) != (M, D) or not x.is_contiguous():
            raise ValueError(f"gate_layernorm_fwd: {name} must be contiguous [M,D]")
    _need(gamma, F32, "gamma"); _need(beta, F32, "beta"); _need(out, BF16, "out")
    if tuple(out.shape) != (M, D) or not out.is_contiguous() or gamma.numel() != D or beta.numel() != D:
        raise ValueError("gate_layernorm_fwd: inconsistent shapes")
    if tuple(g.shape) != (M, D):
        raise ValueError("gate_layernorm_fwd: g must be [M,D]")
    if g.dtype == BF16:
        _need(g, BF16, "g")
        check(_lib.load().va_gate_layernorm_fwd_g16(_p(o), _p(gate), _p(g), _ld(g), _p(gamma), _p(beta), _p(out), M, D, eps,
                                                    _stream()), "va_gate_layernorm_fwd_g16")
        return out
    _need(g, F32, "g")
    if not g.is_contiguous():
        raise ValueError("gate_layernorm_fwd: an f32 g must be contiguous [M,D]")
    check(_lib.load().va_gate_layernorm_fwd(_p(o), _p(gate), _p(g), _p(gamma), _p(beta), _p(out), M, D, eps, _stream()),
          "va_gate_layernorm_fwd")
    return out


# ------------------------------------------------------------------ loss heads (value and gradient per call)
def infonce_sym(p1: torch.Tensor, p2: torch.Tensor, temperature: float, normalize: bool = True):
    """Symmetric InfoNCE of two [B, D] f32 projection sets (s1:434-446).  Returns (loss [1], dloss/dp1, dloss/dp2)."""
    _need(p1, F32, "p1"); _need(p2, F32, "p2")
    if p1.dim() != 2 or p1.shape != p2.shape or not (p1.is_contiguous() and p2.is_contiguous()):
        raise ValueError("infonce_sym: p1, p2 must be contiguous [B, D] of one shape")
    B, D = p1.shape
    lib = _lib.load()
    ws = torch.empty(lib.va_infonce_workspace_floats(B, D), dtype=F32, device=p1.device)
    loss = torch.empty(1, dtype=F32, device=p1.device)
    g1, g2 = torch.empty_like(p1), torch.empty_like(p2)
    check(lib.va_infonce_sym(_p(p1), _p(p2), _p(loss), _p(g1), _p(g2), _p(ws), B, D, float(temperature), int(bool(normalize)),
                             _stream()), "va_infonce_sym")
    return loss, g1, g2


def normalize_rows(p: torch.Tensor):
    """F.normalize(p, dim=1) of a contiguous f32 [B, D] matrix -> (z, inv), inv [B] the factor applied to each row."""
    _need(p, F32, "p")
    if p.dim() != 2 or not p.is_contiguous():
        raise ValueError("normalize_rows: p must be a contiguous matrix")
    B, D = p.shape
    z, inv = torch.empty_like(p), torch.empty(B, dtype=F32, device=p.device)
    check(_lib.load().va_normalize_rows_fwd(_p(p), _p(z), _p(inv), B, D, _stream()), "va_normalize_rows_fwd")
    return z, inv


def normalize_rows_bwd(dz: torch.Tensor, z: torch.Tensor, inv: torch.Tensor):
    _need(dz, F32, "dz"); _need(z, F32, "z"); _need(inv, F32, "inv")
    if dz.shape != z.shape or not (dz.is_contiguous() and z.is_contiguous()) or inv.numel() != z.shape[0]:
        raise ValueError("normalize_rows_bwd: inconsistent shapes")
    dp = torch.empty_like(z)
    check(_lib.load().va_normalize_rows_bwd(_p(dz), _p(z), _p(inv), _p(dp), z.shape[0], z.shape[1], _stream()),
          "va_normalize_rows_bwd")
    return dp


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, grad_scale: Optional[float] = None):
    """Mean cross-entropy of [B, C] f32 logits against int64 labels (s1:428).  Returns (loss [1], (softmax - onehot) *
    grad_scale); grad_scale None = 1/B, the gradient of the mean."""
    _need(logits, F32, "logits")
    if logits.dim() != 2 or not logits.is_contiguous() or labels.dtype != torch.int64 or labels.shape != logits.shape[:1] \
            or not labels.is_cuda or not labels.is_contiguous():
        raise ValueError("cross_entropy: logits contiguous [B, C] f32, labels contiguous int64 [B] on the device")
    B, C = logits.shape
    ws = torch.empty(B, dtype=F32, device=logits.device)
    loss = torch.empty(1, dtype=F32, device=logits.device)
    dx = torch.empty_like(logits)
    check(_lib.load().va_cross_entropy(_p(logits), _p(labels), _p(loss), _p(dx), _p(ws), B, C,
                                       1.0 / B if grad_scale is None else float(grad_scale), _stream()), "va_cross_entropy")
    return loss, dx


def mse_loss(x: torch.Tensor, y: torch.Tensor, grad_scale: Optional[float] = None):
    """mean((x - y)^2) (s1:459).  Returns (loss [1], (x - y) * grad_scale); grad_scale None = 2/n, the gradient of the mean."""
    _need(x, F32, "x"); _need(y, F32, "y")
    if x.shape != y.shape or not (x.is_contiguous() and y.is_contiguous()):
        raise ValueError("mse_loss: x, y must be contiguous and of one shape")
    n = x.numel()
    lib = _lib.load()
    ws = torch.empty(lib.va_mse_workspace_floats(n), dtype=F32, device=x.device)
    loss = torch.empty(1, dtype=F32, device=x.device)
    dx = torch.empty_like(x)
    check(lib.va_mse_loss(_p(x), _p(y), _p(loss), _p(dx), _p(ws), n, 2.0 / n if grad_scale is None else float(grad_scale),
                          _stream()), "va_mse_loss")
    return loss, dx


def delta_gates_fwd(x2d, wa, wb, x_bf16=None):
    """alpha, beta = sigmoid(x @ wa^T), sigmoid(x @ wb^T): x f32 [M, 1024]; wa, wb f32 [H, 1024] -> two f32 [M, H]
    (swm:494-495).  x_bf16: optional contiguous bf16 [M, 1024] that receives the cast of x from the same pass."""
    _need(x2d, F32, "x"); _need(wa, F32, "wa"); _need(wb, F32, "wb")
    M, D = x2d.shape
    H = wa.shape[0]
    if tuple(wa.shape) != (H, D) or tuple(wb.shape) != (H, D) or not (x2d.is_contiguous() and wa.is_contiguous()
                                                                      and wb.is_contiguous()):
        raise ValueError("delta_gates_fwd: inconsistent shapes")
    alpha = torch.empty(M, H, dtype=F32, device=x2d.device)
    beta = torch.empty(M, H, dtype=F32, device=x2d.device)
    if x_bf16 is not None:
        _need(x_bf16, BF16, "x_bf16")
        if tuple(x_bf16.shape) != (M, D) or not x_bf16.is_contiguous():
            raise ValueError("delta_gates_fwd: x_bf16 must be contiguous [M, D]")
    check(_lib.load().va_delta_gates_fwd(_p(x2d), _p(wa), _p(wb), _p(alpha), _p(beta), _p(x_bf16), M, D, H, _stream()),
          "va_delta_gates_fwd")
    return alpha, beta


def dwconv1d_causal_silu_bf16(x2d, B: int, L: int, weight, bias, normalize: bool):
    """x2d bf16 [B*L, 1024] with any row stride (a column block of the merged projection output), rows ordered (b, t);
    weight f32 [C,1,4] or [C,4]; bias [C] -> f32 [B, L, C] = silu(causal conv) [, row-L2-normalised]."""
    _need(x2d, BF16, "x"); _need(weight, F32, "weight"); _need(bias, F32, "bias")
    M, C = x2d.shape
    if M != B * L or weight.numel() != 4 * C or bias.numel() != C or not weight.is_contiguous():
        raise ValueError("dwconv1d_causal_silu_bf16: inconsistent shapes")
    out = torch.empty(B, L, C, dtype=F32, device=x2d.device)
    check(_lib.load().va_dwconv1d_causal_silu_bf16(_p(x2d), _ld(x2d), _p(weight), _p(bias), _p(out), B, L, C,
                                                   int(bool(normalize)), _stream()), "va_dwconv1d_causal_silu_bf16")
    return out


# ------------------------------------------------------------------ casts
def cast_bf16(src: torch.Tensor, dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need(src, F32, "src")
    if not src.is_contiguous():
        raise ValueError("cast_bf16: src must be contiguous")
    if dst is None:
        dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    check(_lib.load().va_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "va_cast_f32_bf16")
    return dst


def cast_transpose_bf16(w: torch.Tensor, dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need(w, F32, "w")
    if w.dim() != 2 or not w.is_contiguous():
        raise ValueError("cast_transpose_bf16: w must be a contiguous matrix")
    R, C = w.shape
    if dst is None:
        dst = torch.empty(C, R, dtype=BF16, device=w.device)
    check(_lib.load().va_cast_transpose_f32_bf16(_p(w), _p(dst), R, C, _stream()), "va_cast_transpose_f32_bf16")
    return dst


class CastPlan:
    """A fixed list of (f32 matrix -> bf16 copy [, bf16 transposed copy]) jobs executed by one launch.  The job
    table lives on the device; build a new plan when a source or destination buffer moves."""

    def __init__(self, triples):
        """triples: (src f32 [R,C] contiguous, dst bf16 [R,C] or None, dst_t bf16 [C,R] or None), R, C % 64 == 0."""
        import struct
        rec, first = [], 0
        self._keep = []
        for src, dst, dst_t in triples:
            _need(src, F32, "src")
            if src.dim() != 2 or not src.is_contiguous() or src.shape[0] % 64 or src.shape[1] % 64:
                raise ValueError("CastPlan: sources must be contiguous matrices with sides that are multiples of 64")
            R, C = src.shape
            for t, shp, nm in ((dst, (R, C), "dst"), (dst_t, (C, R), "dst_t")):
                if t is not None:
                    _need(t, BF16, nm)
                    if tuple(t.shape) != shp or not t.is_contiguous() or t.device != src.device:
                        raise ValueError(f"CastPlan: {nm} must be a contiguous bf16 {shp} tensor on the source's device")
            rec.append(struct.pack("<QQQiiq", src.data_ptr(), _p(dst) or 0, _p(dst_t) or 0, R, C, first))
            first += (R // 64) * (C // 64)
            self._keep.append((src, dst, dst_t))
        if not rec:
            raise ValueError("CastPlan: no jobs")
        self.njobs, self.tiles = len(rec), first
        blob = torch.frombuffer(bytearray(b"".join(rec)), dtype=torch.uint8)
        self._table = blob.to(self._keep[0][0].device)

    def run(self) -> None:
        check(_lib.load().va_multi_cast_f32_bf16(_p(self._table), self.njobs, self.tiles, _stream()),
              "va_multi_cast_f32_bf16")


# ------------------------------------------------------------------ input pipeline
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize_u8(frames: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """uint8 frames (B,T,H,W,3) on the device -> f32 clip (B,3,T,H,W), (u/255 - mean)/std."""
    if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[-1] != 3 or not frames.is_contiguous():
        raise ValueError("normalize_u8: frames must be contiguous uint8 (B,T,H,W,3)")
    if not frames.is_cuda:
        raise _lib.VitAmdError("normalize_u8: tensor must live on a HIP device")
    B, T, H, W, _ = frames.shape
    out = torch.empty(B, 3, T, H, W, dtype=F32, device=frames.device)
    import ctypes
    m3 = (ctypes.c_float * 3)(*mean)
    s3 = (ctypes.c_float * 3)(*std)
    check(_lib.load().va_normalize_u8(_p(frames), _p(out), B, T, H, W, ctypes.addressof(m3), ctypes.addressof(s3),
                                      _stream()), "va_normalize_u8")
    return out


# ------------------------------------------------------------------ masking / JEPA pieces
def future_mask(x: torch.Tensor, first_masked_row: int):
    if x.dim() != 3 or not x.is_contiguous() or x.dtype not in (BF16, F32):
        raise ValueError("future_mask: x must be a contiguous (B,N,D) bf16/f32 tensor")
    if not x.is_cuda:
        raise _lib.VitAmdError("future_mask: tensor must live on a HIP device")
    B, N, D = x.shape
    out = torch.empty_like(x)
    mask = torch.empty(B, N, dtype=torch.uint8, device=x.device)
    check(_lib.load().va_future_mask(_p(x), _p(out), _p(mask), B, N, D, x.element_size(), first_masked_row, _stream()),
          "va_future_mask")
    return out, mask.view(torch.bool)


def token_gather(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    if x.dim() != 3 or not x.is_contiguous() or not x.is_cuda:
        raise ValueError("token_gather: x must be a contiguous (B,N,D) device tensor")
    _need(idx, torch.int32, "idx")
    B, N, D = x.shape
    if idx.dim() != 2 or idx.shape[0] != B or not idx.is_contiguous():
        raise ValueError("token_gather: idx must be contiguous (B, N_keep) int32")
    out = torch.empty(B, idx.shape[1], D, dtype=x.dtype, device=x.device)
    check(_lib.load().va_token_gather(_p(x), _p(idx), _p(out), B, N, idx.shape[1], D * x.element_size(), _stream()),
          "va_token_gather")
    return out


def token_scatter(src: torch.Tensor, idx: torch.Tensor, n_tokens: int) -> torch.Tensor:
    if src.dim() != 3 or not src.is_contiguous() or not src.is_cuda:
        raise ValueError("token_scatter: src must be a contiguous (B,N_keep,D) device tensor")
    _need(idx, torch.int32, "idx")
    B, Nk, D = src.shape
    if tuple(idx.shape) != (B, Nk) or not idx.is_contiguous():
        raise ValueError("token_scatter: idx must be contiguous (B, N_keep) int32")
    out = torch.empty(B, n_tokens, D, dtype=src.dtype, device=src.device)
    check(_lib.load().va_token_scatter(_p(src), _p(idx), _p(out), B, n_tokens, Nk, D * src.element_size(), _stream()),
          "va_token_scatter")
    return out


def token_pool(x: torch.Tensor, n_first: Optional[int] = None) -> torch.Tensor:
    """Mean over the first n_first tokens of x f32 (B, N, D) -> (B, D); n_first=None pools all tokens (`.mean(dim=1)`).
    n_first = 0 gives zeros (the reference's clamp_min(count, 1) denominator)."""
    _need(x, F32, "x")
    if x.dim() != 3 or not x.is_contiguous() or x.shape[2] % 4:
        raise ValueError("token_pool: x must be a contiguous (B, N, D) f32 tensor with D a multiple of 4")
    B, N, D = x.shape
    n_first = N if n_first is None else int(n_first)
    lib = _lib.load()
    out = torch.empty(B, D, dtype=F32, device=x.device)
    ws = torch.empty(lib.va_token_pool_workspace_floats(B, D), dtype=F32, device=x.device)
    check(lib.va_token_pool_fwd(_p(x), _p(out), _p(ws), B, N, D, n_first, 1.0 / max(n_first, 1), _stream()),
          "va_token_pool_fwd")
    return out


def token_pool_bwd(g: torch.Tensor, n_tokens: int, n_first: Optional[int] = None) -> torch.Tensor:
    """Gradient of `token_pool` with respect to x: (B, D) -> (B, N, D)."""
    _need(g, F32, "g")
    if g.dim() != 2 or not g.is_contiguous():
        raise ValueError("token_pool_bwd: g must be a contiguous (B, D) f32 tensor")
    B, D = g.shape
    n_first = n_tokens if n_first is None else int(n_first)
    dx = torch.empty(B, n_tokens, D, dtype=F32, device=g.device)
    check(_lib.load().va_token_pool_bwd(_p(g), _p(dx), B, n_tokens, D, n_first, 1.0 / max(n_first, 1), _stream()),
          "va_token_pool_bwd")
    return dx


def ema_update_(target: torch.Tensor, online: torch.Tensor, momentum: float, shadow_bf16: Optional[torch.Tensor] = None):
    _need(target, F32, "target"); _need(online, F32, "online")
    if target.numel() != online.numel() or not (target.is_contiguous() and online.is_contiguous()):
        raise ValueError("ema_update_: target/online must be contiguous and equally sized")
    if shadow_bf16 is not None:
        _need(shadow_bf16, BF16, "shadow_bf16")
    check(_lib.load().va_ema_update(_p(target), _p(online), _p(shadow_bf16), target.numel(), momentum, _stream()),
          "va_ema_update")
    return target


def l1_loss(pred: torch.Tensor, target: torch.Tensor, want_grad: bool = True, grad_dtype=None):
    """Returns (mean |pred - target| as a 0-d f32 tensor, d loss / d pred or None)."""
    if pred.dtype != target.dtype or pred.dtype not in (BF16, F32) or pred.shape != target.shape:
        raise ValueError("l1_loss: pred/target must share shape and dtype (bf16 or f32)")
    if not (pred.is_cuda and pred.is_contiguous() and target.is_contiguous()):
        raise ValueError("l1_loss: contiguous device tensors required")
    n = pred.numel()
    loss = torch.empty(1, dtype=F32, device=pred.device)
    grad = torch.empty(pred.shape, dtype=grad_dtype or pred.dtype, device=pred.device) if want_grad else None
    check(_lib.load().va_l1_loss(_p(pred), _p(target), int(pred.dtype == BF16), _p(loss), _p(grad),
                                 int(grad is not None and grad.dtype == BF16), 1.0 / n, n, _stream()), "va_l1_loss")
    return loss[0] / n, grad
