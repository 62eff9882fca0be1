"""Autograd functions behind the stand-alone `forward`s of the encoder's building blocks.

The training scripts only ever call `VideoViTEncoder.forward` (one fused schedule, `vit_amd.encoder_engine`), but the
reference's `TubeletEmbedding`, `RoPEMultiheadAttention` and `VideoViTBlock` (surgical_world_model.py:92-109, 283-336,
705-714) are ordinary modules a caller may run on their own.  The functions here give those modules the same
kernels, one launch sequence per module, forward and backward:

    TubeletEmbedFn   gather -> GEMM(+bias)                                  backward: dW/db GEMM, dX GEMM -> scatter
    LayerNormFn      wave-per-row LayerNorm, bf16 or f32 output
    RopeAttentionFn  QKV GEMM -> rotation (any cache / positions) -> flash attention -> out-proj GEMM [+residual]
    MlpFn            fc1 GEMM (+bias, GELU [, dropout]) -> fc2 GEMM (+bias [, dropout] [+residual])

bf16 operands, f32 accumulation, f32 residual stream -- the arithmetic of the fused encoder schedule.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import ops
from ._lib import VitAmdError

BF16, F32 = torch.bfloat16, torch.float32


def _as_operand(h: torch.Tensor) -> torch.Tensor:
    """[M, K] contiguous bf16 GEMM operand of an f32 or bf16 activation."""
    if h.dtype == BF16:
        return h.contiguous()
    return ops.cast_bf16(h.to(F32).contiguous())


def _grad_bf16(g: torch.Tensor, key: int = 0, p: float = 0.0) -> torch.Tensor:
    """bf16 copy of an incoming gradient, optionally through the dropout mask (key, p) of the layer it leaves."""
    g = g.contiguous()
    if p > 0:
        return ops.dropout(g if g.dtype in (BF16, F32) else g.to(F32), key, p,
                           out=torch.empty(g.shape, dtype=BF16, device=g.device))
    return g if g.dtype == BF16 else ops.cast_bf16(g.to(F32))


class TubeletEmbedFn(torch.autograd.Function):
    """Conv3d with kernel = stride = tubelet as patch gather + GEMM, flattened to (B, T'H'W', D) (swm:92-109)."""

    @staticmethod
    def forward(ctx, video, weight, bias, tubelet, cache, key):
        B = video.shape[0]
        v = video.detach().to(F32).contiguous()
        patches = ops.tubelet_gather(v, tubelet)
        M, D = patches.shape[0], weight.shape[0]
        out = torch.empty(M, D, dtype=F32, device=video.device)
        ops.gemm_nt(patches, cache.fwd(key, weight), ops.EPI_F32, out, bias=bias.detach())
        ctx.save_for_backward(patches, weight)
        ctx.meta = (tuple(v.shape), tuple(tubelet), cache, key)
        return out.view(B, M // B, D)

    @staticmethod
    def backward(ctx, g):
        patches, weight = ctx.saved_tensors
        vshape, tubelet, cache, key = ctx.meta
        D = weight.shape[0]
        g_bf = _grad_bf16(g.reshape(-1, D))
        dw = torch.zeros(D, patches.shape[1], dtype=F32, device=g.device)
        db = torch.zeros(D, dtype=F32, device=g.device)
        ops.gemm_tn_accum(g_bf, patches, dw, dbias=db)
        dvideo = None
        if ctx.needs_input_grad[0]:
            d_patches = torch.empty(patches.shape, dtype=F32, device=g.device)
            ops.gemm_nt(g_bf, cache.bwd(key, weight), ops.EPI_F32, d_patches)
            dvideo = ops.tubelet_scatter(d_patches, vshape, tubelet)
        return dvideo, dw.view(weight.shape), db, None, None, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over rows of 1024 (swm:686, 695); output bf16 (a GEMM operand) or f32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_bf16):
        shape = x.shape
        x2 = x.detach().to(F32).contiguous().view(-1, shape[-1])
        M = x2.shape[0]
        out = torch.empty(x2.shape, dtype=BF16 if out_bf16 else F32, device=x.device)
        mean = torch.empty(M, dtype=F32, device=x.device)
        rstd = torch.empty(M, dtype=F32, device=x.device)
        ops.layernorm_fwd(x2, gamma.detach(), beta.detach(), out, mean, rstd, eps)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shape = shape
        return out.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        dy2 = dy.contiguous().view(x2.shape)
        if dy2.dtype not in (BF16, F32):
            dy2 = dy2.to(F32)
        dx = torch.empty_like(x2)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        ops.layernorm_bwd(dy2, x2, gamma.detach(), mean, rstd, None, dx, None, dgamma, dbeta)
        return dx.view(ctx.shape), dgamma, dbeta, None, None


@dataclass
class AttnSpec:
    """What one attention call needs besides tensors."""
    heads: int
    cache: object                      # WeightCache of the owning module
    key: str                           # name prefix of its entries
    rope: Optional[Tuple] = None       # (tables ((cos,sin) x 3), pos int32 [3, N]) or None
    tokens_per_timestep: int = 0       # > 0: block-causal mask (forward only)
    mask: object = None                # ops.PackedMask: an explicit boolean mask (forward and backward, no dropout)
    drop_key: int = 0
    drop_p: float = 0.0


class RopeAttentionFn(torch.autograd.Function):
    """RoPEMultiheadAttention.forward (swm:283-336) [+ residual]: out = resid + Wout . attn(rope(Wqkv . h))."""

    @staticmethod
    def forward(ctx, h, wqkv, wout, resid, spec: AttnSpec):
        B, N, D = h.shape
        H = spec.heads
        hb = _as_operand(h.detach().reshape(B * N, D))
        qkv = torch.empty(B * N, 3 * D, dtype=BF16, device=h.device)
        ops.gemm_nt(hb, spec.cache.fwd(spec.key + "qkv", wqkv), ops.EPI_BF16, qkv)
        if spec.rope is not None:
            ops.rope3d_qk_indexed_(qkv, spec.rope[0], spec.rope[1], N)
        attn = torch.empty(B * N, D, dtype=BF16, device=h.device)
        lse2 = torch.empty(B * H, N, dtype=F32, device=h.device)
        ops.attn_fwd(qkv, B, N, H, attn, lse2, tokens_per_timestep=spec.tokens_per_timestep,
                     drop_key=spec.drop_key, drop_p=spec.drop_p, mask=spec.mask)
        out = torch.empty(B * N, D, dtype=F32, device=h.device)
        r2 = None if resid is None else resid.detach().to(F32).contiguous().view(B * N, D)
        ops.gemm_nt(attn, spec.cache.fwd(spec.key + "out", wout), ops.EPI_F32, out, resid=r2)
        ctx.save_for_backward(hb, qkv, attn, lse2, wqkv, wout)
        ctx.spec, ctx.dims, ctx.h_dtype, ctx.has_resid = spec, (B, N, D), h.dtype, resid is not None
        return out.view(B, N, D)

    @staticmethod
    def backward(ctx, g):
        hb, qkv, attn, lse2, wqkv, wout = ctx.saved_tensors
        spec, (B, N, D) = ctx.spec, ctx.dims
        if spec.tokens_per_timestep > 0:
            raise VitAmdError("attention with a block-causal mask has a forward kernel only (the predictor's inference "
                              "path); run it under torch.no_grad()")
        M, H = B * N, spec.heads
        g = g.contiguous()
        g_bf = _grad_bf16(g.view(M, D))
        d_o = torch.empty(M, D, dtype=BF16, device=g.device)
        ops.gemm_nt(g_bf, spec.cache.bwd(spec.key + "out", wout), ops.EPI_BF16, d_o)
        dwout = torch.zeros(D, D, dtype=F32, device=g.device)
        ops.gemm_tn_accum(g_bf, attn, dwout)
        dqkv = torch.empty(M, 3 * D, dtype=BF16, device=g.device)
        ops.attn_bwd(qkv, attn, d_o, lse2, dqkv, B, N, H, drop_key=spec.drop_key, drop_p=spec.drop_p, mask=spec.mask)
        if spec.rope is not None:
            ops.rope3d_qk_indexed_(dqkv, spec.rope[0], spec.rope[1], N, transpose=True)
        dh = None
        if ctx.needs_input_grad[0]:
            as_bf16 = ctx.h_dtype == BF16
            dh = torch.empty(M, D, dtype=BF16 if as_bf16 else F32, device=g.device)
            ops.gemm_nt(dqkv, spec.cache.bwd(spec.key + "qkv", wqkv), ops.EPI_BF16 if as_bf16 else ops.EPI_F32, dh)
            dh = dh.view(B, N, D)
        dwqkv = torch.zeros(3 * D, D, dtype=F32, device=g.device)
        ops.gemm_tn_accum(dqkv, hb, dwqkv)
        return dh, dwqkv, dwout, (g if ctx.has_resid else None), None


@dataclass
class MlpSpec:
    cache: object
    key: str
    drop1: Tuple[int, float] = (0, 0.0)     # (key, p) of the dropout after GELU (swm:700)
    drop2: Tuple[int, float] = (0, 0.0)     # ... after fc2 (swm:702)


class MlpFn(torch.autograd.Function):
    """Linear -> GELU(erf) -> Dropout -> Linear -> Dropout [+ residual] (swm:697-703, 713)."""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2, resid, spec: MlpSpec):
        lead, D = h.shape[:-1], h.shape[-1]
        hb = _as_operand(h.detach().reshape(-1, D))
        M, hidden = hb.shape[0], w1.shape[0]
        act = torch.empty(M, hidden, dtype=BF16, device=h.device)
        (k1, p1), (k2, p2) = spec.drop1, spec.drop2
        if any(ctx.needs_input_grad[:6]):
            # `pre` receives GELU'(pre-activation), which is all the backward pass needs of it
            pre = torch.empty(M, hidden, dtype=BF16, device=h.device)
            ops.gemm_nt(hb, spec.cache.fwd(spec.key + "fc1", w1), ops.EPI_BIAS_GELU_GRAD, pre, out2=act, bias=b1.detach(),
                        drop_key=k1, drop_p=p1)
        else:       # a pass that keeps nothing for a backward (no_grad, frozen module): half the output bytes
            pre = None
            ops.gemm_nt(hb, spec.cache.fwd(spec.key + "fc1", w1), ops.EPI_GELU_ONLY, act, bias=b1.detach(),
                        drop_key=k1, drop_p=p1)
        out = torch.empty(M, w2.shape[0], dtype=F32, device=h.device)
        r2 = None if resid is None else resid.detach().to(F32).contiguous().view(M, w2.shape[0])
        ops.gemm_nt(act, spec.cache.fwd(spec.key + "fc2", w2), ops.EPI_F32, out, bias=b2.detach(), resid=r2,
                    drop_key=k2, drop_p=p2)
        if pre is not None:
            ctx.save_for_backward(hb, pre, act, w1, w2)
        ctx.spec, ctx.lead, ctx.h_dtype, ctx.has_resid = spec, lead, h.dtype, resid is not None
        return out.view(*lead, w2.shape[0])

    @staticmethod
    def backward(ctx, g):
        hb, pre, act, w1, w2 = ctx.saved_tensors
        spec = ctx.spec
        (k1, p1), (k2, p2) = spec.drop1, spec.drop2
        M, D = hb.shape
        hidden, Dout = w1.shape[0], w2.shape[0]
        g = g.contiguous()
        g_bf = _grad_bf16(g.view(M, Dout), k2, p2)             # gradient behind the second dropout
        db2 = torch.zeros(Dout, dtype=F32, device=g.device)
        ops.colsum_accum(g_bf, db2)
        d_wide = torch.empty(M, hidden, dtype=BF16, device=g.device)
        ops.gemm_nt(g_bf, spec.cache.bwd(spec.key + "fc2", w2), ops.EPI_MUL_AUX, d_wide, aux=pre, drop_key=k1, drop_p=p1)
        dw2 = torch.zeros(Dout, hidden, dtype=F32, device=g.device)
        ops.gemm_tn_accum(g_bf, act, dw2)
        dh = None
        if ctx.needs_input_grad[0]:
            as_bf16 = ctx.h_dtype == BF16
            dh = torch.empty(M, D, dtype=BF16 if as_bf16 else F32, device=g.device)
            ops.gemm_nt(d_wide, spec.cache.bwd(spec.key + "fc1", w1), ops.EPI_BF16 if as_bf16 else ops.EPI_F32, dh)
            dh = dh.view(*ctx.lead, D)
        dw1 = torch.zeros(hidden, D, dtype=F32, device=g.device)
        db1 = torch.zeros(hidden, dtype=F32, device=g.device)
        ops.gemm_tn_accum(d_wide, hb, dw1, dbias=db1)
        return dh, dw1, db1, dw2, db2, (g if ctx.has_resid else None), None
