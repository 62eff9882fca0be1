"""Inference launch schedule of one predictor block (HybridPredictorBlock.forward, swm:975-997) on the HIP
kernels: LayerNorm -> in-proj GEMM (+bias) -> block-causal attention -> out-proj GEMM (+bias +residual) ->
[gated delta memory: one projection GEMM for q/k/v/g, causal conv + SiLU (+normalise), delta-rule recurrence,
gating + LayerNorm, o-proj GEMM (+residual)] -> LayerNorm -> fc1 (+bias, GELU) -> fc2 (+bias +residual).

Same numeric scheme as the encoder: f32 residual stream and statistics, bf16 GEMM/attention operands with f32
accumulation; the two 8-wide gate projections (a_proj, b_proj) are f32 dot products.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops
from .encoder_engine import WeightCache

BF16, F32 = torch.bfloat16, torch.float32


def block_supported(blk, x: torch.Tensor) -> bool:
    """The LayerNorm kernel is 1024 wide, so both the block width and the memory layer's inner width must be 1024;
    any other predictor size takes the torch formulation."""
    D = x.shape[-1]
    H = blk.spatial_attn.num_heads
    if blk.use_deltanet and (blk.temporal_layer.num_heads * blk.temporal_layer.head_dim != 1024
                             or blk.temporal_layer.num_heads > 8):
        return False
    return (D == 1024 and D // H in (64, 128) and blk.spatial_attn.in_proj_bias is not None
            and blk.spatial_attn._qkv_same_embed_dim)


def block_forward(blk, x: torch.Tensor, tokens_per_timestep: int, state: Optional[torch.Tensor],
                  wc: WeightCache) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """x f32 [B, L, D] on the device; tokens_per_timestep 0 = no mask.  Returns (x_out [B,L,D], new state)."""
    B, L, D = x.shape
    M = B * L
    dev = x.device
    a = blk.spatial_attn
    H = a.num_heads
    x2 = x.detach().reshape(M, D).contiguous()

    def new(cols, dtype):
        return torch.empty(M, cols, dtype=dtype, device=dev)

    ln = new(D, BF16)
    ops.layernorm_fwd(x2, blk.norm1.weight, blk.norm1.bias, ln, eps=blk.norm1.eps)       # inference: no statistics kept
    qkv = new(3 * D, BF16)
    ops.gemm_nt(ln, wc.fwd("in_proj", a.in_proj_weight), ops.EPI_BF16, qkv, bias=a.in_proj_bias.detach())
    attn = new(D, BF16)
    lse2 = torch.empty(B * H, L, dtype=F32, device=dev)
    ops.attn_fwd(qkv, B, L, H, attn, lse2, tokens_per_timestep=tokens_per_timestep)
    del qkv
    x_mid = new(D, F32)
    ops.gemm_nt(attn, wc.fwd("out_proj", a.out_proj.weight), ops.EPI_F32, x_mid, bias=a.out_proj.bias.detach(), resid=x2)

    new_state = state
    if blk.use_deltanet:
        t = blk.temporal_layer
        Hd, Dh = t.num_heads, t.head_dim
        inner = Hd * Dh
        # the two gate projections and, from the same pass over x, its bf16 copy for the projection GEMM
        xb = new(D, BF16)
        alpha, beta = ops.delta_gates_fwd(x_mid, t.a_proj.weight.detach(), t.b_proj.weight.detach(), xb)
        alpha, beta = alpha.view(B, L, Hd), beta.view(B, L, Hd)
        # q, k, v (value half and gate half, swm:500) and g as ONE product against the stacked weights; every kernel
        # that follows reads its 1024-wide column block of the bf16 result in place
        proj = new(5 * inner, BF16)
        ops.gemm_nt(xb, wc.fwd_cat("qkvg_proj", (t.q_proj.weight, t.k_proj.weight, t.v_proj.weight, t.g_proj.weight)),
                    ops.EPI_BF16, proj)
        conv_w, conv_b = t.v_conv1d.weight.detach().reshape(2 * inner, 4), t.v_conv1d.bias.detach()

        def col(i):
            return proj[:, i * inner:(i + 1) * inner]

        q = ops.dwconv1d_causal_silu_bf16(col(0), B, L, t.q_conv1d.weight.detach(), t.q_conv1d.bias.detach(), True)
        k = ops.dwconv1d_causal_silu_bf16(col(1), B, L, t.k_conv1d.weight.detach(), t.k_conv1d.bias.detach(), True)
        v = ops.dwconv1d_causal_silu_bf16(col(2), B, L, conv_w[:inner], conv_b[:inner], False)
        v_gate = ops.dwconv1d_causal_silu_bf16(col(3), B, L, conv_w[inner:], conv_b[inner:], False)
        out, new_state = ops.gated_delta_seq(q.view(B, L, Hd, Dh), k.view(B, L, Hd, Dh), v.view(B, L, Hd, Dh), alpha, beta,
                                             None if state is None else state.detach().contiguous())
        del q, k, v
        on = new(inner, BF16)
        ops.gate_layernorm_fwd(out.view(M, inner), v_gate.view(M, inner), col(4), t.o_norm.weight.detach(),
                               t.o_norm.bias.detach(), on, t.o_norm.eps)
        del proj, out, v_gate
        x_dn = new(D, F32)
        ops.gemm_nt(on, wc.fwd("o_proj", t.o_proj.weight), ops.EPI_F32, x_dn, resid=x_mid)
        x_mid = x_dn

    ops.layernorm_fwd(x_mid, blk.norm2.weight, blk.norm2.bias, ln, eps=blk.norm2.eps)
    act = new(4 * D, BF16)
    ops.gemm_nt(ln, wc.fwd("fc1", blk.mlp[0].weight), ops.EPI_GELU_ONLY, act, bias=blk.mlp[0].bias.detach())
    x_out = new(D, F32)
    ops.gemm_nt(act, wc.fwd("fc2", blk.mlp[2].weight), ops.EPI_F32, x_out, bias=blk.mlp[2].bias.detach(), resid=x_mid)
    return x_out.view(B, L, D), new_state


def linear(x2: torch.Tensor, lin, wc: WeightCache, name: str, gelu: bool = False) -> torch.Tensor:
    """nn.Linear on rows x2 ([M, K], f32 or bf16) through the GEMM kernel: f32 [M, N], or with gelu=True the bf16
    GELU of it (the operand of the next Linear)."""
    xb = x2 if x2.dtype == BF16 else ops.cast_bf16(x2.contiguous())
    M, N = xb.shape[0], lin.weight.shape[0]
    bias = None if lin.bias is None else lin.bias.detach()
    out = torch.empty(M, N, dtype=BF16 if gelu else F32, device=xb.device)
    if gelu:
        if bias is None:
            bias = torch.zeros(N, dtype=F32, device=xb.device)
        return ops.gemm_nt(xb, wc.fwd(name, lin.weight), ops.EPI_GELU_ONLY, out, bias=bias)
    return ops.gemm_nt(xb, wc.fwd(name, lin.weight), ops.EPI_F32, out, bias=bias)


def linear_supported(lin) -> bool:
    return isinstance(lin, torch.nn.Linear) and lin.weight.shape[0] % 128 == 0 and lin.weight.shape[1] % 64 == 0


def sequential_forward(seq, x: torch.Tensor, wc: WeightCache, name: str) -> torch.Tensor:
    """Inference forward of an nn.Sequential of LayerNorm / Linear / GELU / Dropout / anything else over the last
    dimension of x (f32): every Linear whose sides the GEMM kernel takes runs there (a following GELU inside its
    epilogue), a 1024-wide LayerNorm on the LayerNorm kernel, the rest (the 7-wide output layers, Softplus) in torch."""
    lead = x.shape[:-1]
    cur = x.reshape(-1, x.shape[-1])
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Dropout):
            i += 1
            continue
        if linear_supported(m):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.GELU) and mods[i + 1].approximate == "none"
            cur = linear(cur, m, wc, f"{name}.{i}", gelu=fuse)
            i += 2 if fuse else 1
            continue
        if cur.dtype == BF16:
            cur = cur.float()
        if (isinstance(m, torch.nn.LayerNorm) and cur.shape[-1] == 1024 and m.elementwise_affine and m.bias is not None
                and i + 1 < len(mods) and linear_supported(mods[i + 1])):
            M = cur.shape[0]
            ln = torch.empty(M, 1024, dtype=BF16, device=cur.device)
            ops.layernorm_fwd(cur.contiguous(), m.weight.detach(), m.bias.detach(), ln, eps=m.eps)
            cur = ln
        else:
            cur = m(cur)
        i += 1
    if cur.dtype == BF16:
        cur = cur.float()
    return cur.reshape(*lead, cur.shape[-1])
