"""Inference launch schedule of one predictor block (HybridPredictorBlock.forward, swm:975-997) on the HIP
kernels: LayerNorm -> in-proj GEMM (+bias) -> block-causal attention -> out-proj GEMM (+bias +residual) ->
[gated delta memory: projection GEMMs, causal conv + SiLU (+normalise), delta-rule recurrence, gating +
LayerNorm, o-proj GEMM (+residual)] -> LayerNorm -> fc1 (+bias, GELU) -> fc2 (+bias +residual).

Same numeric scheme as the encoder: f32 residual stream and statistics, bf16 GEMM/attention operands with f32
accumulation.  The two 8-wide gate projections (a_proj, b_proj) stay a torch product.
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
    if blk.use_deltanet and blk.temporal_layer.num_heads * blk.temporal_layer.head_dim != 1024:
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

    mean, rstd = torch.empty(M, dtype=F32, device=dev), torch.empty(M, dtype=F32, device=dev)
    ln = new(D, BF16)
    ops.layernorm_fwd(x2, blk.norm1.weight, blk.norm1.bias, ln, mean, rstd, blk.norm1.eps)
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
        xb = ops.cast_bf16(x_mid)
        # v_proj's two halves (the value and its gate, swm:500) as two GEMMs and two convolutions, so that each half
        # is a contiguous operand of the kernels that follow
        wv = wc.fwd("v_proj", t.v_proj.weight)
        conv_w, conv_b = t.v_conv1d.weight.detach().reshape(2 * inner, 4), t.v_conv1d.bias.detach()
        lin = {}
        for name, w in (("q", wc.fwd("q_proj", t.q_proj.weight)), ("k", wc.fwd("k_proj", t.k_proj.weight)),
                        ("v", wv[:inner]), ("gate", wv[inner:]), ("g", wc.fwd("g_proj", t.g_proj.weight))):
            lin[name] = new(inner, F32)
            ops.gemm_nt(xb, w, ops.EPI_F32, lin[name])
        # the two 8-wide gate projections as one product
        ab = torch.sigmoid(x_mid @ torch.cat([t.a_proj.weight.detach(), t.b_proj.weight.detach()]).t())
        alpha, beta = ab[:, :Hd].contiguous().view(B, L, Hd), ab[:, Hd:].contiguous().view(B, L, Hd)
        q = ops.dwconv1d_causal_silu(lin["q"].view(B, L, inner), t.q_conv1d.weight.detach(), t.q_conv1d.bias.detach(), True)
        k = ops.dwconv1d_causal_silu(lin["k"].view(B, L, inner), t.k_conv1d.weight.detach(), t.k_conv1d.bias.detach(), True)
        v = ops.dwconv1d_causal_silu(lin["v"].view(B, L, inner), conv_w[:inner], conv_b[:inner], False)
        v_gate = ops.dwconv1d_causal_silu(lin["gate"].view(B, L, inner), conv_w[inner:], conv_b[inner:], False)
        out, new_state = ops.gated_delta_seq(q.view(B, L, Hd, Dh), k.view(B, L, Hd, Dh), v.view(B, L, Hd, Dh), alpha, beta,
                                             None if state is None else state.detach().contiguous())
        del q, k, v
        on = new(inner, BF16)
        ops.gate_layernorm_fwd(out.view(M, inner), v_gate.view(M, inner), lin["g"], t.o_norm.weight.detach(),
                               t.o_norm.bias.detach(), on, t.o_norm.eps)
        del lin, out, v_gate
        x_dn = new(D, F32)
        ops.gemm_nt(on, wc.fwd("o_proj", t.o_proj.weight), ops.EPI_F32, x_dn, resid=x_mid)
        x_mid = x_dn

    ops.layernorm_fwd(x_mid, blk.norm2.weight, blk.norm2.bias, ln, mean, rstd, blk.norm2.eps)
    act = new(4 * D, BF16)
    ops.gemm_nt(ln, wc.fwd("fc1", blk.mlp[0].weight), ops.EPI_GELU_ONLY, act, bias=blk.mlp[0].bias.detach())
    x_out = new(D, F32)
    ops.gemm_nt(act, wc.fwd("fc2", blk.mlp[2].weight), ops.EPI_F32, x_out, bias=blk.mlp[2].bias.detach(), resid=x_mid)
    return x_out.view(B, L, D), new_state
