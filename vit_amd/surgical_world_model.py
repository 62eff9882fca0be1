"""Drop-in module surface of the reference's `surgical_world_model.py`, backed by the MI355X engine.

Same class names, constructor keywords, attributes, parameter names/shapes and construction order (so
`torch.manual_seed` reproduces the reference's initial weights and checkpoints interchange), but
`VideoViTEncoder.forward` -- the path the training scripts spend their time in -- runs the hand-written
HIP kernels of libvit_amd.so through `vit_amd.encoder_engine` instead of eager torch ops.

The encoder's building blocks (`TubeletEmbedding`, `RoPEMultiheadAttention`, `VideoViTBlock`) also run on
their own, forward and backward, through the same kernels (`vit_amd.module_ops`).  The action-conditioned
predictor runs its inference path (no autograd graph) on the kernels of `vit_amd.predictor_engine`; with autograd
it uses a torch formulation of the same arithmetic.  Nothing on the encoder side has a CPU path: CPU tensors raise.

Reference citations: swm = surgical_world_model.py of HananKibria/VIT.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import module_ops, ops, predictor_engine
from ._lib import VitAmdError
from .encoder_engine import DropPlan, EncoderConfig, EncoderEngine, WeightCache, encoder_param_names

__all__ = ["rotate_half", "apply_rotary_pos_emb", "TubeletEmbedding", "VideoRoPE3D", "RoPEMultiheadAttention",
           "GatedDeltaLayer", "VideoViTBlock", "VideoViTEncoder", "ActionStateEmbedding", "HybridPredictorBlock",
           "BlockCausalMask", "SpatialAttentionPooling", "AutoregressivePredictor", "DualSterilityClassifier",
           "SurgicalActionConditionedWorldModel"]

_ROPE_SPLIT = {"t_dim": 256, "h_dim": 384, "w_dim": 384}          # swm:731-735


def _inv_freq(d: int, base: float = 10000.0) -> torch.Tensor:
    return base ** (-torch.arange(0, d, 2).float() / d)           # swm:184-186


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """(x1 | x2) -> (-x2 | x1) (swm:33-36)."""
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def apply_rotary_pos_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x*cos + rotate_half(x)*sin (swm:39-51)."""
    return x * cos + rotate_half(x) * sin


_rotate = apply_rotary_pos_emb


def _need_device(x: torch.Tensor, who: str) -> None:
    if not x.is_cuda:
        raise VitAmdError(f"{who} needs a HIP device tensor: vit_amd has no CPU fallback")


def _draw_key() -> int:
    """A dropout key from torch's default CPU generator (torch.manual_seed makes training runs repeatable)."""
    return int(torch.randint(0, 2 ** 32, (1,), dtype=torch.int64))


def _own_cache(module: nn.Module) -> WeightCache:
    """bf16 operand copies of a stand-alone module's weights (refreshed when a parameter's version changes)."""
    c = module.__dict__.get("_operand_cache")
    if c is None:
        c = module.__dict__["_operand_cache"] = WeightCache()
    return c


# =====================================================================================================
# encoder: parameter containers + engine dispatch
# =====================================================================================================
class TubeletEmbedding(nn.Module):
    """The Conv3d patch projection (swm:58-109), computed as patch gather + MFMA GEMM."""

    def __init__(self, img_size=224, num_frames=16, tubelet_size=(2, 16, 16), in_channels=3, embed_dim=1024):
        super().__init__()
        self.tubelet_size, self.img_size, self.num_frames = tubelet_size, img_size, num_frames
        self.projection = nn.Conv3d(in_channels, embed_dim, kernel_size=tubelet_size, stride=tubelet_size, padding=0)
        tt, th, tw = tubelet_size
        self.num_temporal_patches = num_frames // tt
        self.num_spatial_patches_h = img_size // th
        self.num_spatial_patches_w = img_size // tw
        self.num_patches = self.num_temporal_patches * self.num_spatial_patches_h * self.num_spatial_patches_w

    def forward(self, video):
        """(B,C,T,H,W) -> (B, T'H'W', D), T' slowest (swm:92-109): patch gather + MFMA GEMM (+bias)."""
        _need_device(video, "TubeletEmbedding.forward")
        w = self.projection.weight
        if w.shape[1] != video.shape[1]:
            raise ValueError(f"expected {w.shape[1]} channels, got {video.shape[1]}")
        return module_ops.TubeletEmbedFn.apply(video, w, self.projection.bias, tuple(self.tubelet_size),
                                               _own_cache(self), "pe")

    def inflate_weights_from_2d(self, pretrained_2d_weights: torch.Tensor):
        """Centre-frame inflation of a (out,in,h,w) filter bank into the 3-D projection (swm:111-151)."""
        w2 = pretrained_2d_weights
        if w2.dim() != 4:
            raise ValueError(f"Expected 4D tensor, got {w2.dim()}D")
        tt, th, tw = self.tubelet_size
        if (w2.shape[2], w2.shape[3]) != (th, tw):
            raise ValueError(f"Spatial dimensions mismatch: expected ({th}, {tw}), got ({w2.shape[2]}, {w2.shape[3]})")
        w3 = w2.new_zeros(w2.shape[0], w2.shape[1], tt, th, tw)
        w3[:, :, tt // 2] = w2
        self.projection.weight.data = w3


class VideoRoPE3D(nn.Module):
    """Inverse-frequency buffers of the 3-D rotary embedding (swm:158-222).  `forward` returns the same
    ((cos,sin) x 3) tables as the reference for callers that want them; the encoder kernels use the
    half-width device tables built by `vit_amd.ops.rope_tables_device`."""

    def __init__(self, embed_dim, temporal_dims=256, spatial_dims_h=384, spatial_dims_w=384, base=10000,
                 temporal_scale=0.6):
        super().__init__()
        self.d_t, self.d_h, self.d_w, self.temporal_scale = temporal_dims, spatial_dims_h, spatial_dims_w, temporal_scale
        assert self.d_t + self.d_h + self.d_w == embed_dim, \
            f"Dimension mismatch: {self.d_t}+{self.d_h}+{self.d_w} != {embed_dim}"
        self.register_buffer("inv_freq_t", _inv_freq(self.d_t, base))
        self.register_buffer("inv_freq_h", _inv_freq(self.d_h, base))
        self.register_buffer("inv_freq_w", _inv_freq(self.d_w, base))

    def forward(self, seq_len_t, seq_len_h, seq_len_w):
        dev = self.inv_freq_t.device
        out = []
        for n, inv, sc in ((seq_len_t, self.inv_freq_t, self.temporal_scale), (seq_len_h, self.inv_freq_h, 1.0),
                           (seq_len_w, self.inv_freq_w, 1.0)):
            ang = torch.outer(torch.arange(n, device=dev).float() * sc, inv)
            ang = torch.cat([ang, ang], dim=-1)
            out.append((ang.cos(), ang.sin()))
        return tuple(out)


class RoPEMultiheadAttention(nn.Module):
    """Multi-head attention with the 3-D rotation applied to the full-width Q and K before the head split
    (swm:225-336)."""

    def __init__(self, embed_dim, num_heads, rope_dims=None, dropout=0.1):
        super().__init__()
        assert embed_dim % num_heads == 0, "embed_dim must be divisible by num_heads"
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.scale = self.head_dim ** -0.5
        if rope_dims is None:
            t, h = embed_dim // 4, (embed_dim * 3) // 8
            rope_dims = {"t_dim": t, "h_dim": h, "w_dim": embed_dim - t - h}
        self.rope_dims = rope_dims
        total = rope_dims["t_dim"] + rope_dims["h_dim"] + rope_dims["w_dim"]
        assert total == embed_dim, f"RoPE dimensions must sum to embed_dim: {total} != {embed_dim}"
        self.qkv_proj = nn.Linear(embed_dim, embed_dim * 3, bias=False)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        self.dropout = nn.Dropout(dropout)

    def apply_rope_to_qk(self, qk, rope_cache, token_pos):
        """Per column segment x*cos[idx] + rotate_half(x)*sin[idx] (swm:265-281), as torch ops (a helper for callers;
        `forward` runs the rotation kernel)."""
        if rope_cache is None or token_pos is None:
            return qk
        out, c0 = [], 0
        for (cos, sin), key, d in zip(rope_cache, ("t_idx", "h_idx", "w_idx"),
                                      (self.rope_dims["t_dim"], self.rope_dims["h_dim"], self.rope_dims["w_dim"])):
            out.append(apply_rotary_pos_emb(qk[..., c0:c0 + d], cos[token_pos[key]], sin[token_pos[key]]))
            c0 += d
        return torch.cat(out, dim=-1)

    def _rope_operands(self, rope_cache, token_positions, n_tokens: int, device):
        """Device tables + int32 [3, N] positions for the rotation kernel, validated on the host (swm:265-281)."""
        if rope_cache is None or token_positions is None:
            return None
        rd = self.rope_dims
        if (rd["t_dim"], rd["h_dim"], rd["w_dim"]) != (256, 384, 384):
            raise VitAmdError("the rotation kernel is built for the 256/384/384 split of a 1024-wide model "
                              "(the only split the reference encoder constructs, swm:731-735)")
        tables = tuple(tuple(t.detach().to(device=device, dtype=torch.float32).contiguous() for t in pair)
                       for pair in rope_cache)
        pos = torch.stack([token_positions[k].to(device) for k in ("t_idx", "h_idx", "w_idx")]).to(torch.int32).contiguous()
        if tuple(pos.shape) != (3, n_tokens):
            raise ValueError(f"token positions must have {n_tokens} entries per axis, got {tuple(pos.shape)}")
        lens = torch.tensor([[tables[a][0].shape[0]] for a in range(3)], device=device)
        if bool(((pos < 0) | (pos >= lens)).any()):
            raise IndexError("token position outside the rotary table")
        return tables, pos

    @staticmethod
    def _block_causal_period(attn_mask, L: int) -> int:
        """tokens-per-timestep if `attn_mask` is a block-causal mask (BlockCausalMask.create_mask), else 0."""
        tpt = getattr(attn_mask, "tokens_per_timestep", None)
        if tpt is None and attn_mask.dim() == 2 and tuple(attn_mask.shape) == (L, L) and attn_mask.dtype == torch.bool:
            cand = int((~attn_mask[0]).sum())
            if cand > 0 and L % cand == 0 and torch.equal(attn_mask, BlockCausalMask.create_mask(L, cand, attn_mask.device)):
                tpt = cand
        return int(tpt) if tpt and L % tpt == 0 else 0

    def _mask_operands(self, attn_mask, B: int, L: int, needs_grad: bool, p: float):
        """(tokens_per_timestep, packed mask) for the kernels (swm:316-323, True = masked).  A block-causal mask takes
        the kernel that derives it from the token index when no gradient is wanted (that kernel is forward-only); every
        other boolean mask -- 2-D, 3-D (per batch entry) or 4-D -- and a block-causal one under autograd is bit-packed."""
        if attn_mask is None:
            return 0, None
        if attn_mask.dtype != torch.bool:
            raise VitAmdError("attn_mask must be boolean (True = do not attend), as `masked_fill` requires (swm:322)")
        if not needs_grad:
            tpt = self._block_causal_period(attn_mask, L)
            if tpt:
                return tpt, None
        if p > 0:
            raise VitAmdError("no kernel combines an explicit attention mask with training-mode dropout on the "
                              "probabilities; call .eval() or build the module with dropout=0")
        if tuple(attn_mask.shape[-2:]) != (L, L):
            raise VitAmdError(f"attn_mask of shape {tuple(attn_mask.shape)} does not fit {L} tokens")
        return 0, ops.attn_mask_pack(attn_mask.to(self.qkv_proj.weight.device), B, self.num_heads)

    def forward(self, x, rope_cache=None, token_positions=None, attn_mask=None, _resid=None):
        """swm:283-336 on the kernels: QKV GEMM, rotation of the full-width Q and K, flash attention, output GEMM.
        `attn_mask` (True = masked): None, or any boolean mask `masked_fill` would broadcast over (B, heads, N, N)."""
        _need_device(x, "RoPEMultiheadAttention.forward")
        B, N, C = x.shape
        if C != self.embed_dim or self.head_dim not in (32, 64, 128) or C % 128:
            raise VitAmdError(f"no attention kernel for embed_dim {C} with head_dim {self.head_dim} (32, 64 or 128 needed)")
        p = self.dropout.p if self.dropout.training else 0.0
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self.qkv_proj.weight.requires_grad or
                                                  self.out_proj.weight.requires_grad)
        tpt, packed = self._mask_operands(attn_mask, B, N, needs_grad, p)
        spec = module_ops.AttnSpec(heads=self.num_heads, cache=_own_cache(self), key="",
                                   rope=self._rope_operands(rope_cache, token_positions, N, x.device),
                                   tokens_per_timestep=tpt, mask=packed,
                                   drop_key=_draw_key() if p > 0 else 0, drop_p=p)
        return module_ops.RopeAttentionFn.apply(x, self.qkv_proj.weight, self.out_proj.weight, _resid, spec)


class VideoViTBlock(nn.Module):
    """Pre-LN attention + MLP block (swm:681-714)."""

    def __init__(self, embed_dim, num_heads, rope_dims=None, mlp_ratio=4.0, dropout=0.1):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dim, eps=1e-6)
        self.attn = RoPEMultiheadAttention(embed_dim, num_heads, rope_dims, dropout)
        self.norm2 = nn.LayerNorm(embed_dim, eps=1e-6)
        hidden = int(embed_dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(embed_dim, hidden), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden, embed_dim), nn.Dropout(dropout))

    def forward(self, x, rope_cache=None, token_positions=None, attn_mask=None):
        """x + attn(LN1(x)), then x + mlp(LN2(x)) (swm:705-714); the residual adds ride in the GEMM epilogues."""
        _need_device(x, "VideoViTBlock.forward")
        if x.shape[-1] != 1024:
            raise VitAmdError("the LayerNorm kernel is built for width 1024 (the only width the reference encoder admits)")
        h = module_ops.LayerNormFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, True)
        x = self.attn(h, rope_cache=rope_cache, token_positions=token_positions, attn_mask=attn_mask, _resid=x)
        h = module_ops.LayerNormFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, True)
        fc1, d1, fc2, d2 = self.mlp[0], self.mlp[2], self.mlp[3], self.mlp[4]
        p1, p2 = (d1.p if d1.training else 0.0), (d2.p if d2.training else 0.0)
        spec = module_ops.MlpSpec(cache=_own_cache(self), key="", drop1=(_draw_key() if p1 > 0 else 0, p1),
                                  drop2=(_draw_key() if p2 > 0 else 0, p2))
        return module_ops.MlpFn.apply(h, fc1.weight, fc1.bias, fc2.weight, fc2.bias, x, spec)


class _EncoderFn(torch.autograd.Function):
    """Autograd boundary around the engine: inputs (video, *parameters) -> features (B,N,D)."""

    @staticmethod
    def forward(ctx, module: "VideoViTEncoder", video: torch.Tensor, *params: torch.Tensor):
        P = module._tensor_table(params)
        save = any(ctx.needs_input_grad[1:])
        out, ectx = module._engine.forward(P, video, save=save, drop=module._dropout_plan(),
                                           checkpoint=save and module._checkpointing(video))
        ctx.module, ctx.ectx, ctx.P = module, ectx, P
        return out

    @staticmethod
    def backward(ctx, grad_out):
        module = ctx.module
        reducer = module._grad_reducer
        with torch.cuda.device(grad_out.device):
            grads = module._engine.backward(ctx.P, ctx.ectx, grad_out,
                                            on_block_done=reducer.submit if reducer is not None else None,
                                            input_grad=ctx.needs_input_grad[1])
            if reducer is not None:
                reducer.finish()
        ctx.ectx = None
        names = module._param_order
        return (None, grads.get("video")) + tuple(grads[n] if need else None
                                                  for n, need in zip(names, ctx.needs_input_grad[2:]))


class VideoViTEncoder(nn.Module):
    """Tubelet embedding + 3-D RoPE transformer encoder (swm:717-817) on the MI355X engine."""

    def __init__(self, img_size=224, num_frames=16, tubelet_size=(2, 16, 16), in_channels=3, embed_dim=1024,
                 depth=12, num_heads=12, use_grad_checkpoint=False):
        super().__init__()
        self.tubelet_embed = TubeletEmbedding(img_size, num_frames, tubelet_size, in_channels, embed_dim)
        self.rope = VideoRoPE3D(embed_dim, _ROPE_SPLIT["t_dim"], _ROPE_SPLIT["h_dim"], _ROPE_SPLIT["w_dim"],
                                temporal_scale=0.6)
        self.blocks = nn.ModuleList([VideoViTBlock(embed_dim, num_heads, dict(_ROPE_SPLIT), 4.0, 0.1)
                                     for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        # Activation checkpointing per block (swm:807-811: when the flag is set AND the module is in training mode).
        # 288 GB of HBM hold every activation of ViT-L at batch 32 (~60 GB), so under the default policy "auto" the
        # flag only takes effect when the activations would not fit the memory that is free; "always" recomputes
        # whenever the reference would, "never" ignores the flag.
        self.use_grad_checkpoint = use_grad_checkpoint
        self.checkpoint_policy = "auto"
        self._cfg = EncoderConfig(depth=depth, heads=num_heads, dim=embed_dim, tubelet=tuple(tubelet_size),
                                  in_channels=in_channels)
        self._param_order = encoder_param_names(depth)
        self._engine_obj: Optional[EncoderEngine] = None
        self._grad_reducer = None

    # -- engine plumbing -------------------------------------------------------------------------
    @property
    def _engine(self) -> EncoderEngine:
        if self._engine_obj is None:
            self._engine_obj = EncoderEngine(self._cfg)     # raises for head counts without a kernel
        return self._engine_obj

    def refresh_weights(self) -> None:
        """Drop the cached bf16 operand copies of the weights.  The copies are keyed on each parameter's autograd
        version counter, which optimizers and in-place tensor methods bump -- but writes through `param.data`
        (`p.data.copy_(...)`, `p.data.mul_(...)`, the usual EMA idiom) or by a kernel do not.  Call this after such a
        write (vit_amd's own EMA update and parameter broadcast do)."""
        if self._engine_obj is not None:
            self._engine_obj.weights.invalidate()

    def set_grad_reducer(self, reducer) -> None:
        """Attach a `vit_amd.ddp.BucketReducer`: gradients are then averaged across ranks inside backward."""
        self._grad_reducer = reducer

    ACTIVATION_BYTES_PER_TOKEN_AND_BLOCK = 36 * 1024 + 16     # x 4D, LN1 2D, qkv 6D, attn 2D, x_mid 4D, LN2 2D, MLP 16D; stats

    def _checkpointing(self, video: torch.Tensor) -> bool:
        """Whether this forward keeps only the blocks' inputs and lets the backward recompute the rest."""
        if not (self.use_grad_checkpoint and self.training) or self.checkpoint_policy == "never":
            return False
        if self.checkpoint_policy == "always":
            return True
        if self.checkpoint_policy != "auto":
            raise ValueError(f"checkpoint_policy must be 'auto', 'always' or 'never', not {self.checkpoint_policy!r}")
        tt, th, tw = self._cfg.tubelet
        tokens = video.shape[0] * (video.shape[2] // tt) * (video.shape[3] // th) * (video.shape[4] // tw)
        need = tokens * self._cfg.depth * self.ACTIVATION_BYTES_PER_TOKEN_AND_BLOCK
        free, _ = torch.cuda.mem_get_info(video.device)
        free += torch.cuda.memory_reserved(video.device) - torch.cuda.memory_allocated(video.device)   # torch's own free blocks
        return need > 0.8 * free

    def _dropout_plan(self) -> Optional[DropPlan]:
        """Training mode: the probabilities of each block's nn.Dropout modules (swm:263, 700-702; 0.1 as built,
        a block whose modules are in eval mode or set to p = 0 contributes zeros) and fresh keys for this forward."""
        ps = [tuple(d.p if d.training else 0.0 for d in (blk.attn.dropout, blk.mlp[2], blk.mlp[4]))
              for blk in self.blocks]
        if not any(v > 0 for t in ps for v in t):
            return None
        return DropPlan.draw(ps)

    def _tensor_table(self, params) -> Dict[str, torch.Tensor]:
        P = dict(zip(self._param_order, params))
        P["rope.inv_freq_t"], P["rope.inv_freq_h"], P["rope.inv_freq_w"] = \
            self.rope.inv_freq_t, self.rope.inv_freq_h, self.rope.inv_freq_w
        return P

    def _create_token_positions(self, T, H, W, device):
        """(t,h,w) index of every token, t slowest and w fastest (swm:760-770); the kernels derive the same
        coordinates arithmetically from the row index."""
        n = torch.arange(T * H * W, device=device)
        return {"t_idx": n // (H * W), "h_idx": (n // W) % H, "w_idx": n % W}

    def forward(self, video: torch.Tensor) -> torch.Tensor:
        if not video.is_cuda:
            raise VitAmdError("VideoViTEncoder.forward needs a HIP device tensor: vit_amd has no CPU fallback")
        if video.dim() == 5 and video.shape[0] == 0:
            # an empty batch has nothing to launch; the reference returns an empty (0, N, D) tensor here as well
            tt, th, tw = self._cfg.tubelet
            n_tok = (video.shape[2] // tt) * (video.shape[3] // th) * (video.shape[4] // tw)
            return video.new_zeros((0, n_tok, self._cfg.dim), dtype=torch.float32)
        named = dict(self.named_parameters())
        params = [named[n] for n in self._param_order]
        for n, p in zip(self._param_order, params):
            if p.device != video.device or p.dtype != torch.float32:
                raise VitAmdError(f"parameter {n} must be float32 on {video.device} (master weights stay f32; "
                                  "the engine keeps its own bf16 operand copies)")
        with torch.cuda.device(video.device):        # the kernels go to the current device's stream
            return _EncoderFn.apply(self, video, *params)


# =====================================================================================================
# predictor side.  Inference (no autograd graph, tensors on the device) runs the memory layer's causal
# conv + SiLU (+ normalise), the delta-rule recurrence and the block-causal attention on the HIP kernels;
# projections and the small heads are library GEMMs through torch.  With autograd the torch formulation of
# the same arithmetic is used (the predictor is not trained by the stage-1 scripts).
# =====================================================================================================
def _hip_inference(x: torch.Tensor, module: Optional[nn.Module] = None) -> bool:
    """True when the inference kernels may run: device f32 input and no autograd graph to build -- neither through
    the input nor (grad mode on) through a trainable parameter of `module`, whose gradient would otherwise be lost."""
    if not (x.is_cuda and x.dtype == torch.float32):
        return False
    if not torch.is_grad_enabled():
        return True
    if x.requires_grad:
        return False
    return module is None or not any(p.requires_grad for p in module.parameters())


_warned_torch_path = False


def _note_torch_path(x: torch.Tensor, who: str) -> None:
    """Tell the caller, once per process, that a predictor call on the device is being computed by the torch
    formulation instead of the HIP kernels because an autograd graph is wanted (grad mode on and trainable parameters,
    or an input that requires grad).  A model called without `torch.no_grad()` would otherwise take the slow path
    silently."""
    global _warned_torch_path
    if _warned_torch_path or not (x.is_cuda and x.dtype == torch.float32):
        return
    _warned_torch_path = True
    import warnings
    warnings.warn(f"{who}: grad mode is on and the predictor has trainable parameters (or its input requires grad), so "
                  "this call is computed by the torch formulation (a per-token Python loop for the memory layer), not "
                  "by the HIP kernels. For inference wrap the call in torch.no_grad(); the kernels have no predictor "
                  "backward.", RuntimeWarning, stacklevel=3)


class GatedDeltaLayer(nn.Module):
    """Gated delta-rule memory layer (swm:431-555).  Only the sequential recurrence is provided: it is the
    form inference uses and the one the reference's other ("parallel") path fails to reproduce."""

    def __init__(self, hidden_size, num_heads=4, head_dim=128, chunk_size=64):
        super().__init__()
        self.hidden_size, self.num_heads, self.head_dim, self.chunk_size = hidden_size, num_heads, head_dim, chunk_size
        inner = num_heads * head_dim
        self.q_proj = nn.Linear(hidden_size, inner, bias=False)
        self.k_proj = nn.Linear(hidden_size, inner, bias=False)
        self.v_proj = nn.Linear(hidden_size, inner * 2, bias=False)
        self.a_proj = nn.Linear(hidden_size, num_heads, bias=False)
        self.b_proj = nn.Linear(hidden_size, num_heads, bias=False)
        self.q_conv1d = nn.Conv1d(inner, inner, kernel_size=4, padding=3, groups=inner)
        self.k_conv1d = nn.Conv1d(inner, inner, kernel_size=4, padding=3, groups=inner)
        self.v_conv1d = nn.Conv1d(inner * 2, inner * 2, kernel_size=4, padding=3, groups=inner * 2)
        self.g_proj = nn.Linear(hidden_size, inner, bias=False)
        self.o_norm = nn.LayerNorm(inner, eps=1e-5)
        self.o_proj = nn.Linear(inner, hidden_size, bias=False)
        self.silu = nn.SiLU()

    @staticmethod
    def _causal_conv(conv: nn.Conv1d, x: torch.Tensor) -> torch.Tensor:
        L = x.shape[1]
        return conv(x.transpose(1, 2))[..., :L].transpose(1, 2)

    def _sequential_gated_delta(self, q, k, v, alpha, beta, state):
        """S <- alpha*(S - beta*(S k) k^T) + beta*v k^T ; o = S q  -- the rank-1 form of
        S <- S*alpha*(I - beta*k k^T) + beta*v k^T (swm:523-555)."""
        B, L, H, Dh = q.shape
        if state is None:
            state = q.new_zeros(B, H, Dh, Dh)
        outs = []
        for t in range(L):
            kt, vt, qt = k[:, t], v[:, t], q[:, t]
            a, b = alpha[:, t, :, None, None], beta[:, t, :, None, None]
            sk = torch.einsum("bhij,bhj->bhi", state, kt)
            state = a * (state - b * sk[..., None] * kt[..., None, :]) + b * vt[..., None] * kt[..., None, :]
            outs.append(torch.einsum("bhij,bhj->bhi", state, qt))
        return torch.stack(outs, dim=1), state

    def forward(self, x, state=None, use_parallel=None):
        B, L, _ = x.shape
        H, Dh = self.num_heads, self.head_dim
        alpha, beta = torch.sigmoid(self.a_proj(x)), torch.sigmoid(self.b_proj(x))
        if _hip_inference(x, self) and Dh in (64, 128):
            q = ops.dwconv1d_causal_silu(self.q_proj(x), self.q_conv1d.weight, self.q_conv1d.bias, True)
            k = ops.dwconv1d_causal_silu(self.k_proj(x), self.k_conv1d.weight, self.k_conv1d.bias, True)
            v = ops.dwconv1d_causal_silu(self.v_proj(x), self.v_conv1d.weight, self.v_conv1d.bias, False)
            v, v_gate = v.chunk(2, dim=-1)
            out, new_state = ops.gated_delta_seq(q.view(B, L, H, Dh), k.view(B, L, H, Dh),
                                                 v.contiguous().view(B, L, H, Dh), alpha.contiguous(), beta.contiguous(),
                                                 None if state is None else state.contiguous())
        else:
            if Dh in (64, 128):
                _note_torch_path(x, "GatedDeltaLayer.forward")
            q = F.normalize(self.silu(self._causal_conv(self.q_conv1d, self.q_proj(x))), p=2, dim=-1)
            k = F.normalize(self.silu(self._causal_conv(self.k_conv1d, self.k_proj(x))), p=2, dim=-1)
            v = self.silu(self._causal_conv(self.v_conv1d, self.v_proj(x)))
            v, v_gate = v.chunk(2, dim=-1)
            out, new_state = self._sequential_gated_delta(q.view(B, L, H, Dh), k.view(B, L, H, Dh), v.view(B, L, H, Dh),
                                                          alpha, beta, state)
        out = (out * v_gate.view(B, L, H, Dh)).reshape(B, L, -1) * self.silu(self.g_proj(x))
        return self.o_proj(self.o_norm(out)), new_state


class ActionStateEmbedding(nn.Module):
    """Action / state token embedding with linear resampling to the tubelet rate and temporal RoPE on the
    first quarter of the channels (swm:824-945)."""

    def __init__(self, action_dim=7, state_dim=7, hidden_dim=1024, temporal_rope_base=10000):
        super().__init__()
        self.action_proj = nn.Linear(action_dim, hidden_dim)
        self.state_proj = nn.Linear(state_dim, hidden_dim)
        self.hidden_dim, self.d_t = hidden_dim, hidden_dim // 4
        self.register_buffer("inv_freq_t", _inv_freq(self.d_t, temporal_rope_base))

    def interpolate_to_tubelet_resolution(self, x, target_length):
        if x.shape[1] == target_length:
            return x
        return F.interpolate(x.transpose(1, 2), size=target_length, mode="linear", align_corners=True).transpose(1, 2)

    def create_temporal_rope(self, seq_len, device, temporal_scale=0.6):
        ang = torch.outer(torch.arange(seq_len, device=device).float() * temporal_scale, self.inv_freq_t)
        ang = torch.cat([ang, ang], dim=-1)
        return ang.cos(), ang.sin()

    def apply_temporal_rope(self, x, timestep_indices, cos_t, sin_t):
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(1)
        out = torch.cat([_rotate(x[..., :self.d_t], cos_t[timestep_indices], sin_t[timestep_indices]),
                         x[..., self.d_t:]], dim=-1)
        return out.squeeze(1) if squeeze else out

    def forward(self, actions, states, target_temporal_length=None, timestep_indices=None):
        T = actions.shape[1]
        if target_temporal_length is not None and target_temporal_length != T:
            actions = self.interpolate_to_tubelet_resolution(actions, target_temporal_length)
            states = self.interpolate_to_tubelet_resolution(states, target_temporal_length)
            T = target_temporal_length
        if timestep_indices is None:
            timestep_indices = torch.arange(T, device=actions.device)
        cos_t, sin_t = self.create_temporal_rope(T, actions.device)
        return (self.apply_temporal_rope(self.action_proj(actions), timestep_indices, cos_t, sin_t),
                self.apply_temporal_rope(self.state_proj(states), timestep_indices, cos_t, sin_t))


class HybridPredictorBlock(nn.Module):
    """Pre-LN masked attention, optional gated-delta memory, MLP (swm:952-997)."""

    def __init__(self, hidden_dim, num_heads, use_deltanet=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(hidden_dim, eps=1e-6)
        self.spatial_attn = nn.MultiheadAttention(hidden_dim, num_heads, batch_first=True)
        self.use_deltanet = use_deltanet
        if use_deltanet:
            self.temporal_layer = GatedDeltaLayer(hidden_dim, num_heads, head_dim=hidden_dim // num_heads)
        self.norm2 = nn.LayerNorm(hidden_dim, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(hidden_dim, hidden_dim * 4), nn.GELU(), nn.Linear(hidden_dim * 4, hidden_dim))

    def _masked_attention(self, h: torch.Tensor, attn_mask: Optional[torch.Tensor]) -> torch.Tensor:
        a = self.spatial_attn
        B, L, D = h.shape
        H = a.num_heads
        qkv = F.linear(h, a.in_proj_weight, a.in_proj_bias)
        tpt = 0 if attn_mask is None else getattr(attn_mask, "tokens_per_timestep", -1)
        if _hip_inference(h, self) and D // H in (64, 128) and tpt >= 0 and (tpt == 0 or L % tpt == 0):
            # packed [q | k | v] rows are exactly the attention kernel's layout; the mask is its block-causal mode
            o = torch.empty(B * L, D, dtype=torch.bfloat16, device=h.device)
            lse = torch.empty(B * H, L, dtype=torch.float32, device=h.device)
            ops.attn_fwd(qkv.reshape(B * L, 3 * D).to(torch.bfloat16), B, L, H, o, lse, tokens_per_timestep=tpt)
            return a.out_proj(o.view(B, L, D).float())
        qkv = qkv.view(B, L, 3, H, D // H).permute(2, 0, 3, 1, 4)
        allowed = None if attn_mask is None else ~attn_mask            # reference: True = masked (swm:1004)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=allowed)
        return a.out_proj(o.transpose(1, 2).reshape(B, L, D))

    def forward(self, x, attn_mask=None, state=None):
        tpt = 0 if attn_mask is None else getattr(attn_mask, "tokens_per_timestep", -1)
        if (_hip_inference(x, self) and tpt >= 0 and (tpt == 0 or x.shape[1] % tpt == 0)
                and predictor_engine.block_supported(self, x)):
            return predictor_engine.block_forward(self, x, tpt, state, _own_cache(self))
        if not _hip_inference(x, self):
            _note_torch_path(x, "HybridPredictorBlock.forward")
        x = x + self._masked_attention(self.norm1(x), attn_mask)
        if self.use_deltanet:
            delta, state = self.temporal_layer(x, state)
            x = x + delta
        return x + self.mlp(self.norm2(x)), state


class BlockCausalMask:
    """Bool (L,L), True = masked: a token of timestep t sees every token of timesteps <= t (swm:1000-1028)."""

    @staticmethod
    def create_mask(seq_len, tokens_per_timestep, device):
        step = torch.arange(seq_len, device=device) // tokens_per_timestep
        mask = step[None, :] > step[:, None]
        whole = (seq_len // tokens_per_timestep) * tokens_per_timestep
        if whole < seq_len:                       # rows of a trailing partial timestep stay fully masked
            mask[whole:, :] = True
        mask.tokens_per_timestep = tokens_per_timestep      # lets the attention pick its block-causal mode
        return mask


class SpatialAttentionPooling(nn.Module):
    """Four learned queries attend over the patch tokens; outputs are normalised and flattened (swm:1035-1080)."""

    def __init__(self, hidden_dim, num_queries=4):
        super().__init__()
        self.num_queries = num_queries
        self.queries = nn.Parameter(torch.randn(num_queries, hidden_dim))
        self.attention = nn.MultiheadAttention(hidden_dim, num_heads=8, batch_first=True)
        self.norm = nn.LayerNorm(hidden_dim)

    def forward(self, patch_tokens):
        B = patch_tokens.shape[0]
        a = self.attention
        E, H = a.embed_dim, a.num_heads
        if (_hip_inference(patch_tokens, self) and a._qkv_same_embed_dim and a.in_proj_bias is not None and E % 128 == 0
                and not (self.training and a.dropout > 0)):
            # the key/value projection of every patch token is the one sizeable product here: on the GEMM kernel;
            # the four queries, their 4 x n scores and the output projection of 4 rows per clip stay torch (f32)
            n = patch_tokens.shape[1]
            w, b = a.in_proj_weight.detach(), a.in_proj_bias.detach()
            wc = _own_cache(self)
            kv = torch.empty(B * n, 2 * E, dtype=torch.float32, device=patch_tokens.device)
            ops.gemm_nt(ops.cast_bf16(patch_tokens.reshape(B * n, E).contiguous()), wc.fwd("in_proj", a.in_proj_weight)[E:],
                        ops.EPI_F32, kv, bias=b[E:].contiguous())
            q = F.linear(self.queries.detach(), w[:E], b[:E]).view(1, -1, H, E // H).transpose(1, 2).expand(B, -1, -1, -1)
            k = kv[:, :E].view(B, n, H, E // H).transpose(1, 2)
            v = kv[:, E:].view(B, n, H, E // H).transpose(1, 2)
            o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, -1, E)
            return self.norm(a.out_proj(o)).reshape(B, -1)
        pooled, _ = a(self.queries.unsqueeze(0).expand(B, -1, -1), patch_tokens, patch_tokens)
        return self.norm(pooled).reshape(B, -1)


class AutoregressivePredictor(nn.Module):
    """Per-horizon heads and the append-and-reprocess rollout (swm:1083-1277)."""

    def __init__(self, predictor_dim, encoder_dim, num_patch_tokens):
        super().__init__()
        self.predictor_dim, self.encoder_dim, self.num_patch_tokens = predictor_dim, encoder_dim, num_patch_tokens
        P = predictor_dim
        self.spatial_pooling = SpatialAttentionPooling(P, num_queries=4)
        self.spatial_predictor = nn.Sequential(nn.LayerNorm(P), nn.Linear(P, P * 2), nn.GELU(), nn.Dropout(0.1),
                                               nn.Linear(P * 2, P))
        self.to_encoder_features = nn.Sequential(nn.Linear(P, encoder_dim * 2), nn.GELU(),
                                                 nn.Linear(encoder_dim * 2, encoder_dim))

        def regress():
            return nn.Sequential(nn.Linear(P * 4, P), nn.GELU(), nn.Dropout(0.1), nn.Linear(P, P // 2), nn.GELU(),
                                 nn.Linear(P // 2, 7))

        def variance():
            return nn.Sequential(nn.Linear(P * 4, P // 4), nn.GELU(), nn.Linear(P // 4, 7), nn.Softplus())

        self.action_predictor = regress()
        self.state_predictor = regress()
        self.action_uncertainty = variance()
        self.state_uncertainty = variance()

    def forward(self, x, tokens_per_timestep, predict_horizon=1, future_actions=None, future_states=None,
                action_state_embed=None, predictor_blocks=None, attn_mask=None):
        B = x.shape[0]
        seq = x
        states = [None] * len(predictor_blocks) if predictor_blocks is not None else None
        results: List[Dict[str, torch.Tensor]] = []
        if _hip_inference(x, self):
            # every head's Linear layers on the GEMM kernel (the 7-wide output layers stay torch products)
            wc = _own_cache(self)

            def run(seq_module, name, t):
                return predictor_engine.sequential_forward(seq_module, t, wc, name)
        else:
            def run(seq_module, name, t):
                return seq_module(t)
        for h in range(predict_horizon):
            patches = seq[:, -tokens_per_timestep:][:, :self.num_patch_tokens]
            next_patches = run(self.spatial_predictor, "spatial_predictor", patches)
            ctx = self.spatial_pooling(patches)
            act, sta = run(self.action_predictor, "action_predictor", ctx), run(self.state_predictor, "state_predictor", ctx)
            results.append({"features": run(self.to_encoder_features, "to_encoder_features", next_patches.mean(dim=1)),
                            "action": act, "state": sta,
                            "action_uncertainty": run(self.action_uncertainty, "action_uncertainty", ctx),
                            "state_uncertainty": run(self.state_uncertainty, "state_uncertainty", ctx)})
            if h == predict_horizon - 1:
                break
            nxt_a = future_actions[:, h] if future_actions is not None and h < future_actions.shape[1] else act
            nxt_s = future_states[:, h] if future_states is not None and h < future_states.shape[1] else sta
            if action_state_embed is not None:
                a_emb, s_emb = action_state_embed(nxt_a.unsqueeze(1), nxt_s.unsqueeze(1))
            else:
                a_emb = s_emb = x.new_zeros(B, 1, self.predictor_dim)
            step = torch.cat([next_patches, a_emb.reshape(B, 1, -1), s_emb.reshape(B, 1, -1)], dim=1)
            if predictor_blocks is not None and attn_mask is not None:
                old, new = seq.shape[1], seq.shape[1] + tokens_per_timestep
                grown = torch.ones(new, new, dtype=torch.bool, device=x.device)
                grown[:old, :old] = attn_mask
                grown[old:, :new] = False
                if hasattr(attn_mask, "tokens_per_timestep"):     # one more whole timestep: still block-causal
                    grown.tokens_per_timestep = attn_mask.tokens_per_timestep
                attn_mask = grown
            seq = torch.cat([seq, step], dim=1)
            if predictor_blocks is not None:
                # the whole grown sequence is re-processed; the memory state carries over (swm:1266-1273)
                for i, blk in enumerate(predictor_blocks):
                    seq, states[i] = blk(seq, attn_mask=attn_mask, state=states[i])
        return results


class DualSterilityClassifier(nn.Module):
    """Two small LN-MLP heads: on pooled encoder features and on pooled predictor features (swm:1284-1331)."""

    def __init__(self, encoder_dim, predictor_dim, num_classes=2):
        super().__init__()

        def head(d):
            return nn.Sequential(nn.LayerNorm(d), nn.Linear(d, d // 2), nn.GELU(), nn.Dropout(0.5),
                                 nn.Linear(d // 2, num_classes))

        self.current_classifier = head(encoder_dim)
        self.future_classifier = head(predictor_dim)

    def forward(self, current_features, future_features=None):
        fut = self.future_classifier(future_features) if future_features is not None else None
        return self.current_classifier(current_features), fut


class SurgicalActionConditionedWorldModel(nn.Module):
    """Encoder + action-conditioned predictor (swm:1338-1557); constructor signature unchanged."""

    def __init__(self, img_size=224, num_frames=16, tubelet_size=(2, 16, 16), in_channels=3, encoder_dim=1024,
                 encoder_depth=12, encoder_heads=12, predictor_dim=1024, predictor_depth=12, predictor_heads=8,
                 action_dim=7, state_dim=7, num_anomaly_classes=2, use_grad_checkpoint=False,
                 pretrained_2d_weights=None):
        super().__init__()
        self.encoder = VideoViTEncoder(img_size=img_size, num_frames=num_frames, tubelet_size=tubelet_size,
                                       in_channels=in_channels, embed_dim=encoder_dim, depth=encoder_depth,
                                       num_heads=encoder_heads, use_grad_checkpoint=use_grad_checkpoint)
        if pretrained_2d_weights is not None:
            self.encoder.tubelet_embed.inflate_weights_from_2d(pretrained_2d_weights)
        self.video_proj = nn.Linear(encoder_dim, predictor_dim)
        self.action_state_embed = ActionStateEmbedding(action_dim, state_dim, predictor_dim)
        self.predictor_blocks = nn.ModuleList([
            HybridPredictorBlock(predictor_dim, predictor_heads, use_deltanet=(i < predictor_depth * 2 // 3))
            for i in range(predictor_depth)])
        self.predictor_norm = nn.LayerNorm(predictor_dim, eps=1e-6)
        self.predictor_to_encoder = nn.Linear(predictor_dim, encoder_dim)
        self.sterility_classifier = DualSterilityClassifier(encoder_dim, predictor_dim, num_anomaly_classes)
        self.num_frames, self.encoder_dim, self.predictor_dim = num_frames, encoder_dim, predictor_dim
        te = self.encoder.tubelet_embed
        self.tokens_per_frame = te.num_patches // te.num_temporal_patches
        self.use_grad_checkpoint = use_grad_checkpoint
        self.autoregressive_predictor = AutoregressivePredictor(predictor_dim, encoder_dim,
                                                                num_patch_tokens=self.tokens_per_frame)

    def encode_video(self, video):
        return self.encoder(video)

    def interleave_tokens(self, video_features, action_emb, state_emb):
        """(B,T,n,D) patches + (B,T,D) action + (B,T,D) state -> (B, T*(n+2), D), per timestep
        [patches, action, state] (swm:1428-1438)."""
        B, T, n, D = video_features.shape
        return torch.cat([video_features, action_emb[:, :, None], state_emb[:, :, None]], dim=2).reshape(B, T * (n + 2), D)

    def forward(self, video, actions, states, predict_horizon=1, encoder_frozen=True):
        B = video.shape[0]
        if encoder_frozen:
            with torch.no_grad():
                enc = self.encoder(video)
        else:
            enc = self.encoder(video)
        pooled_now = enc.mean(dim=1)
        Tp, tpf = self.encoder.tubelet_embed.num_temporal_patches, self.tokens_per_frame
        if _hip_inference(enc, self.video_proj) and predictor_engine.linear_supported(self.video_proj):
            tokens = predictor_engine.linear(enc.reshape(-1, enc.shape[-1]), self.video_proj, _own_cache(self), "video_proj")
            tokens = tokens.view(B, Tp, tpf, -1)
        else:
            tokens = self.video_proj(enc).view(B, Tp, tpf, -1)
        a_emb, s_emb = self.action_state_embed(actions, states, target_temporal_length=Tp,
                                               timestep_indices=torch.arange(Tp, device=actions.device))
        x = self.interleave_tokens(tokens, a_emb, s_emb)
        per_step = tpf + 2
        mask = BlockCausalMask.create_mask(x.shape[1], per_step, x.device)
        for blk in self.predictor_blocks:
            x, _ = blk(x, attn_mask=mask, state=None)
        x = self.predictor_norm(x)
        per_t = x.view(B, Tp, per_step, -1)[:, :, :tpf].mean(dim=2)                     # (B,T,D)
        w = torch.softmax(torch.arange(Tp, device=x.device, dtype=torch.float32) / 2.0, dim=0).view(1, -1, 1)
        pooled_future = (per_t * w).sum(dim=1)
        preds: List[Dict[str, torch.Tensor]] = []
        if predict_horizon > 0:
            fa = actions[:, Tp:Tp + predict_horizon] if actions.shape[1] > Tp else None
            fs = states[:, Tp:Tp + predict_horizon] if states.shape[1] > Tp else None
            preds = self.autoregressive_predictor(x, per_step, predict_horizon, future_actions=fa, future_states=fs,
                                                  action_state_embed=self.action_state_embed,
                                                  predictor_blocks=self.predictor_blocks, attn_mask=mask)
        now_logits, fut_logits = self.sterility_classifier(pooled_now, pooled_future)
        return preds, now_logits, fut_logits
