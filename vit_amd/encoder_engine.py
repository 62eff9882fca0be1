"""Host-side schedule of the encoder forward and backward over the C-ABI kernels.

The engine replaces the eager op sequence of the reference's `VideoViTEncoder.forward`
(surgical_world_model.py:772-817) and its autograd backward with explicit launches:

  forward, per clip batch (M = B*N tokens, D = 1024):
    tubelet gather -> patch GEMM(+bias, f32 out)                       [swm:92-109]
    per block:  LN1 -> QKV GEMM -> 3D-RoPE in place -> flash attention  [swm:283-331]
                -> out-proj GEMM (+residual, f32)                       [swm:334, 707]
                -> LN2 -> fc1 GEMM (+bias, GELU) -> fc2 GEMM (+bias +residual, f32)  [swm:698-713]
    final LN (f32 out)                                                  [swm:815]

  backward mirrors it; every data-gradient GEMM runs against a transposed bf16 copy of the weight,
  every weight-gradient GEMM accumulates in f32 into one flat gradient buffer per block, which is what
  the data-parallel reducer all-reduces as soon as the block's backward has been enqueued.

The residual stream, LayerNorm statistics, softmax statistics and all accumulators are f32; GEMM and
attention operands are bf16.

Dropout (p = 0.1 in the reference's blocks, swm:752) is the identity in eval mode.  In training mode the three
dropouts of a block (attention probabilities swm:326, after GELU swm:700, after fc2 swm:702) use counter-based
masks: each site of each block gets a 32-bit key per forward, the forward kernels derive keep/drop from
(key, element index), and the backward regenerates the same decisions from the saved keys -- no mask is stored.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
LN_EPS = 1e-6
# What the fc1 forward keeps for the backward pass: GELU'(pre-activation), consumed by a multiply epilogue (default), or,
# with VIT_AMD_KEEP_PREACT=1, the pre-activation itself, consumed by the epilogue that evaluates GELU' (for A/B timing).
import os as _os
_KEEP_PREACT = _os.environ.get("VIT_AMD_KEEP_PREACT", "0") == "1"
_EPI_FC1_KEEP = ops.EPI_BIAS_GELU if _KEEP_PREACT else ops.EPI_BIAS_GELU_GRAD
_EPI_FC2_BWD = ops.EPI_GELU_BWD if _KEEP_PREACT else ops.EPI_MUL_AUX

BLOCK_PARAM_NAMES = ("norm1.weight", "norm1.bias", "attn.qkv_proj.weight", "attn.out_proj.weight",
                     "norm2.weight", "norm2.bias", "mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias")


def encoder_param_names(depth: int) -> List[str]:
    """Parameter order used across the autograd boundary (state-dict order of the reference encoder)."""
    names = ["tubelet_embed.projection.weight", "tubelet_embed.projection.bias"]
    for i in range(depth):
        names += [f"blocks.{i}.{n}" for n in BLOCK_PARAM_NAMES]
    names += ["norm.weight", "norm.bias"]
    return names


@dataclass
class EncoderConfig:
    depth: int
    heads: int
    dim: int = 1024
    tubelet: Tuple[int, int, int] = (2, 16, 16)
    in_channels: int = 3
    temporal_scale: float = 0.6

    def check(self):
        if self.dim != 1024:
            raise ops._lib.VitAmdError("the encoder kernels are specialised for width 1024 (the only width the "
                                       "reference's 256/384/384 RoPE split admits)")
        hd = self.dim // self.heads
        if self.dim % self.heads or hd not in (32, 64, 128):
            raise ops._lib.VitAmdError(f"head dim {hd} (heads={self.heads}) has no attention kernel; use 32, 16 or 8 heads")


class WeightCache:
    """bf16 (and transposed bf16) copies of the f32 master weights, refreshed when a parameter changes."""

    def __init__(self):
        self._fwd: Dict[str, Tuple[int, int, torch.Tensor]] = {}
        self._bwd: Dict[str, Tuple[int, int, torch.Tensor]] = {}
        self._plan_key = None
        self._plan = None

    @staticmethod
    def _key(p: torch.Tensor):
        return (p.data_ptr(), p._version)

    def invalidate(self) -> None:
        """Forget which parameter versions the copies were made from (the buffers are kept and reused)."""
        self._fwd = {k: (0, -1, v[2]) for k, v in self._fwd.items()}
        self._bwd = {k: (0, -1, v[2]) for k, v in self._bwd.items()}

    def prepare(self, items, transposed: bool) -> None:
        """Refresh, in ONE launch, the copies of every listed (name, parameter) that changed since they were made
        (after an optimizer step: all of them).  fwd() / bwd() then find them current; whatever is not handled
        here (sides that are not multiples of 64) is still made on first use."""
        jobs = []
        for name, p in items:
            key = self._key(p)
            f, b = self._fwd.get(name), self._bwd.get(name)
            need_f = f is None or (f[0], f[1]) != key
            need_b = transposed and (b is None or (b[0], b[1]) != key)
            w2d = p.detach().reshape(p.shape[0], -1)
            if not (need_f or need_b) or w2d.shape[0] % 64 or w2d.shape[1] % 64 or not w2d.is_contiguous():
                continue
            R, C = w2d.shape
            fbuf = bbuf = None
            if need_f:
                fbuf = f[2] if f is not None and tuple(f[2].shape) == (R, C) else torch.empty(R, C, dtype=BF16, device=p.device)
            if need_b:
                bbuf = b[2] if b is not None and tuple(b[2].shape) == (C, R) else torch.empty(C, R, dtype=BF16, device=p.device)
            jobs.append((name, key, w2d, fbuf, bbuf))
        if not jobs:
            return
        plan_key = tuple((n, w.data_ptr(), None if f is None else f.data_ptr(), None if b is None else b.data_ptr())
                         for n, _, w, f, b in jobs)
        if plan_key != self._plan_key:
            self._plan = ops.CastPlan([(w, f, b) for _, _, w, f, b in jobs])
            self._plan_key = plan_key
        self._plan.run()
        for name, key, _, fbuf, bbuf in jobs:
            if fbuf is not None:
                self._fwd[name] = (*key, fbuf)
            if bbuf is not None:
                self._bwd[name] = (*key, bbuf)

    def fwd_cat(self, name: str, params) -> torch.Tensor:
        """bf16 copy of the row-wise concatenation of several [*, K] weights (one GEMM then serves all of them)."""
        key = tuple(self._key(p) for p in params)
        ent = self._fwd.get(name)
        if ent is None or ent[0] != key:
            rows = sum(p.shape[0] for p in params)
            buf = ent[2] if ent is not None and ent[2].shape[0] == rows else torch.empty(
                rows, params[0].shape[1], dtype=BF16, device=params[0].device)
            at = 0
            for p in params:
                ops.cast_bf16(p.detach(), buf[at:at + p.shape[0]])
                at += p.shape[0]
            ent = (key, 0, buf)
            self._fwd[name] = ent
        return ent[2]

    def fwd(self, name: str, p: torch.Tensor) -> torch.Tensor:
        ent = self._fwd.get(name)
        if ent is None or (ent[0], ent[1]) != self._key(p):
            w2d = p.detach().reshape(p.shape[0], -1)
            buf = ent[2] if ent is not None and ent[2].shape == w2d.shape else None
            ent = (*self._key(p), ops.cast_bf16(w2d, buf))
            self._fwd[name] = ent
        return ent[2]

    def bwd(self, name: str, p: torch.Tensor) -> torch.Tensor:
        ent = self._bwd.get(name)
        if ent is None or (ent[0], ent[1]) != self._key(p):
            w2d = p.detach().reshape(p.shape[0], -1)
            buf = ent[2] if ent is not None and ent[2].shape == (w2d.shape[1], w2d.shape[0]) else None
            ent = (*self._key(p), ops.cast_transpose_bf16(w2d, buf))
            self._bwd[name] = ent
        return ent[2]


@dataclass
class _BlockSaved:
    x_in: torch.Tensor
    mean1: torch.Tensor
    rstd1: torch.Tensor
    ln1: torch.Tensor
    qkv: torch.Tensor
    attn: torch.Tensor
    lse2: torch.Tensor
    x_mid: torch.Tensor
    mean2: torch.Tensor
    rstd2: torch.Tensor
    ln2: torch.Tensor
    pre: torch.Tensor
    act: torch.Tensor


@dataclass
class EncoderContext:
    cfg: EncoderConfig
    B: int
    N: int
    grid: Tuple[int, int, int]
    tables: torch.Tensor
    patches: torch.Tensor
    blocks: List[_BlockSaved] = field(default_factory=list)
    x_last: Optional[torch.Tensor] = None
    mean_f: Optional[torch.Tensor] = None
    rstd_f: Optional[torch.Tensor] = None
    drop: Optional["DropPlan"] = None
    video_shape: Optional[Tuple[int, ...]] = None
    checkpointed: bool = False          # blocks hold only x_in; the backward recomputes the rest


@dataclass
class DropPlan:
    """Training-mode dropout of one forward: per block the probabilities and keys of its three sites
    (attention probabilities, after GELU, after fc2)."""
    p: List[Tuple[float, float, float]]
    keys: List[Tuple[int, int, int]]

    @staticmethod
    def draw(p: List[Tuple[float, float, float]]) -> "DropPlan":
        """Keys from torch's default CPU generator, so torch.manual_seed makes a training run repeatable."""
        k = torch.randint(0, 2 ** 32, (len(p), 3), dtype=torch.int64).tolist()
        return DropPlan([tuple(float(v) for v in t) for t in p], [tuple(int(v) for v in t) for t in k])


class EncoderEngine:
    def __init__(self, cfg: EncoderConfig):
        cfg.check()
        self.cfg = cfg
        self.weights = WeightCache()
        self._tables: Dict[Tuple, torch.Tensor] = {}

    # ---------------------------------------------------------------- helpers
    def rope_pos(self, batch: int, grid, device) -> torch.Tensor:
        """Per token row its packed grid coordinates (what the fused QKV + rotary GEMM indexes the tables with)."""
        key = ("pos", batch, grid, str(device))
        t = self._tables.get(key)
        if t is None:
            t = self._tables[key] = ops.rope_pos_code(batch, grid, device)
        return t

    def rope_tables(self, P: Dict[str, torch.Tensor], grid, device) -> torch.Tensor:
        key = (grid, str(device))
        t = self._tables.get(key)
        if t is None:
            t = ops.rope_tables_device(P["rope.inv_freq_t"], P["rope.inv_freq_h"], P["rope.inv_freq_w"], grid,
                                       self.cfg.temporal_scale, device)
            self._tables[key] = t
        return t

    def _geometry(self, video: torch.Tensor):
        if video.dim() != 5:
            raise ValueError(f"video must be (B,C,T,H,W), got {tuple(video.shape)}")
        B, C, T, H, W = video.shape
        tt, th, tw = self.cfg.tubelet
        if C != self.cfg.in_channels:
            raise ValueError(f"expected {self.cfg.in_channels} channels, got {C}")
        grid = (T // tt, H // th, W // tw)
        N = grid[0] * grid[1] * grid[2]
        if N == 0:
            raise ValueError("clip smaller than one tubelet")
        return B, grid, N

    # ---------------------------------------------------------------- forward
    def forward(self, P: Dict[str, torch.Tensor], video: torch.Tensor, save: bool, drop: Optional[DropPlan] = None,
                checkpoint: bool = False):
        """P: name -> f32 device tensor (parameters and the rope.inv_freq_* buffers).
        drop: training-mode dropout plan, None for the eval-mode arithmetic.
        checkpoint (with save): keep only every block's input and let the backward run the block's forward again
        (swm:807-811: activation checkpointing per block); the dropout masks are functions of their keys, so the second
        forward reproduces the first bit for bit.
        Returns (features (B,N,D) f32, EncoderContext or None)."""
        cfg = self.cfg
        dev = video.device
        if not video.is_cuda:
            raise ops._lib.VitAmdError("VideoViTEncoder runs on a HIP device only (no CPU path)")
        B, grid, N = self._geometry(video)
        M, D, H = B * N, cfg.dim, cfg.heads
        video = video.detach().to(F32).contiguous()
        Wc = self.weights
        tables = self.rope_tables(P, grid, dev)

        def new(cols, dtype):
            return torch.empty(M, cols, dtype=dtype, device=dev)

        weight_items = [("pe", P["tubelet_embed.projection.weight"])]
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            weight_items += [(p + "qkv", P[p + "attn.qkv_proj.weight"]), (p + "out", P[p + "attn.out_proj.weight"]),
                             (p + "fc1", P[p + "mlp.0.weight"]), (p + "fc2", P[p + "mlp.3.weight"])]
        Wc.prepare(weight_items, transposed=save)

        patches = ops.tubelet_gather(video, cfg.tubelet)
        x = new(D, F32)
        ops.gemm_nt(patches, Wc.fwd("pe", P["tubelet_embed.projection.weight"]), ops.EPI_F32, x,
                    bias=P["tubelet_embed.projection.bias"])
        ctx = EncoderContext(cfg, B, N, grid, tables, patches, drop=drop, video_shape=tuple(video.shape)) if save else None
        if drop is not None and (len(drop.p) != cfg.depth or len(drop.keys) != cfg.depth):
            raise ValueError("dropout plan does not match the encoder depth")

        # without a backward -- or when the backward recomputes them -- the activations of one block are recycled for
        # the next
        scratch: Dict[str, torch.Tensor] = {}
        keep = save and not checkpoint
        if ctx is not None:
            ctx.checkpointed = checkpoint
        for i in range(cfg.depth):
            x_next, saved = self._block_forward(P, i, x, B, N, grid, tables, drop, keep, scratch,
                                                fresh_output=save)
            if save:
                ctx.blocks.append(saved if keep else _BlockSaved(x, *([None] * 12)))
            x = x_next

        y = new(D, F32)
        mean_f = rstd_f = None
        if save:
            mean_f, rstd_f = torch.empty(M, dtype=F32, device=dev), torch.empty(M, dtype=F32, device=dev)
        ops.layernorm_fwd(x, P["norm.weight"], P["norm.bias"], y, mean_f, rstd_f, LN_EPS)
        if save:
            ctx.x_last, ctx.mean_f, ctx.rstd_f = x, mean_f, rstd_f
        return y.view(B, N, D), ctx

    def _block_forward(self, P, i: int, x: torch.Tensor, B: int, N: int, grid, tables, drop: Optional[DropPlan],
                       keep: bool, scratch: Dict[str, torch.Tensor], fresh_output: bool, need_output: bool = True):
        """One block (swm:705-714) from its input x (M, D) f32.  keep: every intermediate gets its own buffer and comes
        back as a _BlockSaved; otherwise intermediates live in `scratch` (recycled from block to block) and the
        pre-activation is not written at all.  fresh_output: the block's output gets a buffer of its own (it is some
        later computation's saved input) instead of one of two alternating scratch buffers.  need_output = False
        stops after the first MLP layer (the recomputation in a checkpointed backward has no use for the block's
        output)."""
        cfg = self.cfg
        dev = x.device
        M, D, H = B * N, cfg.dim, cfg.heads
        Wc = self.weights
        p = f"blocks.{i}."

        def buf(name, cols, dtype):
            if keep:
                return torch.empty(M, cols, dtype=dtype, device=dev)
            t = scratch.get(name)
            if t is None:
                t = scratch[name] = torch.empty(M, cols, dtype=dtype, device=dev)
            return t

        def stat(name):
            # LayerNorm statistics exist for the backward only: a pass that keeps nothing does not have them written
            return torch.empty(M, dtype=F32, device=dev) if keep else None

        mean1, rstd1 = stat("mean1"), stat("rstd1")
        ln1 = buf("ln1", D, BF16)
        ops.layernorm_fwd(x, P[p + "norm1.weight"], P[p + "norm1.bias"], ln1, mean1, rstd1, LN_EPS)
        qkv = buf("qkv", 3 * D, BF16)
        # the projection with the rotation in its epilogue; small problems take the two-pass route
        if not ops.gemm_nt_qkv_rope(ln1, Wc.fwd(p + "qkv", P[p + "attn.qkv_proj.weight"]), qkv, tables, self.rope_pos(B, grid, dev), grid):
            ops.gemm_nt(ln1, Wc.fwd(p + "qkv", P[p + "attn.qkv_proj.weight"]), ops.EPI_BF16, qkv)
            ops.rope3d_qk_(qkv, tables, grid)
        attn = buf("attn", D, BF16)
        lse2 = torch.empty(B * H, N, dtype=F32, device=dev) if keep else scratch.setdefault(
            "lse2", torch.empty(B * H, N, dtype=F32, device=dev))
        pa, p1, p2 = drop.p[i] if drop is not None else (0.0, 0.0, 0.0)
        ka, k1, k2 = drop.keys[i] if drop is not None else (0, 0, 0)
        ops.attn_fwd(qkv, B, N, H, attn, lse2, drop_key=ka, drop_p=pa)
        x_mid = buf("x_mid", D, F32)
        ops.gemm_nt(attn, Wc.fwd(p + "out", P[p + "attn.out_proj.weight"]), ops.EPI_F32, x_mid, resid=x)
        mean2, rstd2 = stat("mean2"), stat("rstd2")
        ln2 = buf("ln2", D, BF16)
        ops.layernorm_fwd(x_mid, P[p + "norm2.weight"], P[p + "norm2.bias"], ln2, mean2, rstd2, LN_EPS)
        act = buf("act", 4 * D, BF16)
        # both MLP dropouts ride in the GEMM epilogues; the saved activation is the dropped one
        if keep:
            # what the backward pass needs of the pre-activation is GELU'(pre): the epilogue that has exp and rcp of it in
            # registers anyway stores that (same bytes), and the backward epilogue is a multiplication
            pre = torch.empty(M, 4 * D, dtype=BF16, device=dev)
            ops.gemm_nt(ln2, Wc.fwd(p + "fc1", P[p + "mlp.0.weight"]), _EPI_FC1_KEEP, pre, out2=act,
                        bias=P[p + "mlp.0.bias"], drop_key=k1, drop_p=p1)
        else:           # nothing is kept for a backward: the pre-activation (half of this GEMM's output bytes) is not written
            pre = None
            ops.gemm_nt(ln2, Wc.fwd(p + "fc1", P[p + "mlp.0.weight"]), ops.EPI_GELU_ONLY, act,
                        bias=P[p + "mlp.0.bias"], drop_key=k1, drop_p=p1)
        x_next = None
        if need_output:
            # the output must not land in the buffer the input lives in: alternate between two when recycling
            x_next = torch.empty(M, D, dtype=F32, device=dev) if (keep or fresh_output) else buf("x_out" + str(i & 1), D, F32)
            ops.gemm_nt(act, Wc.fwd(p + "fc2", P[p + "mlp.3.weight"]), ops.EPI_F32, x_next,
                        bias=P[p + "mlp.3.bias"], resid=x_mid, drop_key=k2, drop_p=p2)
        saved = _BlockSaved(x, mean1, rstd1, ln1, qkv, attn, lse2, x_mid, mean2, rstd2, ln2, pre, act) if keep else None
        return x_next, saved

    # ---------------------------------------------------------------- backward
    def backward(self, P: Dict[str, torch.Tensor], ctx: EncoderContext, grad_out: torch.Tensor,
                 on_block_done: Optional[Callable[[torch.Tensor], None]] = None,
                 input_grad: bool = False) -> Dict[str, torch.Tensor]:
        """grad_out: (B,N,D).  Returns name -> f32 gradient (views into one flat buffer per block); with
        `input_grad` also "video" -> gradient of the clip (B,C,T,H,W), the input half of the Conv3d backward.
        `on_block_done(flat)` is called right after a block's (and finally the stem's) gradient launches
        have been enqueued; the data-parallel reducer hooks in there."""
        cfg = self.cfg
        B, N, grid, tables = ctx.B, ctx.N, ctx.grid, ctx.tables
        M, D, H = B * N, cfg.dim, cfg.heads
        dev = grad_out.device
        Wc = self.weights
        gy = grad_out.detach().to(F32).contiguous().view(M, D)
        grads: Dict[str, torch.Tensor] = {}

        def carve(names_shapes):
            total = sum(int(torch.Size(s).numel()) for _, s in names_shapes)
            total_padded = (total + 3) // 4 * 4
            flat = torch.zeros(total_padded, dtype=F32, device=dev)
            off = 0
            for nme, shp in names_shapes:
                n = int(torch.Size(shp).numel())
                grads[nme] = flat[off:off + n].view(shp)
                off += n
            return flat

        g_a = torch.empty(M, D, dtype=F32, device=dev)
        g_b = torch.empty(M, D, dtype=F32, device=dev)
        g_bf = torch.empty(M, D, dtype=BF16, device=dev)
        d_small = torch.empty(M, D, dtype=BF16, device=dev)       # dLN inputs / dO
        d_wide = torch.empty(M, 4 * D, dtype=BF16, device=dev)    # d(pre-activation)
        dqkv = torch.empty(M, 3 * D, dtype=BF16, device=dev)
        drop = ctx.drop

        # The bf16 copy of the residual-stream gradient that enters block i from above is read only by block i's fc2
        # gradient GEMMs, which sit behind the fc2 dropout: the kernel that writes the copy applies that mask.
        def fc2_drop(i):
            return (drop.keys[i][2], drop.p[i][2]) if drop is not None and i >= 0 else (0, 0.0)

        # ... and its column sums are that block's fc2 bias gradient, so the block's gradient buffer is carved one
        # step ahead of its backward and the weight-gradient GEMM of fc2 runs without the bias sums.
        def carve_block(i):
            pre = f"blocks.{i}."
            return carve([(pre + n, tuple(P[pre + n].shape)) for n in BLOCK_PARAM_NAMES])

        tail = carve([("norm.weight", (D,)), ("norm.bias", (D,))])
        flat_next = carve_block(cfg.depth - 1) if cfg.depth > 0 else None
        kd, pd = fc2_drop(cfg.depth - 1)
        ops.layernorm_bwd(gy, ctx.x_last, P["norm.weight"], ctx.mean_f, ctx.rstd_f, None, g_a, g_bf,
                          grads["norm.weight"], grads["norm.bias"], drop_key=kd, drop_p=pd,
                          copy_colsum=grads[f"blocks.{cfg.depth - 1}.mlp.3.bias"] if cfg.depth > 0 else None)
        if on_block_done is not None:
            on_block_done(tail)
        ctx.x_last = None

        for i in reversed(range(cfg.depth)):
            p = f"blocks.{i}."
            s = ctx.blocks[i]
            if ctx.checkpointed:        # only the block's input was kept: run its forward again, now keeping everything
                _, s = self._block_forward(P, i, s.x_in, B, N, grid, tables, drop, True, {}, fresh_output=False,
                                           need_output=False)
            flat = flat_next
            pa, p1, p2 = drop.p[i] if drop is not None else (0.0, 0.0, 0.0)
            ka, k1, k2 = drop.keys[i] if drop is not None else (0, 0, 0)
            # ---- MLP (g_bf arrives with the fc2 dropout's mask already applied; g_a is the undropped gradient)
            g_mlp = g_bf
            ops.gemm_nt(g_mlp, Wc.bwd(p + "fc2", P[p + "mlp.3.weight"]), _EPI_FC2_BWD, d_wide, aux=s.pre,
                        drop_key=k1, drop_p=p1)       # s.pre holds GELU'(pre-activation); the mask commutes with it
            ops.gemm_tn_accum(g_mlp, s.act, grads[p + "mlp.3.weight"])     # the bias gradient came with g_bf
            ops.gemm_nt(d_wide, Wc.bwd(p + "fc1", P[p + "mlp.0.weight"]), ops.EPI_BF16, d_small)
            ops.gemm_tn_accum(d_wide, s.ln2, grads[p + "mlp.0.weight"], dbias=grads[p + "mlp.0.bias"])
            ops.layernorm_bwd(d_small, s.x_mid, P[p + "norm2.weight"], s.mean2, s.rstd2, g_a, g_b, g_bf,
                              grads[p + "norm2.weight"], grads[p + "norm2.bias"])
            # ---- attention
            ops.gemm_nt(g_bf, Wc.bwd(p + "out", P[p + "attn.out_proj.weight"]), ops.EPI_BF16, d_small)
            ops.gemm_tn_accum(g_bf, s.attn, grads[p + "attn.out_proj.weight"])
            ops.attn_bwd(s.qkv, s.attn, d_small, s.lse2, dqkv, B, N, H, drop_key=ka, drop_p=pa)
            ops.rope3d_qk_(dqkv, tables, grid, inverse=True)
            ops.gemm_nt(dqkv, Wc.bwd(p + "qkv", P[p + "attn.qkv_proj.weight"]), ops.EPI_BF16, d_small)
            ops.gemm_tn_accum(dqkv, s.ln1, grads[p + "attn.qkv_proj.weight"])
            kd, pd = fc2_drop(i - 1)
            flat_next = carve_block(i - 1) if i > 0 else None
            ops.layernorm_bwd(d_small, s.x_in, P[p + "norm1.weight"], s.mean1, s.rstd1, g_b, g_a, g_bf,
                              grads[p + "norm1.weight"], grads[p + "norm1.bias"], drop_key=kd, drop_p=pd,
                              copy_colsum=grads[f"blocks.{i - 1}.mlp.3.bias"] if i > 0 else None)
            if on_block_done is not None:
                on_block_done(flat)
            ctx.blocks[i] = None   # release this block's activations

        pw = P["tubelet_embed.projection.weight"]
        stem = carve([("tubelet_embed.projection.weight", tuple(pw.shape)), ("tubelet_embed.projection.bias", (D,))])
        ops.gemm_tn_accum(g_bf, ctx.patches, grads["tubelet_embed.projection.weight"].view(D, -1),
                          dbias=grads["tubelet_embed.projection.bias"])
        if on_block_done is not None:
            on_block_done(stem)
        if input_grad:
            d_patches = torch.empty(M, ctx.patches.shape[1], dtype=F32, device=dev)
            ops.gemm_nt(g_bf, Wc.bwd("pe", pw), ops.EPI_F32, d_patches)
            grads["video"] = ops.tubelet_scatter(d_patches, ctx.video_shape, cfg.tubelet)
        return grads
