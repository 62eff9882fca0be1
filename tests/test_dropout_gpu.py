"""GPU: training-mode dropout (attention probabilities swm:326, MLP dropouts swm:700-702).

The library's masks come from a counter-based hash instead of torch's generator, so the reference's random
stream cannot be reproduced ("parity unpinned" for the stream itself).  What is checked instead:
  * the kernels' keep/drop decisions equal the oracle's numpy restatement of the documented hash, bit for bit;
  * with those identical masks, forward values and every gradient agree with the oracle's reference
    arithmetic (mask * x / (1 - p)) to the same tolerances as the eval-mode tests;
  * p = 0 reproduces the eval-mode result exactly, equal keys give equal results, the keep rate is 1 - p.
"""
import math

import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from vit_amd import ops as _ops
    return _ops


def rnd(shape, dev, seed, scale=1.0, dtype=F32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


# ------------------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("shape,p,key", [((8, 1024), 0.1, 1), ((777, 4096), 0.1, 0xDEADBEEF), ((64, 1024), 0.5, 42),
                                         ((3, 8), 0.9, 7)])
def test_dropout_f32_bit_exact(ops, dev, shape, p, key):
    x = rnd(shape, dev, 3)
    mask = R.drop_mask(key, shape, p).to(dev)
    out = ops.dropout(x, key, p, out=torch.empty_like(x))
    assert torch.equal(out, x * mask)
    resid = rnd(shape, dev, 4)
    out2 = ops.dropout(x, key, p, out=torch.empty_like(x), resid=resid)
    assert torch.equal(out2, x * mask + resid)
    inplace = ops.dropout(x.clone(), key, p)
    assert torch.equal(inplace, out)


def test_dropout_dtypes_and_keep_rate(ops, dev):
    shape, p, key = (2048, 4096), 0.1, 99
    x = rnd(shape, dev, 5, dtype=BF16)
    mask = R.drop_mask(key, shape, p).to(dev)
    out = ops.dropout(x, key, p, out=torch.empty_like(x))
    assert torch.equal(out, (x.float() * mask).to(BF16))
    out32 = ops.dropout(x, key, p, out=torch.empty(shape, dtype=F32, device=dev))
    assert torch.equal(out32, x.float() * mask)
    xf = rnd(shape, dev, 6)
    outb = ops.dropout(xf, key, p, out=torch.empty(shape, dtype=BF16, device=dev))
    assert torch.equal(outb, (xf * mask).to(BF16))
    kept = (mask > 0).float().mean().item()
    assert abs(kept - 0.9) < 1e-3
    # another key gives another, equally dense mask
    other = R.drop_mask(key + 1, shape, p)
    assert abs((other > 0).float().mean().item() - 0.9) < 1e-3
    assert ((other > 0) != (mask.cpu() > 0)).float().mean().item() > 0.15


def test_dropout_rejects_bad_arguments(ops, dev):
    from vit_amd._lib import VitAmdError
    x = rnd((4, 8), dev, 1)
    with pytest.raises(VitAmdError):
        ops.dropout(x, 1, 1.0)
    with pytest.raises(VitAmdError):
        ops.dropout(rnd((3, 5), dev, 1), 1, 0.1)      # 15 elements: not a multiple of 8


# ------------------------------------------------------------------------------ dropout inside the GEMM epilogues
@pytest.mark.parametrize("M,tile", [(700, 128), (2048 + 40, 256), (256 * 9, 0)])
def test_gemm_epilogue_dropout(ops, dev, M, tile):
    """The three fused sites against GEMM-then-va_dropout: identical masks (the documented hash over the contiguous
    output), f32 site bit-identical, bf16 sites equal up to the one rounding the fusion saves."""
    N, K, p, key = 1024, 512, 0.1, 0x1234ABCD
    x = rnd((M, K), dev, 11, dtype=BF16)
    w = rnd((N, K), dev, 12, 1.0 / math.sqrt(K), dtype=BF16)
    bias = rnd((N,), dev, 13, 0.5)
    resid = rnd((M, N), dev, 14)
    mask = R.drop_mask(key, (M, N), p).to(dev)            # 0 or 1 / (1 - p)
    # f32 + bias, dropout, + residual
    plain = torch.empty(M, N, dtype=F32, device=dev)
    ops.gemm_nt(x, w, ops.EPI_F32, plain, bias=bias, tile=tile)
    want = ops.dropout(plain, key, p, out=torch.empty_like(plain), resid=resid)
    got = torch.empty_like(plain)
    ops.gemm_nt(x, w, ops.EPI_F32, got, bias=bias, resid=resid, tile=tile, drop_key=key, drop_p=p)
    assert torch.equal(got, want)
    # bias + GELU: first output untouched, second one dropped
    pre0, act0 = torch.empty(M, N, dtype=BF16, device=dev), torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_BIAS_GELU, pre0, out2=act0, bias=bias, tile=tile)
    pre1, act1 = torch.empty_like(pre0), torch.empty_like(act0)
    ops.gemm_nt(x, w, ops.EPI_BIAS_GELU, pre1, out2=act1, bias=bias, tile=tile, drop_key=key, drop_p=p)
    assert torch.equal(pre1, pre0)
    assert torch.equal(act1 == 0, (mask == 0) | (act0 == 0))
    ref_act = R.gelu_erf((x.float() @ w.float().t() + bias).cpu()).to(dev) * mask
    assert rel_err(act1, ref_act) < 4e-3
    assert (act1.float() - act0.float() * mask).abs().max().item() <= 2 ** -7 * (act0.float().abs() * mask).max().item()
    # GELU backward
    d0, d1 = torch.empty(M, N, dtype=BF16, device=dev), torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_GELU_BWD, d0, aux=pre0, tile=tile)
    ops.gemm_nt(x, w, ops.EPI_GELU_BWD, d1, aux=pre0, tile=tile, drop_key=key, drop_p=p)
    assert torch.equal(d1 == 0, (mask == 0) | (d0 == 0))
    assert (d1.float() - d0.float() * mask).abs().max().item() <= 2 ** -7 * (d0.float().abs() * mask).max().item()
    # the derivative-keeping forward and the multiply epilogue: the same decisions, the derivative output untouched
    dg0, a0 = torch.empty_like(pre0), torch.empty_like(act0)
    ops.gemm_nt(x, w, ops.EPI_BIAS_GELU_GRAD, dg0, out2=a0, bias=bias, tile=tile)
    dg1, a1 = torch.empty_like(pre0), torch.empty_like(act0)
    ops.gemm_nt(x, w, ops.EPI_BIAS_GELU_GRAD, dg1, out2=a1, bias=bias, tile=tile, drop_key=key, drop_p=p)
    assert torch.equal(dg1, dg0) and torch.equal(a0, act0) and torch.equal(a1, act1)
    m0, m1 = torch.empty_like(d0), torch.empty_like(d0)
    ops.gemm_nt(x, w, ops.EPI_MUL_AUX, m0, aux=dg0, tile=tile)
    ops.gemm_nt(x, w, ops.EPI_MUL_AUX, m1, aux=dg0, tile=tile, drop_key=key, drop_p=p)
    assert torch.equal(m1 == 0, (mask == 0) | (m0 == 0))
    assert (m1.float() - m0.float() * mask).abs().max().item() <= 2 ** -7 * (m0.float().abs() * mask).max().item()
    # refusals: no dropout site in the plain epilogue, strided output, the four-wave kernel
    with pytest.raises(Exception):
        ops.gemm_nt(x, w, ops.EPI_BF16, d0, drop_key=key, drop_p=p)
    wide = torch.empty(M, 2 * N, dtype=F32, device=dev)
    with pytest.raises(Exception):
        ops.gemm_nt(x, w, ops.EPI_F32, wide[:, :N], drop_key=key, drop_p=p)
    if M >= 2048:
        with pytest.raises(Exception):
            ops.gemm_nt(x, w, ops.EPI_F32, got, tile=4256, drop_key=key, drop_p=p)


def test_layernorm_bwd_dropped_copy(ops, dev):
    """The bf16 copy of LayerNorm backward with a mask folded in: the f32 output and the parameter gradients are
    those of the plain call, the copy is dropout(f32 output) rounded once."""
    M, D, p, key = 517, 1024, 0.1, 0xBEEF
    x, dy, resid = rnd((M, D), dev, 21), rnd((M, D), dev, 22, dtype=BF16), rnd((M, D), dev, 23)
    gamma = rnd((D,), dev, 24) + 1.0
    mean = x.mean(1)
    rstd = torch.rsqrt(x.var(1, unbiased=False) + 1e-6)
    outs = []
    for kw in ({}, {"drop_key": key, "drop_p": p}):
        dx, dxb = torch.empty(M, D, dtype=F32, device=dev), torch.empty(M, D, dtype=BF16, device=dev)
        dg, db = torch.empty(D, dtype=F32, device=dev), torch.empty(D, dtype=F32, device=dev)
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, resid, dx, dxb, dg, db, **kw)
        outs.append((dx, dxb, dg, db))
    (dx0, dxb0, dg0, db0), (dx1, dxb1, dg1, db1) = outs
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(dxb0, dx0.to(BF16))
    mask = R.drop_mask(key, (M, D), p).to(dev)
    assert torch.equal(dxb1, (dx0 * mask).to(BF16))
    # column sums of the copy (the bias gradient of the layer whose gradient GEMMs read it), with and without a mask
    for kw, copy in (({}, dxb0), ({"drop_key": key, "drop_p": p}, dxb1)):
        dx, dxb = torch.empty(M, D, dtype=F32, device=dev), torch.empty(M, D, dtype=BF16, device=dev)
        dg, db = torch.empty(D, dtype=F32, device=dev), torch.empty(D, dtype=F32, device=dev)
        cs = torch.full((D,), 7.0, dtype=F32, device=dev)
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, resid, dx, dxb, dg, db, copy_colsum=cs, **kw)
        assert torch.equal(dxb, copy) and torch.equal(dg, dg0) and torch.equal(db, db0)
        want = copy.double().sum(0)
        assert (cs.double() - want).abs().max().item() <= 1e-5 * copy.double().abs().sum(0).max().item()


# ------------------------------------------------------------------------------ attention
def attn_reference(qkv, B, N, H, mask):
    D = qkv.shape[1] // 3
    hd = D // H
    q, k, v = (t.reshape(B, N, H, hd).permute(0, 2, 1, 3) for t in qkv.split(D, dim=1))
    s = (q @ k.transpose(-1, -2)) * hd ** -0.5
    p = s.softmax(-1)
    o = ((p * mask) @ v).permute(0, 2, 1, 3).reshape(B * N, D)
    return o, (torch.logsumexp(s, dim=-1) * math.log2(math.e)).reshape(B * H, N)


@pytest.mark.parametrize("B,N,H,p", [(2, 8, 16, 0.1), (1, 193, 16, 0.1), (2, 256, 16, 0.1), (1, 640, 16, 0.25),
                                     (2, 18, 8, 0.1), (1, 300, 8, 0.1), (1, 512, 8, 0.5), (1, 193, 32, 0.1), (2, 256, 32, 0.25)])
def test_attention_dropout_fwd_bwd(ops, dev, B, N, H, p):
    D, key = 1024, 0x1234ABCD + N
    qkv = rnd((B * N, 3 * D), dev, 21, dtype=BF16)
    mask = R.attn_drop_mask(key, B, H, N, p).to(dev)
    out = torch.full((B * N, D), float("nan"), dtype=BF16, device=dev)
    lse2 = torch.empty(B * H, N, dtype=F32, device=dev)
    ops.attn_fwd(qkv, B, N, H, out, lse2, drop_key=key, drop_p=p)
    leaf = qkv.float().requires_grad_(True)
    ref_o, ref_lse = attn_reference(leaf, B, N, H, mask)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, ref_o) < 8e-3
    assert (lse2 - ref_lse).abs().max().item() < 2e-2          # the normaliser ignores the mask
    d_out = rnd((B * N, D), dev, 22, dtype=BF16)
    ref_o.backward(d_out.float())
    dqkv = torch.full((B * N, 3 * D), float("nan"), dtype=BF16, device=dev)
    ops.attn_bwd(qkv, out, d_out, lse2, dqkv, B, N, H, drop_key=key, drop_p=p)
    assert torch.isfinite(dqkv.float()).all()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert rel_err(dqkv[:, sl], leaf.grad[:, sl]) < 1.5e-2, name


def test_attention_dropout_mask_is_the_documented_one(ops, dev):
    """V = identity columns turn the output row into the masked probabilities themselves: a dropped key gives an
    exact zero, a kept one a positive value -- the kernel's decisions are read off directly."""
    B, N, H, D, p, key = 1, 64, 16, 1024, 0.3, 77
    qkv = torch.zeros(N, 3 * D, dtype=BF16, device=dev)        # q = k = 0: uniform probabilities 1/64
    for h in range(H):
        qkv[torch.arange(N), 2 * D + h * 64 + torch.arange(N)] = 1.0
    out = torch.empty(N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(H, N, dtype=F32, device=dev)
    ops.attn_fwd(qkv, B, N, H, out, lse2, drop_key=key, drop_p=p)
    got = out.float().reshape(N, H, 64).permute(1, 0, 2) > 0   # [H, q, k]
    want = R.attn_drop_mask(key, B, H, N, p)[0] > 0
    assert torch.equal(got.cpu(), want)


def test_attention_dropout_zero_p_and_determinism(ops, dev):
    B, N, H, D = 1, 320, 16, 1024
    qkv = rnd((B * N, 3 * D), dev, 23, dtype=BF16)
    lse2 = torch.empty(B * H, N, dtype=F32, device=dev)
    base = ops.attn_fwd(qkv, B, N, H, torch.empty(B * N, D, dtype=BF16, device=dev), lse2)
    zero = ops.attn_fwd(qkv, B, N, H, torch.empty(B * N, D, dtype=BF16, device=dev), lse2, drop_key=5, drop_p=0.0)
    assert torch.equal(base, zero)
    a = ops.attn_fwd(qkv, B, N, H, torch.empty(B * N, D, dtype=BF16, device=dev), lse2, drop_key=5, drop_p=0.1)
    b = ops.attn_fwd(qkv, B, N, H, torch.empty(B * N, D, dtype=BF16, device=dev), lse2, drop_key=5, drop_p=0.1)
    c = ops.attn_fwd(qkv, B, N, H, torch.empty(B * N, D, dtype=BF16, device=dev), lse2, drop_key=6, drop_p=0.1)
    assert torch.equal(a, b)
    assert not torch.equal(a, c) and not torch.equal(a, base)


# ------------------------------------------------------------------------------ encoder, training mode
def test_encoder_training_mode_matches_oracle_with_same_masks(dev):
    from vit_amd import surgical_world_model as swm
    from vit_amd.encoder_engine import DropPlan
    depth, heads = 2, 16
    enc = swm.VideoViTEncoder(img_size=64, num_frames=4, embed_dim=1024, depth=depth, num_heads=heads)
    enc.load_state_dict(R.fill_state_dict_(enc.state_dict()))
    enc = enc.to(dev).train()
    video = R.hashed_uniform("dropout.video", (2, 3, 4, 64, 64), 1.0)
    ps = [(0.1, 0.1, 0.1)] * depth
    keys = [(11, 22, 33), (0xFFFFFFF0, 0x80000001, 12345)]
    enc._dropout_plan = lambda: DropPlan(ps, keys)
    y = enc(video.to(dev))
    (y ** 2).mean().backward()
    P = R.make_encoder_params(depth, requires_grad=True)
    ref = R.encoder_forward(P, video, depth, heads, drop_p=ps, drop_keys=keys)
    (ref ** 2).mean().backward()
    assert rel_err(y.cpu(), ref) < 2e-2
    eval_ref = R.encoder_forward(P, video, depth, heads)
    assert rel_err(eval_ref, ref) > 0.05                        # the masks really change the result
    got = dict(enc.named_parameters())
    for name in ("blocks.0.attn.qkv_proj.weight", "blocks.0.attn.out_proj.weight", "blocks.0.mlp.0.weight",
                 "blocks.0.mlp.0.bias", "blocks.1.mlp.3.weight", "blocks.1.mlp.3.bias", "blocks.1.norm2.weight",
                 "tubelet_embed.projection.weight", "norm.weight"):
        assert rel_err(got[name].grad.cpu(), P[name].grad) < 3e-2, name


def test_encoder_training_mode_seeding(dev):
    from vit_amd import surgical_world_model as swm
    enc = swm.VideoViTEncoder(img_size=64, num_frames=4, embed_dim=1024, depth=2, num_heads=16)
    enc.load_state_dict(R.fill_state_dict_(enc.state_dict()))
    enc = enc.to(dev)
    video = R.hashed_uniform("dropout.video2", (1, 3, 4, 64, 64), 1.0).to(dev)
    with torch.no_grad():
        enc.eval()
        e1 = enc(video)
        enc.train()
        torch.manual_seed(7)
        t1 = enc(video)
        t2 = enc(video)
        torch.manual_seed(7)
        t3 = enc(video)
        for blk in enc.blocks:                    # p = 0 everywhere: training mode equals eval mode
            blk.attn.dropout.p = blk.mlp[2].p = blk.mlp[4].p = 0.0
        t0 = enc(video)
    assert torch.equal(t1, t3)
    assert not torch.equal(t1, t2)
    assert not torch.equal(t1, e1)
    assert torch.equal(t0, e1)
