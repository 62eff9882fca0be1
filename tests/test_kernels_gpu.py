"""GPU: every C-ABI kernel against a plain fp32 PyTorch / oracle computation of the same op.

Inputs are rounded to bf16 first where the kernel consumes bf16, so differences come only from the
accumulation order (f32 MFMA accumulate) and from the bf16 rounding of the stored output.
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


# ------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("M,N,K,tile", [(8, 1024, 1024, 128), (300, 3072, 1024, 128), (1000, 1024, 1536, 128),
                                        (2048, 4096, 1024, 256), (2048 + 17, 1024, 4096, 256), (513, 3072, 1024, 256),
                                        (300, 256, 128, 256), (2048, 1024, 1536, 256), (2048, 4096, 1024, 1256), (2048, 3072, 1024, 2256), (700, 1024, 128, 256), (700, 1024, 192, 256),
                                        (4096, 1024, 1024, 0), (10240, 3072, 256, 256), (256 * 37, 1792, 128, 256),
                                        (2048, 4096, 1024, 4256), (2048 + 17, 1024, 4096, 4256), (513, 3072, 128, 4256),
                                        (300, 256, 192, 4256), (256 * 300 + 5, 512, 320, 4256), (65536, 1024, 1024, 4256)])
def test_gemm_nt_plain(ops, dev, M, N, K, tile):
    x = rnd((M, K), dev, 1, dtype=BF16)
    w = rnd((N, K), dev, 2, 1.0 / math.sqrt(K), dtype=BF16)
    out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=tile)
    ref = x.float() @ w.float().t()
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, ref) < 4e-3
    # element-wise: bf16 rounding of an O(1) value
    assert (out.float() - ref).abs().max().item() < 0.06


@pytest.mark.parametrize("tile", [256, 4256])
def test_gemm_nt_strided_operands(ops, dev, tile):
    """Row strides larger than the row length on every operand (views into wider buffers), ragged M."""
    M, N, K = 2048 + 100, 512, 320
    xs = rnd((M, 2 * K + 64), dev, 31, dtype=BF16)
    ws = rnd((N, K + 8), dev, 32, 1.0 / math.sqrt(K), dtype=BF16)
    x, w = xs[:, 64:64 + K], ws[:, :K]
    ref = x.float() @ w.float().t()
    wide = torch.full((M, N + 256), 7.0, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_BF16, wide[:, 128:128 + N], tile=tile)
    assert rel_err(wide[:, 128:128 + N], ref) < 4e-3
    assert (wide[:, :128] == 7).all() and (wide[:, 128 + N:] == 7).all()      # nothing outside the view is touched
    wide32 = torch.full((M, N + 64), 7.0, dtype=F32, device=dev)
    resid = rnd((M, N + 64), dev, 33)
    bias = rnd((N,), dev, 34)
    ops.gemm_nt(x, w, ops.EPI_F32, wide32[:, :N], bias=bias, resid=resid[:, :N], tile=tile)
    assert (wide32[:, :N] - (ref + bias + resid[:, :N])).abs().max().item() < 2e-3
    assert (wide32[:, N:] == 7).all()


def test_gemm_nt_persistent_tickets(ops, dev):
    """The persistent kernel hands tiles out from counters that it rewinds itself: back-to-back launches and launches
    running at the same time on two streams must each cover every tile exactly once (bit-identical results), for
    tile counts below, at and above the number of CUs and not a multiple of 8."""
    K = 256
    w = rnd((768, K), dev, 41, 1.0 / math.sqrt(K), dtype=BF16)
    cases = []
    for M in (256 * 3 + 17, 256 * 86, 256 * 171 + 5):       # 12, 258 and 516 tiles
        x = rnd((M, K), dev, 42 + M, dtype=BF16)
        ref = torch.empty(M, 768, dtype=BF16, device=dev)
        ops.gemm_nt(x, w, ops.EPI_BF16, ref, tile=256)       # one workgroup per tile, no counters
        cases.append((x, ref))
    side = torch.cuda.Stream()
    outs = [[torch.zeros_like(ref) for _ in range(2)] for _, ref in cases]
    torch.cuda.synchronize()
    for rep in range(6):
        for (x, ref), (o_main, o_side) in zip(cases, outs):
            o_main.zero_(); o_side.zero_()
            side.wait_stream(torch.cuda.current_stream())
            ops.gemm_nt(x, w, ops.EPI_BF16, o_main, tile=4256)
            with torch.cuda.stream(side):
                ops.gemm_nt(x, w, ops.EPI_BF16, o_side, tile=4256)
            ops.gemm_nt(x, w, ops.EPI_BF16, o_main, tile=4256)          # again, right behind the first
            torch.cuda.current_stream().wait_stream(side)
            assert torch.equal(o_main, ref) and torch.equal(o_side, ref)


def test_gemm_nt_layout_is_not_transposed(ops, dev):
    """x = [I | 0] picks rows of w^T: catches a swapped row/column map that a random check could miss."""
    M, N, K = 128, 256, 128
    x = torch.zeros(M, K, dtype=BF16, device=dev)
    x[torch.arange(M), torch.arange(M) % K] = 1
    w = (torch.arange(N, device=dev)[:, None] * 3 + torch.arange(K, device=dev)[None, :] % 7).float()
    w = (w % 61 - 30).to(BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=128)
    assert torch.equal(out.float(), (x.float() @ w.float().t()))


@pytest.mark.parametrize("tile,M", [(128, 700), (256, 700), (1256, 700), (2256, 700), (256, 256 * 9), (4256, 700),
                                    (4256, 256 * 9), (4256, 256 * 1100 + 3), (0, 1), (0, 255), (0, 257), (4256, 1), (256, 1)])
def test_gemm_nt_epilogues(ops, dev, tile, M):
    N, K = 1024, 1024
    x = rnd((M, K), dev, 3, dtype=BF16)
    w = rnd((N, K), dev, 4, 1.0 / math.sqrt(K), dtype=BF16)
    bias = rnd((N,), dev, 5, 0.5)
    resid = rnd((M, N), dev, 6)
    acc = x.float() @ w.float().t()
    # f32 + bias + residual
    out = torch.empty(M, N, dtype=F32, device=dev)
    ops.gemm_nt(x, w, ops.EPI_F32, out, bias=bias, resid=resid, tile=tile)
    assert (out - (acc + bias + resid)).abs().max().item() < 2e-3
    out = torch.empty(M, N, dtype=F32, device=dev)
    ops.gemm_nt(x, w, ops.EPI_F32, out, tile=tile)
    assert (out - acc).abs().max().item() < 2e-3
    # bias + exact GELU
    pre = torch.empty(M, N, dtype=BF16, device=dev)
    act = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_BIAS_GELU, pre, out2=act, bias=bias, tile=tile)
    assert rel_err(pre, acc + bias) < 4e-3
    assert rel_err(act, torch.nn.functional.gelu(acc + bias)) < 5e-3
    # the same without the pre-activation output (no-grad passes): the activation must be the very same bits
    act_only = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_GELU_ONLY, act_only, bias=bias, tile=tile)
    assert torch.equal(act_only, act)
    # GELU backward epilogue
    g = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_GELU_BWD, g, aux=pre, tile=tile)
    p = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(p).backward(acc)
    assert rel_err(g, p.grad) < 5e-3
    # the forward that keeps GELU'(pre-activation) instead of the pre-activation, and the multiply epilogue that uses it:
    # the activation is the very same bits, the derivative is the f32 pre-activation's (not the rounded one's)
    dact = torch.empty(M, N, dtype=BF16, device=dev)
    act2 = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_BIAS_GELU_GRAD, dact, out2=act2, bias=bias, tile=tile)
    assert torch.equal(act2, act)
    h = (acc + bias).double()
    dref = 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0))) + h * torch.exp(-0.5 * h * h) / math.sqrt(2.0 * math.pi)
    assert (dact.double() - dref).abs().max().item() < 6e-3          # bf16 rounding of values up to ~1.13
    assert rel_err(dact, dref.float()) < 3e-3
    g2 = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_MUL_AUX, g2, aux=dact, tile=tile)
    assert rel_err(g2, acc * dact.float()) < 3e-3
    assert rel_err(g2, acc * dref.float()) < 5e-3


def test_gemm_nt_rejects_bad_shapes(ops, dev):
    from vit_amd._lib import VitAmdError
    x = torch.zeros(8, 100, dtype=BF16, device=dev)
    w = torch.zeros(128, 100, dtype=BF16, device=dev)
    with pytest.raises(VitAmdError):
        ops.gemm_nt(x, w, ops.EPI_BF16, torch.empty(8, 128, dtype=BF16, device=dev))


# ------------------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize("M,N,K,split", [(8, 1024, 1024, 0), (64, 128, 128, 1), (1000, 3072, 1024, 0),
                                         (4096 + 13, 128, 256, 1), (4096 + 13, 128, 256, 7), (8192, 1024, 4096, 0),
                                         (4096 + 13, 256, 256, 0), (4096 + 13, 256, 512, 3), (129, 256, 256, 1),
                                         (128, 256, 256, 1), (2048, 1024, 1536, 0), (2048, 1024, 1024, -4)])
def test_gemm_tn(ops, dev, M, N, K, split):
    dy = rnd((M, N), dev, 7, dtype=BF16)
    x = rnd((M, K), dev, 8, dtype=BF16)
    dw = torch.ones(N, K, dtype=F32, device=dev)
    db = torch.full((N,), 2.0, dtype=F32, device=dev)
    ops.gemm_tn_accum(dy, x, dw, dbias=db, split_k=split)
    ref = dy.float().t() @ x.float() + 1.0
    assert torch.isfinite(dw).all()
    assert rel_err(dw, ref) < 1e-4
    assert (dw - ref).abs().max().item() < 1e-3 * math.sqrt(M)
    assert (db - (dy.float().sum(0) + 2.0)).abs().max().item() < 2e-4 * math.sqrt(M) + 1e-3


def test_gemm_tn_layout(ops, dev):
    """dy = one-hot rows: dw[n, :] must equal the x row selected for n (exact in f32)."""
    M, N, K = 128, 128, 256
    dy = torch.zeros(M, N, dtype=BF16, device=dev)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(0)).to(dev)
    dy[torch.arange(M, device=dev), perm] = 1
    x = rnd((M, K), dev, 9, dtype=BF16)
    dw = torch.zeros(N, K, dtype=F32, device=dev)
    ops.gemm_tn_accum(dy, x, dw, split_k=1)
    ref = torch.zeros(N, K, device=dev)
    ref[perm] = x.float()
    assert torch.equal(dw, ref)


def test_gemm_tn_ring_layout(ops, dev):
    """Same one-hot check on the 256x256-tile kernel (M = 4 stages of 32 rows)."""
    M, N, K = 128, 256, 512
    dy = torch.zeros(M, N, dtype=BF16, device=dev)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))[:M].to(dev)
    dy[torch.arange(M, device=dev), perm] = 1
    x = rnd((M, K), dev, 9, dtype=BF16)
    dw = torch.zeros(N, K, dtype=F32, device=dev)
    db = torch.zeros(N, dtype=F32, device=dev)
    ops.gemm_tn_accum(dy, x, dw, dbias=db, split_k=1)
    ref = torch.zeros(N, K, device=dev)
    ref[perm] = x.float()
    assert torch.equal(dw, ref)
    assert torch.equal(db, dy.float().sum(0))


def test_colsum(ops, dev):
    y = rnd((3000, 4096), dev, 10, dtype=BF16)
    out = torch.zeros(4096, dtype=F32, device=dev)
    ops.colsum_accum(y, out)
    assert (out - y.float().sum(0)).abs().max().item() < 1e-2


# ------------------------------------------------------------------------------ attention
def attn_reference(qkv, B, N, H):
    D = qkv.shape[1] // 3
    hd = D // H
    q, k, v = (t.reshape(B, N, H, hd).permute(0, 2, 1, 3) for t in qkv.float().split(D, dim=1))
    s = (q @ k.transpose(-1, -2)) * hd ** -0.5
    p = s.softmax(-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * N, D)
    lse2 = torch.logsumexp(s, dim=-1) * math.log2(math.e)
    return o, lse2.reshape(B * H, N)


@pytest.mark.parametrize("B,grid", [(2, (8, 16, 16)), (1, (4, 24, 22)), (3, (8, 16, 16)), (2, (5, 20, 30))])
def test_qkv_projection_with_the_rotation_in_its_epilogue(ops, dev, B, grid):
    """va_gemm_nt_qkv_rope against the two-pass route (va_gemm_nt, then va_rope3d_qk_inplace) and against the f32
    formula: the fused kernel rotates the f32 accumulators, the two-pass route the rounded product, so they differ by
    bf16 roundings; the value third is untouched by the rotation and must be bit-identical."""
    Tp, Hp, Wp = grid
    N = Tp * Hp * Wp
    M, K = B * N, 1024
    x = R.hashed_uniform("t.qr.x", (M, K), 1.0).to(dev).to(BF16)
    w = R.hashed_uniform("t.qr.w", (3072, K), 0.03).to(dev).to(BF16)
    inv = {k: R.hashed_uniform("t.qr." + k, (n,), 0.5, 0.6) for k, n in (("t", 128), ("h", 192), ("w", 192))}
    tables = ops.rope_tables_device(inv["t"], inv["h"], inv["w"], grid, 2.0, dev)
    pos = ops.rope_pos_code(B, grid, dev)
    fused = torch.full((M, 3072), float("nan"), dtype=BF16, device=dev)
    assert ops.gemm_nt_qkv_rope(x, w, fused, tables, pos, grid)
    two = torch.empty(M, 3072, dtype=BF16, device=dev)
    ops.gemm_nt(x, w, ops.EPI_BF16, two)
    plain = two.clone()
    ops.rope3d_qk_(two, tables, grid)
    assert torch.isfinite(fused.float()).all()
    assert torch.equal(fused[:, 2048:], two[:, 2048:])
    # f32 formula: rotate the unrounded product with the same tables
    y = x.float() @ w.float().t()
    Tsz = Tp * 128 + (Hp + Wp) * 192
    cos, sin = tables[:Tsz], tables[Tsz:]
    n = torch.arange(M, device=dev) % N
    tt, hh, ww = n // (Hp * Wp), (n // Wp) % Hp, n % Wp
    ref = y.clone()
    for part in (0, 1024):
        for a0, half, cnt, tab in ((0, 128, 128, tt * 128), (256, 192, 192, Tp * 128 + hh * 192), (640, 192, 192, Tp * 128 + Hp * 192 + ww * 192)):
            idx = tab[:, None] + torch.arange(cnt, device=dev)[None, :]
            c, sn = cos[idx], sin[idx]
            a, b = y[:, part + a0:part + a0 + cnt], y[:, part + a0 + half:part + a0 + half + cnt]
            ref[:, part + a0:part + a0 + cnt] = a * c - b * sn
            ref[:, part + a0 + half:part + a0 + half + cnt] = b * c + a * sn
    assert rel_err(fused, ref) < 3e-3, rel_err(fused, ref)
    assert rel_err(two, ref) < 5e-3
    assert rel_err(fused, ref) <= rel_err(two, ref)                 # one rounding instead of two
    assert rel_err(fused[:, :2048], two[:, :2048]) < 6e-3
    assert not torch.equal(plain[:, :2048], two[:, :2048])          # (the rotation did something)


@pytest.mark.parametrize("B,N,H", [(2, 8, 16), (1, 200, 16), (2, 1568, 16), (1, 2048, 16), (2, 18, 8), (1, 300, 8),
                                   (1, 1024, 8), (1, 8192, 16), (1, 8200, 8),   # the last two: 64-frame clips
                                   (2, 8, 32), (1, 300, 32), (1, 2048, 32),      # 32 heads: head dim 32
                                   (3, 1, 16), (2, 33, 8), (1, 65, 16), (2, 127, 32), (1, 129, 16),   # around the tile edges
                                   (3, 64, 16), (2, 192, 16), (2, 320, 16), (2, 128, 8), (2, 64, 32)])  # whole 64-row tiles, but
                                   # fewer keys than a backward workgroup covers: several clips, so a store past a clip's
                                   # last key would land in the next clip's rows
def test_attention_fwd_bwd(ops, dev, B, N, H):
    D = 1024
    qkv = rnd((B * N, 3 * D), dev, 11, dtype=BF16)
    out = torch.full((B * N, D), float("nan"), dtype=BF16, device=dev)
    lse2 = torch.empty(B * H, N, dtype=F32, device=dev)
    ops.attn_fwd(qkv, B, N, H, out, lse2)
    leaf = qkv.float().requires_grad_(True)
    ref_o, ref_lse = attn_reference(leaf, B, N, H)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, ref_o) < 8e-3
    assert (lse2 - ref_lse).abs().max().item() < 2e-2
    d_out = rnd((B * N, D), dev, 12, dtype=BF16)
    ref_o.backward(d_out.float())
    dqkv = torch.full((B * N, 3 * D), float("nan"), dtype=BF16, device=dev)
    ops.attn_bwd(qkv, out, d_out, lse2, dqkv, B, N, H)
    assert torch.isfinite(dqkv.float()).all()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        if N == 1 and name != "dv":      # a single key: the softmax is the constant 1, so dQ = dK = 0 exactly
            assert dqkv[:, sl].float().abs().max().item() < 1e-5, name
        else:
            assert rel_err(dqkv[:, sl], leaf.grad[:, sl]) < 1.5e-2, name


def _masks(kind, B, N, H, dev):
    """Boolean masks (True = masked) in the shapes `masked_fill` broadcasts over (B, H, N, N); every query keeps at
    least one key."""
    gen = torch.Generator(device="cpu").manual_seed(5)
    i = torch.arange(N)
    if kind == "random":            # 2-D, 60 % masked, diagonal kept
        m = torch.rand(N, N, generator=gen) < 0.6
        m[i, i] = False
    elif kind == "future_only":     # a query sees itself and LATER keys: its first tiles are masked out entirely
        m = i[None, :] < i[:, None]
    elif kind == "causal7":         # block-causal, 7 tokens per timestep (N % 7 == 0 not needed here)
        m = (i[None, :] // 7) > (i[:, None] // 7)
    elif kind == "per_batch":       # 3-D
        m = torch.rand(B, N, N, generator=gen) < 0.5
        m[:, i, i] = False
    elif kind == "per_head":        # 4-D, shared over the batch
        m = torch.rand(1, H, N, N, generator=gen) < 0.5
        m[:, :, i, i] = False
    elif kind == "band":            # only a far-away band of keys is visible: whole leading AND trailing tiles masked
        m = ~(((i[None, :] - i[:, None]) % N) // 8 == (N // 16))
    return m.to(dev)


@pytest.mark.parametrize("B,N,H,kind", [(2, 8, 16, "random"), (1, 200, 16, "random"), (2, 300, 8, "per_batch"),
                                        (2, 256, 16, "per_head"), (1, 1024, 16, "future_only"), (1, 1000, 8, "future_only"),
                                        (2, 210, 16, "causal7"), (1, 2048, 16, "band"), (1, 2048, 8, "random"),
                                        (2, 300, 32, "per_batch"), (1, 1024, 32, "future_only")])
def test_attention_with_explicit_mask_fwd_bwd(ops, dev, B, N, H, kind):
    """`scores.masked_fill(attn_mask, -inf)` before the softmax (swm:316-323) against the fp32 formula, forward and
    backward, for masks of every broadcast shape; same tolerances as the unmasked kernels."""
    D = 1024
    hd = D // H
    qkv = rnd((B * N, 3 * D), dev, 21, dtype=BF16)
    mask = _masks(kind, B, N, H, dev)
    packed = ops.attn_mask_pack(mask, B, H)
    out = torch.full((B * N, D), float("nan"), dtype=BF16, device=dev)
    lse2 = torch.empty(B * H, N, dtype=F32, device=dev)
    ops.attn_fwd(qkv, B, N, H, out, lse2, mask=packed)
    leaf = qkv.float().requires_grad_(True)
    q, k, v = (t.reshape(B, N, H, hd).permute(0, 2, 1, 3) for t in leaf.split(D, dim=1))
    full = mask if mask.dim() == 4 else mask[:, None] if mask.dim() == 3 else mask[None, None]
    s = ((q @ k.transpose(-1, -2)) * hd ** -0.5).masked_fill(full, float("-inf"))
    ref_o = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * N, D)
    ref_lse = (torch.logsumexp(s, dim=-1) * math.log2(math.e)).reshape(B * H, N)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, ref_o) < 8e-3
    assert (lse2 - ref_lse).abs().max().item() < 2e-2
    d_out = rnd((B * N, D), dev, 22, dtype=BF16)
    ref_o.backward(d_out.float())
    dqkv = torch.full((B * N, 3 * D), float("nan"), dtype=BF16, device=dev)
    ops.attn_bwd(qkv, out, d_out, lse2, dqkv, B, N, H, mask=packed)
    assert torch.isfinite(dqkv.float()).all()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert rel_err(dqkv[:, sl], leaf.grad[:, sl]) < 1.5e-2, name


def test_attention_mask_packing_is_bit_exact(ops, dev):
    """The packed words against a plain restatement: bit j of word w of row q = mask[q, 32 w + j], padding set; the
    by-key words are the same of the transposed mask."""
    N = 200
    mask = torch.rand(2, N, N, generator=torch.Generator().manual_seed(9)) < 0.4
    packed = ops.attn_mask_pack(mask.to(dev), 2, 16)
    W = 2 * ((N + 63) // 64)
    assert packed.stride_b == N * W and packed.stride_h == 0

    def restate(m):
        padded = torch.ones(2, N, W * 32, dtype=torch.bool)
        padded[:, :, :N] = m
        weights = (1 << torch.arange(32, dtype=torch.int64))
        words = (padded.reshape(2, N, W, 32).long() * weights).sum(-1)
        return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    assert torch.equal(packed.by_query.cpu().view(2, N, W), restate(mask))
    assert torch.equal(packed.by_key.cpu().view(2, N, W), restate(mask.transpose(1, 2)))


def test_attention_fully_masked_query_is_nan_like_the_reference(ops, dev):
    """softmax over a row of -inf is NaN in the reference (swm:322-325); so is that query's output here, and the
    other queries are untouched."""
    B, N, H, D = 1, 128, 16, 1024
    qkv = rnd((N, 3 * D), dev, 23, dtype=BF16)
    mask = torch.zeros(N, N, dtype=torch.bool, device=dev)
    mask[17, :] = True
    out = torch.empty(N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(H, N, dtype=F32, device=dev)
    ops.attn_fwd(qkv, B, N, H, out, lse2, mask=ops.attn_mask_pack(mask, B, H))
    assert torch.isnan(out[17].float()).all()
    ref_o, _ = attn_reference(qkv, B, N, H)
    keep = torch.arange(N, device=dev) != 17
    assert rel_err(out[keep], ref_o[keep]) < 8e-3


def test_attention_online_softmax_rescale(ops, dev):
    """One key per tile spiked against one query forces the running-max rescale late in the sweep."""
    B, N, H, D = 1, 512, 16, 1024
    qkv = rnd((N, 3 * D), dev, 13, 0.3, dtype=BF16)
    qkv[5, :64] = 4.0                   # query 5, head 0
    qkv[300, D:D + 64] = 4.0            # key 300, head 0  (tile 4): score 64*16/8 = 128
    qkv[450, D:D + 64] = 6.0            # key 450 (tile 7): even larger
    out = torch.empty(N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(H, N, dtype=F32, device=dev)
    ops.attn_fwd(qkv, B, N, H, out, lse2)
    ref_o, _ = attn_reference(qkv, B, N, H)
    assert rel_err(out, ref_o) < 8e-3
    assert (out[5, :64].float() - ref_o[5, :64]).abs().max().item() < 3e-2


# ------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M", [8, 1001, 16384])
def test_layernorm_fwd_bwd(ops, dev, M):
    D = 1024
    x = rnd((M, D), dev, 14, 2.0) + 0.5
    gamma = 1 + 0.1 * rnd((D,), dev, 15)
    beta = 0.1 * rnd((D,), dev, 16)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    y32 = torch.empty(M, D, device=dev)
    ops.layernorm_fwd(x, gamma, beta, y32, mean, rstd)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    assert (y32 - ref).abs().max().item() < 2e-5
    yb = torch.empty(M, D, dtype=BF16, device=dev)
    ops.layernorm_fwd(x, gamma, beta, yb, mean, rstd)
    assert rel_err(yb, ref) < 4e-3
    dy = rnd((M, D), dev, 17)
    ref.backward(dy)
    resid = rnd((M, D), dev, 18)
    dx = torch.empty(M, D, device=dev); dxb = torch.empty(M, D, dtype=BF16, device=dev)
    dg = torch.empty(D, device=dev); db = torch.empty(D, device=dev)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, resid, dx, dxb, dg, db)
    assert (dx - (xr.grad + resid)).abs().max().item() < 5e-5
    assert rel_err(dxb, xr.grad + resid) < 4e-3
    assert rel_err(dg, gr.grad) < 1e-5 and rel_err(db, br.grad) < 1e-5
    # bf16 dy, no residual
    dyb = dy.to(BF16)
    xr.grad = None
    torch.nn.functional.layer_norm(xr, (D,), gamma, beta, 1e-6).backward(dyb.float())
    ops.layernorm_bwd(dyb, x, gamma, mean, rstd, None, dx, None, dg, db)
    assert (dx - xr.grad).abs().max().item() < 5e-5


# ------------------------------------------------------------------------------ RoPE / patch gather / casts
@pytest.mark.parametrize("grid,B", [((2, 2, 2), 2), ((8, 14, 14), 1), ((8, 16, 16), 2)])
def test_rope(ops, dev, grid, B):
    N = grid[0] * grid[1] * grid[2]
    qkv = rnd((B * N, 3072), dev, 19, dtype=BF16)
    tables = ops.rope_tables_device(R.rope_inv_freq(256), R.rope_inv_freq(384), R.rope_inv_freq(384), grid, 0.6, dev)
    orig = qkv.clone()
    ops.rope3d_qk_(qkv, tables, grid)
    x = orig.float().cpu().reshape(B, N, 3072)
    ref_q = R.rope_rotate(x[..., :1024], *grid)
    ref_k = R.rope_rotate(x[..., 1024:2048], *grid)
    got = qkv.float().cpu().reshape(B, N, 3072)
    assert (got[..., :1024] - ref_q).abs().max().item() < 0.04      # one bf16 rounding of |v| < 6
    assert (got[..., 1024:2048] - ref_k).abs().max().item() < 0.04
    assert rel_err(got[..., :2048], torch.cat([ref_q, ref_k], -1)) < 3e-3
    assert torch.equal(got[..., 2048:], x[..., 2048:])              # V untouched
    ops.rope3d_qk_(qkv, tables, grid, inverse=True)                   # rotation by -theta undoes it
    assert rel_err(qkv, orig) < 6e-3


def test_tubelet_gather(ops, dev):
    video = rnd((2, 3, 4, 48, 32), dev, 20)
    patches = ops.tubelet_gather(video, (2, 16, 16))
    ref = R.tubelet_patches(video.cpu()).reshape(-1, 1536).to(BF16)
    assert torch.equal(patches.cpu(), ref)


def test_casts(ops, dev):
    w = rnd((1024, 4096), dev, 21)
    assert torch.equal(ops.cast_bf16(w), w.to(BF16))
    assert torch.equal(ops.cast_transpose_bf16(w), w.t().contiguous().to(BF16))
    v = rnd((1003,), dev, 22)
    buf = torch.empty(1008, dtype=BF16, device=dev)[:1003]
    assert torch.equal(ops.cast_bf16(v, buf), v.to(BF16))


# ------------------------------------------------------------------------------ masking / JEPA pieces
@pytest.mark.parametrize("dtype", [BF16, F32])
def test_future_mask_bit_exact(ops, dev, dtype):
    for Tp, ratio in ((4, 0.5), (5, 0.5), (8, 0.75), (8, 0.9), (32, 0.5)):
        x = rnd((3, Tp * 6, 1024), dev, 23).to(dtype)
        ref_m, ref_mask = R.future_masking(x.cpu().float(), Tp, ratio)
        cutoff = int(Tp * (1 - ratio))
        out, mask = ops.future_mask(x, cutoff * 6)
        assert torch.equal(mask.cpu(), ref_mask)
        assert torch.equal(out.cpu().float(), ref_m)


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_token_gather_scatter_bit_exact(ops, dev, dtype):
    B, grid, D = 4, (8, 16, 16), 1024
    N = 2048
    ctx, tgt = R.jepa_block_masks(11, B, grid)
    x = rnd((B, N, D), dev, 24).to(dtype)
    for idx in (ctx, tgt):
        g = ops.token_gather(x, idx.to(dev))
        assert torch.equal(g.cpu(), R.token_gather(x.cpu(), idx))
        s = ops.token_scatter(g, idx.to(dev), N)
        assert torch.equal(s.cpu(), R.token_scatter(g.cpu(), idx, N))


def test_ema_and_l1(ops, dev):
    n = 3_000_003
    tgt = rnd((n,), dev, 25); onl = rnd((n,), dev, 26)
    ref = R.ema_update(tgt.clone(), onl, 0.996)
    shadow = torch.empty(n + 5, dtype=BF16, device=dev)[:n]
    ops.ema_update_(tgt, onl, 0.996, shadow)
    assert (tgt - ref).abs().max().item() < 1e-6
    assert torch.equal(shadow, tgt.to(BF16))
    for dtype in (F32, BF16):
        p = rnd((7, 333, 1024), dev, 27).to(dtype); t = rnd((7, 333, 1024), dev, 28).to(dtype)
        loss, grad = ops.l1_loss(p, t)
        rl, rg = R.l1_latent_loss(p.float().cpu(), t.float().cpu())
        assert abs(loss.item() - rl.item()) < 1e-5 * max(1.0, rl.item())
        assert torch.equal(grad.float().cpu(), rg.to(dtype).float())


# ------------------------------------------------------------------------------ input pipeline
@pytest.mark.parametrize("B,T,H,W", [(1, 1, 2, 2), (2, 4, 32, 32), (1, 3, 18, 22)])
def test_normalize_u8_bit_exact(ops, dev, B, T, H, W):
    from oracle import ref_cpu as R
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (B, T, H, W, 3), generator=g, dtype=torch.uint8)
    frames[0, 0, 0, 0] = torch.tensor([0, 255, 128], dtype=torch.uint8)
    out = ops.normalize_u8(frames.to(dev))
    assert torch.equal(out.cpu(), R.normalize_u8(frames))


def test_multi_cast_plan(ops, dev):
    """Several matrices, one launch: both copies bit-identical to torch's rounding; null destinations skipped."""
    shapes = [(64, 64), (1024, 1536), (192, 4096), (3072, 1024)]
    srcs = [rnd(sh, dev, 70 + i) for i, sh in enumerate(shapes)]
    dst = [torch.zeros(sh, dtype=BF16, device=dev) for sh in shapes]
    dst_t = [torch.zeros(sh[::-1], dtype=BF16, device=dev) for sh in shapes]
    dst[2], dst_t[1] = None, None
    plan = ops.CastPlan(list(zip(srcs, dst, dst_t)))
    plan.run()
    for s_, d, t in zip(srcs, dst, dst_t):
        if d is not None:
            assert torch.equal(d, s_.to(BF16))
        if t is not None:
            assert torch.equal(t, s_.t().contiguous().to(BF16))
    srcs[0].mul_(2.0)
    plan.run()                                   # same plan, new values
    assert torch.equal(dst[0], srcs[0].to(BF16)) and torch.equal(dst_t[0], srcs[0].t().contiguous().to(BF16))
    with pytest.raises(ValueError):
        ops.CastPlan([(rnd((100, 64), dev, 1), None, None)])


@pytest.mark.parametrize("B,N,D,n_first", [(3, 37, 1024, None), (2, 64, 128, 32), (4, 2048, 1024, 1024), (2, 8, 1024, 0), (1, 5, 4, 5)])
def test_token_pool_forward_backward(B, N, D, n_first):
    """The pooling of the stage-1 heads: `.mean(dim=1)` (s1:426-432) and the mean over the tokens future_masking leaves
    unmasked (s1:452-460: zero-fill, multiply by the keep mask, divide by clamp_min(count, 1)) against the oracle's
    future_masking; the backward against autograd.  Tolerance: f32 summation order (1e-5)."""
    from vit_amd import ops
    x = R.hashed_uniform(f"pool.x.{B}.{N}.{D}", (B, N, D), 1.0)
    got = ops.token_pool(x.cuda(), n_first)
    xr = x.clone().requires_grad_(True)
    if n_first is None:
        ref = xr.mean(dim=1)
    else:
        keep = torch.zeros(B, N, dtype=torch.bool)
        keep[:, :n_first] = True
        masked = xr * keep.unsqueeze(-1)                        # the zero-filled copy of future_masking
        ref = (masked * keep.unsqueeze(-1)).sum(dim=1) / keep.sum(dim=1, keepdim=True).clamp_min(1)
    assert torch.allclose(got.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(got, ops.token_pool(x.cuda(), n_first))          # fixed summation order
    g = R.hashed_uniform("pool.g", (B, D), 1.0)
    (ref * g).sum().backward()
    dx = ops.token_pool_bwd(g.cuda(), N, n_first)
    assert torch.allclose(dx.cpu(), xr.grad, rtol=1e-6, atol=1e-7)


def test_masked_pool_matches_future_masking_of_the_oracle():
    from vit_amd import ops
    from vit_amd.masking import StructuredTemporalMasking as STM
    for Tp, per_t, ratio in ((8, 6, 0.5), (5, 4, 0.5), (8, 3, 0.9), (4, 8, 0.75)):
        x = R.hashed_uniform("pool.fm", (2, Tp * per_t, 64), 1.0)
        masked, mask = R.future_masking(x, Tp, ratio)
        keep = ~mask
        ref = (masked * keep.unsqueeze(-1)).sum(dim=1) / keep.sum(dim=1, keepdim=True).clamp_min(1)
        n_first = STM.first_masked_row(Tp * per_t, Tp, ratio)
        assert n_first == int(keep[0].sum())
        assert torch.allclose(ops.token_pool(x.cuda(), n_first).cpu(), ref, rtol=1e-5, atol=1e-6)


def test_c_abi_from_a_plain_host_program():
    """examples/c_abi_demo (built by __graft_entry__.build()): a C++ host program with hipMalloc'd buffers calls
    va_layernorm_fwd and va_gemm_nt through include/vit_amd.h -- no Python, no torch in that process -- checks them
    against a double-precision host computation and checks that an invalid call returns an error code."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "c_abi_demo")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    env = {k: v for k, v in os.environ.items() if k != "LD_LIBRARY_PATH"}     # its own HIP runtime, not torch's
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "rejected as expected" in r.stdout
