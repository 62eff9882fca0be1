"""GPU: predictor kernels (delta-rule recurrence, causal conv, block-causal attention) and the whole-model
inference forward, against the CPU oracle and the reference's golden outputs (tests/golden/predictor.npz)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
REL_TOL = 2e-2          # bf16 attention / bf16 encoder against the reference's fp32


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    from vit_amd import ops
    from vit_amd import surgical_world_model as swm
    return ops, swm


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "predictor.npz"), allow_pickle=False)


def gd_inputs(B=2, L=21, H=2, D=128):
    F = torch.nn.functional
    q = F.normalize(R.hashed_uniform("golden.gd.q", (B, L, H * D), 1.0), dim=-1).view(B, L, H, D)
    k = F.normalize(R.hashed_uniform("golden.gd.k", (B, L, H * D), 1.0), dim=-1).view(B, L, H, D)
    v = R.hashed_uniform("golden.gd.v", (B, L, H, D), 1.0)
    alpha = R.hashed_uniform("golden.gd.alpha", (B, L, H), 0.45, 0.5)
    beta = R.hashed_uniform("golden.gd.beta", (B, L, H), 0.45, 0.5)
    s0 = R.hashed_uniform("golden.gd.state", (B, H, D, D), 0.1)
    return q, k, v, alpha, beta, s0


def test_gated_delta_matches_reference_vectors(env, gold):
    ops, _ = env
    q, k, v, a, b, s0 = gd_inputs()
    out, s1 = ops.gated_delta_seq(q.cuda(), k.cuda(), v.cuda(), a.cuda(), b.cuda(), s0.cuda())
    assert (out.cpu() - torch.from_numpy(gold["gd_out"])).abs().max().item() < 2e-5
    assert (s1.cpu()[:, :, ::8, ::8] - torch.from_numpy(gold["gd_state_sample"])).abs().max().item() < 2e-5
    out0, _ = ops.gated_delta_seq(q.cuda(), k.cuda(), v.cuda(), a.cuda(), b.cuda(), None)
    assert (out0.cpu() - torch.from_numpy(gold["gd_out_zero_state"])).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,L,H,D", [(1, 1, 1, 128), (3, 16, 8, 128), (2, 33, 3, 64), (1, 200, 8, 128)])
def test_gated_delta_vs_oracle(env, B, L, H, D):
    ops, _ = env
    F = torch.nn.functional
    q = F.normalize(R.hashed_uniform("t.gd.q", (B, L, H, D), 1.0), dim=-1)
    k = F.normalize(R.hashed_uniform("t.gd.k", (B, L, H, D), 1.0), dim=-1)
    v = R.hashed_uniform("t.gd.v", (B, L, H, D), 1.0)
    a = R.hashed_uniform("t.gd.a", (B, L, H), 0.5, 0.5)
    b = R.hashed_uniform("t.gd.b", (B, L, H), 0.5, 0.5)
    ref_o, ref_s = R.gated_delta_sequential(q, k, v, a, b, None)
    out, st = ops.gated_delta_seq(q.cuda(), k.cuda(), v.cuda(), a.cuda(), b.cuda())
    assert (out.cpu() - ref_o).abs().max().item() < 5e-5
    assert (st.cpu() - ref_s).abs().max().item() < 5e-5
    # a sequence processed in two calls, state carried over, equals one call (the rollout's usage)
    cut = max(1, L // 3)
    o1, s1 = ops.gated_delta_seq(q[:, :cut].contiguous().cuda(), k[:, :cut].contiguous().cuda(), v[:, :cut].contiguous().cuda(),
                                 a[:, :cut].contiguous().cuda(), b[:, :cut].contiguous().cuda())
    if cut < L:
        o2, s2 = ops.gated_delta_seq(q[:, cut:].contiguous().cuda(), k[:, cut:].contiguous().cuda(), v[:, cut:].contiguous().cuda(),
                                     a[:, cut:].contiguous().cuda(), b[:, cut:].contiguous().cuda(), s1)
        assert torch.allclose(torch.cat([o1, o2], dim=1), out, atol=1e-5) and torch.allclose(s2, st, atol=1e-5)


@pytest.mark.parametrize("D", [64, 128])
def test_gated_delta_small_and_zero_decays(env, D):
    """Chunks of 16 steps whose decays multiply to less than 1e-20 (or hold a zero) take the kernel's explicit
    update, the others the form with the decay folded into a running product; both against the sequential oracle,
    with decays that are tiny, exactly zero, exactly one, and ordinary, chunk by chunk."""
    ops, _ = env
    F = torch.nn.functional
    B, L, H = 2, 83, 3
    q = F.normalize(R.hashed_uniform("t.gd2.q", (B, L, H, D), 1.0), dim=-1)
    k = F.normalize(R.hashed_uniform("t.gd2.k", (B, L, H, D), 1.0), dim=-1)
    v = R.hashed_uniform("t.gd2.v", (B, L, H, D), 1.0)
    a = R.hashed_uniform("t.gd2.a", (B, L, H), 0.5, 0.5)
    b = R.hashed_uniform("t.gd2.b", (B, L, H), 0.5, 0.5)
    a[:, 16:32] = 0.04                       # product 4e-23: explicit
    a[:, 32:48] = 0.06                       # product 3e-20: folded, close to the threshold
    a[0, 50] = 0.0                           # a zero inside an ordinary chunk
    a[1, 64:80] = 1.0                        # no decay at all
    a[:, 80, 1] = 1e-30
    s0 = R.hashed_uniform("t.gd2.s", (B, H, D, D), 0.3)
    ref_o, ref_s = R.gated_delta_sequential(q, k, v, a, b, s0)
    out, st = ops.gated_delta_seq(q.cuda(), k.cuda(), v.cuda(), a.cuda(), b.cuda(), s0.cuda())
    assert torch.isfinite(out).all() and torch.isfinite(st).all()
    assert (out.cpu() - ref_o).abs().max().item() < 5e-5
    assert (st.cpu() - ref_s).abs().max().item() < 5e-5


def test_gated_delta_long_sequence_stays_close(env):
    """2064 steps (the rollout's length at 8 frames of context): the folded form must not drift from the oracle."""
    ops, _ = env
    F = torch.nn.functional
    B, L, H, D = 1, 2064, 2, 128
    q = F.normalize(R.hashed_uniform("t.gd3.q", (B, L, H, D), 1.0), dim=-1)
    k = F.normalize(R.hashed_uniform("t.gd3.k", (B, L, H, D), 1.0), dim=-1)
    v = R.hashed_uniform("t.gd3.v", (B, L, H, D), 1.0)
    a = R.hashed_uniform("t.gd3.a", (B, L, H), 0.45, 0.5)
    b = R.hashed_uniform("t.gd3.b", (B, L, H), 0.5, 0.5)
    ref_o, ref_s = R.gated_delta_sequential(q, k, v, a, b, None)
    out, st = ops.gated_delta_seq(q.cuda(), k.cuda(), v.cuda(), a.cuda(), b.cuda())
    assert (out.cpu() - ref_o).abs().max().item() < 5e-5
    assert (st.cpu() - ref_s).abs().max().item() < 5e-5


@pytest.mark.parametrize("M", [1, 5, 1030])
def test_gate_layernorm_vs_torch(env, M):
    """LayerNorm((o * gate) * silu(g)) in one pass (swm:511-519) against the torch f32 formulation; bf16 output."""
    ops, _ = env
    F = torch.nn.functional
    o = R.hashed_uniform("t.gl.o", (M, 1024), 2.0).cuda()
    gate = R.hashed_uniform("t.gl.gate", (M, 1024), 1.5).cuda()
    g = R.hashed_uniform("t.gl.g", (M, 1024), 4.0).cuda()
    gamma = R.hashed_uniform("t.gl.gamma", (1024,), 0.5, 1.0).cuda()
    beta = R.hashed_uniform("t.gl.beta", (1024,), 0.3).cuda()
    out = torch.empty(M, 1024, dtype=torch.bfloat16, device="cuda")
    ops.gate_layernorm_fwd(o, gate, g, gamma, beta, out, 1e-5)
    ref = F.layer_norm((o * gate) * F.silu(g), (1024,), gamma, beta, 1e-5)
    assert (out.float() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6      # bf16 rounding
    assert (out.float() - ref.bfloat16().float()).abs().gt(0).float().mean().item() < 0.01             # and mostly the same bits
    with pytest.raises(Exception):
        ops.gate_layernorm_fwd(o[:, :512].contiguous(), gate[:, :512].contiguous(), g[:, :512].contiguous(), gamma[:512],
                               beta[:512], out[:, :512].contiguous(), 1e-5)


@pytest.mark.parametrize("B,L,C,norm", [(2, 1, 1024, True), (2, 7, 1024, True), (3, 50, 2048, False), (1, 5, 256, True)])
def test_dwconv_silu_vs_oracle(env, B, L, C, norm):
    ops, _ = env
    x = R.hashed_uniform("t.dw.x", (B, L, C), 2.0)
    w = R.hashed_uniform("t.dw.w", (C, 1, 4), 0.5)
    bias = R.hashed_uniform("t.dw.b", (C,), 0.1)
    ref = R.dwconv1d_causal_silu(x, w, bias, norm)
    out = ops.dwconv1d_causal_silu(x.cuda(), w.cuda(), bias.cuda(), norm)
    assert (out.cpu() - ref).abs().max().item() < 2e-6 * (1 if norm else 10)


@pytest.mark.parametrize("B,L,norm", [(2, 1, True), (2, 7, True), (3, 50, False), (1, 131, True)])
def test_dwconv_silu_on_a_column_block_of_a_bf16_matrix(env, B, L, norm):
    """The layer's conv inputs are 1024-wide column blocks of one bf16 projection output [B*L, 5120] (swm:484-500):
    the kernel reads block `blk` in place; against the oracle on the same (bf16-exact) numbers, and the rest of the
    matrix is poison that must not leak in."""
    ops, _ = env
    C, blk = 1024, 2
    x = R.hashed_uniform("t.dw16.x", (B, L, C), 2.0).bfloat16()
    w = R.hashed_uniform("t.dw16.w", (C, 1, 4), 0.5)
    bias = R.hashed_uniform("t.dw16.b", (C,), 0.1)
    ref = R.dwconv1d_causal_silu(x.float(), w, bias, norm)
    wide = torch.full((B * L, 5 * C), float("nan"), dtype=torch.bfloat16, device="cuda")
    wide[:, blk * C:(blk + 1) * C] = x.view(B * L, C).cuda()
    out = ops.dwconv1d_causal_silu_bf16(wide[:, blk * C:(blk + 1) * C], B, L, w.cuda(), bias.cuda(), norm)
    assert out.shape == (B, L, C) and bool(torch.isfinite(out).all())
    assert (out.cpu() - ref).abs().max().item() < 2e-6 * (1 if norm else 10)
    # the f32 kernel on the same numbers: the same taps in the same order
    same = ops.dwconv1d_causal_silu(x.float().cuda(), w.cuda(), bias.cuda(), False)
    if not norm:
        assert torch.equal(out, same)
    with pytest.raises(Exception):
        ops.dwconv1d_causal_silu_bf16(wide[:, 4:4 + C], B, L, w.cuda(), bias.cuda(), norm)      # 8-byte aligned block only


@pytest.mark.parametrize("M,H", [(1, 8), (7, 8), (1030, 8), (513, 4), (66, 3)])
def test_delta_gates_vs_torch(env, M, H):
    """alpha = sigmoid(x a_proj^T), beta = sigmoid(x b_proj^T) (swm:494-495) against the f64 product, and a row's
    result does not depend on which rows surround it."""
    ops, _ = env
    x = R.hashed_uniform("t.ab.x", (M, 1024), 2.0)
    wa = R.hashed_uniform("t.ab.wa", (H, 1024), 0.06)
    wb = R.hashed_uniform("t.ab.wb", (H, 1024), 0.06)
    xb = torch.empty(M, 1024, dtype=torch.bfloat16, device="cuda")
    alpha, beta = ops.delta_gates_fwd(x.cuda(), wa.cuda(), wb.cuda(), xb)
    assert alpha.shape == (M, H) and beta.shape == (M, H)
    assert torch.equal(xb.cpu(), x.bfloat16())                    # the cast of x that the same pass writes
    ref_a = torch.sigmoid(x.double() @ wa.double().t()).float()
    ref_b = torch.sigmoid(x.double() @ wb.double().t()).float()
    assert (alpha.cpu() - ref_a).abs().max().item() < 2e-6 and (beta.cpu() - ref_b).abs().max().item() < 2e-6
    if M > 4:
        perm = torch.randperm(M, generator=torch.Generator().manual_seed(M))
        a2, b2 = ops.delta_gates_fwd(x[perm].contiguous().cuda(), wa.cuda(), wb.cuda())
        assert torch.equal(a2.cpu(), alpha.cpu()[perm]) and torch.equal(b2.cpu(), beta.cpu()[perm])
    with pytest.raises(Exception):
        ops.delta_gates_fwd(x[:, :512].contiguous().cuda(), wa[:, :512].contiguous().cuda(), wb[:, :512].contiguous().cuda())


@pytest.mark.parametrize("M", [1, 5, 1030])
def test_gate_layernorm_with_g_as_a_bf16_column_block(env, M):
    """g read in place from the bf16 projection output (row stride 5120): identical to the f32 entry point on the
    same numbers."""
    ops, _ = env
    o = R.hashed_uniform("t.gl16.o", (M, 1024), 2.0).cuda()
    gate = R.hashed_uniform("t.gl16.gate", (M, 1024), 1.5).cuda()
    g = R.hashed_uniform("t.gl16.g", (M, 1024), 4.0).bfloat16()
    gamma = R.hashed_uniform("t.gl16.gamma", (1024,), 0.5, 1.0).cuda()
    beta = R.hashed_uniform("t.gl16.beta", (1024,), 0.3).cuda()
    wide = torch.full((M, 5120), float("nan"), dtype=torch.bfloat16, device="cuda")
    wide[:, 4096:] = g.cuda()
    out = torch.empty(M, 1024, dtype=torch.bfloat16, device="cuda")
    ops.gate_layernorm_fwd(o, gate, wide[:, 4096:], gamma, beta, out, 1e-5)
    want = torch.empty_like(out)
    ops.gate_layernorm_fwd(o, gate, g.float().cuda(), gamma, beta, want, 1e-5)
    assert torch.equal(out, want)


@pytest.mark.parametrize("B,T,tpt,H", [(2, 4, 6, 8), (1, 9, 18, 8), (2, 5, 66, 16), (1, 16, 129, 8)])
def test_block_causal_attention_vs_oracle(env, B, T, tpt, H):
    ops, _ = env
    D, L = 1024, T * tpt
    hd = D // H
    qkv = (R.hashed_uniform("t.bc.qkv", (B * L, 3 * D), 1.5)).to(torch.bfloat16)
    out = torch.empty(B * L, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * H, L, device="cuda")
    ops.attn_fwd(qkv.cuda(), B, L, H, out, lse, tokens_per_timestep=tpt)
    q, k, v = (t.float().view(B, L, H, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(R.block_causal_mask(L, tpt), float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, D)
    assert rel(out, ref) < 8e-3
    ref_lse = torch.logsumexp(s, -1) / math.log(2.0)
    assert (lse.cpu().view(B, H, L) - ref_lse).abs().max().item() < 2e-2


def test_block_causal_rejects_partial_timestep(env):
    ops, _ = env
    from vit_amd._lib import VitAmdError
    qkv = torch.zeros(10, 3 * 1024, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(VitAmdError):
        ops.attn_fwd(qkv, 1, 10, 8, torch.empty(10, 1024, dtype=torch.bfloat16, device="cuda"),
                     torch.empty(8, 10, device="cuda"), tokens_per_timestep=4)


def test_predictor_block_matches_reference_vectors(env, gold):
    _, swm = env
    dim, heads, tpt, T = 1024, 8, int(gold["block_tpt"]), 4
    L = tpt * T
    x = R.hashed_uniform("golden.pb.x", (2, L, dim), 1.0).cuda()
    mask = swm.BlockCausalMask.create_mask(L, tpt, x.device)
    for use_dn in (True, False):
        blk = swm.HybridPredictorBlock(dim, heads, use_deltanet=use_dn).eval()
        P = R.fill_state_dict_({"pb." + k: v for k, v in blk.state_dict().items()})
        blk.load_state_dict({k[3:]: v for k, v in P.items()})
        blk = blk.cuda()
        with torch.no_grad():
            y, st = blk(x, attn_mask=mask)
        assert rel(y, torch.from_numpy(gold[f"block_out_dn{int(use_dn)}"])) < REL_TOL
        if use_dn:
            assert rel(st[:, :, ::8, ::8], torch.from_numpy(gold["block_state_sample"])) < REL_TOL
        # the autograd formulation of the same block (torch ops) agrees with the kernel path
        xg = x.clone().requires_grad_(True)
        y2, _ = blk(xg, attn_mask=mask)
        assert rel(y2.detach(), y) < REL_TOL


@pytest.mark.parametrize("heads", [16, 8])
def test_predictor_block_kernel_path_equals_torch_formulation(env, heads):
    """A block with 16 heads (head dim 64 in the attention and in the recurrence) and with 8 (head dim 128), on a
    sequence of 5 steps x 33 tokens that is not a multiple of any tile: the inference path on the kernels against
    the module's own torch formulation (the one autograd uses), outputs and recurrence state, also with a state
    carried in from a previous call."""
    _, swm = env
    dim, tpt, T = 1024, 33, 5
    L = tpt * T
    x = R.hashed_uniform("t.pb16.x", (2, L, dim), 1.0).cuda()
    mask = swm.BlockCausalMask.create_mask(L, tpt, x.device)
    blk = swm.HybridPredictorBlock(dim, heads, use_deltanet=True).eval()
    P = R.fill_state_dict_({"pb." + k: v for k, v in blk.state_dict().items()})
    blk.load_state_dict({k[3:]: v for k, v in P.items()})
    blk = blk.cuda()
    with torch.no_grad():
        y, st = blk(x, attn_mask=mask)
        y_next, st_next = blk(y, attn_mask=mask, state=st)
    y_t, st_t = blk(x.clone().requires_grad_(True), attn_mask=mask)
    assert rel(y, y_t.detach()) < REL_TOL and rel(st, st_t.detach()) < REL_TOL
    y_t2, st_t2 = blk(y.clone().requires_grad_(True), attn_mask=mask, state=st)
    assert rel(y_next, y_t2.detach()) < REL_TOL and rel(st_next, st_t2.detach()) < REL_TOL


def test_pooling_and_heads_kernel_path_equals_torch_formulation(env):
    """SpatialAttentionPooling (swm:1035-1080) and the per-horizon heads (swm:1083-1150) without an autograd graph run
    their Linear layers on the GEMM kernel (bf16 operands); with one, torch's f32 layers.  Same numbers to the bf16
    tolerance."""
    _, swm = env
    torch.manual_seed(11)
    ar = swm.AutoregressivePredictor(1024, 1024, num_patch_tokens=16).eval().cuda()
    x = R.hashed_uniform("t.pool.x", (3, 2 * 18, 1024), 1.0).cuda()
    with torch.no_grad():
        fast = ar(x, 18, predict_horizon=1)[0]
        pooled = ar.spatial_pooling(x[:, -18:][:, :16])
    slow = ar(x, 18, predict_horizon=1)[0]                       # grad mode on, trainable parameters: torch layers
    pooled_t = ar.spatial_pooling(x[:, -18:][:, :16])
    assert rel(pooled, pooled_t.detach()) < REL_TOL
    for k in fast:
        assert fast[k].shape == slow[k].shape and rel(fast[k], slow[k].detach()) < REL_TOL, k


def test_world_model_forward_matches_reference_vectors(env, gold):
    _, swm = env
    torch.manual_seed(0)
    model = swm.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                                    predictor_depth=3, predictor_heads=8).eval()
    model.load_state_dict(R.fill_state_dict_(model.state_dict()))
    model = model.cuda()
    video = R.hashed_uniform("golden.wm.video", (2, 3, 4, 32, 32), 1.0).cuda()
    actions = R.hashed_uniform("golden.wm.actions", (2, 4, 7), 1.0).cuda()
    states = R.hashed_uniform("golden.wm.states", (2, 4, 7), 1.0).cuda()
    with torch.no_grad():
        preds, cur, fut = model(video, actions, states, predict_horizon=2)
    assert len(preds) == int(gold["wm_horizon"])
    assert sorted(preds[0].keys()) == [str(k) for k in gold["wm_keys"]]
    assert rel(cur, torch.from_numpy(gold["wm_current_logits"])) < REL_TOL
    assert rel(fut, torch.from_numpy(gold["wm_future_logits"])) < REL_TOL
    for i, p in enumerate(preds):
        for k, v in p.items():
            assert rel(v, torch.from_numpy(gold[f"wm_pred{i}_{k}"])) < 2 * REL_TOL, (i, k)
