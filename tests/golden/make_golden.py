"""Generate the committed golden vectors by running the REFERENCE modules.

Run only where /root/reference exists (never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's `surgical_world_model` (torch + numpy only) and, with inert
placeholder modules for the absent `cv2`/`torchvision` (only the dataset classes touch them),
`stage1_surgvu_pretraining` for `StructuredTemporalMasking.future_masking`.  Every reference
output is first compared with the CPU oracle (`oracle/ref_cpu.py`); the script aborts if they
disagree, so a committed fixture also certifies the oracle.  Outputs: small `.npz` files in this
directory holding inputs, expected outputs and gradient checksums.  Model weights are NOT stored
(the encoder only exists at width 1024): they are re-created from the closed-form fill
`oracle.ref_cpu.fill_state_dict_`, keyed by state-dict name.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import ref_cpu as R  # noqa: E402

import surgical_world_model as swm  # noqa: E402  (the reference)

torch.set_num_threads(8)
TOL = dict(rtol=2e-5, atol=2e-5)


def close(a, b, what, rtol=2e-5, atol=2e-5):
    if not torch.allclose(a, b, rtol=rtol, atol=atol):
        raise SystemExit(f"oracle != reference for {what}: max abs diff {(a - b).abs().max().item():.3e}")
    print(f"  ok {what}: max abs diff {(a - b).abs().max().item():.3e}")


def grad_digest(g: torch.Tensor, name: str):
    """sum, abs-sum and 64 sampled elements (sample positions from the name hash)."""
    flat = g.reshape(-1).double()
    n = flat.numel()
    rng = np.random.default_rng(R._name_seed(name) % (1 << 32))
    pos = np.sort(rng.integers(0, n, size=min(64, n)))
    return np.array([flat.sum().item(), flat.abs().sum().item()]), pos.astype(np.int64), flat[pos].numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {name} ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_rope():
    out = {}
    for grid in [(2, 2, 2), (8, 14, 14), (8, 16, 16), (32, 16, 16)]:
        rope = swm.VideoRoPE3D(1024, 256, 384, 384, temporal_scale=0.6)
        ref = rope(*grid)
        mine = R.rope_tables(*grid)
        tag = "x".join(map(str, grid))
        for (rc, rs), (mc, ms), ax in zip(ref, mine, "thw"):
            assert torch.equal(rc, mc) and torch.equal(rs, ms), f"rope table {tag} {ax}"
            # full tables are tiny for (2,2,2); for others keep digests
            if grid == (2, 2, 2):
                out[f"cos_{ax}_{tag}"] = rc
                out[f"sin_{ax}_{tag}"] = rs
            out[f"digest_{ax}_{tag}"] = np.array([rc.double().sum().item(), rs.double().sum().item(),
                                                  rc.double().abs().sum().item(), rs.double().abs().sum().item()])
        enc = swm.VideoViTEncoder.__new__(swm.VideoViTEncoder)
        pos = swm.VideoViTEncoder._create_token_positions(enc, *grid, device="cpu")
        t, h, w = R.token_coords(*grid)
        assert torch.equal(pos["t_idx"], t) and torch.equal(pos["h_idx"], h) and torch.equal(pos["w_idx"], w)
        out[f"pos_digest_{tag}"] = np.array([t.sum().item(), h.sum().item(), w.sum().item(),
                                             (t * torch.arange(len(t))).sum().item(),
                                             (h * torch.arange(len(t))).sum().item(),
                                             (w * torch.arange(len(t))).sum().item()], dtype=np.int64)
    print("  ok rope tables + token positions: bit-exact")
    save("rope.npz", **out)


def gen_tubelet():
    te = swm.TubeletEmbedding(img_size=32, num_frames=4, tubelet_size=(2, 16, 16), in_channels=3, embed_dim=1024)
    sd = R.fill_state_dict_({"tubelet_embed." + k: v for k, v in te.state_dict().items()})
    te.load_state_dict({k[len("tubelet_embed."):]: v for k, v in sd.items()})
    video = R.hashed_uniform("golden.tubelet.video", (1, 3, 4, 32, 32), 1.0)
    ref = te(video)
    mine = R.tubelet_embed(video, sd["tubelet_embed.projection.weight"], sd["tubelet_embed.projection.bias"])
    close(ref, mine, "tubelet embed")
    save("tubelet.npz", video=video, out=ref)


def gen_attention():
    out = {}
    grid = (2, 2, 2)
    N = 8
    for heads in (16, 8):
        for masked in (False, True):
            att = swm.RoPEMultiheadAttention(1024, heads, rope_dims={"t_dim": 256, "h_dim": 384, "w_dim": 384}).eval()
            sd = R.fill_state_dict_({"blocks.0.attn." + k: v for k, v in att.state_dict().items()})
            att.load_state_dict({k[len("blocks.0.attn."):]: v for k, v in sd.items()})
            x = R.hashed_uniform(f"golden.attn.x.{heads}", (2, N, 1024), 1.0).requires_grad_(True)
            rope = swm.VideoRoPE3D(1024)(*grid)
            enc = swm.VideoViTEncoder.__new__(swm.VideoViTEncoder)
            pos = swm.VideoViTEncoder._create_token_positions(enc, *grid, device="cpu")
            mask = None
            if masked:
                mask = torch.triu(torch.ones(N, N, dtype=torch.bool), diagonal=1)
            y = att(x, rope_cache=rope, token_positions=pos, attn_mask=mask)
            gy = R.hashed_uniform("golden.attn.gy", y.shape, 1.0)
            (y * gy).sum().backward()
            x2 = x.detach().clone().requires_grad_(True)
            wq = sd["blocks.0.attn.qkv_proj.weight"].clone().requires_grad_(True)
            wo = sd["blocks.0.attn.out_proj.weight"].clone().requires_grad_(True)
            y2 = R.rope_attention(x2, wq, wo, heads, grid, mask)
            (y2 * gy).sum().backward()
            tag = f"h{heads}_{'mask' if masked else 'nomask'}"
            close(y, y2, f"attention out {tag}")
            close(x.grad, x2.grad, f"attention dx {tag}", atol=1e-4)
            close(att.qkv_proj.weight.grad, wq.grad, f"attention dWqkv {tag}", atol=1e-4)
            close(att.out_proj.weight.grad, wo.grad, f"attention dWout {tag}", atol=1e-4)
            out[f"x_{tag}"] = x.detach()
            out[f"gy_{tag}"] = gy
            out[f"y_{tag}"] = y.detach()
            out[f"dx_{tag}"] = x.grad
            for nm, g in (("dwqkv", att.qkv_proj.weight.grad), ("dwout", att.out_proj.weight.grad)):
                d, p, v = grad_digest(g, f"{nm}_{tag}")
                out[f"{nm}_{tag}_digest"], out[f"{nm}_{tag}_pos"], out[f"{nm}_{tag}_val"] = d, p, v
    save("attention.npz", **out)


def gen_encoder():
    for depth, heads, B, frames, img in ((1, 16, 1, 4, 32), (2, 16, 2, 4, 32), (2, 8, 1, 4, 48)):
        enc = swm.VideoViTEncoder(img_size=img, num_frames=frames, embed_dim=1024, depth=depth, num_heads=heads).eval()
        shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items() if "inv_freq" not in k}
        assert shapes == R.encoder_param_shapes(depth), "state-dict keys/shapes differ from oracle list"
        sd = R.fill_state_dict_(enc.state_dict())
        enc.load_state_dict(sd)
        video = R.hashed_uniform(f"golden.enc.video.{depth}.{heads}", (B, 3, frames, img, img), 1.0).requires_grad_(True)
        y = enc(video)
        gy = R.hashed_uniform("golden.enc.gy", y.shape, 1.0)
        (y * gy).sum().backward()
        P = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "inv_freq" not in k}
        v2 = video.detach().clone().requires_grad_(True)
        y2 = R.encoder_forward(P, v2, depth, heads)
        (y2 * gy).sum().backward()
        tag = f"d{depth}h{heads}"
        close(y, y2, f"encoder out {tag}", atol=1e-4)
        close(video.grad, v2.grad, f"encoder dvideo {tag}", atol=1e-4, rtol=1e-4)
        out = dict(video=video.detach(), gy=gy, y=y.detach(), dvideo=video.grad,
                   cfg=np.array([depth, heads, B, frames, img], dtype=np.int64))
        for k, p in enc.named_parameters():
            close(p.grad, P[k].grad, f"encoder grad {tag} {k}", atol=2e-4, rtol=1e-4)
            d, pos, val = grad_digest(p.grad, k)
            out[f"g:{k}:digest"], out[f"g:{k}:pos"], out[f"g:{k}:val"] = d, pos, val
        save(f"encoder_{tag}.npz", **out)


def gen_seed_canary():
    torch.manual_seed(0)
    enc = swm.VideoViTEncoder(img_size=32, num_frames=4, embed_dim=1024, depth=1, num_heads=16).eval()
    x = torch.randn(1, 3, 4, 32, 32, generator=torch.Generator().manual_seed(1))
    y = enc(x)
    loss = (y ** 2).mean()
    loss.backward()
    save("seed_canary.npz",
         y_abs_sum=y.detach().double().abs().sum().item(), y_head=y[0, 0, :8].detach(),
         conv_w_sum=enc.tubelet_embed.projection.weight.double().sum().item(),
         qkv_w_sum=enc.blocks[0].attn.qkv_proj.weight.double().sum().item(),
         fc1_w_sum=enc.blocks[0].mlp[0].weight.double().sum().item(),
         fc2_b_sum=enc.blocks[0].mlp[3].bias.double().sum().item(),
         dqkv_l1=enc.blocks[0].attn.qkv_proj.weight.grad.double().abs().sum().item(),
         keys=np.array(list(enc.state_dict().keys())))



def gen_predictor():
    """Predictor pieces and one whole-model forward (inference), through the reference classes."""
    # (a) the recurrence alone, with a non-zero incoming state
    B, L, H, D = 2, 21, 2, 128
    layer = swm.GatedDeltaLayer(256, num_heads=H, head_dim=D)
    q = torch.nn.functional.normalize(R.hashed_uniform("golden.gd.q", (B, L, H * D), 1.0), dim=-1).view(B, L, H, D)
    k = torch.nn.functional.normalize(R.hashed_uniform("golden.gd.k", (B, L, H * D), 1.0), dim=-1).view(B, L, H, D)
    v = R.hashed_uniform("golden.gd.v", (B, L, H, D), 1.0)
    alpha = R.hashed_uniform("golden.gd.alpha", (B, L, H), 0.45, 0.5)
    beta = R.hashed_uniform("golden.gd.beta", (B, L, H), 0.45, 0.5)
    s0 = R.hashed_uniform("golden.gd.state", (B, H, D, D), 0.1)
    with torch.no_grad():
        out, s1 = layer._sequential_gated_delta(q, k, v, alpha, beta, s0.clone())
        out0, _ = layer._sequential_gated_delta(q, k, v, alpha, beta, None)
    o_out, o_s1 = R.gated_delta_sequential(q, k, v, alpha, beta, s0)
    close(o_out, out, "gated delta recurrence out")
    close(o_s1, s1, "gated delta recurrence state")
    close(R.gated_delta_sequential(q, k, v, alpha, beta, None)[0], out0, "gated delta recurrence out (zero state)")
    arrs = dict(gd_out=out, gd_out_zero_state=out0, gd_state_sample=s1[:, :, ::8, ::8])

    # (b) one HybridPredictorBlock with and without the memory layer, block-causal mask, heads 8 (hd 128)
    dim, heads, tpt, T = 1024, 8, 6, 4
    Lb = tpt * T
    x = R.hashed_uniform("golden.pb.x", (2, Lb, dim), 1.0)
    mask = swm.BlockCausalMask.create_mask(Lb, tpt, x.device)
    assert torch.equal(mask, R.block_causal_mask(Lb, tpt))
    for use_dn in (True, False):
        blk = swm.HybridPredictorBlock(dim, heads, use_deltanet=use_dn).eval()
        P = R.fill_state_dict_({"pb." + kk: vv for kk, vv in blk.state_dict().items()})
        blk.load_state_dict({kk[3:]: vv for kk, vv in P.items()})
        with torch.no_grad():
            y, st = blk(x, attn_mask=mask)
        oy, ost = R.hybrid_predictor_block(x, P, "pb.", heads, use_dn, mask)
        close(oy, y, f"predictor block (deltanet={use_dn})", rtol=1e-4, atol=1e-4)
        arrs[f"block_out_dn{int(use_dn)}"] = y
        if use_dn:
            close(ost, st, "predictor block state", rtol=1e-4, atol=1e-4)
            arrs["block_state_sample"] = st[:, :, ::8, ::8]
    arrs["block_tpt"] = tpt

    # (c) the whole model, horizon 2 (img 32, 4 frames: 2 temporal x 4 spatial patches, tokens_per_timestep 6)
    torch.manual_seed(0)
    model = swm.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                                    predictor_depth=3, predictor_heads=8).eval()
    model.load_state_dict(R.fill_state_dict_(model.state_dict()))
    video = R.hashed_uniform("golden.wm.video", (2, 3, 4, 32, 32), 1.0)
    actions = R.hashed_uniform("golden.wm.actions", (2, 4, 7), 1.0)
    states = R.hashed_uniform("golden.wm.states", (2, 4, 7), 1.0)
    with torch.no_grad():
        preds, cur, fut = model(video, actions, states, predict_horizon=2)
    arrs["wm_current_logits"] = cur
    arrs["wm_future_logits"] = fut
    arrs["wm_horizon"] = len(preds)
    for i, p in enumerate(preds):
        for kk, vv in p.items():
            arrs[f"wm_pred{i}_{kk}"] = vv
    arrs["wm_keys"] = np.array(sorted(preds[0].keys()))
    save("predictor.npz", **arrs)


def gen_masking():
    s1 = _import_reference_stage1()
    out = {}
    for Tp in (4, 5, 8, 32):
        for ratio in (0.5, 0.75, 0.9):
            per = 3
            x = R.hashed_uniform("golden.mask.x", (2, Tp * per, 8), 1.0)
            rm, rmask = s1.StructuredTemporalMasking.future_masking(x, Tp, ratio)
            mm, mmask = R.future_masking(x, Tp, ratio)
            assert torch.equal(rmask, mmask) and torch.equal(rm, mm), (Tp, ratio)
            out[f"mask_T{Tp}_r{ratio}"] = rmask[0].numpy()
    print("  ok future_masking: bit-exact for all (T, ratio)")
    save("future_masking.npz", **out)

    # one stage-1 loss evaluation on fixed features (dropout disabled) through the reference heads
    torch.manual_seed(3)
    B, N, D, Tp = 4, 8, 1024, 2
    tc = torch.nn.Sequential(torch.nn.Linear(D, D // 2), torch.nn.ReLU(), torch.nn.Dropout(0.0), torch.nn.Linear(D // 2, 2))
    cp = torch.nn.Sequential(torch.nn.Linear(D, D), torch.nn.ReLU(), torch.nn.Linear(D, 128))
    dh = torch.nn.Sequential(torch.nn.LayerNorm(D), torch.nn.Linear(D, D), torch.nn.GELU(), torch.nn.Linear(D, D))
    heads = {}
    for pre, mod in (("temporal_classifier.", tc), ("contrastive_proj.", cp), ("denoise_head.", dh)):
        filled = R.fill_state_dict_({pre + k: v for k, v in mod.state_dict().items()})
        mod.load_state_dict({k[len(pre):]: v for k, v in filled.items()})
        heads.update(filled)
    feats = [R.hashed_uniform(f"golden.s1.f{i}", (B, N, D), 1.0) for i in range(4)]
    label = torch.tensor([0, 1, 1, 0])
    import torch.nn.functional as F
    l_t = F.cross_entropy(tc(feats[0].mean(1)), label)
    z1 = F.normalize(cp(feats[1].mean(1)), dim=1)
    z2 = F.normalize(cp(feats[2].mean(1)), dim=1)
    lab = torch.arange(B)
    l_c = 0.5 * (F.cross_entropy(z1 @ z2.t() / 0.07, lab) + F.cross_entropy(z2 @ z1.t() / 0.07, lab))
    masked, mb = s1.StructuredTemporalMasking.future_masking(feats[3], Tp, 0.5)
    keep = ~mb
    pooled = (masked * keep.unsqueeze(-1)).sum(1) / keep.sum(1, keepdim=True).clamp_min(1)
    l_m = F.mse_loss(dh(pooled), feats[3].mean(1))
    total = l_t + l_c + 0.1 * l_m
    mine, parts = R.stage1_losses(feats[0], feats[1], feats[2], feats[3], label, heads, Tp)
    close(total, mine, "stage-1 total loss")
    save("stage1_losses.npz", total=total.item(), l_t=l_t.item(), l_c=l_c.item(), l_m=l_m.item(), label=label.numpy())


def _import_reference_stage1():
    for m in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    import stage1_surgvu_pretraining as s1  # the reference script (dataset classes unused)
    return s1


def stage1_loop_batch(B=4):
    """The one synthetic batch of the stage-1 loop fixture (regenerated from names by the tests)."""
    shape = (B, 3, 4, 32, 32)
    return {"video": R.hashed_uniform("golden.s1loop.video", shape, 1.0),
            "video_augmented": R.hashed_uniform("golden.s1loop.aug", shape, 1.0),
            "shuffled_video": R.hashed_uniform("golden.s1loop.shuf", shape, 1.0),
            "order_label": torch.tensor([1, 0, 0, 1][:B])}


def _manifest(obj):
    """Nested keys / shapes / dtypes of a checkpoint dictionary (no values)."""
    if torch.is_tensor(obj):
        return {"tensor": list(obj.shape), "dtype": str(obj.dtype).replace("torch.", "")}
    if isinstance(obj, dict):
        return {"dict": {str(k): _manifest(v) for k, v in obj.items()}}
    if isinstance(obj, (list, tuple)):
        return {"list": [_manifest(v) for v in obj]}
    return {"value": type(obj).__name__}


def gen_stage1_loop():
    """Run the REFERENCE's own training loop, `train_stage1_surgvu` (stage1_surgvu_pretraining.py:340-530), for one
    epoch of one batch on the CPU: list-backed loader, every nn.Dropout made the identity for the duration (the
    loop calls model.train()), seeded head initialisation.  Twice: with learning rate 0 (the saved heads are then
    the initial heads, which certifies that a seed reproduces them) and with the default 4e-4 (post-step weights).
    Commits the loss the loop wrote into `checkpoint_epoch_01.pth`, samples of the post-step weights, and a manifest
    (nested keys, shapes, dtypes) of both checkpoint files the loop writes."""
    import json
    import tempfile
    s1 = _import_reference_stage1()
    SEED, B = 123, 4

    class ListLoader(list):
        batch_size = B

    def run(lr):
        model = swm.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                                        predictor_depth=1, predictor_heads=8)
        model.encoder.load_state_dict(R.fill_state_dict_(model.encoder.state_dict()))
        out = tempfile.mkdtemp(prefix="s1loop_")
        keep = torch.nn.Dropout.forward
        torch.nn.Dropout.forward = lambda self, x: x
        try:
            torch.manual_seed(SEED)                       # the loop builds its heads from this stream
            s1.train_stage1_surgvu(model, ListLoader([stage1_loop_batch(B)]), num_epochs=1, device="cpu", save_dir=out,
                                   learning_rate=lr)
        finally:
            torch.nn.Dropout.forward = keep
        return (torch.load(os.path.join(out, "checkpoint_epoch_01.pth"), weights_only=True),
                torch.load(os.path.join(out, "best_encoder_surgvu.pth"), weights_only=True))

    frozen, _ = run(0.0)
    stepped, best = run(4e-4)
    assert frozen["loss"] == stepped["loss"]
    # (1) a seed reproduces the loop's head initialisation in the layer order the product's Stage1Heads uses
    from vit_amd.train_step import Stage1Heads
    torch.manual_seed(SEED)
    mine = Stage1Heads(1024)
    for part in ("temporal_classifier", "contrastive_proj", "denoise_head"):
        for k, v in getattr(mine, part).state_dict().items():
            assert torch.equal(v, frozen[part][k]), (part, k)
    enc0 = R.fill_state_dict_(stepped["encoder_state_dict"])
    for k, v in frozen["encoder_state_dict"].items():
        assert torch.equal(v, enc0[k]), k
    # (2) the oracle's restatement of the step math gives the loop's loss
    batch = stage1_loop_batch(B)
    P = {k: v for k, v in enc0.items() if "inv_freq" not in k}
    f = [R.encoder_forward(P, batch[k], 1, 16) for k in ("shuffled_video", "video", "video_augmented", "video")]
    heads = {f"{part}.{k}": v for part in ("temporal_classifier", "contrastive_proj", "denoise_head")
             for k, v in frozen[part].items()}
    total, parts = R.stage1_losses(f[0], f[1], f[2], f[3], batch["order_label"], heads, Tp=2)
    close(total.detach(), torch.tensor(stepped["loss"]), "stage-1 loop loss (oracle vs reference loop)", atol=1e-5)

    arrs = dict(loss=stepped["loss"], seed=SEED, batch=B, lr=4e-4, weight_decay=0.1,
                l_parts=np.array([float(v) for v in parts]))
    for name, t in (("encoder.blocks.0.attn.qkv_proj.weight", stepped["encoder_state_dict"]["blocks.0.attn.qkv_proj.weight"]),
                    ("encoder.blocks.0.mlp.3.bias", stepped["encoder_state_dict"]["blocks.0.mlp.3.bias"]),
                    ("encoder.norm.weight", stepped["encoder_state_dict"]["norm.weight"]),
                    ("temporal_classifier.3.weight", stepped["temporal_classifier"]["3.weight"]),
                    ("contrastive_proj.2.weight", stepped["contrastive_proj"]["2.weight"]),
                    ("denoise_head.3.weight", stepped["denoise_head"]["3.weight"])):
        flat = t.reshape(-1).double()
        rng = np.random.default_rng(R._name_seed(name) % (1 << 32))
        pos = np.sort(rng.integers(0, flat.numel(), size=min(256, flat.numel())))
        arrs[f"w:{name}:pos"], arrs[f"w:{name}:val"] = pos.astype(np.int64), flat[pos].numpy()
    save("stage1_loop.npz", **arrs)
    with open(os.path.join(HERE, "stage1_checkpoint_manifest.json"), "w") as fh:
        json.dump({"checkpoint_epoch_01.pth": _manifest(stepped), "best_encoder_surgvu.pth": _manifest(best)}, fh, indent=1,
                  sort_keys=False)
    print("wrote stage1_checkpoint_manifest.json")


def gen_finetune_loop():
    """Run the REFERENCE's fine-tuning loop, `train_stage1_with_vjepa2` (stage1_vjepa2_finetuning.py:141-397), for one
    epoch of one batch on the CPU, with the encoder frozen and unfrozen: same list-backed loader, dropout made the
    identity, seeded heads.  The loop's optimizer has one group per sub-module (encoder at a tenth of the rate when
    it trains, ft:196-232) under a warm-up schedule (ft:235-241: the first epoch runs at half the base rate).  Commits
    the loss in `best_vjepa2_finetuned.pth`, samples of the post-step weights, the group learning rates and a manifest
    of the file."""
    import json
    import tempfile
    _import_reference_stage1()
    import stage1_vjepa2_finetuning as ft        # the reference
    SEED, B, LR = 321, 4, 4e-4

    class ListLoader(list):
        batch_size = B

    def run(frozen):
        model = swm.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                                        predictor_depth=1, predictor_heads=8)
        model.encoder.load_state_dict(R.fill_state_dict_(model.encoder.state_dict()))
        out = tempfile.mkdtemp(prefix="ftloop_")
        keep = torch.nn.Dropout.forward
        torch.nn.Dropout.forward = lambda self, x: x
        try:
            torch.manual_seed(SEED)
            ft.train_stage1_with_vjepa2(model, ListLoader([stage1_loop_batch(B)]), num_epochs=1, device="cpu", save_dir=out,
                                        learning_rate=LR, encoder_frozen=frozen)
        finally:
            torch.nn.Dropout.forward = keep
        return torch.load(os.path.join(out, "best_vjepa2_finetuned.pth"), weights_only=True)

    arrs = dict(seed=SEED, batch=B, lr=LR, weight_decay=0.1)
    manifests = {}
    for frozen in (True, False):
        ck = run(frozen)
        tag = "frozen" if frozen else "unfrozen"
        manifests[tag] = _manifest(ck)
        enc0 = R.fill_state_dict_(dict(ck["encoder_state_dict"]))
        if frozen:
            for k, v in ck["encoder_state_dict"].items():
                assert torch.equal(v, enc0[k]), k
        # the loss is a function of the initial weights: the oracle's restatement must give it (heads by seed)
        from vit_amd.train_step import Stage1Heads
        torch.manual_seed(SEED)
        init = Stage1Heads(1024)
        heads = {f"{part}.{k}": v for part in ("temporal_classifier", "contrastive_proj", "denoise_head")
                 for k, v in getattr(init, part).state_dict().items()}
        batch = stage1_loop_batch(B)
        P = {k: v for k, v in enc0.items() if "inv_freq" not in k}
        f = [R.encoder_forward(P, batch[k], 1, 16) for k in ("shuffled_video", "video", "video_augmented", "video")]
        total, parts = R.stage1_losses(f[0], f[1], f[2], f[3], batch["order_label"], heads, Tp=2)
        close(total.detach(), torch.tensor(ck["loss"]), f"fine-tune loop loss ({tag}; oracle vs reference loop)", atol=1e-5)
        arrs[f"{tag}:loss"] = ck["loss"]
        arrs[f"{tag}:l_parts"] = np.array([float(v) for v in parts])
        arrs[f"{tag}:group_lr"] = np.array([g["lr"] for g in ck["optimizer"]["param_groups"]])
        arrs[f"{tag}:group_initial_lr"] = np.array([g["initial_lr"] for g in ck["optimizer"]["param_groups"]])
        arrs[f"{tag}:group_sizes"] = np.array([len(g["params"]) for g in ck["optimizer"]["param_groups"]])
        for name, t in (("encoder.blocks.0.attn.qkv_proj.weight", ck["encoder_state_dict"]["blocks.0.attn.qkv_proj.weight"]),
                        ("encoder.norm.weight", ck["encoder_state_dict"]["norm.weight"]),
                        ("temporal_classifier.3.weight", ck["temporal_classifier"]["3.weight"]),
                        ("contrastive_proj.2.weight", ck["contrastive_proj"]["2.weight"]),
                        ("denoise_head.3.weight", ck["denoise_head"]["3.weight"])):
            flat = t.reshape(-1).double()
            rng = np.random.default_rng(R._name_seed(name) % (1 << 32))
            pos = np.sort(rng.integers(0, flat.numel(), size=min(256, flat.numel())))
            arrs[f"{tag}:w:{name}:pos"], arrs[f"{tag}:w:{name}:val"] = pos.astype(np.int64), flat[pos].numpy()
    save("finetune_loop.npz", **arrs)
    with open(os.path.join(HERE, "finetune_checkpoint_manifest.json"), "w") as fh:
        json.dump(manifests, fh, indent=1, sort_keys=False)
    print("wrote finetune_checkpoint_manifest.json")


def gen_pretrained_manifest():
    """The reference's writer and reader of the adapted-checkpoint file: `save_adapted_checkpoint`
    (load_vjepa2_pretrained.py:279-300) writes a reference model -> manifest committed; the reference's
    `load_pretrained_model` (stage1_vjepa2_finetuning.py:57-138) then loads a file written by THIS package's
    `checkpoint.save_pretrained_model` and every tensor must come back equal (interchange in the direction the
    tests cannot run without the reference)."""
    import json
    import tempfile
    _import_reference_stage1()
    import load_vjepa2_pretrained as lp          # the reference (its network code is never called)
    import stage1_vjepa2_finetuning as ft        # the reference
    out = tempfile.mkdtemp(prefix="lp_")
    cfg = {"embed_dim": 1024, "depth": 1, "num_heads": 16}
    ref_model = swm.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                                        predictor_depth=2, predictor_heads=8)
    ref_model.load_state_dict(R.fill_state_dict_(ref_model.state_dict()))
    lp.save_adapted_checkpoint(ref_model, os.path.join(out, "ref.pth"), cfg, {"source": "test"})
    man = _manifest(torch.load(os.path.join(out, "ref.pth"), weights_only=True))
    with open(os.path.join(HERE, "pretrained_checkpoint_manifest.json"), "w") as fh:
        json.dump(man, fh, indent=1)
    print("wrote pretrained_checkpoint_manifest.json")

    from vit_amd import checkpoint as C
    from vit_amd import surgical_world_model as mine
    my_model = mine.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                                        predictor_depth=2, predictor_heads=8)
    my_model.load_state_dict(R.fill_state_dict_(my_model.state_dict()))
    C.save_pretrained_model(my_model, os.path.join(out, "mine.pth"), cfg, {"source": "test"})
    loaded, config = ft.load_pretrained_model(os.path.join(out, "mine.pth"), img_size=32, num_frames=4, predictor_depth=2,
                                              device="cpu")
    assert config == {"encoder_dim": 1024, "encoder_depth": 1, "encoder_heads": 16}
    ref_sd, got_sd = ref_model.state_dict(), loaded.state_dict()
    assert list(ref_sd) == list(got_sd)
    for k in ref_sd:
        assert torch.equal(ref_sd[k], got_sd[k]), k
    print(f"  ok the reference's load_pretrained_model read this package's file: {len(ref_sd)} tensors equal")
    # and the other direction: this package reads the reference writer's file
    mine_loaded, my_cfg = C.load_pretrained_model(os.path.join(out, "ref.pth"), img_size=32, num_frames=4, predictor_depth=2,
                                                  device="cpu")
    assert my_cfg == config and not mine_loaded.load_result.missing_keys and not mine_loaded.load_result.unexpected_keys
    for k, v in mine_loaded.state_dict().items():
        assert torch.equal(v, ref_sd[k]), k
    print("  ok this package's load_pretrained_model read the reference writer's file")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pretrained":
        gen_pretrained_manifest()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "finetune_loop":
        gen_finetune_loop()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stage1_loop":  # only the loop fixture
        gen_stage1_loop()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "predictor":     # only the predictor fixture
        gen_predictor()
        raise SystemExit(0)
    with torch.enable_grad():
        gen_predictor()
        gen_rope()
        gen_tubelet()
        gen_attention()
        gen_encoder()
        gen_seed_canary()
        gen_masking()
        gen_stage1_loop()
        gen_finetune_loop()
        gen_pretrained_manifest()
    print("all golden vectors written")
