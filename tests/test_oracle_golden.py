"""CPU: the oracle (oracle/ref_cpu.py) against the committed reference outputs in tests/golden."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_rope_tables_bit_exact(golden_dir):
    g = _load(golden_dir, "rope.npz")
    for grid in [(2, 2, 2), (8, 14, 14), (8, 16, 16), (32, 16, 16)]:
        tag = "x".join(map(str, grid))
        tabs = R.rope_tables(*grid)
        for (c, s), ax in zip(tabs, "thw"):
            d = np.array([c.double().sum().item(), s.double().sum().item(),
                          c.double().abs().sum().item(), s.double().abs().sum().item()])
            np.testing.assert_array_equal(d, g[f"digest_{ax}_{tag}"])
        t, h, w = R.token_coords(*grid)
        ar = torch.arange(len(t))
        d = np.array([t.sum().item(), h.sum().item(), w.sum().item(),
                      (t * ar).sum().item(), (h * ar).sum().item(), (w * ar).sum().item()], dtype=np.int64)
        np.testing.assert_array_equal(d, g[f"pos_digest_{tag}"])
    for ax, (c, s) in zip("thw", R.rope_tables(2, 2, 2)):
        np.testing.assert_array_equal(c.numpy(), g[f"cos_{ax}_2x2x2"])
        np.testing.assert_array_equal(s.numpy(), g[f"sin_{ax}_2x2x2"])


def test_tubelet(golden_dir):
    g = _load(golden_dir, "tubelet.npz")
    sd = R.fill_state_dict_({"tubelet_embed.projection.weight": torch.empty(1024, 3, 2, 16, 16),
                             "tubelet_embed.projection.bias": torch.empty(1024)})
    y = R.tubelet_embed(torch.from_numpy(g["video"]), sd["tubelet_embed.projection.weight"],
                        sd["tubelet_embed.projection.bias"])
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("heads", [16, 8])
@pytest.mark.parametrize("masked", [False, True])
def test_attention(golden_dir, heads, masked):
    g = _load(golden_dir, "attention.npz")
    tag = f"h{heads}_{'mask' if masked else 'nomask'}"
    sd = R.fill_state_dict_({"blocks.0.attn.qkv_proj.weight": torch.empty(3072, 1024),
                             "blocks.0.attn.out_proj.weight": torch.empty(1024, 1024)})
    x = torch.from_numpy(g[f"x_{tag}"]).requires_grad_(True)
    wq = sd["blocks.0.attn.qkv_proj.weight"].requires_grad_(True)
    wo = sd["blocks.0.attn.out_proj.weight"].requires_grad_(True)
    mask = torch.triu(torch.ones(8, 8, dtype=torch.bool), diagonal=1) if masked else None
    y = R.rope_attention(x, wq, wo, heads, (2, 2, 2), mask)
    (y * torch.from_numpy(g[f"gy_{tag}"])).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), g[f"y_{tag}"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(x.grad.numpy(), g[f"dx_{tag}"], rtol=1e-4, atol=1e-4)
    for nm, p in (("dwqkv", wq), ("dwout", wo)):
        flat = p.grad.reshape(-1).double()
        np.testing.assert_allclose(flat[g[f"{nm}_{tag}_pos"]].numpy(), g[f"{nm}_{tag}_val"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(flat.abs().sum().item(), g[f"{nm}_{tag}_digest"][1], rtol=1e-5)


@pytest.mark.parametrize("tag", ["d1h16", "d2h16", "d2h8"])
def test_encoder_fwd_bwd(golden_dir, tag):
    g = _load(golden_dir, f"encoder_{tag}.npz")
    depth, heads, B, frames, img = (int(v) for v in g["cfg"])
    P = R.make_encoder_params(depth, requires_grad=True)
    video = torch.from_numpy(g["video"]).requires_grad_(True)
    y = R.encoder_forward(P, video, depth, heads)
    (y * torch.from_numpy(g["gy"])).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(video.grad.numpy(), g["dvideo"], rtol=1e-4, atol=1e-4)
    for k, p in P.items():
        flat = p.grad.reshape(-1).double()
        np.testing.assert_allclose(flat[g[f"g:{k}:pos"]].numpy(), g[f"g:{k}:val"], rtol=2e-4, atol=2e-4, err_msg=k)
        np.testing.assert_allclose(flat.abs().sum().item(), g[f"g:{k}:digest"][1], rtol=1e-4, err_msg=k)


def test_future_masking_bit_exact(golden_dir):
    g = _load(golden_dir, "future_masking.npz")
    for Tp in (4, 5, 8, 32):
        for ratio in (0.5, 0.75, 0.9):
            x = R.hashed_uniform("golden.mask.x", (2, Tp * 3, 8), 1.0)
            masked, mask = R.future_masking(x, Tp, ratio)
            np.testing.assert_array_equal(mask[0].numpy(), g[f"mask_T{Tp}_r{ratio}"])
            np.testing.assert_array_equal(mask[1].numpy(), g[f"mask_T{Tp}_r{ratio}"])
            assert torch.equal(masked[~mask], x[~mask]) and (masked[mask] == 0).all()
    # the cut-off arithmetic: int(8*(1-0.9)) == 0 masks everything
    assert g["mask_T8_r0.9"].all()


def test_stage1_losses(golden_dir):
    g = _load(golden_dir, "stage1_losses.npz")
    B, N, D, Tp = 4, 8, 1024, 2
    shapes = {"temporal_classifier.0.weight": (D // 2, D), "temporal_classifier.0.bias": (D // 2,),
              "temporal_classifier.3.weight": (2, D // 2), "temporal_classifier.3.bias": (2,),
              "contrastive_proj.0.weight": (D, D), "contrastive_proj.0.bias": (D,),
              "contrastive_proj.2.weight": (128, D), "contrastive_proj.2.bias": (128,),
              "denoise_head.0.weight": (D,), "denoise_head.0.bias": (D,),
              "denoise_head.1.weight": (D, D), "denoise_head.1.bias": (D,),
              "denoise_head.3.weight": (D, D), "denoise_head.3.bias": (D,)}
    heads = R.fill_state_dict_({k: torch.empty(v) for k, v in shapes.items()})
    feats = [R.hashed_uniform(f"golden.s1.f{i}", (B, N, D), 1.0) for i in range(4)]
    total, (l_t, l_c, l_m) = R.stage1_losses(*feats, torch.from_numpy(g["label"]), heads, Tp)
    np.testing.assert_allclose(total.item(), g["total"], rtol=1e-5)
    np.testing.assert_allclose([l_t.item(), l_c.item(), l_m.item()], [g["l_t"], g["l_c"], g["l_m"]], rtol=1e-5)


def test_stage1_loss_of_the_reference_training_loop(golden_dir):
    """stage1_loop.npz holds the loss the REFERENCE's own loop (`train_stage1_surgvu`, s1:340-530) wrote into
    checkpoint_epoch_01.pth for one batch: heads initialised from torch.manual_seed(seed) inside the loop, encoder on
    the closed-form weights, dropout disabled.  The oracle's restatement of the step math must give that number."""
    from conftest import stage1_loop_batch
    from vit_amd.train_step import Stage1Heads
    g = _load(golden_dir, "stage1_loop.npz")
    torch.manual_seed(int(g["seed"]))
    heads = {k: v for k, v in Stage1Heads(1024).state_dict().items()}       # same layer order = same random stream
    batch = stage1_loop_batch(int(g["batch"]))
    P = R.make_encoder_params(1)
    f = [R.encoder_forward(P, batch[k], 1, 16) for k in ("shuffled_video", "video", "video_augmented", "video")]
    total, parts = R.stage1_losses(f[0], f[1], f[2], f[3], batch["order_label"], heads, Tp=2)
    np.testing.assert_allclose(total.item(), float(g["loss"]), rtol=2e-6)
    np.testing.assert_allclose([float(v) for v in parts], g["l_parts"], rtol=2e-5)


def test_jepa_masks_are_seeded_and_disjoint():
    c1, t1 = R.jepa_block_masks(7, 4, (8, 16, 16))
    c2, t2 = R.jepa_block_masks(7, 4, (8, 16, 16))
    assert torch.equal(c1, c2) and torch.equal(t1, t2)
    assert c1.dtype == torch.int32 and t1.dtype == torch.int32
    for b in range(4):
        assert not set(c1[b].tolist()) & set(t1[b].tolist())
        assert (c1[b][1:] > c1[b][:-1]).all() and (t1[b][1:] > t1[b][:-1]).all()
    x = R.hashed_uniform("x", (4, 2048, 16), 1.0)
    back = R.token_scatter(R.token_gather(x, t1), t1, 2048)
    assert torch.equal(R.token_gather(back, t1), R.token_gather(x, t1))


def test_finetune_schedule_and_groups_follow_the_reference_loop(golden_dir):
    """`warmup_cosine` and `finetune_param_groups` against what the reference's fine-tuning loop left in its optimizer
    (tests/golden/finetune_loop.npz: group sizes and learning rates before and after the first `scheduler.step()`)."""
    import os
    import numpy as np
    import torch
    from vit_amd import surgical_world_model as swm
    from vit_amd import train_step as ts
    g = np.load(os.path.join(golden_dir, "finetune_loop.npz"), allow_pickle=False)
    lam = ts.warmup_cosine(2, 10)
    assert [lam(0), lam(1), lam(2)] == [0.5, 1.0, 1.0] and abs(lam(6) - 0.5) < 1e-12 and abs(lam(10)) < 1e-12
    for frozen in (True, False):
        tag = "frozen" if frozen else "unfrozen"
        model = swm.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                                        predictor_depth=1, predictor_heads=8)
        heads = ts.Stage1Heads(1024)
        groups = ts.finetune_param_groups(model, heads, float(g["lr"]), encoder_frozen=frozen)
        assert [len(x["params"]) for x in groups] == g[f"{tag}:group_sizes"].tolist()
        np.testing.assert_allclose([x["lr"] for x in groups], g[f"{tag}:group_initial_lr"], rtol=1e-12)
        assert all(p.requires_grad != frozen for p in model.encoder.parameters())
        opt = torch.optim.AdamW(groups, weight_decay=0.1)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, ts.warmup_cosine(2, 1))
        opt.step(); sched.step()
        np.testing.assert_allclose([x["lr"] for x in opt.param_groups], g[f"{tag}:group_lr"], rtol=1e-12)
