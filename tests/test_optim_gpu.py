"""GPU: the multi-tensor AdamW + clip step against torch.optim.AdamW + clip_grad_norm_ (the reference's
optimizer calls, stage1_surgvu_pretraining.py:390, 469-470)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1024, 1024), (3,), (70000,), (2, 5), (4096,), (65536,), (1,), (131073,)]
    return [torch.randn(s, generator=g) for s in shapes]


@pytest.mark.parametrize("max_norm", [None, 1.0, 1e6])
def test_fused_adamw_matches_torch(max_norm):
    from vit_amd.optim import FusedAdamW
    ref_p = [torch.nn.Parameter(t.clone().cuda()) for t in make_params(0)]
    my_p = [torch.nn.Parameter(t.clone().cuda()) for t in make_params(0)]
    ref = torch.optim.AdamW(ref_p, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    mine = FusedAdamW(my_p, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, max_grad_norm=max_norm)
    for step in range(4):
        grads = make_params(100 + step)
        for i, (a, b, g) in enumerate(zip(ref_p, my_p, grads)):
            if step == 2 and i == 1:           # a parameter without a gradient in one step is skipped by both
                a.grad = b.grad = None
                continue
            a.grad = (g * 3).cuda()
            b.grad = (g * 3).cuda()
        if max_norm is not None:
            total = torch.nn.utils.clip_grad_norm_(ref_p, max_norm=max_norm)
        ref.step()
        mine.step()
        if max_norm is not None:
            assert abs(mine.grad_norm().item() - total.item()) < 1e-4 * total.item()
        for a, b in zip(ref_p, my_p):
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (step, a.shape, (a - b).abs().max().item())
    assert all(p._version > 0 for p in my_p)      # cached operand copies will notice the update


def test_fused_adamw_drives_the_encoder_step():
    from oracle import ref_cpu as R
    from vit_amd import surgical_world_model as swm
    from vit_amd.optim import FusedAdamW
    enc = swm.VideoViTEncoder(img_size=32, num_frames=4, embed_dim=1024, depth=1, num_heads=16)
    enc.load_state_dict(R.fill_state_dict_(enc.state_dict()))
    enc = enc.cuda()
    opt = FusedAdamW(enc.parameters(), lr=1e-2, weight_decay=0.1, max_grad_norm=1.0)
    video = R.hashed_uniform("opt.video", (2, 3, 4, 32, 32), 1.0).cuda()
    with torch.no_grad():
        before = enc(video).clone()
    opt.zero_grad()
    enc(video).pow(2).mean().backward()
    opt.step()
    with torch.no_grad():
        after = enc(video)
    assert torch.isfinite(after).all() and not torch.equal(before, after)


def test_two_param_groups_scheduler_and_state_dict_interchange(tmp_path):
    """The fine-tuning recipe (stage1_vjepa2_finetuning.py:216-239): two groups, the encoder at 0.1x the learning
    rate, a LambdaLR on top; one global clip norm over both groups.  Then checkpoint interchange: the state dict
    has torch.optim.AdamW's layout, loads into a fresh FusedAdamW and into torch.optim.AdamW (and back), and the
    resumed optimizers keep producing the same parameters."""
    from vit_amd.optim import FusedAdamW

    def groups(ps, lr):
        return [{"params": ps[:3], "lr": lr * 0.1}, {"params": ps[3:], "lr": lr}]

    ref_p = [torch.nn.Parameter(t.clone().cuda()) for t in make_params(0)]
    my_p = [torch.nn.Parameter(t.clone().cuda()) for t in make_params(0)]
    ref = torch.optim.AdamW(groups(ref_p, 2e-3), weight_decay=0.05)
    mine = FusedAdamW(groups(my_p, 2e-3), weight_decay=0.05, max_grad_norm=1.0)
    assert isinstance(mine, torch.optim.Optimizer) and mine.param_groups is mine.param_groups
    sched = lambda e: (e + 1) / 2 if e < 2 else 0.5       # warm-up then a lower plateau
    s_ref = torch.optim.lr_scheduler.LambdaLR(ref, sched)
    s_mine = torch.optim.lr_scheduler.LambdaLR(mine, sched)

    def one_step(opt_ref, opt_mine, pr, pm, seed):
        grads = make_params(seed)
        for a, b, g in zip(pr, pm, grads):
            a.grad, b.grad = (g * 2).cuda(), (g * 2).cuda()
        torch.nn.utils.clip_grad_norm_(pr, max_norm=1.0)
        opt_ref.step()
        opt_mine.step()

    for step in range(3):
        one_step(ref, mine, ref_p, my_p, 200 + step)
        s_ref.step()
        s_mine.step()
        assert [g["lr"] for g in mine.param_groups] == pytest.approx([g["lr"] for g in ref.param_groups])
        for a, b in zip(ref_p, my_p):
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (step, a.shape)

    # ---- save / load: same keys as torch.optim.AdamW's state, through torch.save like s1:504
    sd = mine.state_dict()
    sd_ref = ref.state_dict()
    assert set(sd["state"].keys()) == set(sd_ref["state"].keys())
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd["state"][0]["step"]) == float(sd_ref["state"][0]["step"]) == 3.0
    torch.save({"optimizer": sd}, tmp_path / "ck.pth")
    loaded = torch.load(tmp_path / "ck.pth", weights_only=True)["optimizer"]

    p2 = [torch.nn.Parameter(p.detach().clone()) for p in my_p]
    resumed = FusedAdamW(groups(p2, 1.0), weight_decay=0.0, max_grad_norm=1.0)
    resumed.load_state_dict(loaded)                                 # hyper-parameters come from the file
    p3 = [torch.nn.Parameter(p.detach().clone()) for p in my_p]
    torch_resumed = torch.optim.AdamW(groups(p3, 1.0))
    torch_resumed.load_state_dict(loaded)                           # FusedAdamW file -> torch.optim.AdamW
    p4 = [torch.nn.Parameter(p.detach().clone()) for p in my_p]
    from_torch = FusedAdamW(groups(p4, 1.0), max_grad_norm=1.0)
    from_torch.load_state_dict(ref.state_dict())                    # torch.optim.AdamW file -> FusedAdamW

    grads = make_params(999)
    for plist in (my_p, p2, p3, p4, ref_p):
        for p, g in zip(plist, grads):
            p.grad = (g * 2).cuda()
    torch.nn.utils.clip_grad_norm_(p3, max_norm=1.0)
    torch.nn.utils.clip_grad_norm_(ref_p, max_norm=1.0)
    for o in (mine, resumed, torch_resumed, from_torch, ref):
        o.step()
    for a, b, c, d, e in zip(my_p, p2, p3, p4, ref_p):
        assert torch.equal(a, b)                                    # resumed == never stopped, bit for bit
        assert torch.allclose(a, c, rtol=2e-6, atol=2e-7)
        assert torch.allclose(a, d, rtol=2e-6, atol=2e-7)
        assert torch.allclose(a, e, rtol=2e-6, atol=2e-7)
    assert float(resumed.state[p2[0]]["step"]) == 4.0


def test_frozen_parameters_keep_their_place_in_the_state_dict():
    """A group that holds a `requires_grad=False` parameter (the frozen encoder of the fine-tuning recipe when all
    parameters are handed over): torch.optim.AdamW keeps it in the group, so its state dict numbers the parameters
    with it.  FusedAdamW must number them the same way, skip it in step(), and interchange state both ways."""
    from vit_amd.optim import FusedAdamW

    def build(seed):
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in make_params(seed)]
        ps[1].requires_grad_(False)                      # frozen, in the middle of the first group
        return ps, [{"params": ps[:3], "lr": 1e-3}, {"params": ps[3:], "lr": 2e-3}]

    ref_p, ref_g = build(0)
    my_p, my_g = build(0)
    ref = torch.optim.AdamW(ref_g, weight_decay=0.05)
    mine = FusedAdamW(my_g, weight_decay=0.05)
    assert [len(g["params"]) for g in mine.param_groups] == [len(g["params"]) for g in ref.param_groups]
    frozen_before = my_p[1].detach().clone()
    for step in range(2):
        grads = make_params(300 + step)
        for i, (a, b, g) in enumerate(zip(ref_p, my_p, grads)):
            if i != 1:
                a.grad, b.grad = g.cuda(), g.cuda()
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7)
    assert torch.equal(my_p[1], frozen_before)
    import copy
    # deep copies, as a checkpoint file would hold: a state dict's tensors are references to the live moments
    sd, sd_ref = copy.deepcopy(mine.state_dict()), copy.deepcopy(ref.state_dict())
    assert [g["params"] for g in sd["param_groups"]] == [g["params"] for g in sd_ref["param_groups"]]
    assert set(sd["state"].keys()) == set(sd_ref["state"].keys()) and 1 not in sd["state"]
    # both directions of the interchange, then one more step everywhere
    p2, g2 = build(0)
    p3, g3 = build(0)
    for dst, src in zip(p2 + p3, my_p + my_p):
        dst.data.copy_(src.data)
    from_torch = FusedAdamW(g2, weight_decay=0.05)
    from_torch.load_state_dict(sd_ref)
    to_torch = torch.optim.AdamW(g3, weight_decay=0.05)
    to_torch.load_state_dict(sd)
    grads = make_params(777)
    for plist in (my_p, p2, p3):
        for i, (p, g) in enumerate(zip(plist, grads)):
            if i != 1:
                p.grad = g.cuda()
    for o in (mine, from_torch, to_torch):
        o.step()
    for a, b, c in zip(my_p, p2, p3):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7) and torch.allclose(a, c, rtol=2e-6, atol=2e-7)
    # unfreezing after construction is refused with a message instead of writing moments nowhere
    my_p[1].requires_grad_(True)
    my_p[1].grad = torch.zeros_like(my_p[1])
    with pytest.raises(ValueError, match="frozen"):
        mine.step()
