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
