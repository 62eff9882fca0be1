"""The stage-1 step's loss kernels (value + gradient per call) against the oracle's restatement of
stage1_surgvu_pretraining.py:426-462, evaluated in f64 with autograd on the CPU."""
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from vit_amd import ops, train_step
    return ops, train_step


@pytest.mark.parametrize("B,D,normalize", [(1, 128, True), (2, 128, True), (32, 128, True), (96, 128, True), (300, 128, True),
                                           (33, 64, True), (256, 128, False), (7, 20, True)])
def test_infonce_value_and_gradients(mods, B, D, normalize):
    ops, _ = mods
    p1 = R.hashed_uniform("t.nce.p1", (B, D), 1.5)
    p2 = (0.6 * p1 + R.hashed_uniform("t.nce.p2", (B, D), 1.0)).contiguous()
    if not normalize:
        p1, p2 = torch.nn.functional.normalize(p1, dim=1), torch.nn.functional.normalize(p2, dim=1)
    a, b = p1.double().requires_grad_(True), p2.double().requires_grad_(True)
    ref = R.infonce_sym(a, b, 0.07, normalize)
    ref.backward()
    loss, g1, g2 = ops.infonce_sym(p1.cuda(), p2.cuda(), 0.07, normalize)
    assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
    scale = max(a.grad.abs().max().item(), b.grad.abs().max().item(), 1e-12)
    # (B = 1: the true gradient is exactly zero; the kernel's is f32 rounding of exp(0) + exp(0) - 2 scaled by 1 / tau)
    assert (g1.cpu().double() - a.grad).abs().max().item() <= 2e-5 * scale + 5e-7
    assert (g2.cpu().double() - b.grad).abs().max().item() <= 2e-5 * scale + 5e-7
    again = ops.infonce_sym(p1.cuda(), p2.cuda(), 0.07, normalize)
    assert torch.equal(again[0], loss) and torch.equal(again[1], g1) and torch.equal(again[2], g2)      # fixed summation order


@pytest.mark.parametrize("B,C", [(1, 2), (5, 2), (32, 2), (33, 10), (7, 1000)])
def test_cross_entropy_value_and_gradient(mods, B, C):
    ops, _ = mods
    x = R.hashed_uniform("t.ce.x", (B, C), 4.0)
    lab = (torch.arange(B) * 7 + 1) % C
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xd, lab)
    ref.backward()
    loss, dx = ops.cross_entropy(x.cuda(), lab.cuda())
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    assert (dx.cpu().double() - xd.grad).abs().max().item() <= 1e-6
    with pytest.raises(Exception):
        ops.cross_entropy(x.cuda(), lab.int().cuda())


@pytest.mark.parametrize("shape", [(7,), (32, 1024), (3, 341)])
def test_mse_value_and_gradient(mods, shape):
    ops, _ = mods
    x, y = R.hashed_uniform("t.mse.x", shape, 2.0), R.hashed_uniform("t.mse.y", shape, 2.0)
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(xd, y.double())
    ref.backward()
    loss, dx = ops.mse_loss(x.cuda(), y.cuda())
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert (dx.cpu().double() - xd.grad).abs().max().item() <= 1e-6 * xd.grad.abs().max().item() + 1e-12


def test_loss_functions_carry_upstream_gradients(mods):
    """Through autograd: the saved gradients are scaled by whatever multiplies the loss (lambda, the data-parallel weights)."""
    _, ts = mods
    p1 = R.hashed_uniform("t.nce2.p1", (16, 128), 1.0).cuda().requires_grad_(True)
    p2 = R.hashed_uniform("t.nce2.p2", (16, 128), 1.0).cuda().requires_grad_(True)
    x = R.hashed_uniform("t.nce2.x", (16, 2), 2.0).cuda().requires_grad_(True)
    lab = (torch.arange(16) % 2).cuda()
    u = R.hashed_uniform("t.nce2.u", (16, 64), 1.0).cuda().requires_grad_(True)
    v = R.hashed_uniform("t.nce2.v", (16, 64), 1.0).cuda()
    total = 3.0 * ts.infonce(p1, p2, 0.07) + 0.5 * ts.cross_entropy(x, lab) + 0.1 * ts.mse(u, v)
    total.backward()
    q1, q2, y, w = (t.detach().clone().requires_grad_(True) for t in (p1, p2, x, u))
    F = torch.nn.functional
    z1, z2 = F.normalize(q1, dim=1), F.normalize(q2, dim=1)
    labels = torch.arange(16, device="cuda")
    ref = 3.0 * 0.5 * (F.cross_entropy(z1 @ z2.t() / 0.07, labels) + F.cross_entropy(z2 @ z1.t() / 0.07, labels)) \
        + 0.5 * F.cross_entropy(y, lab) + 0.1 * F.mse_loss(w, v)
    ref.backward()
    assert abs(total.item() - ref.item()) < 1e-4
    for got, want in ((p1.grad, q1.grad), (p2.grad, q2.grad), (x.grad, y.grad), (u.grad, w.grad)):
        assert (got - want).abs().max().item() <= 1e-4 * want.abs().max().item() + 1e-8
