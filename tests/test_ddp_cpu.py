"""CPU, world_size 2 over gloo: the data-parallel pieces (batch sharding, bucket reducer, gathered
InfoNCE term) give the same gradients as one rank on the concatenated batch.

The encoder engine itself needs the GPU; here the per-rank compute is the CPU oracle (test use only) on a
tiny encoder, and the pieces under test are vit_amd.ddp's sharding / reduction semantics.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vit_amd.ddp import BucketReducer, all_gather_with_grad, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_partitions_every_batch():
    for gb in (1, 2, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _tiny_model_loss(params, x):
    """A stand-in with the same reduction structure as the stage-1 objective: a per-clip mean loss plus an
    in-batch contrastive term over projections."""
    w1, w2 = params
    feats = torch.tanh(x @ w1)                       # (B, 16)
    per_clip = (feats ** 2).mean()
    return feats, per_clip, w2


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        gb = 6
        x_all = torch.randn(gb, 8)
        w1 = torch.randn(8, 16, requires_grad=True)
        w2 = torch.randn(16, 4, requires_grad=True)

        # ---- single-process result on the whole batch
        feats = torch.tanh(x_all @ w1)
        z = torch.nn.functional.normalize(feats @ w2, dim=1)
        logits = z @ z.t() / 0.07
        ref_loss = (feats ** 2).mean() + torch.nn.functional.cross_entropy(logits, torch.arange(gb))
        ref_g1, ref_g2 = torch.autograd.grad(ref_loss, (w1, w2))

        # ---- this rank's shard; mean-type losses are averaged by the reducer, the gathered (global)
        #      term is identical on every rank and therefore scaled by world before the averaging
        b, e = shard_range(gb, rank, world)
        feats_l = torch.tanh(x_all[b:e] @ w1)
        z_l = torch.nn.functional.normalize(feats_l @ w2, dim=1)
        z_g = all_gather_with_grad(z_l)
        assert z_g.shape == (gb, 4)
        nce = torch.nn.functional.cross_entropy(z_g @ z_g.t() / 0.07, torch.arange(gb))
        loss = (feats_l ** 2).mean() + world * nce
        g1, g2 = torch.autograd.grad(loss, (w1, w2))

        # gradients travel as flat buckets, submitted in "reverse layer order"
        reducer = BucketReducer()
        flat2, flat1 = g2.reshape(-1).clone(), g1.reshape(-1).clone()
        reducer.submit(flat2)
        reducer.submit(flat1)
        reducer.finish()
        assert reducer.buckets_reduced == 2 and reducer.bytes_reduced == (flat1.numel() + flat2.numel()) * 4
        torch.testing.assert_close(flat1.view_as(g1), ref_g1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(flat2.view_as(g2), ref_g2, rtol=1e-5, atol=1e-6)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_equal_single_rank(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_reducer_is_a_noop_for_one_rank(tmp_path):
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        r = BucketReducer()
        t = torch.ones(5)
        r.submit(t)
        r.finish()
        assert torch.equal(t, torch.ones(5)) and r.buckets_reduced == 0
    finally:
        dist.destroy_process_group()


def _missing_grad_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = torch.nn.Parameter(torch.zeros(3))
        b = torch.nn.Parameter(torch.zeros(2, 2))
        frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
        a.grad = torch.full((3,), float(rank + 1))
        if rank == 0:                                    # rank 1 never produced a gradient for `b`
            b.grad = torch.full((2, 2), 4.0)
        reducer = BucketReducer()
        reducer.reduce_gradients([a, b, frozen])
        assert reducer.buckets_reduced == 1 and reducer.bytes_reduced == (3 + 4) * 4
        torch.testing.assert_close(a.grad, torch.full((3,), 1.5))
        torch.testing.assert_close(b.grad, torch.full((2, 2), 2.0))      # (4 + 0) / 2 on BOTH ranks
        assert frozen.grad is None
        open(os.path.join(tmpdir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_head_bucket_has_the_same_size_on_every_rank_when_a_gradient_is_missing(tmp_path):
    """A parameter without a gradient on one rank (an empty shard, a head that rank did not use) must not change the
    size of the bucket or skip the collective there: the ranks would disagree and hang.  Zeros stand in."""
    world = 2
    mp.spawn(_missing_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


# ---------------------------------------------------------------------------------------------------------------
# the whole stage-1 step (heads, gathered InfoNCE, clip, AdamW) under data parallelism
# ---------------------------------------------------------------------------------------------------------------
class _StandInModel(torch.nn.Module):
    """CPU stand-in for the world model with the surface `stage1_step` uses: `encode_video` (clips independent,
    as through the real encoder) and `encoder.tubelet_embed.num_temporal_patches`."""

    class _Enc(torch.nn.Module):
        def __init__(self, dim, tokens):
            super().__init__()
            self.proj = torch.nn.Linear(12, tokens * dim)
            self.tubelet_embed = type("T", (), {"num_temporal_patches": 2})()
            self.tokens, self.dim = tokens, dim
            self._grad_reducer = None

        def forward(self, video):
            return torch.tanh(self.proj(video.flatten(1))).view(video.shape[0], self.tokens, self.dim)

    def __init__(self, dim=32, tokens=4):
        super().__init__()
        self.encoder = self._Enc(dim, tokens)

    def encode_video(self, video):
        return self.encoder(video)


def _stage1_setup(seed):
    from vit_amd import train_step as ts
    from oracle import ref_cpu as R
    # the pooling kernels need a device; on CPU a torch restatement (mean / the oracle's future_masking + masked mean,
    # s1:449-460) stands in for them -- test only, what is under test here is the reduction semantics across ranks
    def pool_unmasked(tokens, Tp, ratio=0.5):
        masked, mask = R.future_masking(tokens, Tp, ratio)
        keep = ~mask
        return (masked * keep.unsqueeze(-1)).sum(dim=1) / keep.sum(dim=1, keepdim=True).clamp_min(1), tokens.mean(dim=1)
    ts._pool_tokens = lambda x: x.mean(dim=1)
    ts._pool_unmasked = pool_unmasked
    torch.manual_seed(seed)
    model = _StandInModel()
    heads = ts.Stage1Heads(dim=32).eval()              # eval: the classifier's nn.Dropout(0.1) is the identity
    return ts, model, heads


def _stage1_batch(gb):
    g = torch.Generator().manual_seed(7)
    return {"video": torch.randn(gb, 3, 2, 2, generator=g), "video_augmented": torch.randn(gb, 3, 2, 2, generator=g),
            "shuffled_video": torch.randn(gb, 3, 2, 2, generator=g), "order_label": torch.randint(0, 2, (gb,), generator=g)}


def _stage1_worker(rank, world, port, tmpdir, gb):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vit_amd.ddp import broadcast_parameters
        ts, model, heads = _stage1_setup(100 + rank)            # different on purpose: the broadcast makes them equal
        broadcast_parameters(model)
        broadcast_parameters(heads)
        params = list(model.parameters()) + list(heads.parameters())
        opt = torch.optim.SGD(params, lr=0.5)      # plain SGD: the step is proportional to the clip coefficient
        reducer = BucketReducer()
        full = _stage1_batch(gb)
        b, e = shard_range(gb, rank, world)
        shard = {k: v[b:e] for k, v in full.items()}
        for _ in range(2):
            loss, parts = ts.stage1_step(model, heads, opt, shard, max_norm=0.05, reducer=reducer)
        torch.save({"params": [p.detach().clone() for p in params], "loss": loss, "parts": [float(v) for v in parts]},
                   os.path.join(tmpdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gb", [6, 5])       # 5 clips over 2 ranks: shards of 3 and 2
def test_stage1_step_two_ranks_equal_one_rank(tmp_path, gb):
    """After two steps of `stage1_step` (heads + gathered InfoNCE + a global-norm clip that actually clips + SGD, whose update, unlike Adam's,
    scales with the clip coefficient)
    both ranks hold the parameters one rank gets on the concatenated batch: head gradients are averaged, the clip
    norm is the global one, unequal shards are weighted."""
    world = 2
    mp.spawn(_stage1_worker, args=(world, _free_port(), str(tmp_path), gb), nprocs=world, join=True)
    got = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(world)]

    ts, model, heads = _stage1_setup(100)                       # rank 0's initial weights = what was broadcast
    params = list(model.parameters()) + list(heads.parameters())
    opt = torch.optim.SGD(params, lr=0.5)
    full = _stage1_batch(gb)
    for _ in range(2):
        loss, parts = ts.stage1_step(model, heads, opt, full, max_norm=0.05)
    for a, b in zip(got[0]["params"], got[1]["params"]):
        assert torch.equal(a, b), "replicas drifted apart"
    for a, ref in zip(got[0]["params"], params):
        torch.testing.assert_close(a, ref.detach(), rtol=2e-4, atol=2e-6)
    # the logged contrastive term is the global one on every rank
    assert got[0]["parts"][1] == pytest.approx(float(parts[1]), rel=1e-4)
    assert got[1]["parts"][1] == pytest.approx(float(parts[1]), rel=1e-4)
