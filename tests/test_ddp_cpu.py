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
