"""GPU: the gradient reducer's stream hand-over with the RCCL backend, in a world of one process (the collectives
run, on the side stream, and must leave the gradients of a normal backward unchanged)."""
import os

import pytest
import torch
import torch.distributed as dist

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def test_bucket_reducer_on_rccl_world_of_one():
    from vit_amd import surgical_world_model as swm
    from vit_amd.ddp import BucketReducer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        enc = swm.VideoViTEncoder(img_size=32, num_frames=4, embed_dim=1024, depth=2, num_heads=16).eval()
        enc.load_state_dict(R.fill_state_dict_(enc.state_dict()))
        enc = enc.cuda()
        video = R.hashed_uniform("ddp.video", (2, 3, 4, 32, 32), 1.0).cuda()
        enc(video).pow(2).mean().backward()
        plain = [p.grad.clone() for p in enc.parameters()]
        enc.zero_grad(set_to_none=True)
        reducer = BucketReducer(always_reduce=True)
        enc.set_grad_reducer(reducer)
        enc(video).pow(2).mean().backward()
        torch.cuda.synchronize()
        assert reducer.buckets_reduced == 2 + 2           # two blocks, final norm, stem
        assert reducer.bytes_reduced == 4 * sum(p.numel() for p in enc.parameters())
        for a, p in zip(plain, enc.parameters()):
            assert torch.equal(a, p.grad)
    finally:
        dist.destroy_process_group()


def test_broadcast_after_a_forward_drops_every_cached_operand_copy():
    """`broadcast_parameters` writes through `.data`, which does not bump the version counters the bf16 operand
    caches key on.  A forward that ran BEFORE the broadcast (a warm-up, a re-sync after loading a checkpoint) must
    not leave stale copies in use: after the broadcast of the whole world model the encoder and the predictor blocks
    must compute with the new parameters.  World of one: the parameters are rewritten the way a broadcast from
    another rank rewrites them, then the real broadcast (and its invalidation) runs."""
    from vit_amd import surgical_world_model as swm
    from vit_amd.ddp import broadcast_parameters
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        def build(seed):
            torch.manual_seed(seed)
            return swm.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                                           predictor_depth=2, predictor_heads=8).eval().cuda()
        model, other = build(1), build(2)
        video = R.hashed_uniform("ddp.bc.video", (2, 3, 4, 32, 32), 1.0).cuda()
        actions = R.hashed_uniform("ddp.bc.a", (2, 4, 7), 1.0).cuda()
        states = R.hashed_uniform("ddp.bc.s", (2, 4, 7), 1.0).cuda()

        def run(m):
            with torch.no_grad():
                preds, cur, fut = m(video, actions, states, predict_horizon=1)
            return torch.cat([m.encode_video(video).flatten(), preds[0]["features"].flatten(), cur.flatten(), fut.flatten()])

        stale = run(model)                                  # fills every operand cache from the seed-1 weights
        want = run(other)
        assert not torch.allclose(stale, want)
        with torch.no_grad():                               # what receiving a broadcast does: raw writes
            for dst, src in zip(list(model.parameters()) + list(model.buffers()),
                                list(other.parameters()) + list(other.buffers())):
                dst.data.copy_(src.data)
        broadcast_parameters(model)
        assert torch.equal(run(model), want)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# the whole stage-1 step with the real encoder on two ranks that share this GPU (gloo: vit_amd.ddp stages device
# buffers through host memory; RCCL refuses ranks that share a device) against one rank on the concatenated batch
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model_and_heads(seed):
    from vit_amd import surgical_world_model as swm
    from vit_amd import train_step as ts
    torch.manual_seed(seed)
    model = swm.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=2, encoder_heads=16,
                                                    predictor_depth=1, predictor_heads=8).eval()
    heads = ts.Stage1Heads(1024).eval()                  # eval: every nn.Dropout is the identity
    return ts, model.cuda(), heads.cuda()


def _batch(gb):
    return {"video": R.hashed_uniform("ddp.s1.video", (gb, 3, 4, 32, 32), 1.0),
            "video_augmented": R.hashed_uniform("ddp.s1.aug", (gb, 3, 4, 32, 32), 1.0),
            "shuffled_video": R.hashed_uniform("ddp.s1.shuf", (gb, 3, 4, 32, 32), 1.0),
            "order_label": torch.tensor([1, 0, 0, 1, 1, 0][:gb])}


def _trained(model, heads):
    return list(model.encoder.parameters()) + list(heads.parameters())


def _stage1_rank(rank, world, port, tmpdir, gb):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vit_amd.ddp import BucketReducer, broadcast_parameters, shard_range
        ts, model, heads = _model_and_heads(100 + rank)      # different on purpose: the broadcast makes them equal
        broadcast_parameters(model)
        broadcast_parameters(heads)
        params = _trained(model, heads)
        opt = torch.optim.SGD(params, lr=0.05)               # SGD: the step scales with the clip coefficient
        reducer = BucketReducer()
        model.encoder.set_grad_reducer(reducer)              # encoder buckets from inside backward; heads at the join
        b, e = shard_range(gb, rank, world)
        shard = {k: v[b:e].cuda() for k, v in _batch(gb).items()}
        snaps = []
        for _ in range(2):
            loss, parts = ts.stage1_step(model, heads, opt, shard, max_norm=0.5, reducer=reducer)
            snaps.append([p.detach().cpu() for p in params])
        torch.cuda.synchronize()
        torch.save({"after1": snaps[0], "after2": snaps[1], "parts": [float(v) for v in parts],
                    "buckets": reducer.buckets_reduced}, os.path.join(tmpdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gb", [4, 5])       # 5 clips over 2 ranks: shards of 3 and 2
def test_stage1_step_two_ranks_on_one_gpu_equal_one_rank(tmp_path, gb):
    """Two optimisation steps (encoder buckets averaged inside backward, heads as one more bucket, gathered InfoNCE,
    a global-norm clip that clips, SGD): both ranks end with the SAME parameters.  After the first step they are the
    parameters one rank reaches on the concatenated batch up to f32 summation order (1e-5 of each tensor's norm; for
    the zero-initialised LayerNorm biases that is 1e-5 of the update itself).  After the second step only 5e-3 can
    be asked: the 1e-7 differences of step one flip the bf16 rounding of a few hundred operand copies of the weights,
    and sums that cancel (a bias gradient over 96 rows) amplify that."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_stage1_rank, args=(world, _free_port(), str(tmp_path), gb), nprocs=world, join=True)
    got = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(world)]
    assert got[0]["buckets"] == 2 * (2 + 2 + 1)           # per step: two blocks, final norm, stem, heads
    for a, b in zip(got[0]["after2"], got[1]["after2"]):
        assert torch.equal(a, b), "replicas drifted apart"
    ts, model, heads = _model_and_heads(100)
    params = _trained(model, heads)
    opt = torch.optim.SGD(params, lr=0.05)
    full = {k: v.cuda() for k, v in _batch(gb).items()}
    names = ["encoder." + k for k, _ in model.encoder.named_parameters()] + ["heads." + k for k, _ in heads.named_parameters()]
    for key, tol in (("after1", 1e-5), ("after2", 5e-3)):
        loss, parts = ts.stage1_step(model, heads, opt, full, max_norm=0.5)
        errs = sorted(((((a - ref.detach().cpu()).norm() / ref.detach().cpu().norm().clamp_min(1e-12)).item(), n)
                       for a, ref, n in zip(got[0][key], params, names)), reverse=True)
        assert errs[0][0] < tol, (key, errs[:6])
    assert got[0]["parts"][1] == pytest.approx(float(parts[1]), rel=2e-3)
