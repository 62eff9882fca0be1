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
