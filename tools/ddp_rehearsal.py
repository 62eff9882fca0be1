"""Two (or more) ranks sharing ONE GPU over gloo: exercises broadcast, the bucket reducer's stream hand-over and the
gradient agreement across ranks with real kernels when only one device is available (RCCL refuses ranks that share
a device).  Launch with torch.distributed.run.  Sizes come from DEPTH / IMG / FRAMES / BATCH.  Device buffers are
never handed to gloo: parameters and gradient buckets are staged through host memory by vit_amd.ddp.  A stall
dumps every thread's Python stack after STALL_SECONDS (faulthandler) and every bucket is logged, so a hang names
its frame and its bucket."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def say(*a):
    print(f"[rank {os.environ.get('RANK')}] {time.strftime('%H:%M:%S')}", *a, flush=True)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    faulthandler.dump_traceback_later(int(os.environ.get("STALL_SECONDS", "150")), exit=True)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    say("group up")
    from vit_amd import surgical_world_model as swm
    from vit_amd.ddp import BucketReducer
    depth = int(os.environ.get("DEPTH", "2"))
    img, frames, batch = (int(os.environ.get(k, d)) for k, d in (("IMG", "64"), ("FRAMES", "8"), ("BATCH", "4")))
    torch.manual_seed(42 + rank)          # different on purpose: the broadcast must make them equal
    enc = swm.VideoViTEncoder(img_size=img, num_frames=frames, embed_dim=1024, depth=depth, num_heads=16).eval().to(dev)
    say("model built")
    from vit_amd.ddp import broadcast_parameters
    broadcast_parameters(enc)
    say("broadcast done")
    reducer = BucketReducer(log=say if os.environ.get("BUCKET_LOG", "1") == "1" else None)
    enc.set_grad_reducer(reducer)
    video = torch.randn(batch, 3, frames, img, img, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    for step in range(2):
        enc.zero_grad(set_to_none=True)
        y = enc(video)
        (y ** 2).mean().backward()
        torch.cuda.synchronize()
        say("step", step, "buckets", reducer.buckets_reduced)
    # every rank must now hold the same (averaged) gradients
    digest = torch.stack([p.grad.double().abs().sum() for p in enc.parameters()]).sum().reshape(1).cpu()
    both = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    say("digests", [float(b) for b in both])
    assert all(torch.equal(b, both[0]) for b in both), "ranks disagree on the averaged gradients"
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    say("ok")


if __name__ == "__main__":
    main()
