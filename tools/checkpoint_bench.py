"""Activation checkpointing (use_grad_checkpoint, swm:807-811) at size: step time and peak memory with every
activation kept against block inputs only, ViT-L 16x256^2 B=32; then 1024/40/16 on 64x256^2 clips (8192 tokens) at
batch sizes whose activations do not fit without it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import surgical_world_model as swm
dev = torch.device("cuda:0")

def run(depth, frames, B, policy, steps=3):
    enc = swm.VideoViTEncoder(img_size=256, num_frames=frames, embed_dim=1024, depth=depth, num_heads=16,
                              use_grad_checkpoint=True).to(dev).train()
    for m in enc.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    enc.checkpoint_policy = policy
    video = torch.randn(B, 3, frames, 256, 256, device=dev)
    used = enc._checkpointing(video)
    def step():
        enc.zero_grad(set_to_none=True)
        (enc(video) ** 2).mean().backward()
    step(); torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"depth {depth} frames {frames} B {B} policy {policy:6s} (recomputes: {used}): {dt * 1e3:8.1f} ms/step  {B / dt:7.1f} clips/s  "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:6.1f} GiB", flush=True)
    del enc, video
    torch.cuda.empty_cache()

run(24, 16, 32, "never"); run(24, 16, 32, "always"); run(24, 16, 32, "auto")
run(40, 64, 16, "never"); run(40, 64, 16, "always")
run(40, 64, 32, "auto"); run(40, 64, 48, "auto")
