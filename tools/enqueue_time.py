"""How far ahead of the GPU the host runs: time for the host to enqueue one forward+backward step of the headline
configuration against the time the device takes to execute it (host time << device time = no launch gaps)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import surgical_world_model as swm
dev = torch.device("cuda:0")
for depth, frames, img, B in ((24, 16, 256, 32), (12, 16, 224, 2)):
    enc = swm.VideoViTEncoder(img_size=img, num_frames=frames, embed_dim=1024, depth=depth, num_heads=16).eval().to(dev)
    video = torch.randn(B, 3, frames, img, img, device=dev)
    def step():
        enc.zero_grad(set_to_none=True)
        (enc(video) ** 2).mean().backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(5):
        t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    print(f"depth {depth} B {B}: host enqueue {min(host) * 1e3:.1f} ms, step {min(total) * 1e3:.1f} ms", flush=True)
    del enc, video
