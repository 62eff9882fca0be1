"""Encoder forward+backward throughput at other configurations of the reference (not the headline one):
    python tools/config_bench.py --depth 40 --frames 64 --batch 8          # 1024/40/16, 64x256^2 (8192 tokens)
    python tools/config_bench.py --depth 12 --frames 16 --img 224 --batch 2 # the reference's own CPU-sized case
Prints clips/s and the model FLOP utilisation against the 2.5 PFLOP/s bf16 peak."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import surgical_world_model as swm

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=40)
ap.add_argument("--heads", type=int, default=16)
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--img", type=int, default=256)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
D = 1024
torch.manual_seed(42)
enc = swm.VideoViTEncoder(img_size=a.img, num_frames=a.frames, embed_dim=D, depth=a.depth, num_heads=a.heads).eval().cuda()
video = torch.randn(a.batch, 3, a.frames, a.img, a.img, device="cuda")
N = (a.frames // 2) * (a.img // 16) ** 2
per_tok = 2 * D * 3 * D + 2 * D * D + 4 * D * 4 * D + 4 * N * D
flops = 3.0 * (a.depth * N * per_tok + 2 * 1536 * D * N)
def step():
    enc.zero_grad(set_to_none=True)
    enc._engine.weights.invalidate()
    (enc(video) ** 2).mean().backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(f"depth {a.depth} heads {a.heads} tokens/clip {N} batch {a.batch}: {dt*1e3:.1f} ms/step, {a.batch/dt:.2f} clips/s, "
      f"{flops/1e12:.2f} TFLOP/clip, MFU {a.batch/dt*flops/2.5e15:.3f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
