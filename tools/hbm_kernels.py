"""Launch the HBM-bound kernels of the JEPA-side path at ViT-L sizes (for rocprofv3): EMA over all 303.8 M encoder
parameters, token gather / scatter over a working set beyond the 256 MiB Infinity Cache, L1 latent loss, future mask.
Prints each kernel's algorithmic bytes per launch as JSON so achieved GB/s = bytes / rocprof's average duration."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
out = {}
ITERS = 5

# EMA: target, online f32, 303.8 M parameters (ViT-L encoder); read 2 x 4 B, write 4 B per parameter
n = 303_800_000 // 8 * 8
tgt, onl = torch.randn(n, device=dev), torch.randn(n, device=dev)
for _ in range(ITERS):
    ops.ema_update_(tgt, onl, 0.996)
out["ema_kernel"] = 12 * n
del tgt, onl

# token gather / scatter: 64 clips x 8192 tokens x 1024 f32 (2.1 GB source), half of the tokens kept
B, N, D, keep = 64, 8192, 1024, 4096
x = torch.randn(B, N, D, device=dev)
idx = torch.stack([torch.randperm(N)[:keep].sort().values for _ in range(B)]).int().to(dev)
for _ in range(ITERS):
    g = ops.token_gather(x, idx)
out["token_move_kernel<false> (gather)"] = 2 * B * keep * D * 4 + 4 * B * keep
for _ in range(ITERS):
    s = ops.token_scatter(g, idx, N)
out["token_move_kernel<true> (scatter; the zero-fill of the output is a separate memset)"] = 2 * B * keep * D * 4 + 4 * B * keep
del x, g, s

# L1 latent loss: predictions and targets (B=32 clips x 2048 target tokens x 1024) bf16, gradient bf16
B, Nt = 32, 2048
p = torch.randn(B, Nt, D, device=dev).to(torch.bfloat16)
t = torch.randn(B, Nt, D, device=dev).to(torch.bfloat16)
for _ in range(ITERS):
    ops.l1_loss(p, t)
out["l1_kernel<true,true>"] = 3 * B * Nt * D * 2
del p, t

# future mask: (32, 2048, 1024) f32 features, last half of the timesteps zeroed
f = torch.randn(32, 2048, 1024, device=dev)
for _ in range(ITERS):
    ops.future_mask(f, 1024)
out["future_mask_kernel (masked rows are not read)"] = f.numel() * 4 * 3 // 2 + 32 * 2048
torch.cuda.synchronize()
print(json.dumps(out))
