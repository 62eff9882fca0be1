"""Attention forward at the rollout's predictor shape (B = 64, 8 heads of 128, 2064 tokens = 8 steps of 258) and at the
encoder's (16 heads of 64), with and without the block-causal mask: time and matrix-pipe rate."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, N, H, tpt) in ((64, 2064, 8, 0), (64, 2064, 8, 258), (64, 2064, 16, 0), (64, 2064, 16, 258), (64, 2048, 8, 0), (64, 2048, 16, 0)):
    D = 1024
    qkv = torch.randn(B * N, 3 * D, device=dev).to(torch.bfloat16)
    o = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, N, device=dev)
    ms = timeit(lambda: ops.attn_fwd(qkv, B, N, H, o, lse, tokens_per_timestep=tpt))
    frac = 1.0
    if tpt:
        steps = N // tpt
        frac = (steps + 1) / (2 * steps)
    fl = 4.0 * N * N * (D // H) * B * H * frac
    print(f"B={B} N={N} heads={H} (hd {D//H}) block-causal period {tpt}: {ms:.3f} ms  {fl/ms/1e9:.0f} TFLOP/s of unmasked work", flush=True)
