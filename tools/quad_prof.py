"""Cycle split (waits+barriers vs stage bodies) of the four-wave GEMM, debug build only (tile code 4261)."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
M, D = 65536, 1024
for name, (Nn, K) in {"qkv": (3 * D, D), "fc2": (D, 4 * D)}.items():
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    out = torch.zeros(M, Nn, dtype=torch.bfloat16, device=dev)
    for _ in range(3): ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=4261)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=4261); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    f = out.view(torch.float32).flatten()[:256 * 16].reshape(256, 4, 4).cpu()
    tiles = (M // 256) * (Nn // 256); stages = tiles / 256 * (K // 32)
    print(f"{name}: {ms:.3f} ms; per stage: sync {f[:, :, 0].mean() / stages:.0f}  body {f[:, :, 1].mean() / stages:.0f} cycles; "
          f"epilogue per tile {f[:, :, 3].mean() / (tiles / 256):.0f}; tail {f[:, :, 2].mean():.0f}; total counted {(f[:, :, 0] + f[:, :, 1]).mean():.0f} cycles "
          f"-> {(f[:, :, 0] + f[:, :, 1]).mean() / ms / 1e6:.2f} GHz-equivalent")
