"""Cycle split (waits+barriers / stage bodies / epilogues) of the four-wave GEMM, debug build only (tile code 4261),
for several numbers of workgroups (VA_QUAD_GRID, honoured by the debug build): does a tile's epilogue take as long
when few CUs store at the same time?"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
M, D = 65536, 1024
for name, (Nn, K) in {"qkv": (3 * D, D), "fc2": (D, 4 * D)}.items():
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    out = torch.zeros(M, Nn, dtype=torch.bfloat16, device=dev)
    for grid in (256, 128, 64, 32, 8):
        os.environ["VA_QUAD_GRID"] = str(grid)
        for _ in range(2): ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=4261)
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=4261); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        f = out.view(torch.float32).flatten()[:grid * 16].reshape(grid, 4, 4).cpu()
        tiles = (M // 256) * (Nn // 256); per_wg = tiles / grid; stages = per_wg * (K // 32)
        total = (f[:, :, 0] + f[:, :, 1] + f[:, :, 3]).mean()
        print(f"{name} grid {grid:3d}: {ms:.3f} ms; per stage: sync {f[:, :, 0].mean() / stages:.0f}  body {f[:, :, 1].mean() / stages:.0f} cycles; "
              f"epilogue per tile {f[:, :, 3].mean() / per_wg:.0f}; total counted {total:.0f} cycles -> {total / ms / 1e6:.2f} GHz-equivalent",
              flush=True)
