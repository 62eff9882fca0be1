"""What a tile boundary of the four-wave GEMM costs, taken apart with the stamped diagnostic variants
(tools/gemm_nt_debug.hip): 5 = the shipped kernel + stamps, 8 = epilogue without its stores (timing only), 9 = tiles
handed out statically (no ticket atomics), 10 = epilogue that stores zeros without reading the accumulators (timing
only).  Medians over all waves, cycles per tile; the `wave 0` column is the wave that draws the ticket.
usage: quad_epi_prof.py [shape ...]   shapes: qkv out fc1 fc2"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gemm_debug as G
lib = G.load()
dev = torch.device("cuda:0")
M, D = 65536, 1024
SHAPES = {"qkv": (3 * D, D), "out": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}
grid = 256
for name in (sys.argv[1:] or ["qkv", "fc2"]):
    Nn, K = SHAPES[name]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    out = torch.zeros(M, Nn, dtype=torch.bfloat16, device=dev)
    for dbg in [int(v) for v in os.environ.get("EPI_VARIANTS", "5,9,8,10,11,12,13,14,15,16,17").split(",")]:
        stamps = torch.zeros(grid * 4 * 8, dtype=torch.float32, device=dev)
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        for _ in range(300): G.gemm(lib, x, w, out, dbg=dbg, grid_limit=grid, stamps=stamps)
        s.record()
        for _ in range(20): G.gemm(lib, x, w, out, dbg=dbg, grid_limit=grid, stamps=stamps)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        f = stamps.reshape(grid, 4, 8).cpu().double()
        tiles = (M // 256) * (Nn // 256); per_wg = tiles / grid; stages = per_wg * (K // 32)
        med = lambda t: t.flatten().median().item()
        print(f"{name} variant {dbg:2d}: {ms:.4f} ms | stage: waits {med(f[:, :, 0]) / stages:.0f} + body {med(f[:, :, 1]) / stages:.0f}"
              f" | epilogue {med(f[:, :, 3]) / per_wg:.0f} (wave 0: {med(f[:, 0, 3]) / per_wg:.0f}, waves 1-3: {med(f[:, 1:, 3]) / per_wg:.0f})"
              f" | hand-over {med(f[:, :, 7]) / per_wg:.0f} (wave 0: {med(f[:, 0, 7]) / per_wg:.0f})"
              f" | prologue {med(f[:, :, 6]):.0f} | clock {med(f[:, :, 4] / f[:, :, 5] * 0.1):.3f} GHz", flush=True)
