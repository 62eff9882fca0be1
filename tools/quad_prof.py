"""Where the four-wave GEMM's cycles go and what clock it runs at, from the diagnostic library (tools/gemm_nt_debug.hip,
variant 5: correct results plus s_memtime / s_memrealtime stamps), for several numbers of workgroups.

Per (shape, grid) it prints, medians over the waves of all workgroups:
  * cycles per BK-32 stage split into waits+barriers and stage bodies, cycles per epilogue,
  * the in-kernel shader clock = cycles / 100-MHz ticks over the SAME span (first stage .. last store) -- no wall-clock
    time and no un-stamped span enters this figure,
  * `counted / wall`: the stamped cycles of a wave divided by the kernel's wall time (HIP events) -- this one is lowered by
    everything the stamps do not cover (launch, prologue, waiting for the slowest workgroup) without the clock moving,
  * the share of the kernel's wall time a median wave spends between its first stage and its last store.
Run under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES` to get the counter-derived clock of
the very same launches (GRBM_GUI_ACTIVE / 8 / duration)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gemm_debug as G
lib = G.load()
dev = torch.device("cuda:0")
M, D = 65536, 1024
grids = [int(g) for g in sys.argv[1:]] or [256, 128, 64, 32, 8]
for name, (Nn, K) in {"qkv": (3 * D, D), "fc2": (D, 4 * D)}.items():
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    out = torch.zeros(M, Nn, dtype=torch.bfloat16, device=dev)
    ref = torch.zeros(M, Nn, dtype=torch.bfloat16, device=dev)
    G.gemm(lib, x, w, ref, dbg=0)
    for grid in grids:
        stamps = torch.zeros(grid * 4 * 8, dtype=torch.float32, device=dev)
        # settle the clock: ~1.5 s of back-to-back launches of the same variant on the same data
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        G.gemm(lib, x, w, out, dbg=5, grid_limit=grid, stamps=stamps); torch.cuda.synchronize()
        s.record(); G.gemm(lib, x, w, out, dbg=5, grid_limit=grid, stamps=stamps); e.record(); torch.cuda.synchronize()
        reps = max(3, int(1500.0 / max(s.elapsed_time(e), 0.05)))
        for _ in range(reps): G.gemm(lib, x, w, out, dbg=5, grid_limit=grid, stamps=stamps)
        s.record(); G.gemm(lib, x, w, out, dbg=5, grid_limit=grid, stamps=stamps); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        assert torch.equal(out, ref), "the stamped variant must compute the same result"
        f = stamps.reshape(grid, 4, 8).cpu().double()
        tiles = (M // 256) * (Nn // 256); per_wg = tiles / grid; stages = per_wg * (K // 32)
        sync, body, tail, epi, span, ticks, prologue = (f[:, :, i].flatten() for i in range(7))
        clock = (span / ticks * 0.1).median().item()                       # GHz
        counted = (sync + body + epi).median().item()
        print(f"{name} grid {grid:3d}: {ms:.3f} ms | per stage: waits {sync.median().item() / stages:.0f} + body {body.median().item() / stages:.0f} cycles"
              f" | epilogue {epi.median().item() / per_wg:.0f} per tile | prologue {prologue.median().item():.0f}"
              f" | in-kernel clock {clock:.3f} GHz (min {float((span / ticks * 0.1).min()):.3f}, max {float((span / ticks * 0.1).max()):.3f})"
              f" | counted/wall {counted / ms / 1e6:.3f} GHz | span/wall {(ticks.median().item() * 1e-5) / ms:.3f}"
              f" (slowest wave {(ticks.max().item() * 1e-5) / ms:.3f})", flush=True)
