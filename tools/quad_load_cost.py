"""Timing-only variants of the four-wave GEMM (diagnostic library, tools/gemm_nt_debug.hip; results are wrong by
construction): what the operand loads cost in the main loop.  0 real kernel, 1 no loads at all, 6 the same bytes fetched
as whole 128-byte lines into registers and dropped, 7 the same plus one ds_write_b128 per load into the ring slot."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _gemm_debug as G
lib = G.load()
dev = torch.device("cuda:0")
M = 65536
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, N, K in [("qkv", 3072, 1024), ("dfc1", 1024, 4096)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    line = []
    for dbg in (0, 1, 6, 7, 0):
        t = min(timed(lambda: G.gemm(lib, x, w, out, dbg=dbg)) for _ in range(2))
        line.append(f"variant {dbg}: {t:.4f} ms")
    print(name, " | ".join(line), flush=True)
