"""The bf16-output GEMM kernels side by side at M = 65536 (tile codes: 4256 four-wave persistent BK 32 ring, 256 eight-wave
BK 32 ring with paired loads, 2256 the same unpaired, 1256 eight-wave BK 64 double buffer)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
M = 65536
for name, N, K in [("qkv", 3072, 1024), ("dfc1", 1024, 4096), ("fc1-shape", 4096, 1024)]:
    x = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF16)
    out = torch.zeros(M, N, dtype=BF16, device=dev)
    line = []
    for tile in (4256, 256, 2256, 1256):
        t = min(timed(lambda: ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=tile)) for _ in range(2))
        line.append(f"{tile}: {t:.4f} ms ({2.0 * M * N * K / t / 1e9:.0f} TF)")
    print(name, " | ".join(line), flush=True)
