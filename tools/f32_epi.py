"""f32-output GEMM (N = K = 1024) with and without bias / residual on the 256-tile kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for M in (65536, 132096):
    D = 1024
    x = torch.randn(M, D, device=dev).to(BF16); w = (torch.randn(D, D, device=dev) / 32).to(BF16)
    bias = torch.zeros(D, device=dev); resid = torch.randn(M, D, device=dev); out = torch.empty(M, D, device=dev); outb = torch.empty(M, D, dtype=BF16, device=dev)
    for tile in (256, 2256, 4256):
        print(f"M={M} tile{tile}: bf16 {timeit(lambda: ops.gemm_nt(x, w, ops.EPI_BF16, outb, tile=tile)):.3f}  f32 {timeit(lambda: ops.gemm_nt(x, w, ops.EPI_F32, out, tile=tile)):.3f}  "
              f"f32+bias {timeit(lambda: ops.gemm_nt(x, w, ops.EPI_F32, out, bias=bias, tile=tile)):.3f}  f32+resid {timeit(lambda: ops.gemm_nt(x, w, ops.EPI_F32, out, resid=resid, tile=tile)):.3f}  "
              f"f32+bias+resid {timeit(lambda: ops.gemm_nt(x, w, ops.EPI_F32, out, bias=bias, resid=resid, tile=tile)):.3f} ms", flush=True)
