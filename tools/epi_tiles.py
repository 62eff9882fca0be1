"""GELU-epilogue GEMMs at ViT-L shapes on the 128-tile kernel (two workgroups per CU) vs the 256-tile kernels."""
import math, os, sys
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
M, D = 65536, 1024
x = torch.randn(M, D, device=dev).to(BF16); w = (torch.randn(4 * D, D, device=dev) / 32).to(BF16); bias = torch.zeros(4 * D, device=dev)
pre = torch.empty(M, 4 * D, dtype=BF16, device=dev); act = torch.empty_like(pre)
for tile in (128, 256, 1256, 4256):
    print(f"fc1+bias+gelu tile{tile}: {timeit(lambda: ops.gemm_nt(x, w, ops.EPI_BIAS_GELU, pre, out2=act, bias=bias, tile=tile)):.3f} ms", flush=True)
    print(f"dgrad+gelu_bwd tile{tile}: {timeit(lambda: ops.gemm_nt(x, w, ops.EPI_GELU_BWD, act, aux=pre, tile=tile)):.3f} ms", flush=True)
    print(f"plain bf16     tile{tile}: {timeit(lambda: ops.gemm_nt(x, w, ops.EPI_BF16, act, tile=tile)):.3f} ms", flush=True)
