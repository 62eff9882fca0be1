"""Causal conv + SiLU kernel at the rollout size (B*L = 64*2064 rows of 1024 channels), normalised and not: time per library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
B, L, C = 64, 2064, 1024
x = torch.randn(B, L, C, device=dev); w = torch.randn(C, 1, 4, device=dev) * 0.5; b = torch.randn(C, device=dev) * 0.1
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for norm in (False, True):
    ms = timeit(lambda: ops.dwconv1d_causal_silu(x, w, b, norm))
    out = ops.dwconv1d_causal_silu(x, w, b, norm)
    print(f"{os.environ.get('VIT_AMD_LIB', 'current')}: normalise {norm}: {ms*1e3:.1f} us  {2*x.numel()*4/ms/1e9:.2f} TB/s  checksum {float(out.double().sum()):.6e}", flush=True)
