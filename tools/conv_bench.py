"""Times the causal conv + SiLU kernels at the rollout's shape (64 sequences x 2451 steps x 1024 channels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
B, L, C = 64, 2451, 1024
dev = "cuda"
wide = torch.randn(B * L, 5 * C, device=dev).to(torch.bfloat16)
w = torch.randn(C, 1, 4, device=dev) * 0.5; b = torch.randn(C, device=dev) * 0.1
x32 = wide[:, :C].float().contiguous().view(B, L, C)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for norm in (True, False):
    us16 = t(lambda: ops.dwconv1d_causal_silu_bf16(wide[:, C:2 * C], B, L, w, b, norm))
    us32 = t(lambda: ops.dwconv1d_causal_silu(x32, w, b, norm))
    print(f"normalize={norm}: bf16 column block {us16:.0f} us ({B * L * C * 6 / us16 / 1e6:.2f} TB/s), f32 {us32:.0f} us ({B * L * C * 8 / us32 / 1e6:.2f} TB/s)")
