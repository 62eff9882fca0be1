"""Delta-rule recurrence kernel at the rollout size (B*H = 512 heads of 128, 2064 steps): time per library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
B, L, H, D = 64, 2064, 8, 128
g = torch.Generator().manual_seed(3)
q = torch.nn.functional.normalize(torch.randn(B, L, H * D, generator=g), dim=-1).view(B, L, H, D).to(dev)
k = torch.nn.functional.normalize(torch.randn(B, L, H * D, generator=g), dim=-1).view(B, L, H, D).to(dev)
v = torch.randn(B, L, H, D, generator=g).to(dev)
alpha = torch.sigmoid(torch.randn(B, L, H, generator=g)).to(dev); beta = torch.sigmoid(torch.randn(B, L, H, generator=g)).to(dev)
for _ in range(2): out, st = ops.gated_delta_seq(q, k, v, alpha, beta, None)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): out, st = ops.gated_delta_seq(q, k, v, alpha, beta, None)
e.record(); torch.cuda.synchronize()
print(f"{os.environ.get('VIT_AMD_LIB', 'current')}: {s.elapsed_time(e) / 5:.3f} ms  out checksum {float(out.double().abs().sum()):.6e} state {float(st.double().abs().sum()):.6e}")
