"""Delta-rule recurrence kernel: time against the number of (batch, head) workgroups, heads of 128, 2064 steps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
L, H, D = 2064, 8, 128
for B in (4, 8, 16, 24, 32, 48, 64, 96, 128):
    g = torch.Generator().manual_seed(3)
    q = torch.nn.functional.normalize(torch.randn(B, L, H, D, generator=g), dim=-1).to(dev)
    k = torch.nn.functional.normalize(torch.randn(B, L, H, D, generator=g), dim=-1).to(dev)
    v = torch.randn(B, L, H, D, generator=g).to(dev)
    alpha = torch.sigmoid(torch.randn(B, L, H, generator=g)).to(dev); beta = torch.sigmoid(torch.randn(B, L, H, generator=g)).to(dev)
    for _ in range(2): out, st = ops.gated_delta_seq(q, k, v, alpha, beta, None)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): out, st = ops.gated_delta_seq(q, k, v, alpha, beta, None)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"{os.environ.get('VIT_AMD_LIB', 'current')}: workgroups {B*H:5d}: {ms:.3f} ms  {ms*1e3/L:.3f} us/step", flush=True)
