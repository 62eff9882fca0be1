"""Per-phase cycle stamps of the forward attention kernel (needs a build with -DVA_ATTN_PROFILE)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
B, N, H, D = 32, 2048, 16, 1024
M = B * N
qkv = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
o = torch.empty(M, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, N, device=dev)
for _ in range(3):
    ops.attn_fwd(qkv, B, N, H, o, lse)
torch.cuda.synchronize()
v = lse.view(B * H, N // 32, 32)[:, :, :3].double().mean(dim=(0, 1))
print("cycles per tile-wave (s_memtime ticks): wait+barrier %.0f  scores+max %.0f  softmax+PV %.0f  total %.0f" % (v[0], v[1], v[2], v.sum()))
