"""Times va_attn_fwd at the ViT-L shape (B = 32, N = 2048, 16 heads of 64) and reports its distance from an f32 softmax
attention on one head-batch; with VIT_AMD_LIB pointing at another build of the library this is the A/B of the kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
B, N, H, D = 32, 2048, 16, 1024
M = B * N
qkv = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
o = torch.empty(M, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, N, device=dev)
ops.attn_fwd(qkv, B, N, H, o, lse)
q = qkv[:N, 0:64].float(); k = qkv[:N, 1024:1088].float(); v = qkv[:N, 2048:2112].float()
sc = (q @ k.t()) / 8.0
ref = torch.softmax(sc, -1) @ v
lse_ref = torch.logsumexp(sc, -1) * 1.4426950408889634
print("rel err o", ((o[:N, :64].float() - ref).norm() / ref.norm()).item(), "lse abs err", (lse[0] - lse_ref).abs().max().item())
for rnd in range(3):
    for _ in range(5): ops.attn_fwd(qkv, B, N, H, o, lse)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): ops.attn_fwd(qkv, B, N, H, o, lse)
    e.record(); torch.cuda.synchronize()
    print(os.environ.get("VIT_AMD_LIB", "in-tree build"), "fwd", round(s.elapsed_time(e) / 50, 4), "ms", flush=True)
