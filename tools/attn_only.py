"""Run attention fwd+bwd a few times at ViT-L shape (for counter collection)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
B, N, H, D = 32, 2048, 16, 1024
M = B * N
qkv = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
o = torch.empty(M, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, N, device=dev)
do = torch.randn(M, D, device=dev).to(torch.bfloat16); dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.attn_fwd(qkv, B, N, H, o, lse)
    ops.attn_bwd(qkv, o, do, lse, dqkv, B, N, H)
torch.cuda.synchronize()
