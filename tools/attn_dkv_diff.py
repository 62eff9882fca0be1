"""Where the pipelined dK/dV pass differs from the generic one (debugging aid): error pattern by key position / column."""
import os, sys, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib
lib = _lib.load()
B, N, H, D = int(os.environ.get("B", 2)), int(os.environ.get("N", 512)), 16, 1024
hd, M = D // H, B * N
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
do = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
o = torch.empty(M, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, N, device=dev)
ws = torch.empty(lib.va_attn_bwd_workspace_bytes(B, N, H) // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = qkv.data_ptr()
assert lib.va_attn_fwd(p, p + 2 * D, p + 4 * D, o.data_ptr(), lse.data_ptr(), B, N, H, hd, 3 * D, D, st) == 0
out = torch.zeros_like(qkv)
d = out.data_ptr()
assert lib.va_attn_bwd(p, p + 2 * D, p + 4 * D, o.data_ptr(), do.data_ptr(), lse.data_ptr(), ws.data_ptr(), d, d + 2 * D, d + 4 * D,
                       B, N, H, hd, 3 * D, D, 3 * D, st) == 0
torch.cuda.synchronize()
torch.save(out.cpu(), os.environ["OUT"])
