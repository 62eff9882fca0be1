"""Run one GEMM shape a few times (for counter collection). usage: gemm_only.py <qkv|out|fc1|fc2> <tile> [tn]"""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
shape = sys.argv[1]; tile = int(sys.argv[2]); mode = sys.argv[3] if len(sys.argv) > 3 else "nt"
M, D = 65536, 1024
N, K = {"qkv": (3072, 1024), "out": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}[shape]
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
if mode == "nt":
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(6):
        ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=tile)
else:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    for _ in range(6):
        ops.gemm_tn_accum(dy, x, dw)
torch.cuda.synchronize()
