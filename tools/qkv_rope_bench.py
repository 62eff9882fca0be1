"""QKV projection at the ViT-L shape: plain GEMM, the rotation pass, and the GEMM with the rotation in its epilogue."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = "cuda"
B, grid = 32, (8, 16, 16)
M, K = B * 2048, 1024
x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(3072, K, device=dev) / 32).to(torch.bfloat16)
tables = ops.rope_tables_device(torch.rand(128), torch.rand(192), torch.rand(192), grid, 2.0, dev)
pos = ops.rope_pos_code(B, grid, dev)
out = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for rnd in range(2):
    a = t(lambda: ops.gemm_nt(x, w, ops.EPI_BF16, out))
    b = t(lambda: ops.rope3d_qk_(out, tables, grid))
    c = t(lambda: ops.gemm_nt_qkv_rope(x, w, out, tables, pos, grid))
    print(f"plain GEMM {a:.4f} ms + rotation pass {b:.4f} ms = {a + b:.4f} | fused {c:.4f} ms", flush=True)
