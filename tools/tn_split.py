import sys, torch
sys.path.insert(0, "/root/repo")
from vit_amd import ops
dev = torch.device("cuda:0")
M = 65536
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    row = []
    for split in (0, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
        row.append(f"{split}:{timeit(lambda: ops.gemm_tn_accum(dy, x, dw, split_k=split)):.3f}")
    print(name, " ".join(row), flush=True)
