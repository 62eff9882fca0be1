"""Calibration only: the vendor GEMM library (through torch.mm) at the encoder's GEMM shapes."""
import torch, time
dev = torch.device("cuda:0")
M = 65536
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, N, K in [("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ms = bench(lambda: torch.mm(x, w.t()))
    print(f"NT {name}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s")
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    ms = bench(lambda: torch.mm(dy.t(), x))
    print(f"TN {name}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s")
