"""Times the weight-gradient GEMM (va_gemm_tn_accum) on the encoder step's four shapes at B = 32 and checks it against an
f32 matmul.  VA_GEMM_TN_WAVES=8 in the environment selects the eight-wave form of the large-tile kernel, so two runs of
this script in one gpurun call are the A/B."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, D = int(os.environ.get("AB_M", "65536")), 1024
g = torch.Generator(device="cuda").manual_seed(0)
tag = os.environ.get("VA_GEMM_TN_WAVES", "4")
for name, N, K, bias in [("qkv", 3 * D, D, False), ("out", D, D, False), ("fc1", 4 * D, D, True), ("fc2", D, 4 * D, False),
                         ("fc2+b", D, 4 * D, True)]:
    dy = (torch.randn(M, N, device=dev, generator=g) / math.sqrt(M)).to(torch.bfloat16)
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev) if bias else None
    ops.gemm_tn_accum(dy, x, dw, db)
    ref = dy.float().t() @ x.float()
    err = ((dw - ref).norm() / ref.norm()).item()
    berr = ((db - dy.float().sum(0)).norm() / dy.float().sum(0).norm()).item() if bias else 0.0
    best = 1e9
    for rnd in range(3):
        for _ in range(5):
            ops.gemm_tn_accum(dy, x, dw, db)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.gemm_tn_accum(dy, x, dw, db)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    print(f"waves={tag} {name:6s} {N:5d}x{K:5d}: {best:.4f} ms ({2.0 * M * N * K / best / 1e9:.0f} TF)  rel err {err:.2e} bias {berr:.2e}",
          flush=True)
