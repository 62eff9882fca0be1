"""Correctness + timing of the four-wave GEMM variant (tile code 4256) against the eight-wave ring (256)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
def rnd(shape, scale=1.0, dtype=BF16):
    return (torch.randn(shape, device=dev) * scale).to(dtype)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
tiles = [int(a) for a in sys.argv[1:]] or [256, 4256]
# correctness on small / ragged shapes
for (M, N, K) in [(2048, 1024, 1024), (2048 + 17, 1024, 4096), (513, 3072, 128), (4096, 256, 384), (70000, 512, 320)]:
    x = rnd((M, K)); w = rnd((N, K), 1 / math.sqrt(K))
    ref = x.float() @ w.float().t()
    for tile in tiles:
        out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
        ops.gemm_nt(x, w, ops.EPI_BF16, out, tile=tile)
        err = ((out.float() - ref).norm() / ref.norm()).item()
        print(f"check M={M} N={N} K={K} tile{tile}: rel err {err:.2e}", flush=True)
        assert err < 4e-3 or tile > 4256
M, D = 65536, 1024
x = rnd((M, D)); x4 = rnd((M, 4 * D))
for name, (Nn, K) in {"qkv": (3 * D, D), "out": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}.items():
    w = rnd((Nn, K), 1 / math.sqrt(K)); a = x if K == D else x4
    out = torch.empty(M, Nn, dtype=BF16, device=dev)
    for tile in tiles:
        t = timeit(lambda: ops.gemm_nt(a, w, ops.EPI_BF16, out, tile=tile))
        print(f"gemm_nt {name} tile{tile}: {t*1e3:.3f} ms  {2*M*Nn*K/t/1e12:.1f} TF/s", flush=True)
