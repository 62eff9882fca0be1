"""Per-kernel timing at ViT-L shapes (B=32 clips of 2048 tokens). Dev tool; prints one line per kernel."""
import math, sys, json
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vit_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def rnd(shape, dtype=BF16, scale=1.0):
    return (torch.randn(shape, device=dev) * scale).to(dtype)

B, N, D, H = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 2048, 1024, 16
M = B * N
res = {}
x = rnd((M, D)); x4 = rnd((M, 4 * D))
for name, (Nn, K) in {"qkv": (3 * D, D), "out": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}.items():
    w = rnd((Nn, K), scale=1 / math.sqrt(K)); a = x if K == D else x4
    out = torch.empty(M, Nn, dtype=BF16, device=dev)
    for tile in (4256, 256):
        t = timeit(lambda: ops.gemm_nt(a, w, ops.EPI_BF16, out, tile=tile))
        print(f"gemm_nt {name} tile{tile}: {t*1e3:.3f} ms  {2*M*Nn*K/t/1e12:.1f} TF/s", flush=True)
    dw = torch.zeros(Nn, K, dtype=F32, device=dev)
    dy = rnd((M, Nn))
    for split in (0, 2, 4, 8, 16, -6):
        t = timeit(lambda: ops.gemm_tn_accum(dy, a, dw, split_k=split))
        print(f"gemm_tn {name} split{split}: {t*1e3:.3f} ms  {2*M*Nn*K/t/1e12:.1f} TF/s", flush=True)
    del dw, dy, out
# epilogues
w = rnd((4 * D, D), scale=1 / 32); bias = torch.zeros(4 * D, device=dev)
pre = torch.empty(M, 4 * D, dtype=BF16, device=dev); act = torch.empty_like(pre)
for tile in (256,):
    t = timeit(lambda: ops.gemm_nt(x, w, ops.EPI_BIAS_GELU, pre, out2=act, bias=bias, tile=tile))
    print(f"gemm_nt fc1+bias+gelu tile{tile}: {t*1e3:.3f} ms {2*M*4*D*D/t/1e12:.1f} TF/s")
    t = timeit(lambda: ops.gemm_nt(x, w, ops.EPI_GELU_BWD, act, aux=pre, tile=tile))
    print(f"gemm_nt dgrad+gelu_bwd tile{tile}: {t*1e3:.3f} ms {2*M*4*D*D/t/1e12:.1f} TF/s")
w2 = rnd((D, 4 * D), scale=1 / 64); xo = torch.empty(M, D, dtype=F32, device=dev); res_ = torch.randn(M, D, device=dev); b2 = torch.zeros(D, device=dev)
for tile in (256,):
    t = timeit(lambda: ops.gemm_nt(act, w2, ops.EPI_F32, xo, bias=b2, resid=res_, tile=tile))
    print(f"gemm_nt fc2+bias+resid f32 tile{tile}: {t*1e3:.3f} ms {2*M*4*D*D/t/1e12:.1f} TF/s")

del pre, act, w, w2, xo, res_
# attention
qkv = rnd((M, 3 * D)); o = torch.empty(M, D, dtype=BF16, device=dev); lse = torch.empty(B * H, N, device=dev)
t = timeit(lambda: ops.attn_fwd(qkv, B, N, H, o, lse))
fl = 4 * B * H * N * N * 64
print(f"attn_fwd: {t*1e3:.3f} ms {fl/t/1e12:.1f} TF/s")
do = rnd((M, D)); dqkv = torch.empty_like(qkv)
t = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, N, H))
print(f"attn_bwd: {t*1e3:.3f} ms {2.5*fl/t/1e12:.1f} TF/s (5-product flops)")
t = timeit(lambda: ops.attn_fwd(qkv, B, N, H, o, lse, drop_key=7, drop_p=0.1))
print(f"attn_fwd dropout 0.1: {t*1e3:.3f} ms")
t = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, N, H, drop_key=7, drop_p=0.1))
print(f"attn_bwd dropout 0.1: {t*1e3:.3f} ms")
act = torch.randn(B * N, 4 * D, device=dev).to(torch.bfloat16)
t = timeit(lambda: ops.dropout(act, 7, 0.1))
print(f"dropout bf16 in place [M,4D]: {t*1e3:.3f} ms {2*act.numel()*2/t/1e9:.0f} GB/s")
xf = torch.randn(B * N, D, device=dev); rf = torch.randn(B * N, D, device=dev)
t = timeit(lambda: ops.dropout(xf, 7, 0.1, resid=rf))
print(f"dropout f32 + resid in place [M,D]: {t*1e3:.3f} ms {3*xf.numel()*4/t/1e9:.0f} GB/s")
# elementwise
xf = torch.randn(M, D, device=dev); g = torch.ones(D, device=dev); bt = torch.zeros(D, device=dev)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev); y = torch.empty(M, D, dtype=BF16, device=dev)
t = timeit(lambda: ops.layernorm_fwd(xf, g, bt, y, mean, rstd)); print(f"ln_fwd: {t*1e6:.1f} us {(M*D*6)/t/1e12:.2f} TB/s")
dx = torch.empty(M, D, device=dev); dxb = torch.empty(M, D, dtype=BF16, device=dev); dg = torch.empty(D, device=dev); db = torch.empty(D, device=dev)
t = timeit(lambda: ops.layernorm_bwd(y, xf, g, mean, rstd, xf, dx, dxb, dg, db)); print(f"ln_bwd: {t*1e6:.1f} us {(M*D*(2+4+4+4+2))/t/1e12:.2f} TB/s")
from oracle import ref_cpu as R
tables = ops.rope_tables_device(R.rope_inv_freq(256), R.rope_inv_freq(384), R.rope_inv_freq(384), (8, 16, 16), 0.6, dev)
t = timeit(lambda: ops.rope3d_qk_(qkv, tables, (8, 16, 16))); print(f"rope: {t*1e6:.1f} us {(M*2048*2*2)/t/1e12:.2f} TB/s")
video = torch.randn(B, 3, 16, 256, 256, device=dev)
t = timeit(lambda: ops.tubelet_gather(video, (2, 16, 16))); print(f"tubelet_gather: {t*1e6:.1f} us {(video.numel()*6)/t/1e12:.2f} TB/s")
t = timeit(lambda: ops.colsum_accum(x4, torch.zeros(4 * D, device=dev))); print(f"colsum 4096: {t*1e6:.1f} us {(M*4*D*2)/t/1e12:.2f} TB/s")
n = 303_800_000 // 8 * 8
a = torch.randn(n, device=dev); b_ = torch.randn(n, device=dev)
t = timeit(lambda: ops.ema_update_(a, b_, 0.996)); print(f"ema: {t*1e6:.1f} us {(n*12)/t/1e12:.2f} TB/s")
xb = rnd((B, N, D)); idx = torch.stack([torch.randperm(N)[:1024].sort().values for _ in range(B)]).int().to(dev)
t = timeit(lambda: ops.token_gather(xb, idx)); print(f"gather: {t*1e6:.1f} us {(B*1024*D*2*2)/t/1e12:.2f} TB/s")
p = rnd((B, 1024, D)); q = rnd((B, 1024, D))
t = timeit(lambda: ops.l1_loss(p, q)); print(f"l1: {t*1e6:.1f} us {(B*1024*D*2*3)/t/1e12:.2f} TB/s")
