"""fc1 + bias + GELU with and without the pre-activation output (the no-grad forward writes only the activation),
and the encoder's no-grad forward at ViT-L, B = 32."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
from vit_amd import surgical_world_model as swm
dev = torch.device("cuda:0")
M, N, K = 65536, 4096, 1024
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev); act = torch.empty_like(pre)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for _ in range(2):
    a = timed(lambda: ops.gemm_nt(x, w, ops.EPI_BIAS_GELU, pre, out2=act, bias=bias))
    b = timed(lambda: ops.gemm_nt(x, w, ops.EPI_GELU_ONLY, act, bias=bias))
    print(f"fc1+GELU two outputs {a:.4f} ms, activation only {b:.4f} ms", flush=True)
del x, w, pre, act
enc = swm.VideoViTEncoder(img_size=256, num_frames=16, embed_dim=1024, depth=24, num_heads=16).eval().to(dev)
video = torch.randn(32, 3, 16, 256, 256, device=dev)
with torch.no_grad():
    t = timed(lambda: enc(video), n=5)
print(f"encoder no-grad forward, B=32: {t:.2f} ms ({32 / t * 1e3:.1f} clips/s)")
