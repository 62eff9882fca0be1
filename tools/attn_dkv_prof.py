"""Where the pipelined dK/dV pass spends its cycles: stamped build (tools/attn_debug.hip) at the ViT-L shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _attn_debug as A
from vit_amd import _lib
dbg = A.load(os.environ.get("DKV_DEBUG_LIB"))
CHECK = os.environ.get("DKV_DEBUG_LIB") is None      # ablated builds compute wrong results on purpose
lib = _lib.load()
B, N, H, D = 32, 2048, 16, 1024
hd, M = D // H, B * N
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
do = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
o = torch.empty(M, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, N, device=dev)
dqkv = torch.empty_like(qkv); ref = torch.empty_like(qkv)
ws = torch.empty(lib.va_attn_bwd_workspace_bytes(B, N, H) // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
p, d = qkv.data_ptr(), ref.data_ptr()
assert lib.va_attn_fwd(p, p + 2 * D, p + 4 * D, o.data_ptr(), lse.data_ptr(), B, N, H, hd, 3 * D, D, st) == 0
assert lib.va_attn_bwd(p, p + 2 * D, p + 4 * D, o.data_ptr(), do.data_ptr(), lse.data_ptr(), ws.data_ptr(), d, d + 2 * D, d + 4 * D,
                       B, N, H, hd, 3 * D, D, 3 * D, st) == 0
wgs = (N // 256) * B * H
stamps = torch.zeros(wgs * 4 * 8, device=dev)
d2 = dqkv.data_ptr()
def run():
    rc = dbg.va_dbg_attn_dkv64(p, p + 2 * D, p + 4 * D, do.data_ptr(), ws.data_ptr(), d2 + 2 * D, d2 + 4 * D, B, N, H, 3 * D, D, 3 * D,
                               stamps.data_ptr(), st)
    assert rc == 0, dbg.va_dbg_last_error()
for _ in range(20): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); run(); e.record(); torch.cuda.synchronize()
assert not CHECK or torch.equal(dqkv[:, D:], ref[:, D:]), "stamped build must compute the same dK / dV"
f = stamps.view(wgs, 4, 8).double().cpu()
nt = f[0, 0, 7].item()
names = ["tile top (wait+barrier+staging)", "step (qb0,kb0)", "step (qb0,kb1)", "step (qb1,kb0)", "step (qb1,kb1)"]
print(f"stamped kernel {s.elapsed_time(e):.3f} ms, {nt:.0f} tiles per workgroup; medians over all waves, cycles per tile:")
tot = 0
for i, n in enumerate(names):
    v = (f[:, :, i] / nt).median().item(); tot += v
    print(f"  {n:34s} {v:7.0f}")
print(f"  sum {tot:.0f} per tile (18 MFMAs per step = 576 matrix-pipe cycles; 2304 per tile)")
print(f"  prologue {f[:, :, 5].median().item():.0f}  epilogue {f[:, :, 6].median().item():.0f} cycles per workgroup")
for w in range(4):
    print(f"  wave {w}: top {(f[:, w, 0] / nt).median().item():.0f}  steps " + " ".join(f"{(f[:, w, i] / nt).median().item():.0f}" for i in range(1, 5)))
