"""Same-box A/B of the attention kernels between library builds: python tools/attn_ab.py libA.so libB.so ...
Times forward, backward (both passes together) at the ViT-L shape and compares every build's outputs with the
first build's."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib  # noqa: E402

B, N, H, D = int(os.environ.get("AB_B", "32")), int(os.environ.get("AB_N", "2048")), 16, 1024
hd, M = D // H, B * N
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
do = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)


def bind(path):
    lib = ctypes.CDLL(path)
    for name in ("va_attn_fwd", "va_attn_bwd", "va_attn_bwd_workspace_bytes", "va_last_error"):
        res, args = _lib.SIGNATURES[name]
        getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib


def run(lib, iters=10):
    o = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H, N, device=dev)
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(lib.va_attn_bwd_workspace_bytes(B, N, H) // 4, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    p = qkv.data_ptr()

    def fwd():
        assert lib.va_attn_fwd(p, p + 2 * D, p + 4 * D, o.data_ptr(), lse.data_ptr(), B, N, H, hd, 3 * D, D, st) == 0, lib.va_last_error()

    def bwd():
        d = dqkv.data_ptr()
        assert lib.va_attn_bwd(p, p + 2 * D, p + 4 * D, o.data_ptr(), do.data_ptr(), lse.data_ptr(), ws.data_ptr(), d, d + 2 * D,
                               d + 4 * D, B, N, H, hd, 3 * D, D, 3 * D, st) == 0, lib.va_last_error()

    out = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        out[name] = s.elapsed_time(e) / iters
    return out, o.clone(), dqkv.clone()


base = None
for rnd in range(2):                      # two rounds: the second shows the run-to-run spread
    for path in sys.argv[1:]:
        t, o, dqkv = run(bind(os.path.abspath(path)))
        msg = f"{os.path.basename(path):28s} fwd {t['fwd']:.3f} ms  bwd {t['bwd']:.3f} ms"
        if base is None:
            base = (o, dqkv)
        else:
            eo = (o.float() - base[0].float()).norm() / base[0].float().norm()
            eg = (dqkv.float() - base[1].float()).norm() / base[1].float().norm()
            msg += f"   vs first: o rel {eo.item():.2e}, dqkv rel {eg.item():.2e}, dqkv equal {torch.equal(dqkv, base[1])}"
        print(msg, flush=True)
