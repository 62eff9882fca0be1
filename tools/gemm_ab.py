"""Same-box A/B of the NT GEMM between library builds: python tools/gemm_ab.py libA.so libB.so ...
Every (shape, epilogue) the encoder step launches at B = 32, timed per build in alternation; every build's outputs are
compared with the first build's bit for bit."""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
M, D = int(os.environ.get("AB_M", "65536")), 1024
EPI = {"bf16": 0, "bias+gelu": 1, "f32+bias+resid": 2, "f32+bias": 2, "xgelu'": 3, "gelu-only": 4, "bias+gelu+grad": 5, "xaux": 6}
CASES = [("qkv", 3 * D, D, "bf16"), ("out", D, D, "f32+bias+resid"), ("fc1", 4 * D, D, "bias+gelu"), ("fc1", 4 * D, D, "gelu-only"),
         ("fc2", D, 4 * D, "f32+bias+resid"), ("dact", 4 * D, D, "xgelu'"), ("fc1", 4 * D, D, "bias+gelu+grad"), ("dact", 4 * D, D, "xaux"), ("dln2", D, 4 * D, "bf16"), ("dattn", D, D, "bf16"),
         ("dln1", D, 3 * D, "bf16"), ("out", D, D, "f32+bias"), ("out", D, D, "bf16")]
if os.environ.get("AB_CASES"):
    CASES = [c for c in CASES if any(k in c[0] + ":" + c[3] for k in os.environ["AB_CASES"].split(","))]


def bind(path):
    lib = ctypes.CDLL(path)
    for name in ("va_gemm_nt", "va_last_error"):
        res, args = _lib.SIGNATURES[name]
        getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib


# "lib.so@4256" pins the tile code (4256 four-wave persistent, 256 eight-wave ring) instead of the library's own choice
libs = [(p, bind(p.split("@")[0])) for p in sys.argv[1:]]
tiles = [int(p.split("@")[1]) if "@" in p else 0 for p in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in CASES:
    x = torch.randn(M, K, device=dev, generator=g).to(BF16)
    w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(BF16)
    bias = torch.randn(N, device=dev, generator=g) if "bias" in epi or epi == "gelu-only" else None
    resid = torch.randn(M, N, device=dev, generator=g) if "resid" in epi else None
    aux = torch.randn(M, N, device=dev, generator=g).to(BF16) if epi in ("xgelu'", "xaux") else None
    odt = F32 if epi.startswith("f32") else BF16
    outs, times = [], [[] for _ in libs]

    def launch(lib, out, out2, tile=0):
        rc = lib.va_gemm_nt(x.data_ptr(), w.data_ptr(), M, N, K, K, K, EPI[epi], out.data_ptr(),
                            None if out2 is None else out2.data_ptr(), None if bias is None else bias.data_ptr(),
                            None if resid is None else resid.data_ptr(), None if aux is None else aux.data_ptr(), N, tile, st)
        assert rc == 0, lib.va_last_error()

    bufs = [(torch.zeros(M, N, dtype=odt, device=dev), torch.zeros(M, N, dtype=BF16, device=dev) if epi in ("bias+gelu", "bias+gelu+grad") else None)
            for _ in libs]
    for rnd in range(3):
        for i, (_, lib) in enumerate(libs):
            for _ in range(5):
                launch(lib, *bufs[i], tiles[i])
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                launch(lib, *bufs[i], tiles[i])
            e.record()
            torch.cuda.synchronize()
            times[i].append(s.elapsed_time(e) / 20)
    same = all(torch.equal(bufs[i][0], bufs[0][0]) and (bufs[i][1] is None or torch.equal(bufs[i][1], bufs[0][1]))
               for i in range(1, len(libs)))
    flops = 2.0 * M * N * K
    print(f"{name:6s} {N:5d}x{K:5d} {epi:15s}: " + " | ".join(
        f"{min(t):.4f} ms ({flops / min(t) / 1e9:.0f} TF)" for t in times) + f" | identical: {same}", flush=True)
