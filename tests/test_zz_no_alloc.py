"""GPU: the entry point that allocates nothing (`va_layernorm_bwd_noalloc`, static device scratch) against the
existing `va_layernorm_bwd_dropout` (caller workspace), through the C ABI: bit-identical outputs on two calls in a
row, the second with a smaller input, so stale contents of the shared scratch array would show."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
D = 1024


def _run(lib, M, seed, noalloc, with_colsum, p, key=0x1234):
    from vit_amd._lib import check
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, D, generator=g).cuda()
    dy = torch.randn(M, D, generator=g).to(torch.bfloat16).cuda()
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).cuda()
    resid = torch.randn(M, D, generator=g).cuda()
    mean = x.mean(dim=1).contiguous()
    rstd = torch.rsqrt(x.var(dim=1, unbiased=False) + 1e-6).contiguous()
    dx = torch.full((M, D), float("nan"), device="cuda")
    dx_bf = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
    dgamma, dbeta = torch.full((D,), float("nan"), device="cuda"), torch.full((D,), float("nan"), device="cuda")
    colsum = torch.full((D,), float("nan"), device="cuda") if with_colsum else None
    stream = torch.cuda.current_stream().cuda_stream
    cs = colsum.data_ptr() if with_colsum else None
    if noalloc:
        check(lib.va_layernorm_bwd_noalloc(dy.data_ptr(), 1, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           resid.data_ptr(), dx.data_ptr(), dx_bf.data_ptr(), dgamma.data_ptr(),
                                           dbeta.data_ptr(), M, D, key, p, cs, stream), "va_layernorm_bwd_noalloc")
    else:
        rows = lib.va_layernorm_bwd_partial_rows(M)
        partial = torch.empty(rows * 3 * D, device="cuda")
        check(lib.va_layernorm_bwd_dropout(dy.data_ptr(), 1, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           resid.data_ptr(), dx.data_ptr(), dx_bf.data_ptr(), partial.data_ptr(),
                                           dgamma.data_ptr(), dbeta.data_ptr(), M, D, key, p, cs, stream),
              "va_layernorm_bwd_dropout")
    torch.cuda.synchronize()
    return [t for t in (dx, dx_bf, dgamma, dbeta, colsum) if t is not None]


def test_noalloc_layernorm_backward_equals_workspace_variant():
    from vit_amd import _lib
    lib = _lib.load()
    # large first (fills all 512 partial rows of the scratch), then smaller inputs that use only part of it
    for M, seed, with_colsum, p in ((4099, 1, True, 0.1), (777, 2, True, 0.0), (5, 3, False, 0.0), (2048, 4, False, 0.1)):
        want = _run(lib, M, seed, False, with_colsum, p)
        got = _run(lib, M, seed, True, with_colsum, p)
        for a, b in zip(got, want):
            assert torch.isfinite(a.float()).all()
            assert torch.equal(a, b), (M, a.shape)


def test_noalloc_rejects_sizes_beyond_its_maximum():
    from vit_amd import _lib
    lib = _lib.load()
    t = torch.zeros(8, D, device="cuda")
    rc = lib.va_layernorm_bwd_noalloc(t.data_ptr(), 0, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None,
                                      t.data_ptr(), None, t.data_ptr(), t.data_ptr(), (1 << 20) + 1, D, 0, 0.0, None, None)
    assert rc == -1 and b"maximum" in lib.va_last_error()
