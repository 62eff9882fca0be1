"""ctypes binding of libvit_amd.so (the C ABI declared in include/vit_amd.h).

There is no fallback: if the shared library is missing, `load()` raises, and every op in
`vit_amd.ops` goes through `load()`.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint32, c_void_p

# PyTorch-ROCm ships its own libamdhip64.so.7.  It must be the HIP runtime already resident when
# libvit_amd.so is opened, so that the library's kernels launch into the runtime (and device context)
# that owns torch's allocations and streams; opening the library first would pull in a second runtime.
import torch  # noqa: F401  (import order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIT_AMD_LIB=<path> loads another build of the same library (same-box A/B of kernel variants); there is still no fallback.
LIB_PATH = os.environ.get("VIT_AMD_LIB") or os.path.join(_HERE, "libvit_amd.so")

_lib = None

i64, vp, ci, cf, cu32 = c_int64, c_void_p, c_int, c_float, c_uint32

# name -> (restype, argtypes); mirrors include/vit_amd.h one to one
SIGNATURES = {
    "va_version": (ci, []),
    "va_last_error": (c_char_p, []),
    "va_gemm_nt": (ci, [vp, vp, i64, i64, i64, i64, i64, ci, vp, vp, vp, vp, vp, i64, ci, vp]),
    "va_gemm_nt_dropout": (ci, [vp, vp, i64, i64, i64, i64, i64, ci, vp, vp, vp, vp, vp, i64, ci, cu32, cf, vp]),
    "va_gemm_tn_accum": (ci, [vp, vp, i64, i64, i64, i64, i64, vp, i64, vp, ci, vp]),
    "va_colsum_bf16_accum": (ci, [vp, i64, i64, i64, vp, vp]),
    "va_tubelet_gather": (ci, [vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp]),
    "va_tubelet_scatter": (ci, [vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp]),
    "va_layernorm_fwd": (ci, [vp, vp, vp, vp, ci, vp, vp, i64, i64, cf, vp]),
    "va_layernorm_bwd_partial_rows": (i64, [i64]),
    "va_layernorm_bwd": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp]),
    "va_layernorm_bwd_dropout": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, cu32, cf, vp, vp]),
    "va_layernorm_bwd_noalloc": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, cu32, cf, vp, vp]),
    "va_rope3d_qk_inplace": (ci, [vp, vp, i64, i64, i64, i64, ci, vp]),
    "va_rope3d_qk_indexed": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, ci, vp]),
    "va_attn_fwd": (ci, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, vp]),
    "va_attn_fwd_block_causal": (ci, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, vp]),
    "va_attn_bwd_workspace_bytes": (i64, [i64, i64, i64]),
    "va_attn_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, vp]),
    "va_attn_mask_words": (i64, [i64]),
    "va_attn_mask_pack": (ci, [vp, i64, i64, vp, vp, vp]),
    "va_attn_fwd_masked": (ci, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, i64, i64, vp]),
    "va_attn_bwd_masked": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, vp, vp, i64, i64,
                                vp]),
    "va_attn_fwd_dropout": (ci, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, cu32, cf, vp]),
    "va_attn_bwd_dropout": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, cu32, cf,
                                vp]),
    "va_dropout": (ci, [vp, ci, vp, ci, vp, i64, cu32, cf, vp]),
    "va_cast_f32_bf16": (ci, [vp, vp, i64, vp]),
    "va_cast_transpose_f32_bf16": (ci, [vp, vp, i64, i64, vp]),
    "va_multi_cast_f32_bf16": (ci, [vp, i64, i64, vp]),
    "va_future_mask": (ci, [vp, vp, vp, i64, i64, i64, i64, i64, vp]),
    "va_token_gather": (ci, [vp, vp, vp, i64, i64, i64, i64, vp]),
    "va_token_scatter": (ci, [vp, vp, vp, i64, i64, i64, i64, vp]),
    "va_token_pool_workspace_floats": (i64, [i64, i64]),
    "va_token_pool_fwd": (ci, [vp, vp, vp, i64, i64, i64, i64, cf, vp]),
    "va_token_pool_bwd": (ci, [vp, vp, i64, i64, i64, i64, cf, vp]),
    "va_ema_update": (ci, [vp, vp, vp, i64, cf, vp]),
    "va_l1_loss": (ci, [vp, vp, ci, vp, vp, ci, cf, i64, vp]),
    "va_normalize_u8": (ci, [vp, vp, i64, i64, i64, i64, vp, vp, vp]),
    "va_multi_sumsq": (ci, [vp, vp, i64, vp, vp, vp]),
    "va_multi_adamw": (ci, [vp, vp, i64, cf, cf, cf, cf, cf, vp, cf, vp]),
    "va_gated_delta_seq": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    "va_dwconv1d_causal_silu": (ci, [vp, vp, vp, vp, i64, i64, i64, ci, vp]),
    "va_gate_layernorm_fwd": (ci, [vp, vp, vp, vp, vp, vp, i64, i64, cf, vp]),
    "va_dwconv1d_causal_silu_bf16": (ci, [vp, i64, vp, vp, vp, i64, i64, i64, ci, vp]),
    "va_normalize_rows_fwd": (ci, [vp, vp, vp, i64, i64, vp]),
    "va_normalize_rows_bwd": (ci, [vp, vp, vp, vp, i64, i64, vp]),
    "va_gemm_nt_qkv_rope": (ci, [vp, vp, i64, i64, i64, i64, vp, i64, vp, vp, i64, i64, i64, vp]),
    "va_infonce_workspace_floats": (i64, [i64, i64]),
    "va_infonce_sym": (ci, [vp, vp, vp, vp, vp, vp, i64, i64, cf, ci, vp]),
    "va_cross_entropy": (ci, [vp, vp, vp, vp, vp, i64, i64, cf, vp]),
    "va_mse_workspace_floats": (i64, [i64]),
    "va_mse_loss": (ci, [vp, vp, vp, vp, vp, i64, cf, vp]),
    "va_delta_gates_fwd": (ci, [vp, vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    "va_gate_layernorm_fwd_g16": (ci, [vp, vp, vp, i64, vp, vp, vp, i64, i64, cf, vp]),
}


class VitAmdError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raise loudly when it is absent (no CPU or eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitAmdError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). vit_amd has no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().va_last_error().decode(errors="replace")
        raise VitAmdError(f"{what} failed ({code}): {msg}")
