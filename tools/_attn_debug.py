"""Builds (once) and binds tools/libva_attn_debug.so, the stamped instantiations of the attention kernels."""
import ctypes, os, subprocess
import torch  # noqa: F401  (its HIP runtime must be resident first)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "attn_debug.hip")
LIB = os.path.join(ROOT, "tools", "libva_attn_debug.so")


def build_variant(tag, ablate):
    """A timing-only build with some instruction categories removed from the step (tools/gen_attn_dkv.py, DKV_ABLATE)."""
    inc = os.path.join(ROOT, "tools", f"_dkv64_body_{tag}.inc")
    lib = os.path.join(ROOT, "tools", f"libva_attn_debug_{tag}.so")
    env = dict(os.environ, DKV_OUT=inc, DKV_ABLATE=ablate)
    subprocess.run(["python", os.path.join(ROOT, "tools", "gen_attn_dkv.py")], check=True, env=env, stdout=subprocess.DEVNULL)
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                    "-munsafe-fp-atomics", "-fno-slp-vectorize", "-shared", "-I", os.path.join(ROOT, "include"),
                    f'-DVA_DKV64_BODY="{inc}"', SRC, "-o", lib], check=True)
    return lib


def build():
    deps = [SRC] + [os.path.join(ROOT, "vit_amd", "csrc", f) for f in ("attn.hip", "attn_dkv64_body.inc", "common.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-munsafe-fp-atomics", "-fno-slp-vectorize", "-shared", "-I", os.path.join(ROOT, "include"), SRC, "-o", LIB],
                       check=True)
    return LIB


def load(path=None):
    lib = ctypes.CDLL(path or build())
    lib.va_dbg_attn_dkv64.restype = ctypes.c_int
    lib.va_dbg_attn_dkv64.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int64] * 6 + [ctypes.c_void_p, ctypes.c_void_p]
    lib.va_dbg_last_error.restype = ctypes.c_char_p
    return lib


if __name__ == "__main__":
    print(build())
