"""Builds (once) and binds tools/libva_gemm_debug.so, the diagnostic instantiations of the four-wave GEMM."""
import ctypes, os, subprocess
import torch  # noqa: F401  (its HIP runtime must be resident first)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "gemm_nt_debug.hip")
LIB = os.path.join(ROOT, "tools", "libva_gemm_debug.so")


def build():
    deps = [SRC, os.path.join(ROOT, "vit_amd", "csrc", "gemm_nt.hip"), os.path.join(ROOT, "vit_amd", "csrc", "common.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-munsafe-fp-atomics", "-shared", "-I", os.path.join(ROOT, "include"), SRC, "-o", LIB], check=True)
    return LIB


def load():
    lib = ctypes.CDLL(build())
    lib.va_dbg_gemm_nt_quad.restype = ctypes.c_int
    lib.va_dbg_gemm_nt_quad.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.va_dbg_last_error.restype = ctypes.c_char_p
    return lib


def gemm(lib, x, w, out, dbg=0, grid_limit=0, stamps=None):
    M, K = x.shape
    rc = lib.va_dbg_gemm_nt_quad(x.data_ptr(), w.data_ptr(), M, w.shape[0], K, out.data_ptr(),
                                 None if stamps is None else stamps.data_ptr(), dbg, grid_limit,
                                 torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError(lib.va_dbg_last_error().decode())


if __name__ == "__main__":
    print(build())
