// Diagnostic instantiations of the attention kernels, in a library of their own (tools/libva_attn_debug.so): builds that
// spend cycles on in-kernel stamps and write them to a buffer of their own.  The product library instantiates none of
// them.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared -I include
//               tools/attn_debug.hip -o tools/libva_attn_debug.so   (tools/_attn_debug.py does it)
#include <cstdarg>
#include <cstdio>
#include "../vit_amd/csrc/attn.hip"

static thread_local char dbg_error[256];
extern "C" void va_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dbg_error, sizeof dbg_error, fmt, ap);
    va_end(ap);
}
extern "C" const char* va_dbg_last_error(void) { return dbg_error; }

// The pipelined dK/dV pass (head dim 64) with stamps: `aux` are the records the dQ pass wrote (workspace of va_attn_bwd),
// `stamps` receives 8 floats per (workgroup, wave): cycles in {tile top, step 0..3} summed over the tiles, prologue,
// epilogue, number of tiles.
extern "C" int va_dbg_attn_dkv64(const void* q, const void* k, const void* v, const void* d_o, const void* aux, void* dk, void* dv,
                                 int64_t B, int64_t N, int64_t H, int64_t ld, int64_t ldo, int64_t lddq, float* stamps,
                                 hipStream_t stream) {
    if (!q || !k || !v || !d_o || !aux || !dk || !dv || !stamps || N % 64) { va_set_error("va_dbg_attn_dkv64: bad arguments"); return VA_ERR_INVALID; }
    constexpr int LDS_P = DKV64_SLOTS * DKV64_STAGE + 16;
    static bool flags[64] = {};
    if (int e = set_lds(attn_bwd_dkv64_kernel<true>, LDS_P, flags, "va_dbg_attn_dkv64")) return e;
    const float scale = 1.0f / sqrtf(64.0f), c = 1.4426950408889634f * scale;
    hipLaunchKernelGGL(attn_bwd_dkv64_kernel<true>, dim3((unsigned)((N + 255) / 256), (unsigned)(B * H)), dim3(256), LDS_P, stream,
                       (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, (const u32x4_t*)aux, (bf16_t*)dk,
                       (bf16_t*)dv, (int)N, (int)H, ld, ldo, lddq, c, scale, stamps);
    VA_CHECK_LAUNCH("va_dbg_attn_dkv64");
    return VA_OK;
}
