// Diagnostic instantiations of the four-wave persistent GEMM, in a library of their own (tools/libva_gemm_debug.so).
//
// The product library instantiates the kernel with DBG = 0 only; the variants below either compute WRONG results on
// purpose (timing-only ablations) or spend cycles on stamps, so they are built from this separate source and reached
// through a separate entry point that the product never links:
//   dbg 1  no operand loads in the main loop            dbg 2  no LDS fragment reads
//   dbg 3  no stage barrier                             dbg 5  correct results + s_memtime / s_memrealtime stamps
//   dbg 6  operands fetched as whole lines into registers and dropped       dbg 7  the same + one ds_write_b128 each
//   dbg 8  stamps + epilogue without its stores         dbg 9  stamps + static tile hand-out (no ticket atomics)
//   dbg 10 stamps + epilogue that stores zeros without reading the accumulators
//   dbg 11-17  as 10 with other store shapes (see the kernel)
// Build (done by the tools that use it): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared
//                                        -I vit_amd/csrc -I include tools/gemm_nt_debug.hip -o tools/libva_gemm_debug.so
#include <cstdarg>
#include <cstdio>
#include "../vit_amd/csrc/gemm_nt.hip"

// the error plumbing of the product library (defined in its api.hip) is not linked here
static thread_local char dbg_error[256];
extern "C" void va_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dbg_error, sizeof dbg_error, fmt, ap);
    va_end(ap);
}
extern "C" const char* va_dbg_last_error(void) { return dbg_error; }

// C[M,N] (bf16) = X[M,K] * W[N,K]^T with diagnostic variant `dbg` on at most `grid_limit` workgroups (0: one per CU).
// dbg 5 writes 8 floats per (workgroup, wave) into `stamps`: cycles in waits+barriers, in stage bodies, after the last
// stage, in epilogues; cycles and 100-MHz ticks from the first stage to the end; cycles of the prologue; 0.
extern "C" int va_dbg_gemm_nt_quad(const void* x, const void* w, int64_t M, int64_t N, int64_t K, void* out, float* stamps,
                                   int dbg, int grid_limit, hipStream_t stream) {
    if (!x || !w || !out || M <= 0 || N % 256 || K % 64 || K < 128) { va_set_error("va_dbg_gemm_nt_quad: bad arguments"); return VA_ERR_INVALID; }
    if ((dbg == 5 || dbg >= 8) && !stamps) { va_set_error("va_dbg_gemm_nt_quad: dbg 5 needs a stamp buffer"); return VA_ERR_INVALID; }
    EpiArgs ep{out, stamps, nullptr, nullptr, nullptr, (int)N, 0u, 0u, 1.f, 1, nullptr, nullptr, 0, 0, 0};
    const bf16_t* X = (const bf16_t*)x;
    const bf16_t* W = (const bf16_t*)w;
    const int m = (int)M, n = (int)N, k = (int)K;
    switch (dbg) {
        case 0: return launch_quad<VA_EPI_BF16, false, 0>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 1: return launch_quad<VA_EPI_BF16, false, 1>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 2: return launch_quad<VA_EPI_BF16, false, 2>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 3: return launch_quad<VA_EPI_BF16, false, 3>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 5: return launch_quad<VA_EPI_BF16, false, 5>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 6: return launch_quad<VA_EPI_BF16, false, 6>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 7: return launch_quad<VA_EPI_BF16, false, 7>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 8: return launch_quad<VA_EPI_BF16, false, 8>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 9: return launch_quad<VA_EPI_BF16, false, 9>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 10: return launch_quad<VA_EPI_BF16, false, 10>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 11: return launch_quad<VA_EPI_BF16, false, 11>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 12: return launch_quad<VA_EPI_BF16, false, 12>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 13: return launch_quad<VA_EPI_BF16, false, 13>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 14: return launch_quad<VA_EPI_BF16, false, 14>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 15: return launch_quad<VA_EPI_BF16, false, 15>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 16: return launch_quad<VA_EPI_BF16, false, 16>(X, W, m, n, k, k, k, ep, stream, grid_limit);
        case 17: return launch_quad<VA_EPI_BF16, false, 17>(X, W, m, n, k, k, k, ep, stream, grid_limit);
    }
    va_set_error("va_dbg_gemm_nt_quad: unknown variant %d", dbg);
    return VA_ERR_INVALID;
}
