// How a large GEMM behaves when another stream's kernel holds some CUs (as a collective does during backward):
// a blocker of NB workgroups (64 KiB LDS, 256 threads, spinning for about US microseconds) is started on a second
// stream right before the GEMM; the GEMM's duration is compared with its undisturbed one, for each tile code.
// Build: hipcc --offload-arch=gfx950 -O3 -I../../include -o interfere interfere.hip -L../../vit_amd -lvit_amd -Wl,-rpath,'$ORIGIN/../../vit_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "vit_amd.h"

__global__ __launch_bounds__(256) void blocker(long long ticks, int* sink) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (lds[threadIdx.x] == 2) *sink = 1;
}

int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 32;
    const int US = argc > 2 ? atoi(argv[2]) : 400;
    const int64_t M = 65536, N = 3072, K = 1024;
    void *x, *w, *out, *dy; float* dw; int* sink;
    hipMalloc(&x, M * K * 2); hipMalloc(&w, N * K * 2); hipMalloc(&out, M * N * 2); hipMalloc(&dy, M * N * 2);
    hipMalloc(&dw, N * K * 4); hipMalloc(&sink, 4);
    hipMemset(x, 0x11, M * K * 2); hipMemset(w, 0x11, N * K * 2); hipMemset(dy, 0x11, M * N * 2); hipMemset(dw, 0, N * K * 4);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipFuncSetAttribute((const void*)blocker, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
    const long long ticks = (long long)US * rate / 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int which, int tile, bool blocked) {
        float best = 1e9f, sum = 0;
        const int reps = 6;
        for (int r = 0; r < reps + 2; ++r) {
            hipDeviceSynchronize();
            if (blocked) hipLaunchKernelGGL(blocker, dim3(NB), dim3(256), 65536, sb, ticks, sink);
            hipEventRecord(e0, sa);
            int rc = which == 0 ? va_gemm_nt(x, w, M, N, K, K, K, VA_EPI_BF16, out, nullptr, nullptr, nullptr, nullptr, N, tile, sa)
                                : va_gemm_tn_accum(dy, x, M, N, K, N, K, dw, K, nullptr, tile, sa);
            if (rc) { printf("error: %s\n", va_last_error()); exit(1); }
            hipEventRecord(e1, sa); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        return sum / reps;
    };
    printf("blocker: %d workgroups for ~%d us\n", NB, US);
    for (int tile : {256, 4256}) printf("gemm_nt qkv tile %4d: alone %.3f ms, with blocker %.3f ms\n", tile, run(0, tile, false), run(0, tile, true));
    for (int split : {0, 8, 16}) printf("gemm_tn qkv split %2d: alone %.3f ms, with blocker %.3f ms\n", split, run(1, split, false), run(1, split, true));
    return 0;
}
