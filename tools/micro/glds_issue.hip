// Micro-benchmark: cycles a wave spends ISSUING eight 1-KiB direct-to-LDS loads, (a) each with its own LDS base
// (M0 rewritten before every load) vs (b) one LDS base and eight immediate offsets.  512-thread workgroups, one per
// CU, source lines resident in L2.  Build: hipcc --offload-arch=gfx950 -O3 -o glds_issue glds_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* g = src + ((size_t)blockIdx.x * 65536 + wave * 8192 + lane * 16) + 4096;
    char* base = smem + wave * 8192 + (MODE ? 4096 : 0);
    uint64_t t_issue = 0, t_all = 0;
    for (int it = 0; it < iters; ++it) {
        const uint64_t c0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + p * 1024), (lds_ptr_t)(base + ((p * 5) & 7) * 1024 + (p & 1) * 65536), 16, 0, 0);
        } else {
#define PIECE(p) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g), (lds_ptr_t)(base), 16, (p) * 1024 - 4096, 0);
            PIECE(0) PIECE(1) PIECE(2) PIECE(3) PIECE(4) PIECE(5) PIECE(6) PIECE(7)
#undef PIECE
        }
        const uint64_t c1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint64_t c2 = __builtin_amdgcn_s_memtime();
        t_issue += c1 - c0; t_all += c2 - c0;
    }
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = (float)t_issue / iters; out[(blockIdx.x * 8 + wave) * 2 + 1] = (float)t_all / iters; }
}

int main() {
    const int blocks = 256, iters = 2000;
    char* src; float* out;
    hipMalloc(&src, (size_t)blocks * 65536 + 65536); hipMalloc(&out, blocks * 8 * 2 * 4);
    hipMemset(src, 1, (size_t)blocks * 65536 + 65536);
    float* h = new float[blocks * 16];
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 160 * 1024, 0, src, out, iters); }
            else { hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 160 * 1024, 0, src, out, iters); }
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, blocks * 16 * 4, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < blocks * 8; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
        printf("mode %d (%s): issue of 8 pieces %.0f cycles, issue+land+barrier %.0f cycles\n", mode, mode ? "one base + immediate offsets" : "base rewritten per piece", a / (blocks * 8), b / (blocks * 8));
    }
    return 0;
}
