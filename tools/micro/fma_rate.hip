// Build: hipcc --offload-arch=gfx950 -O3 -o fma_rate fma_rate.hip
// Issue rate of f32 fused multiply-adds on one SIMD: single v_fma_f32 against v_pk_fma_f32, with 1, 2 or 4 waves
// per SIMD.  Each wave runs ITER rounds of 16 independent accumulators (so no dependent-chain stall), and the
// kernel is timed with s_memtime.  Prints cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
constexpr int ITER = 4096;

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, long long* cycles, float a, float b) {
    float acc[16];
    f32x2_t pacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = (float)(threadIdx.x + i); pacc[i] = (f32x2_t){acc[i], acc[i] + 1.f}; }
    const f32x2_t a2 = {a, a}, b2 = {b, b};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(a2), "v"(b2));
            if (KIND == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(pacc[i]) : "v"(a2));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + pacc[i][0] + pacc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    if (hipMalloc(&out, 1024 * 256 * 4) != hipSuccess || hipMalloc(&cyc, 256 * 8) != hipSuccess) return 1;
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_fmac_f32", "v_pk_mul_f32"};
    for (int kind = 0; kind < 4; ++kind)
        for (int waves_per_simd : {1, 2, 4}) {
            const int threads = 256 * waves_per_simd;       // one workgroup on one CU: waves spread over the 4 SIMDs
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
                if (kind == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
                if (kind == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
                if (kind == 3) hipLaunchKernelGGL(rate_kernel<3>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
                hipEventRecord(e1);
                if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double per_simd_instr = (double)ITER * 16 * waves_per_simd;
            printf("%-14s %d wave(s)/SIMD: %8lld counter ticks, %.3f ms -> %.2f ns per instruction per SIMD\n", names[kind],
                   waves_per_simd, c, ms, ms * 1e6 / per_simd_instr);
        }
    return 0;
}
