// Micro-benchmark: sustained bf16 MFMA rate of the two shapes on register operands (no memory traffic),
// 8 waves per CU, random operand bits.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    bf16x8_t a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8_t, in[(t * 12 + i) & 0xffff]);
    for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(bf16x8_t, in[(t * 12 + 4 + i) & 0xffff]);
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4_t acc[4][8];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16_t acc[2][4];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i * 2 + kk], b[j * 2 + kk], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    out[t] = s;
}

int main() {
    const int blocks = 256, iters = 20000;
    uint4* in; float* out;
    hipMalloc(&in, 65536 * 16); hipMalloc(&out, blocks * 512 * 4);
    uint32_t* h = (uint32_t*)malloc(65536 * 16);
    for (int i = 0; i < 65536 * 4; ++i) { uint32_t lo = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15), hi = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15); h[i] = lo | (hi << 16); }
    hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape : {16, 32, 16, 32}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 8 * iters * 32 * 16384.0;   // per wave-iter: 32 x (16x16x32) = 16 x (32x32x16)
            if (rep) printf("shape %dx: %.2f ms  %.0f TF/s\n", shape, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
