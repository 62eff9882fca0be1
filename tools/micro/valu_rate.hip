// Micro-benchmark: issue rate of the integer VALU ops a counter-based mask hash can be built from
// (32-bit multiply, 24-bit multiply / multiply-add, xor, shift-add), 8 independent chains per lane, 8 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(512) void k(uint32_t* __restrict__ out, uint32_t seed, int iters) {
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 9 + i;
    uint32_t c = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 2) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 4) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (OP == 7) asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 8) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(x[i]));
                if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                if (OP == 13 && (i & 1) == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(uint64_t*)&x[i]) : "v"(*(uint64_t*)&x[0]));
                if (OP == 14 && (i & 1) == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(uint64_t*)&x[i]) : "v"(*(uint64_t*)&x[0]));
                if (OP == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c) : "vcc");
                if (OP == 16) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 9) asm volatile("v_cmp_ge_u16_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:WORD_0\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[i]) : "v"(c) : "vcc");
            }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* out, int per) {
    const int blocks = 256, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, 12345u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // wave-instructions per SIMD: 2 waves per SIMD x iters x 64 x per
    const double winst = 2.0 * iters * 64 * per;
    printf("%-28s %.3f ms   %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / winst);
}

int main() {
    uint32_t* out; hipMalloc(&out, 256 * 512 * 4);
    run<3>("v_xor_b32", out, 1);
    run<0>("v_mul_lo_u32", out, 1);
    run<5>("v_mul_hi_u32", out, 1);
    run<1>("v_mul_u32_u24", out, 1);
    run<2>("v_mad_u32_u24", out, 1);
    run<4>("v_lshl_add_u32", out, 1);
    run<7>("v_xad_u32", out, 1);
    run<8>("v_lshrrev_b32", out, 1);
    run<6>("v_exp_f32", out, 1);
    run<9>("v_cmp_u16_sdwa+v_cndmask", out, 2);
    run<10>("v_fma_f32", out, 1);
    run<11>("v_mul_f32", out, 1);
    run<12>("v_rcp_f32", out, 1);
    run<13>("v_pk_fma_f32 (per instr, 2 elems; half as many issued)", out, 1);
    run<14>("v_pk_mul_f32 (per instr, 2 elems; half as many issued)", out, 1);
    run<15>("v_cndmask_b32", out, 1);
    run<16>("v_cvt_pk_bf16_f32", out, 1);
    return 0;
}
