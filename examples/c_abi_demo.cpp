// A host program that uses libvit_amd through its C ABI alone (include/vit_amd.h): no Python, no torch -- device
// memory from hipMalloc, one LayerNorm (f32 -> bf16) feeding one bf16 GEMM with the f32 + bias + residual epilogue,
// the pair a transformer block's attention output path makes (swm:705-714), checked on the host in double precision.
//   hipcc --offload-arch=gfx950 examples/c_abi_demo.cpp -Iinclude -Lvit_amd -l:libvit_amd.so -Wl,-rpath,'$ORIGIN/../vit_amd'
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "vit_amd.h"

#define HIP_OK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)
#define VA_OK(call)                                                                        \
    do {                                                                                   \
        if ((call) != 0) {                                                                 \
            std::fprintf(stderr, "%s: %s\n", #call, va_last_error());                      \
            return 3;                                                                      \
        }                                                                                  \
    } while (0)

static float bf16_to_float(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static uint16_t float_to_bf16(float f) {          // round to nearest even
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float hashed(uint32_t i, uint32_t salt) {  // deterministic values in [-1, 1)
    uint32_t x = i * 0x9E3779B1u + salt;
    x ^= x >> 16; x *= 0x21F0AAADu; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

int main() {
    const int64_t M = 600, D = 1024, N = 1024;    // M is not a multiple of the tile: edge rows are exercised
    std::vector<float> x(M * D), gamma(D), beta(D), bias(N), resid(M * N);
    std::vector<uint16_t> w(N * D);
    for (int64_t i = 0; i < M * D; ++i) x[i] = 2.0f * hashed((uint32_t)i, 1u) + 0.25f;
    for (int64_t i = 0; i < D; ++i) { gamma[i] = 1.0f + 0.1f * hashed((uint32_t)i, 2u); beta[i] = 0.1f * hashed((uint32_t)i, 3u); }
    for (int64_t i = 0; i < N; ++i) bias[i] = 0.5f * hashed((uint32_t)i, 4u);
    for (int64_t i = 0; i < M * N; ++i) resid[i] = hashed((uint32_t)i, 5u);
    for (int64_t i = 0; i < N * D; ++i) w[i] = float_to_bf16(0.03f * hashed((uint32_t)i, 6u));

    float *dx, *dgamma, *dbeta, *dbias, *dresid, *dmean, *drstd, *dout;
    void *dln, *dw;
    HIP_OK(hipMalloc(&dx, M * D * 4)); HIP_OK(hipMalloc(&dgamma, D * 4)); HIP_OK(hipMalloc(&dbeta, D * 4));
    HIP_OK(hipMalloc(&dbias, N * 4)); HIP_OK(hipMalloc(&dresid, M * N * 4)); HIP_OK(hipMalloc(&dmean, M * 4));
    HIP_OK(hipMalloc(&drstd, M * 4)); HIP_OK(hipMalloc(&dout, M * N * 4)); HIP_OK(hipMalloc(&dln, M * D * 2));
    HIP_OK(hipMalloc(&dw, N * D * 2));
    HIP_OK(hipMemcpy(dx, x.data(), M * D * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dgamma, gamma.data(), D * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dbeta, beta.data(), D * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dresid, resid.data(), M * N * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dw, w.data(), N * D * 2, hipMemcpyHostToDevice));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    std::printf("libvit_amd version %d\n", va_version());
    VA_OK(va_layernorm_fwd(dx, dgamma, dbeta, dln, /*out_bf16=*/1, dmean, drstd, M, D, 1e-6f, stream));
    VA_OK(va_gemm_nt(dln, dw, M, N, D, D, D, VA_EPI_F32, dout, nullptr, dbias, dresid, nullptr, N, /*tile=*/0, stream));
    // an invalid call must come back as an error code with a message, not as an exception or a crash
    if (va_gemm_nt(dln, dw, M, N, 100, D, D, VA_EPI_F32, dout, nullptr, dbias, dresid, nullptr, N, 0, stream) == 0) {
        std::fprintf(stderr, "K = 100 was accepted\n");
        return 4;
    }
    std::printf("rejected as expected: %s\n", va_last_error());
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<float> out(M * N);
    std::vector<uint16_t> ln(M * D);
    HIP_OK(hipMemcpy(out.data(), dout, M * N * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ln.data(), dln, M * D * 2, hipMemcpyDeviceToHost));

    double worst_ln = 0.0, worst = 0.0;
    for (int64_t m = 0; m < M; m += 37) {         // a sample of rows, every column
        double mean = 0.0, var = 0.0;
        for (int64_t d = 0; d < D; ++d) mean += x[m * D + d];
        mean /= D;
        for (int64_t d = 0; d < D; ++d) var += (x[m * D + d] - mean) * (x[m * D + d] - mean);
        const double rstd = 1.0 / std::sqrt(var / D + 1e-6);
        for (int64_t d = 0; d < D; ++d) {
            const double ref = (x[m * D + d] - mean) * rstd * gamma[d] + beta[d];
            worst_ln = std::fmax(worst_ln, std::fabs(ref - bf16_to_float(ln[m * D + d])) / (std::fabs(ref) + 1.0));
        }
        for (int64_t n = 0; n < N; ++n) {
            double acc = 0.0;                      // from the bf16 operands the device used: checks the GEMM itself
            for (int64_t d = 0; d < D; ++d) acc += (double)bf16_to_float(ln[m * D + d]) * bf16_to_float(w[n * D + d]);
            const double ref = acc + bias[n] + resid[m * N + n];
            worst = std::fmax(worst, std::fabs(ref - out[m * N + n]));
        }
    }
    std::printf("LayerNorm: worst relative deviation %.2e (bf16 output); GEMM + bias + residual: worst absolute deviation %.2e\n",
                worst_ln, worst);
    const bool ok = worst_ln < 4e-3 && worst < 1e-4;
    std::printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
