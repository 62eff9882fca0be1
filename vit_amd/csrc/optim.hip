// Multi-tensor optimizer kernels: global gradient norm and AdamW over an arbitrary list of f32 tensors in
// two launches (plus a tiny reduction), instead of several elementwise launches per parameter tensor.
// HBM-bound: the norm reads 4 B per element, the update reads 16 B and writes 12 B per element.
//
// Work decomposition: every tensor is cut into chunks of VA_OPT_CHUNK elements; the host passes a table with
// one (tensor index, chunk index) entry per workgroup, so tensors of any mix of sizes load the GPU evenly.
#include "common.h"

namespace {

constexpr int CHUNK = VA_OPT_CHUNK;

VA_DEVINL bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

__global__ __launch_bounds__(256) void multi_sumsq_kernel(const va_opt_tensor* __restrict__ tensors,
                                                          const int32_t* __restrict__ table,
                                                          float* __restrict__ partial) {
    const int t = table[2 * blockIdx.x], chunk = table[2 * blockIdx.x + 1];
    const va_opt_tensor T = tensors[t];
    const int64_t off = (int64_t)chunk * CHUNK;
    const int n = (int)((T.n - off) < CHUNK ? (T.n - off) : CHUNK);
    const float* g = T.g + off;
    float s = 0.f;
    if (aligned16(g)) {
        for (int i = threadIdx.x * 4; i + 3 < n; i += 1024) {
            const f32x4_t v = *(const f32x4_t*)(g + i);
            s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        for (int i = (n & ~3) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    } else {
        for (int i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    }
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup, fixed summation order: bitwise reproducible from run to run.
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
    float lr, beta1, beta2, eps, decay, max_norm;
};

VA_DEVINL void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float gscale, float inv_bc1,
                        float inv_sqrt_bc2) {
    g *= gscale;
    p *= a.decay;                                        // p <- p * (1 - lr * wd)
    m = m + (1.0f - a.beta1) * (g - m);                  // lerp, as torch's exp_avg.lerp_
    v = a.beta2 * v + (1.0f - a.beta2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + a.eps;
    p -= (a.lr * inv_bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void multi_adamw_kernel(const va_opt_tensor* __restrict__ tensors,
                                                          const int32_t* __restrict__ table, AdamArgs a,
                                                          const float* __restrict__ total_sumsq) {
    const int t = table[2 * blockIdx.x], chunk = table[2 * blockIdx.x + 1];
    const va_opt_tensor T = tensors[t];
    const int64_t off = (int64_t)chunk * CHUNK;
    const int n = (int)((T.n - off) < CHUNK ? (T.n - off) : CHUNK);
    float gscale = 1.0f;
    if (total_sumsq != nullptr && a.max_norm > 0.f) {    // torch.nn.utils.clip_grad_norm_: min(1, max / (norm + 1e-6))
        const float coef = a.max_norm / (sqrtf(total_sumsq[0]) + 1e-6f);
        gscale = coef < 1.0f ? coef : 1.0f;
    }
    float* p = T.p + off;
    const float* g = T.g + off;
    float* m = T.m + off;
    float* v = T.v + off;
    if (aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) {
        for (int i = threadIdx.x * 4; i + 3 < n; i += 1024) {
            f32x4_t pv = *(f32x4_t*)(p + i), mv = *(f32x4_t*)(m + i), vv = *(f32x4_t*)(v + i);
            const f32x4_t gv = *(const f32x4_t*)(g + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pv[e], me = mv[e], ve = vv[e];
                adam_one(pe, gv[e], me, ve, a, gscale, T.inv_bc1, T.inv_sqrt_bc2);
                pv[e] = pe; mv[e] = me; vv[e] = ve;
            }
            *(f32x4_t*)(p + i) = pv;
            *(f32x4_t*)(m + i) = mv;
            *(f32x4_t*)(v + i) = vv;
        }
        for (int i = (n & ~3) + threadIdx.x; i < n; i += 256) adam_one(p[i], g[i], m[i], v[i], a, gscale, T.inv_bc1, T.inv_sqrt_bc2);
    } else {
        for (int i = threadIdx.x; i < n; i += 256) adam_one(p[i], g[i], m[i], v[i], a, gscale, T.inv_bc1, T.inv_sqrt_bc2);
    }
}

}  // namespace

extern "C" int va_multi_sumsq(const va_opt_tensor* tensors, const int32_t* chunk_table, int64_t num_chunks,
                              float* partial, float* total, hipStream_t stream) {
    VA_CHECK_ARG(tensors && chunk_table && partial && total, "va_multi_sumsq: null operand");
    VA_CHECK_ARG(num_chunks > 0 && num_chunks < (1ll << 31), "va_multi_sumsq: bad chunk count %lld", (long long)num_chunks);
    hipLaunchKernelGGL(multi_sumsq_kernel, dim3((unsigned)num_chunks), dim3(256), 0, stream, tensors, chunk_table, partial);
    VA_CHECK_LAUNCH("va_multi_sumsq");
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, stream, partial, num_chunks, total);
    VA_CHECK_LAUNCH("va_multi_sumsq(sum)");
    return VA_OK;
}

extern "C" int va_multi_adamw(const va_opt_tensor* tensors, const int32_t* chunk_table, int64_t num_chunks, float lr,
                              float beta1, float beta2, float eps, float weight_decay, const float* total_sumsq,
                              float max_norm, hipStream_t stream) {
    VA_CHECK_ARG(tensors && chunk_table, "va_multi_adamw: null operand");
    VA_CHECK_ARG(num_chunks > 0 && num_chunks < (1ll << 31), "va_multi_adamw: bad chunk count %lld", (long long)num_chunks);
    VA_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "va_multi_adamw: bad hyper-parameter");
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.decay = 1.0f - lr * weight_decay;
    a.max_norm = max_norm;
    hipLaunchKernelGGL(multi_adamw_kernel, dim3((unsigned)num_chunks), dim3(256), 0, stream, tensors, chunk_table, a,
                       total_sumsq);
    VA_CHECK_LAUNCH("va_multi_adamw");
    return VA_OK;
}
