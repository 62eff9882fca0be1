// Loss heads of the stage-1 step (stage1_surgvu_pretraining.py:426-462), value and gradient in one call each:
//   * symmetric InfoNCE between two sets of projections (s1:434-446): row-normalise, logits = z1 z2^T / tau,
//     0.5 * (CE(logits, diag) + CE(logits^T, diag));
//   * cross-entropy with integer labels over a few classes (s1:428: the two-way order classifier);
//   * mean squared error (s1:459).
// Everything is f32 and deterministic: fixed summation orders, no atomics.  The problems are tiny (B rows of 128 or 2
// numbers, B x 1024 for the MSE); what the kernels buy is launch count -- one to three launches instead of the ten to
// twenty elementwise / reduction / GEMM launches of the formulas written out in torch, forward and backward.
#include "common.h"

namespace {

VA_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
VA_DEVINL float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
// sum / max over the 256 threads of a block; every thread gets the result.  `red` holds 4 floats.
VA_DEVINL float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
VA_DEVINL float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- InfoNCE -------------------------------------------------------------------------------------------
// Pass 1, one block per row: z[r] = p[r] / max(|p[r]|, eps) (or a copy of p when the rows come normalised), inv[r] the
// factor.  grid (B, 2): y selects the set.
__global__ __launch_bounds__(256) void nce_normalize_kernel(const float* __restrict__ P1, const float* __restrict__ P2,
                                                            float* __restrict__ Z, float* __restrict__ INV, int B, int D,
                                                            int normalize) {
    __shared__ float red[4];
    const int r = blockIdx.x, set = blockIdx.y;
    const float* p = (set ? P2 : P1) + (int64_t)r * D;
    float* z = Z + ((int64_t)set * B + r) * D;
    float ss = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) ss = fmaf(p[d], p[d], ss);
    ss = block_sum(ss, red);
    const float inv = normalize ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;      // F.normalize: x / max(||x||, eps)
    for (int d = threadIdx.x; d < D; d += 256) z[d] = p[d] * inv;
    if (threadIdx.x == 0) INV[set * B + r] = inv;
}

// Rows of one matrix: forward as above; backward dp = (dz - z (z . dz)) * inv, the same arithmetic as the fused form's.
__global__ __launch_bounds__(256) void rows_normalize_bwd_kernel(const float* __restrict__ DZ, const float* __restrict__ Z,
                                                                 const float* __restrict__ INV, float* __restrict__ DP, int D) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float* dz = DZ + (int64_t)r * D;
    const float* z = Z + (int64_t)r * D;
    float dot = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) dot = fmaf(dz[d], z[d], dot);
    dot = block_sum(dot, red);
    const float inv = INV[r];
    for (int d = threadIdx.x; d < D; d += 256) DP[(int64_t)r * D + d] = (dz[d] - z[d] * dot) * inv;
}

// Pass 2, one block per (row i, direction): the logits of row i against every row j of the other set, their
// log-sum-exp, the diagonal one, and the softmax mass OFF the diagonal (summed as such: "softmax_ii - 1" formed from it
// has no cancellation).  LSE[dir * B + i], OFF[dir * B + i], DIAG[i] (direction 0 only).
__global__ __launch_bounds__(256) void nce_lse_kernel(const float* __restrict__ Z, float* __restrict__ LSE,
                                                      float* __restrict__ OFF, float* __restrict__ DIAG, int B, int D,
                                                      float inv_tau) {
    extern __shared__ float sh[];        // D floats: the row; then 4
    float* row = sh;
    float* red = sh + D;
    const int i = blockIdx.x, dir = blockIdx.y;
    const float* za = Z + ((int64_t)dir * B + i) * D;
    const float* zb = Z + (int64_t)(dir ^ 1) * B * D;
    for (int d = threadIdx.x; d < D; d += 256) row[d] = za[d];
    __syncthreads();
    // two sweeps over j (max, then sum of exponentials): the logits are recomputed, not stored
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < B; j += 256) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(row[d], zb[(int64_t)j * D + d], s);
        s *= inv_tau;
        mx = fmaxf(mx, s);
        if (j == i && dir == 0) DIAG[i] = s;
    }
    mx = block_max(mx, red);
    float se = 0.f, so = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(row[d], zb[(int64_t)j * D + d], s);
        const float e = expf(s * inv_tau - mx);
        se += e;
        if (j != i) so += e;
    }
    se = block_sum(se, red);
    so = block_sum(so, red);
    if (threadIdx.x == 0) {
        LSE[dir * B + i] = mx + logf(se);
        OFF[dir * B + i] = so / se;
    }
}

// Pass 3, one block per (row i, direction): gradient with respect to row i of set `dir`.
//   dS_ij = (softmax_row(S)_ij + softmax_col(S)_ij - 2 delta_ij) * 0.5 / B          (S = set-0 rows x set-1 rows)
//   dz_i = sum_j dS_ij zb_j / tau   (for direction 1: sum over the set-0 rows with the transposed coefficient)
//   dp_i = (dz_i - z_i (z_i . dz_i)) * inv_i                                         (through the normalisation)
// Block 0 of direction 0 also writes the loss: 0.5 / B * sum_i (lse_row_i + lse_col_i - 2 diag_i), in index order.
__global__ __launch_bounds__(256) void nce_grad_kernel(const float* __restrict__ Z, const float* __restrict__ INV,
                                                       const float* __restrict__ LSE, const float* __restrict__ OFF,
                                                       const float* __restrict__ DIAG,
                                                       float* __restrict__ G1, float* __restrict__ G2,
                                                       float* __restrict__ LOSS, int B, int D, float inv_tau, int normalize) {
    extern __shared__ float sh[];        // D (row) | B (coefficients) | D (dz) | 4
    float* row = sh;
    float* w = sh + D;
    float* dz = w + B;
    float* red = dz + D;
    const int i = blockIdx.x, dir = blockIdx.y;
    const float* za = Z + ((int64_t)dir * B + i) * D;
    const float* zb = Z + (int64_t)(dir ^ 1) * B * D;
    for (int d = threadIdx.x; d < D; d += 256) row[d] = za[d];
    __syncthreads();
    const float lse_own = LSE[dir * B + i];
    for (int j = threadIdx.x; j < B; j += 256) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(row[d], zb[(int64_t)j * D + d], s);
        s *= inv_tau;
        // row i of this direction's logits and column i of the other direction's: the same number s
        float c = expf(s - lse_own) + expf(s - LSE[(dir ^ 1) * B + j]);
        if (j == i) c = -(OFF[dir * B + i] + OFF[(dir ^ 1) * B + i]);      // (softmax_ii - 1) of either direction
        w[j] = c * (0.5f / (float)B) * inv_tau;
    }
    __syncthreads();
    float dot = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int j = 0; j < B; ++j) acc = fmaf(w[j], zb[(int64_t)j * D + d], acc);
        dz[d] = acc;
        dot = fmaf(acc, row[d], dot);
    }
    dot = block_sum(dot, red);
    float* g = (dir ? G2 : G1) + (int64_t)i * D;
    const float inv = INV[dir * B + i];
    for (int d = threadIdx.x; d < D; d += 256) g[d] = normalize ? (dz[d] - row[d] * dot) * inv : dz[d];
    if (i == 0 && dir == 0) {
        float part = 0.f;
        for (int r = threadIdx.x; r < B; r += 256) part += (LSE[r] + LSE[B + r]) - 2.0f * DIAG[r];
        part = block_sum(part, red);
        if (threadIdx.x == 0) LOSS[0] = part * (0.5f / (float)B);
    }
}

// ---- cross-entropy over C classes, one wave per row; PART[row] = lse - logit[label]; the mean is taken by the
// reduction kernel below ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ X, const int64_t* __restrict__ LABEL,
                                                 float* __restrict__ PART, float* __restrict__ DX, int B, int C, float gscale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;
    const float* x = X + (int64_t)r * C;
    const int lab = (int)LABEL[r];
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(x[c] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    for (int c = lane; c < C; c += 64) DX[(int64_t)r * C + c] = (expf(x[c] - lse) - (c == lab ? 1.0f : 0.0f)) * gscale;
    if (lane == 0) PART[r] = lse - x[lab];
}

// ---- mean squared error: per-block partial sums in index order, gradient 2 (x - y) / n ------------------------
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                  float* __restrict__ PART, float* __restrict__ DX, int64_t n, float gscale) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 + 4 <= n) {
        const f32x4_t x = *(const f32x4_t*)(X + i0), y = *(const f32x4_t*)(Y + i0);
        f32x4_t g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = x[e] - y[e];
            acc = fmaf(d, d, acc);
            g[e] = d * gscale;
        }
        *(f32x4_t*)(DX + i0) = g;
    } else {
        for (int64_t i = i0; i < n; ++i) {
            const float d = X[i] - Y[i];
            acc = fmaf(d, d, acc);
            DX[i] = d * gscale;
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) PART[blockIdx.x] = acc;
}

// out[0] = scale * sum of part[0 .. n) in index order (one block)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ PART, float* __restrict__ OUT, int64_t n,
                                                        float scale) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += PART[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) OUT[0] = acc * scale;
}

}  // namespace

extern "C" int64_t va_infonce_workspace_floats(int64_t B, int64_t D) { return B > 0 && D > 0 ? 2 * B * D + 7 * B : 0; }

extern "C" int va_infonce_sym(const float* p1, const float* p2, float* loss, float* g1, float* g2, float* workspace,
                              int64_t B, int64_t D, float temperature, int normalize, hipStream_t stream) {
    VA_CHECK_ARG(p1 && p2 && loss && g1 && g2 && workspace, "va_infonce_sym: null operand");
    VA_CHECK_ARG(B > 0 && B <= 8192 && D > 0 && D <= 2048, "va_infonce_sym: B=%lld (1..8192) or D=%lld (1..2048) out of range",
                 (long long)B, (long long)D);
    VA_CHECK_ARG(temperature > 0.f, "va_infonce_sym: temperature must be positive");
    float* Z = workspace;
    float* INV = Z + 2 * B * D;
    float* LSE = INV + 2 * B;
    float* OFF = LSE + 2 * B;
    float* DIAG = OFF + 2 * B;
    const float inv_tau = 1.0f / temperature;
    const dim3 grid((unsigned)B, 2);
    hipLaunchKernelGGL(nce_normalize_kernel, grid, dim3(256), 0, stream, p1, p2, Z, INV, (int)B, (int)D, normalize);
    hipLaunchKernelGGL(nce_lse_kernel, grid, dim3(256), (size_t)(D + 4) * 4, stream, Z, LSE, OFF, DIAG, (int)B, (int)D, inv_tau);
    hipLaunchKernelGGL(nce_grad_kernel, grid, dim3(256), (size_t)(2 * D + B + 4) * 4, stream, Z, INV, LSE, OFF, DIAG, g1, g2, loss,
                       (int)B, (int)D, inv_tau, normalize);
    VA_CHECK_LAUNCH("va_infonce_sym");
    return VA_OK;
}

extern "C" int va_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits, float* workspace,
                                int64_t B, int64_t C, float grad_scale, hipStream_t stream) {
    VA_CHECK_ARG(logits && labels && loss && dlogits && workspace, "va_cross_entropy: null operand");
    VA_CHECK_ARG(B > 0 && B < (1ll << 24) && C > 0 && C <= 65536, "va_cross_entropy: shape out of range");
    hipLaunchKernelGGL(ce_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, logits, labels, workspace, dlogits, (int)B,
                       (int)C, grad_scale);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, stream, workspace, loss, B, 1.0f / (float)B);
    VA_CHECK_LAUNCH("va_cross_entropy");
    return VA_OK;
}

extern "C" int64_t va_mse_workspace_floats(int64_t n) { return n > 0 ? (n + 1023) / 1024 : 0; }

extern "C" int va_mse_loss(const float* x, const float* y, float* loss, float* dx, float* workspace, int64_t n,
                           float grad_scale, hipStream_t stream) {
    VA_CHECK_ARG(x && y && loss && dx && workspace, "va_mse_loss: null operand");
    VA_CHECK_ARG(n > 0 && n < (1ll << 38), "va_mse_loss: element count out of range");
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)y | (uintptr_t)dx) % 16 == 0, "va_mse_loss: misaligned operand");
    const int64_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(mse_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, workspace, dx, n, grad_scale);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, stream, workspace, loss, blocks, 1.0f / (float)n);
    VA_CHECK_LAUNCH("va_mse_loss");
    return VA_OK;
}

extern "C" int va_normalize_rows_fwd(const float* p, float* z, float* inv, int64_t B, int64_t D, hipStream_t stream) {
    VA_CHECK_ARG(p && z && inv, "va_normalize_rows_fwd: null operand");
    VA_CHECK_ARG(B > 0 && B < (1ll << 24) && D > 0 && D < (1ll << 24), "va_normalize_rows_fwd: shape out of range");
    hipLaunchKernelGGL(nce_normalize_kernel, dim3((unsigned)B, 1), dim3(256), 0, stream, p, p, z, inv, (int)B, (int)D, 1);
    VA_CHECK_LAUNCH("va_normalize_rows_fwd");
    return VA_OK;
}

extern "C" int va_normalize_rows_bwd(const float* dz, const float* z, const float* inv, float* dp, int64_t B, int64_t D,
                                     hipStream_t stream) {
    VA_CHECK_ARG(dz && z && inv && dp, "va_normalize_rows_bwd: null operand");
    VA_CHECK_ARG(B > 0 && B < (1ll << 24) && D > 0 && D < (1ll << 24), "va_normalize_rows_bwd: shape out of range");
    hipLaunchKernelGGL(rows_normalize_bwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, dz, z, inv, dp, (int)D);
    VA_CHECK_LAUNCH("va_normalize_rows_bwd");
    return VA_OK;
}
