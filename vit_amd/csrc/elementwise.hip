// HBM-bound kernels of the encoder path: LayerNorm forward/backward, 3D-RoPE rotation, tubelet patch
// gather, parameter casts, bias-gradient column sums, token masking / gather / scatter, EMA update and
// the L1 latent loss.  All of them move 16 bytes per lane per access and keep no state.
#include "common.h"
#include <mutex>
#include <stdlib.h>

namespace {

VA_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

VA_DEVINL void unpack8(const u32x4_t& v, float* f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(v[j] << 16);
        f[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
    }
}
VA_DEVINL u32x4_t pack8(const float* f) {
    return (u32x4_t){pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over rows of D = 1024 f32: one wave per row, 16 elements per lane (4 x float4, stride 256).
// ------------------------------------------------------------------------------------------------
constexpr int LN_D = 1024;

template <bool OUT_BF16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t M,
                                                     float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * LN_D;
    f32x4_t v[4];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        v[g] = *(const f32x4_t*)(xr + g * 256 + lane * 4);
        s += v[g][0] + v[g][1] + v[g][2] + v[g][3];
    }
    const float mu = wave_sum(s) * (1.0f / LN_D);
    float ss = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[g][j] - mu;
            ss += d * d;
        }
    const float rs = rsqrtf(wave_sum(ss) * (1.0f / LN_D) + eps);
    if (lane == 0 && mean != nullptr) {      // a forward that keeps nothing for a backward passes no statistics buffers
        mean[row] = mu;
        rstd[row] = rs;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = g * 256 + lane * 4;
        const f32x4_t gm = *(const f32x4_t*)(gamma + c);
        const f32x4_t bt = *(const f32x4_t*)(beta + c);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[g][j] - mu) * rs * gm[j] + bt[j];
        if (OUT_BF16) {
            *(u32x2_t*)((bf16_t*)y + row * LN_D + c) = (u32x2_t){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        } else {
            *(f32x4_t*)((float*)y + row * LN_D + c) = (f32x4_t){o[0], o[1], o[2], o[3]};
        }
    }
}

constexpr int LN_BWD_BLOCKS = 512;    // upper bound on the number of partial rows

// Each wave walks rows wave_id, wave_id + n_waves, ...; dgamma/dbeta partial sums stay in registers,
// are combined across the 4 waves of the block through LDS and written as one partial row per block.
template <bool DY_BF16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ resid_grad, float* __restrict__ dx_out,
                                                     bf16_t* __restrict__ dx_bf16, float* __restrict__ partial,
                                                     int64_t M, uint32_t drop_key, uint32_t drop_thi, float drop_scale,
                                                     int nsets) {
    // nsets = 3: besides dgamma / dbeta the column sums of the bf16 copy are produced (the bias gradient of the
    // linear layer whose gradient GEMMs read that copy)
    __shared__ float red[3][4][LN_D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4_t gm[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gm[g] = *(const f32x4_t*)(gamma + g * 256 + lane * 4);
    float dg[16], db[16], dc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dg[i] = 0.f; db[i] = 0.f; dc[i] = 0.f; }

    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[16], dyv[16];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = g * 256 + lane * 4;
            const f32x4_t xv = *(const f32x4_t*)(x + row * LN_D + c);
            float d4[4];
            if (DY_BF16) {
                const u32x2_t pk = *(const u32x2_t*)((const bf16_t*)dy + row * LN_D + c);
                d4[0] = __uint_as_float(pk[0] << 16); d4[1] = __uint_as_float(pk[0] & 0xffff0000u);
                d4[2] = __uint_as_float(pk[1] << 16); d4[3] = __uint_as_float(pk[1] & 0xffff0000u);
            } else {
                const f32x4_t dv = *(const f32x4_t*)((const float*)dy + row * LN_D + c);
                d4[0] = dv[0]; d4[1] = dv[1]; d4[2] = dv[2]; d4[3] = dv[3];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h = (xv[j] - mu) * rs;
                xh[4 * g + j] = h;
                dyv[4 * g + j] = d4[j];
                dg[4 * g + j] += d4[j] * h;
                db[4 * g + j] += d4[j];
                const float t = d4[j] * gm[g][j];
                s1 += t;
                s2 += t * h;
            }
        }
        s1 = wave_sum(s1) * (1.0f / LN_D);
        s2 = wave_sum(s2) * (1.0f / LN_D);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = g * 256 + lane * 4;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rs * (dyv[4 * g + j] * gm[g][j] - s1 - xh[4 * g + j] * s2);
            if (resid_grad != nullptr) {
                const f32x4_t r = *(const f32x4_t*)(resid_grad + row * LN_D + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += r[j];
            }
            *(f32x4_t*)(dx_out + row * LN_D + c) = (f32x4_t){o[0], o[1], o[2], o[3]};
            if (dx_bf16 != nullptr) {
                if (drop_thi) {   // the bf16 copy feeds a branch behind a dropout: it carries that dropout's mask
                    const uint32_t w0 = drop_weyl(drop_key, (uint32_t)((row * LN_D + c) >> 1));
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t hsh = drop_mix(w0 + (uint32_t)q * DROP_WEYL);
                        o[2 * q] = drop_keep_even(hsh, drop_thi) ? __fmul_rn(o[2 * q], drop_scale) : 0.f;
                        o[2 * q + 1] = drop_keep_odd(hsh, drop_thi) ? __fmul_rn(o[2 * q + 1], drop_scale) : 0.f;
                    }
                }
                const u32x2_t pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                *(u32x2_t*)(dx_bf16 + row * LN_D + c) = pk;
                if (nsets == 3) {   // sums of the ROUNDED values: what a GEMM over the copy would add up
                    dc[4 * g + 0] += __uint_as_float(pk[0] << 16); dc[4 * g + 1] += __uint_as_float(pk[0] & 0xffff0000u);
                    dc[4 * g + 2] += __uint_as_float(pk[1] << 16); dc[4 * g + 3] += __uint_as_float(pk[1] & 0xffff0000u);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][wave][g * 256 + lane * 4 + j] = dg[4 * g + j];
            red[1][wave][g * 256 + lane * 4 + j] = db[4 * g + j];
            if (nsets == 3) red[2][wave][g * 256 + lane * 4 + j] = dc[4 * g + j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < LN_D; c += 256)
        for (int w = 0; w < nsets; ++w)
            partial[((int64_t)blockIdx.x * nsets + w) * LN_D + c] = red[w][0][c] + red[w][1][c] + red[w][2][c] + red[w][3][c];
}

// dgamma[c] = sum_b partial[b][0][c], dbeta[c] = sum_b partial[b][1][c], colsum[c] = sum_b partial[b][2][c].
// Block = 16 columns x 16 row lanes; grid = nsets * D / 16 blocks (one run of blocks per output).
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nblk, int nsets,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ colsum) {
    __shared__ float red[16][17];
    const int which = blockIdx.x / (LN_D / 16);
    const int c = (blockIdx.x % (LN_D / 16)) * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
    float a = 0.f;
    for (int r = part; r < nblk; r += 16) a += partial[((int64_t)r * nsets + which) * LN_D + c];
    red[part][threadIdx.x & 15] = a;
    __syncthreads();
    if (part == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
        (which == 0 ? dgamma : which == 1 ? dbeta : colsum)[c] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// 3D RoPE on the Q and K thirds of a packed [M, 3072] bf16 buffer, in place.
// One wave per (row, q|k): 512 rotation pairs, 8 per lane.  Pair index pi in [0,512): segment t for
// pi<128 (columns pi, pi+128), h for pi<320 (256+j, 448+j), w else (640+j, 832+j).
// tables: cos [Tp*128 | Hp*192 | Wp*192] then sin with the same layout.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ tables,
                                                   int64_t M, int Tp, int Hp, int Wp, float sin_sign) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 2 + (wave >> 1);
    if (row >= M) return;
    const int which = wave & 1;   // 0 = Q, 1 = K
    const int n = (int)(row % ((int64_t)Tp * Hp * Wp));
    const int t = n / (Hp * Wp), hh = (n / Wp) % Hp, w = n % Wp;
    const int tab_sz = Tp * 128 + Hp * 192 + Wp * 192;
    const int pi = lane * 8;
    int col_a, half, toff;
    if (pi < 128) { col_a = pi; half = 128; toff = t * 128 + pi; }
    else if (pi < 320) { col_a = 256 + (pi - 128); half = 192; toff = Tp * 128 + hh * 192 + (pi - 128); }
    else { col_a = 640 + (pi - 320); half = 192; toff = Tp * 128 + Hp * 192 + w * 192 + (pi - 320); }
    bf16_t* base = qkv + row * 3072 + which * 1024;
    float a[8], b[8], c[8], s[8];
    unpack8(*(const u32x4_t*)(base + col_a), a);
    unpack8(*(const u32x4_t*)(base + col_a + half), b);
    *(f32x4_t*)(c) = *(const f32x4_t*)(tables + toff);
    *(f32x4_t*)(c + 4) = *(const f32x4_t*)(tables + toff + 4);
    *(f32x4_t*)(s) = *(const f32x4_t*)(tables + tab_sz + toff);
    *(f32x4_t*)(s + 4) = *(const f32x4_t*)(tables + tab_sz + toff + 4);
    float oa[8], ob[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float sn = s[j] * sin_sign;
        oa[j] = a[j] * c[j] - b[j] * sn;
        ob[j] = b[j] * c[j] + a[j] * sn;
    }
    *(u32x4_t*)(base + col_a) = pack8(oa);
    *(u32x4_t*)(base + col_a + half) = pack8(ob);
}

// The same rotation in its general form (RoPEMultiheadAttention.apply_rope_to_qk, swm:265-281, called with an
// arbitrary cache and arbitrary token positions): full-width cos / sin tables per segment ([len, 256], [len, 384],
// [len, 384] f32, the layout VideoRoPE3D.forward returns) and explicit per-token coordinates pos[3][N] (t, h, w).
// The two halves of a table row are NOT assumed equal:  y_a = x_a*cos_a - x_b*sin_a,  y_b = x_b*cos_b + x_a*sin_b;
// transpose != 0 applies the transposed map (the gradient):  dx_a = dy_a*cos_a + dy_b*sin_b,  dx_b = dy_b*cos_b - dy_a*sin_a.
// Lane mapping as rope_kernel.
__global__ __launch_bounds__(256) void rope_indexed_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cos_t,
                                                           const float* __restrict__ sin_t, const float* __restrict__ cos_h,
                                                           const float* __restrict__ sin_h, const float* __restrict__ cos_w,
                                                           const float* __restrict__ sin_w, const int32_t* __restrict__ pos,
                                                           int64_t M, int N, int transpose) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 2 + (wave >> 1);
    if (row >= M) return;
    const int which = wave & 1;   // 0 = Q, 1 = K
    const int n = (int)(row % N);
    const int pi = lane * 8;
    int col_a, half;
    const float *ct, *st;
    if (pi < 128) { col_a = pi; half = 128; ct = cos_t + (int64_t)pos[n] * 256 + pi; st = sin_t + (int64_t)pos[n] * 256 + pi; }
    else if (pi < 320) { col_a = 256 + (pi - 128); half = 192; ct = cos_h + (int64_t)pos[N + n] * 384 + (pi - 128); st = sin_h + (int64_t)pos[N + n] * 384 + (pi - 128); }
    else { col_a = 640 + (pi - 320); half = 192; ct = cos_w + (int64_t)pos[2 * N + n] * 384 + (pi - 320); st = sin_w + (int64_t)pos[2 * N + n] * 384 + (pi - 320); }
    bf16_t* base = qkv + row * 3072 + which * 1024;
    float a[8], b[8], ca[8], cb[8], sa[8], sb[8];
    unpack8(*(const u32x4_t*)(base + col_a), a);
    unpack8(*(const u32x4_t*)(base + col_a + half), b);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        *(f32x4_t*)(ca + 4 * q) = *(const f32x4_t*)(ct + 4 * q);
        *(f32x4_t*)(cb + 4 * q) = *(const f32x4_t*)(ct + half + 4 * q);
        *(f32x4_t*)(sa + 4 * q) = *(const f32x4_t*)(st + 4 * q);
        *(f32x4_t*)(sb + 4 * q) = *(const f32x4_t*)(st + half + 4 * q);
    }
    float oa[8], ob[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (transpose) {
            oa[j] = a[j] * ca[j] + b[j] * sb[j];
            ob[j] = b[j] * cb[j] - a[j] * sa[j];
        } else {
            oa[j] = a[j] * ca[j] - b[j] * sa[j];
            ob[j] = b[j] * cb[j] + a[j] * sb[j];
        }
    }
    *(u32x4_t*)(base + col_a) = pack8(oa);
    *(u32x4_t*)(base + col_a + half) = pack8(ob);
}

// ------------------------------------------------------------------------------------------------
// tubelet gather: one thread per 8 consecutive pixels of a patch row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tubelet_gather_kernel(const float* __restrict__ video,
                                                             bf16_t* __restrict__ patches, int64_t total_chunks, int C,
                                                             int T, int H, int W, int tt, int th, int tw) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_chunks) return;
    const int Tp = T / tt, Hp = H / th, Wp = W / tw;
    const int kdim = C * tt * th * tw, cpr = kdim / 8;
    const int64_t token = gid / cpr;
    const int col = (int)(gid % cpr) * 8;
    const int dw = col % tw, dh = (col / tw) % th, dt = (col / (tw * th)) % tt, c = col / (tw * th * tt);
    const int ntok = Tp * Hp * Wp;
    const int64_t b = token / ntok;
    const int n = (int)(token % ntok);
    const int tp = n / (Hp * Wp), hp = (n / Wp) % Hp, wp = n % Wp;
    const float* src = video + ((((b * C + c) * T + (tp * tt + dt)) * H + (hp * th + dh)) * (int64_t)W + wp * tw + dw);
    float f[8];
    *(f32x4_t*)(f) = *(const f32x4_t*)(src);
    *(f32x4_t*)(f + 4) = *(const f32x4_t*)(src + 4);
    *(u32x4_t*)(patches + token * kdim + col) = pack8(f);
}

// The inverse placement for the input gradient: patch-matrix rows (f32) back to (B,C,T,H,W); one thread per 4
// consecutive pixels of a patch row.  Tubelets do not overlap, so every covered pixel is written exactly once.
__global__ __launch_bounds__(256) void tubelet_scatter_kernel(const float* __restrict__ patches, float* __restrict__ video,
                                                              int64_t total_chunks, int C, int T, int H, int W, int tt,
                                                              int th, int tw) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_chunks) return;
    const int Tp = T / tt, Hp = H / th, Wp = W / tw;
    const int kdim = C * tt * th * tw, cpr = kdim / 4;
    const int64_t token = gid / cpr;
    const int col = (int)(gid % cpr) * 4;
    const int dw = col % tw, dh = (col / tw) % th, dt = (col / (tw * th)) % tt, c = col / (tw * th * tt);
    const int ntok = Tp * Hp * Wp;
    const int64_t b = token / ntok;
    const int n = (int)(token % ntok);
    const int tp = n / (Hp * Wp), hp = (n / Wp) % Hp, wp = n % Wp;
    float* dst = video + ((((b * C + c) * T + (tp * tt + dt)) * H + (hp * th + dh)) * (int64_t)W + wp * tw + dw);
    *(f32x4_t*)dst = *(const f32x4_t*)(patches + token * kdim + col);
}

// ------------------------------------------------------------------------------------------------
// casts
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float f[8];
        *(f32x4_t*)(f) = *(const f32x4_t*)(src + i);
        *(f32x4_t*)(f + 4) = *(const f32x4_t*)(src + i + 4);
        *(u32x4_t*)(dst + i) = pack8(f);
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
    }
}

// w[R,C] f32 -> wt[C,R] bf16 through a padded 32x32 LDS tile; block (32, 8).
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ w, bf16_t* __restrict__ wt, int R,
                                                             int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[ty + 8 * j][tx] = w[(int64_t)(r0 + ty + 8 * j) * C + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) wt[(int64_t)(c0 + ty + 8 * j) * R + r0 + tx] = f32_to_bf16(tile[tx][ty + 8 * j]);
}

// Many matrices in one launch: every f32 [rows, cols] source gets a bf16 copy and (optionally) a bf16 transposed
// copy from ONE read.  A block handles one 64 x 64 tile; it finds its matrix by bisecting the jobs' first-tile
// numbers.  Loads and both stores move 128 or 256 contiguous bytes per 16 lanes; the transpose goes through LDS.
__global__ __launch_bounds__(256) void multi_cast_kernel(const va_cast_job* __restrict__ jobs, int njobs) {
    __shared__ float tile[64][65];
    const int64_t b = blockIdx.x;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                      // last job whose first_tile <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_tile <= b) lo = mid;
        else hi = mid - 1;
    }
    const va_cast_job job = jobs[lo];
    const int tiles_c = job.cols / 64;
    const int64_t local = b - job.first_tile;
    const int r0 = (int)(local / tiles_c) * 64, c0 = (int)(local % tiles_c) * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    bf16_t* dst = (bf16_t*)job.dst;
    bf16_t* dst_t = (bf16_t*)job.dst_t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = ty + 16 * k;
        const f32x4_t v = *(const f32x4_t*)(job.src + (int64_t)(r0 + r) * job.cols + c0 + tx * 4);
        if (dst != nullptr)
            *(u32x2_t*)(dst + (int64_t)(r0 + r) * job.cols + c0 + tx * 4) = (u32x2_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        if (dst_t != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[r][tx * 4 + j] = v[j];
        }
    }
    if (dst_t == nullptr) return;          // uniform over the block
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = ty + 16 * k;         // source column = destination row
        const u32x2_t pk = {pack_bf16x2(tile[tx * 4][c], tile[tx * 4 + 1][c]), pack_bf16x2(tile[tx * 4 + 2][c], tile[tx * 4 + 3][c])};
        *(u32x2_t*)(dst_t + (int64_t)(c0 + c) * job.rows + r0 + tx * 4) = pk;
    }
}

// ------------------------------------------------------------------------------------------------
// column sums of a bf16 matrix (bias gradients): block = 64 column groups (8 cols each) x 4 row lanes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ y, int64_t M, int N, int64_t ldy,
                                                     float* __restrict__ out) {
    __shared__ float red[4][64][8];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + cg * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (col < N) {
        for (int64_t r = (int64_t)blockIdx.y * 4 + rl; r < M; r += (int64_t)gridDim.y * 4) {
            float f[8];
            unpack8(*(const u32x4_t*)(y + r * ldy + col), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cg][j] = acc[j];
    __syncthreads();
    if (rl == 0 && col < N) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            unsafeAtomicAdd(out + col + j, red[0][cg][j] + red[1][cg][j] + red[2][cg][j] + red[3][cg][j]);
    }
}

// ------------------------------------------------------------------------------------------------
// token masking / gather / scatter: rows are opaque byte strings moved 16 bytes per lane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void future_mask_kernel(const u32x4_t* __restrict__ x, u32x4_t* __restrict__ out,
                                                          uint8_t* __restrict__ mask, int64_t rows, int N,
                                                          int chunks_per_row, int first_masked) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * chunks_per_row) return;
    const int64_t row = gid / chunks_per_row;
    const bool masked = (int)(row % N) >= first_masked;
    out[gid] = masked ? (u32x4_t){0u, 0u, 0u, 0u} : x[gid];
    if (gid % chunks_per_row == 0) mask[row] = masked ? 1 : 0;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void token_move_kernel(const u32x4_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                         u32x4_t* __restrict__ dst, int64_t total_rows, int n_keep,
                                                         int n_full, int chunks_per_row) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_rows * chunks_per_row) return;
    const int64_t r = gid / chunks_per_row;           // row in the compact (B, n_keep) space
    const int c = (int)(gid % chunks_per_row);
    const int64_t b = r / n_keep;
    const int32_t j = idx[r];
    if ((uint32_t)j >= (uint32_t)n_full) {            // index outside [0, n_full): never touch memory outside the row range
        if (!SCATTER) dst[gid] = (u32x4_t){0u, 0u, 0u, 0u};
        return;
    }
    const int64_t full_row = b * n_full + j;
    if (SCATTER) dst[full_row * chunks_per_row + c] = src[gid];
    else dst[gid] = src[full_row * chunks_per_row + c];
}

// ------------------------------------------------------------------------------------------------
// EMA and L1
// ------------------------------------------------------------------------------------------------
// UNROLL independent 16-byte pieces per stream are requested before the first is used (a thread keeps
// 2 * UNROLL * 16 bytes in flight); the streams are touched once, so loads and stores carry the non-temporal hint.
template <int UNROLL>
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ tgt, const float* __restrict__ onl,
                                                  bf16_t* __restrict__ shadow, int64_t n, float m) {
    const float om = 1.0f - m;
    const int64_t step = (int64_t)gridDim.x * 1024 * UNROLL;
    for (int64_t i0 = ((int64_t)blockIdx.x * UNROLL * 256 + threadIdx.x) * 4; i0 < n; i0 += step) {
        f32x4_t t[UNROLL], o[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = i0 + (int64_t)u * 1024;
            if (i + 4 <= n) {
                t[u] = __builtin_nontemporal_load((const f32x4_t*)(tgt + i));
                o[u] = __builtin_nontemporal_load((const f32x4_t*)(onl + i));
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = i0 + (int64_t)u * 1024;
            if (i + 4 <= n) {
                const f32x4_t r = {m * t[u][0] + om * o[u][0], m * t[u][1] + om * o[u][1], m * t[u][2] + om * o[u][2],
                                   m * t[u][3] + om * o[u][3]};
                __builtin_nontemporal_store(r, (f32x4_t*)(tgt + i));
                if (shadow) *(u32x2_t*)(shadow + i) = (u32x2_t){pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
            } else if (i < n) {
                for (int64_t j = i; j < n; ++j) {
                    const float r = m * tgt[j] + om * onl[j];
                    tgt[j] = r;
                    if (shadow) shadow[j] = f32_to_bf16(r);
                }
            }
        }
    }
}

// A thread handles L1_CHUNKS pieces of 8 elements per iteration, all of their loads requested before the first use.
constexpr int L1_CHUNKS = 4;
template <bool IN_BF16, bool GRAD_BF16>
__global__ __launch_bounds__(256) void l1_kernel(const void* __restrict__ pred, const void* __restrict__ tgt,
                                                 float* __restrict__ loss_sum, void* __restrict__ grad, float gscale,
                                                 int64_t n) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t step = (int64_t)gridDim.x * 2048 * L1_CHUNKS;
    for (int64_t i0 = ((int64_t)blockIdx.x * L1_CHUNKS * 256 + threadIdx.x) * 8; i0 < n; i0 += step) {
        float p[L1_CHUNKS][8], t[L1_CHUNKS][8];
#pragma unroll
        for (int u = 0; u < L1_CHUNKS; ++u) {
            const int64_t i = i0 + (int64_t)u * 2048;
            if (i + 8 <= n) {
                if (IN_BF16) {
                    unpack8(__builtin_nontemporal_load((const u32x4_t*)((const bf16_t*)pred + i)), p[u]);
                    unpack8(__builtin_nontemporal_load((const u32x4_t*)((const bf16_t*)tgt + i)), t[u]);
                } else {
                    *(f32x4_t*)(p[u]) = __builtin_nontemporal_load((const f32x4_t*)((const float*)pred + i));
                    *(f32x4_t*)(p[u] + 4) = __builtin_nontemporal_load((const f32x4_t*)((const float*)pred + i + 4));
                    *(f32x4_t*)(t[u]) = __builtin_nontemporal_load((const f32x4_t*)((const float*)tgt + i));
                    *(f32x4_t*)(t[u] + 4) = __builtin_nontemporal_load((const f32x4_t*)((const float*)tgt + i + 4));
                }
            } else {
                for (int j = 0; j < 8; ++j) {
                    p[u][j] = 0.f;
                    t[u][j] = 0.f;
                    if (i + j < n) {
                        p[u][j] = IN_BF16 ? bf16_to_f32(((const bf16_t*)pred)[i + j]) : ((const float*)pred)[i + j];
                        t[u][j] = IN_BF16 ? bf16_to_f32(((const bf16_t*)tgt)[i + j]) : ((const float*)tgt)[i + j];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < L1_CHUNKS; ++u) {
            const int64_t i = i0 + (int64_t)u * 2048;
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = p[u][j] - t[u][j];
                acc += fabsf(d);
                g[j] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
            }
            if (grad != nullptr) {
                if (i + 8 <= n) {
                    if (GRAD_BF16) __builtin_nontemporal_store(pack8(g), (u32x4_t*)((bf16_t*)grad + i));
                    else {
                        __builtin_nontemporal_store(*(f32x4_t*)(g), (f32x4_t*)((float*)grad + i));
                        __builtin_nontemporal_store(*(f32x4_t*)(g + 4), (f32x4_t*)((float*)grad + i + 4));
                    }
                } else {
                    for (int j = 0; j < 8 && i + j < n; ++j) {
                        if (GRAD_BF16) ((bf16_t*)grad)[i + j] = f32_to_bf16(g[j]);
                        else ((float*)grad)[i + j] = g[j];
                    }
                }
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// Token pooling of the stage-1 heads (stage1_surgvu_pretraining.py:426-432 `.mean(dim=1)`, :456-460 the mean over the
// unmasked tokens): out[b][d] = scale * sum_{n < n_first} x[b][n][d] over x f32 [B, N, D], D % 4 == 0.
// Stage 1: block (b, part) sums a contiguous run of rows into partial[b][part][:] (a thread owns 4 columns, fixed
// order).  Stage 2: the parts are added in index order.  No atomics: bitwise reproducible.
// ------------------------------------------------------------------------------------------------
constexpr int POOL_PARTS = 64;
__global__ __launch_bounds__(256) void pool_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int N,
                                                           int D, int n_first) {
    const int b = blockIdx.x / POOL_PARTS, part = blockIdx.x % POOL_PARTS;
    const int per = (n_first + POOL_PARTS - 1) / POOL_PARTS;
    const int r0 = part * per, r1 = min(n_first, r0 + per);
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const float* col = x + ((int64_t)b * N) * D + c;
        int r = r0;
        for (; r + 8 <= r1; r += 8) {             // eight rows requested before the first is added
            f32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const f32x4_t*)(col + (int64_t)(r + u) * D));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; r < r1; ++r) acc += *(const f32x4_t*)(col + (int64_t)r * D);
        *(f32x4_t*)(partial + ((int64_t)blockIdx.x) * D + c) = acc;
    }
}
__global__ __launch_bounds__(256) void pool_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                          int64_t BD, int D, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= BD) return;
    const int64_t b = i / D, d = i % D;
    float a = 0.f;
#pragma unroll
    for (int p = 0; p < POOL_PARTS; ++p) a += partial[(b * POOL_PARTS + p) * D + d];
    out[i] = a * scale;
}
// backward: dx[b][n][:] = n < n_first ? scale * g[b][:] : 0   (a write-only stream)
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int64_t total4,
                                                       int N, int D, int n_first, float scale) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x);
    if (i >= total4) return;
    const int d4 = D / 4;
    const int64_t row = i / d4;
    const int c = (int)(i % d4) * 4;
    const int64_t b = row / N;
    const int n = (int)(row % N);
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (n < n_first) {
        v = *(const f32x4_t*)(g + b * D + c);
        v *= scale;
    }
    __builtin_nontemporal_store(v, (f32x4_t*)(dx + row * D + c));
}

inline unsigned blocks_for(int64_t items, int per_block) { return (unsigned)((items + per_block - 1) / per_block); }

// uint8 frames [B, T, H, W, 3] -> f32 clip [B, 3, T, H, W], (u / 255 - mean[c]) / std[c] with the same three
// f32 operations, in the same order, as ToTensor + Normalize (stage1_surgvu_pretraining.py:241-244, 279-281).
// A thread converts 4 neighbouring pixels (12 bytes in, three 16-byte stores out); HW % 4 == 0.
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                           int64_t quads, int64_t frame_px, int64_t T, f32x4_t mean,
                                                           f32x4_t stdv) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;     // quad index over (b, t, pixel/4)
    if (q >= quads) return;
    const int64_t px = q * 4;
    const int64_t bt = px / frame_px, off = px % frame_px;
    const int64_t b = bt / T, t = bt % T;
    const uint32_t* src = (const uint32_t*)(in + px * 3);
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
    uint8_t u[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u[i] = (w0 >> (8 * i)) & 0xff;
        u[4 + i] = (w1 >> (8 * i)) & 0xff;
        u[8 + i] = (w2 >> (8 * i)) & 0xff;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4_t v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ((float)u[3 * i + c] / 255.0f - mean[c]) / stdv[c];
        *(f32x4_t*)(out + ((b * 3 + c) * T + t) * frame_px + off) = v;
    }
}

// out = [resid +] keep(in) * scale over a contiguous [rows, cols] matrix (cols % 8 == 0), element index
// row * cols + col.  in / out independently f32 or bf16; in place allowed.  A thread handles 8 elements.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void dropout_kernel(const TIN* __restrict__ in, TOUT* __restrict__ out,
                                                      const float* __restrict__ resid, int64_t n8, uint32_t key,
                                                      uint32_t thresh16, float scale) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n8) return;
    const int64_t e0 = g * 8;
    float v[8];
    if (sizeof(TIN) == 4) {
        const f32x4_t a = *(const f32x4_t*)((const float*)in + e0), b = *(const f32x4_t*)((const float*)in + e0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    } else {
        const u32x4_t a = *(const u32x4_t*)((const bf16_t*)in + e0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(a[j] << 16); v[2 * j + 1] = __uint_as_float(a[j] & 0xffff0000u); }
    }
    const uint32_t thi = thresh16 << 16;
    const uint32_t w0 = drop_weyl(key, (uint32_t)(e0 / 2));
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // rounded products: never fused with the residual add
        const uint32_t hsh = drop_mix(w0 + (uint32_t)j * DROP_WEYL);
        v[2 * j] = drop_keep_even(hsh, thi) ? __fmul_rn(v[2 * j], scale) : 0.f;
        v[2 * j + 1] = drop_keep_odd(hsh, thi) ? __fmul_rn(v[2 * j + 1], scale) : 0.f;
    }
    if (resid != nullptr) {
        const f32x4_t a = *(const f32x4_t*)(resid + e0), b = *(const f32x4_t*)(resid + e0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += a[j]; v[4 + j] += b[j]; }
    }
    if (sizeof(TOUT) == 4) {
        *(f32x4_t*)((float*)out + e0) = (f32x4_t){v[0], v[1], v[2], v[3]};
        *(f32x4_t*)((float*)out + e0 + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
    } else {
        *(u32x4_t*)((bf16_t*)out + e0) = (u32x4_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
}

}  // namespace

// ===================================================================================================
extern "C" int va_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int out_bf16,
                                float* mean, float* rstd, int64_t M, int64_t D, float eps, hipStream_t stream) {
    VA_CHECK_ARG(x && gamma && beta && y, "va_layernorm_fwd: null operand");
    VA_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "va_layernorm_fwd: mean and rstd go together (both, or neither for a forward-only call)");
    VA_CHECK_ARG(D == LN_D, "va_layernorm_fwd: width %lld not supported (the encoder is 1024 wide)", (long long)D);
    VA_CHECK_ARG(M > 0, "va_layernorm_fwd: empty input");
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0, "va_layernorm_fwd: misaligned operand");
    if (out_bf16)
        hipLaunchKernelGGL(ln_fwd_kernel<true>, dim3(blocks_for(M, 4)), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, M, eps);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3(blocks_for(M, 4)), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, M, eps);
    VA_CHECK_LAUNCH("va_layernorm_fwd");
    return VA_OK;
}

extern "C" int64_t va_layernorm_bwd_partial_rows(int64_t M) {
    const int64_t need = (M + 3) / 4;
    return need < LN_BWD_BLOCKS ? need : LN_BWD_BLOCKS;
}

extern "C" int va_layernorm_bwd_dropout(const void* dy, int dy_bf16, const float* x, const float* gamma,
                                        const float* mean, const float* rstd, const float* resid_grad, float* dx_out,
                                        void* dx_bf16, float* partial, float* dgamma, float* dbeta, int64_t M,
                                        int64_t D, uint32_t key, float p, float* copy_colsum, hipStream_t stream) {
    VA_CHECK_ARG(p >= 0.f && p < 1.f, "va_layernorm_bwd: dropout probability %g outside [0, 1)", (double)p);
    const uint32_t thresh = va_drop_thresh16(p);
    VA_CHECK_ARG(thresh < 65536u, "va_layernorm_bwd: dropout probability %g rounds to 1", (double)p);
    VA_CHECK_ARG(thresh == 0 || (dx_bf16 && M * D < (1ll << 32)), "va_layernorm_bwd: dropout needs the bf16 output and fewer than 2^32 elements");
    VA_CHECK_ARG(!copy_colsum || dx_bf16, "va_layernorm_bwd: column sums of the bf16 copy need the bf16 copy");
    const uint32_t thi = thresh << 16;
    const float dscale = va_drop_scale(thresh);
    const int nsets = copy_colsum ? 3 : 2;
    VA_CHECK_ARG(dy && x && gamma && mean && rstd && dx_out && partial && dgamma && dbeta, "va_layernorm_bwd: null operand");
    VA_CHECK_ARG(D == LN_D, "va_layernorm_bwd: width %lld not supported (the encoder is 1024 wide)", (long long)D);
    VA_CHECK_ARG(M > 0, "va_layernorm_bwd: empty input");
    VA_CHECK_ARG(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx_out | (uintptr_t)gamma) % 16 == 0, "va_layernorm_bwd: misaligned operand");
    const int nblk = (int)va_layernorm_bwd_partial_rows(M);
    if (dy_bf16)
        hipLaunchKernelGGL(ln_bwd_kernel<true>, dim3(nblk), dim3(256), 0, stream, dy, x, gamma, mean, rstd, resid_grad,
                           dx_out, (bf16_t*)dx_bf16, partial, M, key, thi, dscale, nsets);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(nblk), dim3(256), 0, stream, dy, x, gamma, mean, rstd, resid_grad,
                           dx_out, (bf16_t*)dx_bf16, partial, M, key, thi, dscale, nsets);
    VA_CHECK_LAUNCH("va_layernorm_bwd");
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(nsets * LN_D / 16), dim3(256), 0, stream, partial, nblk, nsets, dgamma,
                       dbeta, copy_colsum);
    VA_CHECK_LAUNCH("va_layernorm_bwd(reduce)");
    return VA_OK;
}

extern "C" int va_layernorm_bwd(const void* dy, int dy_bf16, const float* x, const float* gamma, const float* mean,
                                const float* rstd, const float* resid_grad, float* dx_out, void* dx_bf16,
                                float* partial, float* dgamma, float* dbeta, int64_t M, int64_t D, hipStream_t stream) {
    return va_layernorm_bwd_dropout(dy, dy_bf16, x, gamma, mean, rstd, resid_grad, dx_out, dx_bf16, partial, dgamma, dbeta,
                                    M, D, 0u, 0.f, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses that
// forbid hipMalloc): the partial-sum workspace is a file-scope device array instead of a caller buffer.
// The array exists once per GPU the library runs on (6 MiB each, held until the process ends); its address on a
// device is resolved once, with that device current, and kept in `ln_scratch_ptr`.  Host code never uses the array's
// name for anything else: on the host the name is only a handle for hipGetSymbolAddress, not a device address.
// ---------------------------------------------------------------------------------------------------------------
constexpr int VA_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[LN_BWD_BLOCKS * 3 * LN_D];
static float* ln_scratch_ptr[VA_MAX_DEVICES];
static std::mutex ln_scratch_lock;

extern "C" int va_layernorm_bwd_noalloc(const void* dy, int dy_bf16, const float* x, const float* gamma,
                                        const float* mean, const float* rstd, const float* resid_grad, float* dx_out,
                                        void* dx_bf16, float* dgamma, float* dbeta, int64_t M, int64_t D, uint32_t key,
                                        float p, float* copy_colsum, hipStream_t stream) {
    VA_CHECK_ARG(M <= VA_LN_NOALLOC_MAX_ROWS, "va_layernorm_bwd_noalloc: M=%lld rows exceed the maximum of %lld of the no-allocation variant",
                 (long long)M, (long long)VA_LN_NOALLOC_MAX_ROWS);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VA_MAX_DEVICES) {
        va_set_error("va_layernorm_bwd_noalloc: no current device (or device index %d beyond %d)", dev, VA_MAX_DEVICES - 1);
        return VA_ERR_INVALID;
    }
    float* ptr;
    {
        std::lock_guard<std::mutex> hold(ln_scratch_lock);
        if (ln_scratch_ptr[dev] == nullptr) {      // first call on this device: resolve this device's copy of the array
            void* addr = nullptr;
            const hipError_t e = hipGetSymbolAddress(&addr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || addr == nullptr) {
                va_set_error("va_layernorm_bwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
                return VA_ERR_LAUNCH;
            }
            ln_scratch_ptr[dev] = (float*)addr;
        }
        ptr = ln_scratch_ptr[dev];
    }
    // every partial row the reduce kernel reads is written by the first kernel of the same call: nothing in the
    // array has to start at zero, and what the previous call left behind is never read
    return va_layernorm_bwd_dropout(dy, dy_bf16, x, gamma, mean, rstd, resid_grad, dx_out, dx_bf16, ptr, dgamma, dbeta, M, D,
                                    key, p, copy_colsum, stream);
}

extern "C" int va_rope3d_qk_inplace(void* qkv, const float* tables, int64_t M, int64_t Tp, int64_t Hp, int64_t Wp,
                                    int inverse, hipStream_t stream) {
    VA_CHECK_ARG(qkv && tables, "va_rope3d_qk_inplace: null operand");
    VA_CHECK_ARG(M > 0 && Tp > 0 && Hp > 0 && Wp > 0, "va_rope3d_qk_inplace: empty input");
    VA_CHECK_ARG(M % (Tp * Hp * Wp) == 0, "va_rope3d_qk_inplace: M=%lld is not a whole number of clips of %lld tokens",
                 (long long)M, (long long)(Tp * Hp * Wp));
    VA_CHECK_ARG(((uintptr_t)qkv | (uintptr_t)tables) % 16 == 0, "va_rope3d_qk_inplace: misaligned operand");
    hipLaunchKernelGGL(rope_kernel, dim3(blocks_for(M, 2)), dim3(256), 0, stream, (bf16_t*)qkv, tables, M, (int)Tp,
                       (int)Hp, (int)Wp, inverse ? -1.0f : 1.0f);
    VA_CHECK_LAUNCH("va_rope3d_qk_inplace");
    return VA_OK;
}

extern "C" int va_rope3d_qk_indexed(void* qkv, const float* cos_t, const float* sin_t, const float* cos_h,
                                    const float* sin_h, const float* cos_w, const float* sin_w, const int32_t* pos,
                                    int64_t M, int64_t N, int transpose, hipStream_t stream) {
    VA_CHECK_ARG(qkv && cos_t && sin_t && cos_h && sin_h && cos_w && sin_w && pos, "va_rope3d_qk_indexed: null operand");
    VA_CHECK_ARG(M > 0 && N > 0 && N < (1ll << 30) && M % N == 0, "va_rope3d_qk_indexed: M=%lld is not a whole number of sequences of %lld tokens",
                 (long long)M, (long long)N);
    VA_CHECK_ARG(((uintptr_t)qkv | (uintptr_t)cos_t | (uintptr_t)sin_t | (uintptr_t)cos_h | (uintptr_t)sin_h | (uintptr_t)cos_w |
                  (uintptr_t)sin_w) % 16 == 0 && (uintptr_t)pos % 4 == 0, "va_rope3d_qk_indexed: misaligned operand");
    hipLaunchKernelGGL(rope_indexed_kernel, dim3(blocks_for(M, 2)), dim3(256), 0, stream, (bf16_t*)qkv, cos_t, sin_t, cos_h,
                       sin_h, cos_w, sin_w, pos, M, (int)N, transpose);
    VA_CHECK_LAUNCH("va_rope3d_qk_indexed");
    return VA_OK;
}

extern "C" int va_tubelet_gather(const float* video, void* patches, int64_t B, int64_t C, int64_t T, int64_t H,
                                 int64_t W, int64_t tt, int64_t th, int64_t tw, hipStream_t stream) {
    VA_CHECK_ARG(video && patches, "va_tubelet_gather: null operand");
    VA_CHECK_ARG(B > 0 && C > 0 && tt > 0 && th > 0 && tw > 0 && T >= tt && H >= th && W >= tw, "va_tubelet_gather: bad shape");
    VA_CHECK_ARG(tw % 8 == 0 && W % 4 == 0, "va_tubelet_gather: tubelet width must be a multiple of 8 and W of 4");
    VA_CHECK_ARG(((uintptr_t)video | (uintptr_t)patches) % 16 == 0, "va_tubelet_gather: misaligned operand");
    const int64_t tokens = B * (T / tt) * (H / th) * (W / tw);
    const int64_t chunks = tokens * (C * tt * th * tw / 8);
    hipLaunchKernelGGL(tubelet_gather_kernel, dim3(blocks_for(chunks, 256)), dim3(256), 0, stream, video,
                       (bf16_t*)patches, chunks, (int)C, (int)T, (int)H, (int)W, (int)tt, (int)th, (int)tw);
    VA_CHECK_LAUNCH("va_tubelet_gather");
    return VA_OK;
}

extern "C" int va_tubelet_scatter(const float* patches, float* video, int64_t B, int64_t C, int64_t T, int64_t H,
                                  int64_t W, int64_t tt, int64_t th, int64_t tw, hipStream_t stream) {
    VA_CHECK_ARG(video && patches, "va_tubelet_scatter: null operand");
    VA_CHECK_ARG(B > 0 && C > 0 && tt > 0 && th > 0 && tw > 0 && T >= tt && H >= th && W >= tw, "va_tubelet_scatter: bad shape");
    VA_CHECK_ARG(tw % 4 == 0 && W % 4 == 0, "va_tubelet_scatter: tubelet width and W must be multiples of 4");
    VA_CHECK_ARG(((uintptr_t)video | (uintptr_t)patches) % 16 == 0, "va_tubelet_scatter: misaligned operand");
    const int64_t tokens = B * (T / tt) * (H / th) * (W / tw);
    const int64_t chunks = tokens * (C * tt * th * tw / 4);
    hipLaunchKernelGGL(tubelet_scatter_kernel, dim3(blocks_for(chunks, 256)), dim3(256), 0, stream, patches, video,
                       chunks, (int)C, (int)T, (int)H, (int)W, (int)tt, (int)th, (int)tw);
    VA_CHECK_LAUNCH("va_tubelet_scatter");
    return VA_OK;
}

extern "C" int va_cast_f32_bf16(const float* src, void* dst, int64_t n, hipStream_t stream) {
    VA_CHECK_ARG(src && dst && n > 0, "va_cast_f32_bf16: null or empty operand");
    VA_CHECK_ARG(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "va_cast_f32_bf16: misaligned operand");
    hipLaunchKernelGGL(cast_kernel, dim3(blocks_for(n, 2048)), dim3(256), 0, stream, src, (bf16_t*)dst, n);
    VA_CHECK_LAUNCH("va_cast_f32_bf16");
    return VA_OK;
}

extern "C" int va_multi_cast_f32_bf16(const va_cast_job* jobs, int64_t njobs, int64_t total_tiles, hipStream_t stream) {
    VA_CHECK_ARG(jobs && njobs > 0 && njobs < (1 << 20), "va_multi_cast_f32_bf16: empty or oversized job list");
    VA_CHECK_ARG(total_tiles > 0 && total_tiles < (1ll << 31), "va_multi_cast_f32_bf16: bad tile count %lld", (long long)total_tiles);
    hipLaunchKernelGGL(multi_cast_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, jobs, (int)njobs);
    VA_CHECK_LAUNCH("va_multi_cast_f32_bf16");
    return VA_OK;
}

extern "C" int va_cast_transpose_f32_bf16(const float* w, void* wt, int64_t R, int64_t C, hipStream_t stream) {
    VA_CHECK_ARG(w && wt, "va_cast_transpose_f32_bf16: null operand");
    VA_CHECK_ARG(R > 0 && C > 0 && R % 32 == 0 && C % 32 == 0, "va_cast_transpose_f32_bf16: R and C must be multiples of 32");
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((unsigned)(C / 32), (unsigned)(R / 32)), dim3(256), 0, stream, w,
                       (bf16_t*)wt, (int)R, (int)C);
    VA_CHECK_LAUNCH("va_cast_transpose_f32_bf16");
    return VA_OK;
}

extern "C" int va_colsum_bf16_accum(const void* y, int64_t M, int64_t N, int64_t ldy, float* out, hipStream_t stream) {
    VA_CHECK_ARG(y && out, "va_colsum_bf16_accum: null operand");
    VA_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && ldy % 8 == 0 && ldy >= N, "va_colsum_bf16_accum: bad shape");
    VA_CHECK_ARG((uintptr_t)y % 16 == 0, "va_colsum_bf16_accum: misaligned operand");
    int64_t gy = (M + 255) / 256;
    if (gy > 128) gy = 128;
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks_for(N, 512), (unsigned)gy), dim3(256), 0, stream, (const bf16_t*)y, M,
                       (int)N, ldy, out);
    VA_CHECK_LAUNCH("va_colsum_bf16_accum");
    return VA_OK;
}

extern "C" int va_future_mask(const void* x, void* out, uint8_t* mask, int64_t B, int64_t N, int64_t D,
                              int64_t elem_bytes, int64_t first_masked_row, hipStream_t stream) {
    VA_CHECK_ARG(x && out && mask, "va_future_mask: null operand");
    VA_CHECK_ARG(B > 0 && N > 0 && D > 0 && (elem_bytes == 2 || elem_bytes == 4), "va_future_mask: bad shape");
    VA_CHECK_ARG((D * elem_bytes) % 16 == 0, "va_future_mask: rows must be a multiple of 16 bytes");
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)out) % 16 == 0, "va_future_mask: misaligned operand");
    VA_CHECK_ARG(first_masked_row >= 0, "va_future_mask: negative cut-off");
    const int cpr = (int)(D * elem_bytes / 16);
    hipLaunchKernelGGL(future_mask_kernel, dim3(blocks_for(B * N * cpr, 256)), dim3(256), 0, stream, (const u32x4_t*)x,
                       (u32x4_t*)out, mask, B * N, (int)N, cpr, (int)(first_masked_row > N ? N : first_masked_row));
    VA_CHECK_LAUNCH("va_future_mask");
    return VA_OK;
}

static int token_move(bool scatter, const void* src, const int32_t* idx, void* dst, int64_t B, int64_t n_full,
                      int64_t n_keep, int64_t row_bytes, hipStream_t stream, const char* who) {
    VA_CHECK_ARG(src && idx && dst, "%s: null operand", who);
    VA_CHECK_ARG(B > 0 && n_full > 0 && n_keep > 0 && n_keep <= n_full, "%s: bad shape", who);
    VA_CHECK_ARG(row_bytes > 0 && row_bytes % 16 == 0, "%s: rows must be a multiple of 16 bytes", who);
    VA_CHECK_ARG(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "%s: misaligned operand", who);
    const int cpr = (int)(row_bytes / 16);
    if (scatter) {
        if (hipMemsetAsync(dst, 0, (size_t)(B * n_full * row_bytes), stream) != hipSuccess) {
            va_set_error("%s: memset failed", who);
            return VA_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(token_move_kernel<true>, dim3(blocks_for(B * n_keep * cpr, 256)), dim3(256), 0, stream,
                           (const u32x4_t*)src, idx, (u32x4_t*)dst, B * n_keep, (int)n_keep, (int)n_full, cpr);
    } else {
        hipLaunchKernelGGL(token_move_kernel<false>, dim3(blocks_for(B * n_keep * cpr, 256)), dim3(256), 0, stream,
                           (const u32x4_t*)src, idx, (u32x4_t*)dst, B * n_keep, (int)n_keep, (int)n_full, cpr);
    }
    VA_CHECK_LAUNCH(who);
    return VA_OK;
}

extern "C" int va_token_gather(const void* x, const int32_t* idx, void* out, int64_t B, int64_t N_src, int64_t N_keep,
                               int64_t row_bytes, hipStream_t stream) {
    return token_move(false, x, idx, out, B, N_src, N_keep, row_bytes, stream, "va_token_gather");
}

extern "C" int va_token_scatter(const void* src, const int32_t* idx, void* out, int64_t B, int64_t N_dst,
                                int64_t N_keep, int64_t row_bytes, hipStream_t stream) {
    return token_move(true, src, idx, out, B, N_dst, N_keep, row_bytes, stream, "va_token_scatter");
}

extern "C" int64_t va_token_pool_workspace_floats(int64_t B, int64_t D) { return B > 0 && D > 0 ? B * POOL_PARTS * D : 0; }

extern "C" int va_token_pool_fwd(const float* x, float* out, float* workspace, int64_t B, int64_t N, int64_t D,
                                 int64_t n_first, float scale, hipStream_t stream) {
    VA_CHECK_ARG(x && out && workspace, "va_token_pool_fwd: null operand");
    VA_CHECK_ARG(B > 0 && N > 0 && D > 0 && D % 4 == 0 && N < (1ll << 31) && B * POOL_PARTS < (1ll << 31), "va_token_pool_fwd: bad shape");
    VA_CHECK_ARG(n_first >= 0 && n_first <= N, "va_token_pool_fwd: n_first=%lld outside [0, N]", (long long)n_first);
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)out | (uintptr_t)workspace) % 16 == 0, "va_token_pool_fwd: misaligned operand");
    hipLaunchKernelGGL(pool_partial_kernel, dim3((unsigned)(B * POOL_PARTS)), dim3(256), 0, stream, x, workspace, (int)N, (int)D,
                       (int)n_first);
    VA_CHECK_LAUNCH("va_token_pool_fwd");
    hipLaunchKernelGGL(pool_reduce_kernel, dim3(blocks_for(B * D, 256)), dim3(256), 0, stream, workspace, out, B * D, (int)D, scale);
    VA_CHECK_LAUNCH("va_token_pool_fwd(reduce)");
    return VA_OK;
}

extern "C" int va_token_pool_bwd(const float* g, float* dx, int64_t B, int64_t N, int64_t D, int64_t n_first, float scale,
                                 hipStream_t stream) {
    VA_CHECK_ARG(g && dx, "va_token_pool_bwd: null operand");
    VA_CHECK_ARG(B > 0 && N > 0 && D > 0 && D % 4 == 0 && N < (1ll << 31), "va_token_pool_bwd: bad shape");
    VA_CHECK_ARG(n_first >= 0 && n_first <= N, "va_token_pool_bwd: n_first=%lld outside [0, N]", (long long)n_first);
    VA_CHECK_ARG(((uintptr_t)g | (uintptr_t)dx) % 16 == 0, "va_token_pool_bwd: misaligned operand");
    const int64_t total4 = B * N * D / 4;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(blocks_for(total4, 256)), dim3(256), 0, stream, g, dx, total4, (int)N, (int)D,
                       (int)n_first, scale);
    VA_CHECK_LAUNCH("va_token_pool_bwd");
    return VA_OK;
}

extern "C" int va_ema_update(float* target, const float* online, void* target_bf16, int64_t n, float momentum,
                             hipStream_t stream) {
    VA_CHECK_ARG(target && online && n > 0, "va_ema_update: null or empty operand");
    VA_CHECK_ARG(((uintptr_t)target | (uintptr_t)online | (uintptr_t)target_bf16) % 16 == 0, "va_ema_update: misaligned operand");
    constexpr int UNROLL = 4;     // measured on 303.8 M parameters: 1 -> 4.8 TB/s, 2 -> 5.7, 4 -> 6.05
    unsigned blocks = blocks_for(n, 1024 * UNROLL);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(ema_kernel<UNROLL>, dim3(blocks), dim3(256), 0, stream, target, online, (bf16_t*)target_bf16, n, momentum);
    VA_CHECK_LAUNCH("va_ema_update");
    return VA_OK;
}

extern "C" int va_l1_loss(const void* pred, const void* target, int in_bf16, float* loss_sum, void* grad,
                          int grad_bf16, float grad_scale, int64_t n, hipStream_t stream) {
    VA_CHECK_ARG(pred && target && loss_sum && n > 0, "va_l1_loss: null or empty operand");
    VA_CHECK_ARG(((uintptr_t)pred | (uintptr_t)target | (uintptr_t)grad) % 16 == 0, "va_l1_loss: misaligned operand");
    if (hipMemsetAsync(loss_sum, 0, sizeof(float), stream) != hipSuccess) {
        va_set_error("va_l1_loss: memset failed");
        return VA_ERR_LAUNCH;
    }
    // every block ends with one atomic add onto the same word, and same-address float atomics serialise at the
    // memory side: few, long-running blocks keep that tail short
    // (measured at 67 M bf16 elements: 8192 blocks 123 us, 1024 blocks 77 us, 256 blocks 139 us)
    unsigned blocks = blocks_for(n, 2048 * L1_CHUNKS);
    if (blocks > 1024) blocks = 1024;
#define VA_L1(IB, GB) hipLaunchKernelGGL((l1_kernel<IB, GB>), dim3(blocks), dim3(256), 0, stream, pred, target, loss_sum, grad, grad_scale, n)
    if (in_bf16) { if (grad_bf16) VA_L1(true, true); else VA_L1(true, false); }
    else { if (grad_bf16) VA_L1(false, true); else VA_L1(false, false); }
#undef VA_L1
    VA_CHECK_LAUNCH("va_l1_loss");
    return VA_OK;
}

extern "C" int va_normalize_u8(const uint8_t* frames, float* out, int64_t B, int64_t T, int64_t H, int64_t W,
                               const float* mean3, const float* std3, hipStream_t stream) {
    VA_CHECK_ARG(frames && out && mean3 && std3, "va_normalize_u8: null operand");
    VA_CHECK_ARG(B > 0 && T > 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "va_normalize_u8: H*W must be a positive multiple of 4");
    VA_CHECK_ARG((uintptr_t)frames % 4 == 0 && (uintptr_t)out % 16 == 0, "va_normalize_u8: misaligned operand");
    VA_CHECK_ARG(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "va_normalize_u8: zero std");
    const int64_t quads = B * T * H * W / 4;
    VA_CHECK_ARG(quads < (1ll << 39), "va_normalize_u8: problem too large");
    const f32x4_t mean = {mean3[0], mean3[1], mean3[2], 0.f}, stdv = {std3[0], std3[1], std3[2], 1.f};
    hipLaunchKernelGGL(normalize_u8_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, frames, out, quads,
                       H * W, T, mean, stdv);
    VA_CHECK_LAUNCH("va_normalize_u8");
    return VA_OK;
}

extern "C" int va_dropout(const void* in, int in_bf16, void* out, int out_bf16, const float* resid, int64_t n,
                          uint32_t key, float p, hipStream_t stream) {
    VA_CHECK_ARG(in && out, "va_dropout: null operand");
    VA_CHECK_ARG(n > 0 && n % 8 == 0 && n / 2 < (1ll << 32), "va_dropout: n=%lld must be a positive multiple of 8 below 2^33", (long long)n);
    VA_CHECK_ARG(p >= 0.f && p < 1.f, "va_dropout: p=%f outside [0,1)", (double)p);
    VA_CHECK_ARG(((uintptr_t)in | (uintptr_t)out | (uintptr_t)resid) % 16 == 0, "va_dropout: misaligned operand");
    const uint32_t thresh = va_drop_thresh16(p);
    const float scale = va_drop_scale(thresh);
    const int64_t n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256)), block(256);
    if (in_bf16 && out_bf16)
        hipLaunchKernelGGL((dropout_kernel<bf16_t, bf16_t>), grid, block, 0, stream, (const bf16_t*)in, (bf16_t*)out, resid, n8, key, thresh, scale);
    else if (!in_bf16 && out_bf16)
        hipLaunchKernelGGL((dropout_kernel<float, bf16_t>), grid, block, 0, stream, (const float*)in, (bf16_t*)out, resid, n8, key, thresh, scale);
    else if (!in_bf16 && !out_bf16)
        hipLaunchKernelGGL((dropout_kernel<float, float>), grid, block, 0, stream, (const float*)in, (float*)out, resid, n8, key, thresh, scale);
    else
        hipLaunchKernelGGL((dropout_kernel<bf16_t, float>), grid, block, 0, stream, (const bf16_t*)in, (float*)out, resid, n8, key, thresh, scale);
    VA_CHECK_LAUNCH("va_dropout");
    return VA_OK;
}
