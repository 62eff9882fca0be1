// Flash-style multi-head self-attention for the encoder (no mask, no dropout): forward and backward.
//
// Layout: element (b, n, h, d) of q/k/v lives at ((b*N + n)*ld + h*HD + d) from the respective base, so
// the three thirds of the packed [M, 3*D] QKV-projection output are used in place; O / dO are [M, D].
//
// All three kernels share one skeleton (gfx950, wave64, MFMA 32x32x16 bf16):
//   * a workgroup is 4 waves; each wave owns 32 "stationary" rows (queries in forward / dQ, keys in
//     dK/dV) whose operand fragments stay in registers; the other side streams through LDS in tiles of
//     64 rows, double-buffered with 16-byte direct-to-LDS loads, one barrier per tile.
//   * scores are produced with the streamed side as the MFMA A operand, so a lane holds a COLUMN of the
//     score tile (one stationary row) and softmax statistics are per-lane scalars (one cross-half
//     exchange per reduction).  The accumulator tile then feeds the next MFMA directly as its B operand
//     (contraction over the streamed rows, which sit in the registers); the matching A operand, a
//     transposed view of the streamed tile, comes from ds_read_b64_tr_b16.
//   * one LDS image serves both the row reads (ds_read_b128) and the transposed reads: 16-byte chunk
//     index XOR f(row), f chosen so both patterns hit all 64 banks (see swz()).
//   * dQ and dK/dV are separate passes (scores recomputed in each): no atomics, bitwise reproducible.
#include "common.h"

namespace {

constexpr int TILE_ROWS = 64;

template <int HD>
VA_DEVINL int swz(int row) {
    if (HD == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);   // 8 chunks per 128-byte row
    return ((row & 3) << 2) | ((row >> 2) & 3);                          // 16 chunks per 256-byte row
}

// Stage a [64 x HD] bf16 tile: rows row0..row0+63 (clamped to n_rows-1) of `src` (row stride ld).
template <int HD>
VA_DEVINL void stage_tile(const bf16_t* __restrict__ src, int64_t ld, int row0, int n_rows, char* dst, int tid,
                          int wave) {
    constexpr int CPR = HD / 8;                 // chunks per row
    constexpr int PER_THREAD = TILE_ROWS * CPR / 256;
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
        const int q = i * 256 + tid;
        const int row = q / CPR;
        const int c = (q % CPR) ^ swz<HD>(row);
        int g = row0 + row;
        g = g < n_rows ? g : n_rows - 1;
        glds16(src + (int64_t)g * ld + c * 8, dst + (i * 256 + wave * 64) * 16);
    }
}

// Row-read fragment (MFMA operand with the tile's rows as the M/N index): lane (r = lane&31, h = lane>>5)
// gets tile[row_base + r][16*ks + 8h .. +7].
template <int HD>
VA_DEVINL bf16x8_t row_frag(const char* tile, int row_base, int ks, int lane) {
    const int row = row_base + (lane & 31);
    const int chunk = (2 * ks + (lane >> 5)) ^ swz<HD>(row);
    return *(const bf16x8_t*)(tile + row * (HD * 2) + (chunk << 4));
}

// Transposed fragment: the A operand X^T[c][k'] of a product that contracts over 16 tile rows
// k' = row_base + {0..15} for the 32 columns cb*32 .. cb*32+31, in the k-order of an accumulator tile used
// as the B operand: element j of lane half h is row  row_base + 8*(j>>2) + 4h + (j&3).
template <int HD>
VA_DEVINL bf16x8_t tr_frag(const char* tile, int row_base, int cb, int lane) {
    const int gi = lane >> 4, cg = gi & 1, h = gi >> 1, q = (lane & 15) >> 2, p = lane & 3;
    const int r0 = row_base + 4 * h + q, r1 = r0 + 8;
    const int chunk = 4 * cb + 2 * cg + (p >> 1);
    const char* a0 = tile + r0 * (HD * 2) + ((chunk ^ swz<HD>(r0)) << 4) + (p & 1) * 8;
    const char* a1 = tile + r1 * (HD * 2) + ((chunk ^ swz<HD>(r1)) << 4) + (p & 1) * 8;
    return join_tr(lds_read_tr16(a0), lds_read_tr16(a1));
}

// Registers 8s..8s+7 of an accumulator tile -> one bf16 fragment (B operand of the following MFMA).
VA_DEVINL bf16x8_t pack_acc(const f32x16_t& x, int s) {
    u32x4_t v = {pack_bf16x2(x[8 * s + 0], x[8 * s + 1]), pack_bf16x2(x[8 * s + 2], x[8 * s + 3]),
                 pack_bf16x2(x[8 * s + 4], x[8 * s + 5]), pack_bf16x2(x[8 * s + 6], x[8 * s + 7])};
    return __builtin_bit_cast(bf16x8_t, v);
}

VA_DEVINL float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }
VA_DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Store a transposed 32-wide accumulator set acc[cb][i] = T[col = cb*32 + acc32_row(i,h)][row = lane&31]
// as bf16 rows: out_row[col..col+3] for each group of 4 registers.
template <int HD>
VA_DEVINL void store_rows(const f32x16_t (&acc)[HD / 32], float mul, bf16_t* out_row, int h) {
#pragma unroll
    for (int cb = 0; cb < HD / 32; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2_t pk = {pack_bf16x2(acc[cb][4 * g] * mul, acc[cb][4 * g + 1] * mul),
                          pack_bf16x2(acc[cb][4 * g + 2] * mul, acc[cb][4 * g + 3] * mul)};
            *(u32x2_t*)(out_row + cb * 32 + 8 * g + 4 * h) = pk;
        }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int HD, bool RAGGED>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                       const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                       float* __restrict__ LSE2, int N, int H, int64_t ld, int64_t ldo,
                                                       float c) {
    constexpr int KS = HD / 16, CB = HD / 32, TB = TILE_ROWS * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, hh = bh % H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bf16_t* Qb = Q + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Kb = K + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Vb = V + (int64_t)b * N * ld + hh * HD;

    const int qrow = q0 + (lane & 31);
    const int qrow_c = qrow < N ? qrow : N - 1;
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8_t*)(Qb + (int64_t)qrow_c * ld + ks * 16 + h * 8);

    f32x16_t o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[cb][i] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int nt = (N + TILE_ROWS - 1) / TILE_ROWS;
    stage_tile<HD>(Kb, ld, 0, N, smem, tid, wave);
    stage_tile<HD>(Vb, ld, 0, N, smem + TB, tid, wave);
    for (int t = 0; t < nt; ++t) {
        wait_vm0();
        __syncthreads();
        if (t + 1 < nt) {
            char* nb = smem + ((t + 1) & 1) * 2 * TB;
            stage_tile<HD>(Kb, ld, (t + 1) * TILE_ROWS, N, nb, tid, wave);
            stage_tile<HD>(Vb, ld, (t + 1) * TILE_ROWS, N, nb + TB, tid, wave);
        }
        const char* kt = smem + (t & 1) * 2 * TB;
        const char* vt = kt + TB;

        f32x16_t st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) st[kb] = mfma32(row_frag<HD>(kt, kb * 32, ks, lane), qf[ks], st[kb]);
        }
        if (RAGGED && t == nt - 1) {   // RAGGED <=> N % 64 != 0: only that build carries the selects
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (t * TILE_ROWS + kb * 32 + acc32_row(i, h) >= N) st[kb][i] = -INFINITY;
        }
        float mx = st[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[kb][i]);
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m, mx * c);
        const float alpha = fast_exp2(m - m_new);
        m = m_new;
        l *= alpha;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[cb][i] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = fast_exp2(fmaf(st[kb][i], c, -m_new));
                st[kb][i] = p;
                l += p;
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t pf = pack_acc(st[kb], s);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) o[cb] = mfma32(tr_frag<HD>(vt, kb * 32 + 16 * s, cb, lane), pf, o[cb]);
            }
    }
    l += __shfl_xor(l, 32);
    if (qrow < N) {
        store_rows<HD>(o, 1.0f / l, O + ((int64_t)b * N + qrow) * ldo + hh * HD, h);
        if (h == 0) LSE2[(int64_t)bh * N + qrow] = m + __log2f(l);
    }
}

// ------------------------------------------------------------------------------------------------
// delta[bh][n] = sum_d dO * O      (one 8-lane group per (token, head) for HD=64, 16 lanes for 128)
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ DO,
                                                         float* __restrict__ delta, int64_t rows, int N, int H,
                                                         int64_t ldo) {
    constexpr int LPH = HD / 8;   // lanes per (row, head)
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t item = gid / LPH;             // (row, head)
    const int part = (int)(gid % LPH);
    float s = 0.f;
    const bool live = item < rows * H;
    const int64_t row = live ? item / H : 0;
    const int hh = live ? (int)(item % H) : 0;
    if (live) {
        const u32x4_t a = *(const u32x4_t*)(O + row * ldo + hh * HD + part * 8);
        const u32x4_t d = *(const u32x4_t*)(DO + row * ldo + hh * HD + part * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s += __uint_as_float(a[j] << 16) * __uint_as_float(d[j] << 16);
            s += __uint_as_float(a[j] & 0xffff0000u) * __uint_as_float(d[j] & 0xffff0000u);
        }
    }
#pragma unroll
    for (int off = LPH / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (live && part == 0) {
        const int64_t b = row / N, n = row % N;
        delta[(b * H + hh) * N + n] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// dQ pass: query-stationary, streams K and V.
// ------------------------------------------------------------------------------------------------
template <int HD, bool RAGGED>
__global__ __launch_bounds__(256, HD == 64 ? 2 : 1) void attn_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ V, const bf16_t* __restrict__ DO,
                                                          const float* __restrict__ LSE2,
                                                          const float* __restrict__ DELTA, bf16_t* __restrict__ DQ,
                                                          int N, int H, int64_t ld, int64_t ldo, int64_t lddq, float c,
                                                          float scale) {
    constexpr int KS = HD / 16, CB = HD / 32, TB = TILE_ROWS * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, hh = bh % H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bf16_t* Qb = Q + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Kb = K + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Vb = V + (int64_t)b * N * ld + hh * HD;
    const bf16_t* DOb = DO + (int64_t)b * N * ldo + hh * HD;

    const int qrow = q0 + (lane & 31);
    const int qrow_c = qrow < N ? qrow : N - 1;
    bf16x8_t qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *(const bf16x8_t*)(Qb + (int64_t)qrow_c * ld + ks * 16 + h * 8);
        dof[ks] = *(const bf16x8_t*)(DOb + (int64_t)qrow_c * ldo + ks * 16 + h * 8);
    }
    const float lse = LSE2[(int64_t)bh * N + qrow_c];
    const float dl = DELTA[(int64_t)bh * N + qrow_c];

    f32x16_t dq[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[cb][i] = 0.f;

    const int nt = (N + TILE_ROWS - 1) / TILE_ROWS;
    stage_tile<HD>(Kb, ld, 0, N, smem, tid, wave);
    stage_tile<HD>(Vb, ld, 0, N, smem + TB, tid, wave);
    for (int t = 0; t < nt; ++t) {
        wait_vm0();
        __syncthreads();
        if (t + 1 < nt) {
            char* nb = smem + ((t + 1) & 1) * 2 * TB;
            stage_tile<HD>(Kb, ld, (t + 1) * TILE_ROWS, N, nb, tid, wave);
            stage_tile<HD>(Vb, ld, (t + 1) * TILE_ROWS, N, nb + TB, tid, wave);
        }
        const char* kt = smem + (t & 1) * 2 * TB;
        const char* vt = kt + TB;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = mfma32(row_frag<HD>(kt, kb * 32, ks, lane), qf[ks], s);
                dp = mfma32(row_frag<HD>(vt, kb * 32, ks, lane), dof[ks], dp);
            }
            const bool ragged = RAGGED && (t == nt - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p = fast_exp2(fmaf(s[i], c, -lse));
                if (ragged && t * TILE_ROWS + kb * 32 + acc32_row(i, h) >= N) p = 0.f;
                s[i] = p * (dp[i] - dl);
            }
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const bf16x8_t dsf = pack_acc(s, sidx);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    dq[cb] = mfma32(tr_frag<HD>(kt, kb * 32 + 16 * sidx, cb, lane), dsf, dq[cb]);
            }
        }
    }
    if (qrow < N) store_rows<HD>(dq, scale, DQ + ((int64_t)b * N + qrow) * lddq + hh * HD, h);
}

// ------------------------------------------------------------------------------------------------
// dK/dV pass: key-stationary, streams Q and dO (plus the per-query lse2 and delta).
// ------------------------------------------------------------------------------------------------
template <int HD, bool RAGGED>
__global__ __launch_bounds__(256, HD == 64 ? 2 : 1) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                           const bf16_t* __restrict__ V, const bf16_t* __restrict__ DO,
                                                           const float* __restrict__ LSE2,
                                                           const float* __restrict__ DELTA, bf16_t* __restrict__ DK,
                                                           bf16_t* __restrict__ DV, int N, int H, int64_t ld,
                                                           int64_t ldo, int64_t lddq, float c, float scale) {
    constexpr int KS = HD / 16, CB = HD / 32, TB = TILE_ROWS * HD * 2;
    constexpr int STAGE = 2 * TB + 2 * TILE_ROWS * 4;   // Q tile | dO tile | lse2[64] | delta[64]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, hh = bh % H;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const bf16_t* Qb = Q + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Kb = K + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Vb = V + (int64_t)b * N * ld + hh * HD;
    const bf16_t* DOb = DO + (int64_t)b * N * ldo + hh * HD;
    const float* lse_b = LSE2 + (int64_t)bh * N;
    const float* dl_b = DELTA + (int64_t)bh * N;

    const int krow = k0 + (lane & 31);
    const int krow_c = krow < N ? krow : N - 1;
    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *(const bf16x8_t*)(Kb + (int64_t)krow_c * ld + ks * 16 + h * 8);
        vf[ks] = *(const bf16x8_t*)(Vb + (int64_t)krow_c * ld + ks * 16 + h * 8);
    }
    f32x16_t dk[CB], dv[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[cb][i] = 0.f; dv[cb][i] = 0.f; }

    // per-query scalars of a tile, also by direct-to-LDS loads (4 bytes per lane): wave 0 brings lse2[64],
    // wave 1 delta[64]; rows past N are clamped here and masked in the compute.
    auto stage_scalars = [&](int t, char* stg) {
        if (wave < 2) {
            int r = t * TILE_ROWS + lane;
            r = r < N ? r : N - 1;
            const float* src = (wave == 0 ? lse_b : dl_b) + r;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(stg + 2 * TB + wave * 256), 4, 0, 0);
        }
    };

    const int nt = (N + TILE_ROWS - 1) / TILE_ROWS;
    stage_tile<HD>(Qb, ld, 0, N, smem, tid, wave);
    stage_tile<HD>(DOb, ldo, 0, N, smem + TB, tid, wave);
    stage_scalars(0, smem);
    for (int t = 0; t < nt; ++t) {
        wait_vm0();
        __syncthreads();
        if (t + 1 < nt) {
            char* nb = smem + ((t + 1) & 1) * STAGE;
            stage_tile<HD>(Qb, ld, (t + 1) * TILE_ROWS, N, nb, tid, wave);
            stage_tile<HD>(DOb, ldo, (t + 1) * TILE_ROWS, N, nb + TB, tid, wave);
            stage_scalars(t + 1, nb);
        }
        const char* qt = smem + (t & 1) * STAGE;
        const char* dot = qt + TB;
        const float* lse_t = (const float*)(qt + 2 * TB);
        const float* dl_t = lse_t + TILE_ROWS;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16_t s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = mfma32(row_frag<HD>(qt, qb * 32, ks, lane), kf[ks], s);      // [query][key]
                dp = mfma32(row_frag<HD>(dot, qb * 32, ks, lane), vf[ks], dp);
            }
            f32x16_t pr;
            const bool ragged = RAGGED && (t == nt - 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t lv = *(const f32x4_t*)(lse_t + qb * 32 + 8 * g + 4 * h);
                const f32x4_t dv4 = *(const f32x4_t*)(dl_t + qb * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float p = fast_exp2(fmaf(s[4 * g + j], c, -lv[j]));
                    if (ragged && t * TILE_ROWS + qb * 32 + 8 * g + 4 * h + j >= N) p = 0.f;
                    pr[4 * g + j] = p;
                    s[4 * g + j] = p * (dp[4 * g + j] - dv4[j]);
                }
            }
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const bf16x8_t pf = pack_acc(pr, sidx);
                const bf16x8_t dsf = pack_acc(s, sidx);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    dv[cb] = mfma32(tr_frag<HD>(dot, qb * 32 + 16 * sidx, cb, lane), pf, dv[cb]);
                    dk[cb] = mfma32(tr_frag<HD>(qt, qb * 32 + 16 * sidx, cb, lane), dsf, dk[cb]);
                }
            }
        }
    }
    if (krow < N) {
        store_rows<HD>(dk, scale, DK + ((int64_t)b * N + krow) * lddq + hh * HD, h);
        store_rows<HD>(dv, 1.0f, DV + ((int64_t)b * N + krow) * lddq + hh * HD, h);
    }
}

template <typename KernT>
int set_lds(KernT kern, int bytes, bool* flags, const char* who) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !flags[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            va_set_error("%s: cannot reserve %d bytes of LDS", who, bytes);
            return VA_ERR_LAUNCH;
        }
        flags[dev] = true;
    }
    return VA_OK;
}

int check_common(const char* who, int64_t B, int64_t N, int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo) {
    VA_CHECK_ARG(B > 0 && N > 0 && H > 0, "%s: empty problem", who);
    VA_CHECK_ARG(hd == 64 || hd == 128, "%s: head dim %lld not supported (64 or 128)", who, (long long)hd);
    VA_CHECK_ARG(ldqkv >= H * hd && ldo >= H * hd, "%s: leading dimension smaller than H*hd", who);
    VA_CHECK_ARG(ldqkv % 8 == 0 && ldo % 8 == 0, "%s: leading dimensions must keep 16-byte alignment", who);
    VA_CHECK_ARG(B * H < 65536 && N < (1 << 24), "%s: B*H or N too large for the launch grid", who);
    return VA_OK;
}

template <int HD>
int fwd_launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, float* lse2, int B, int N, int H,
               int64_t ld, int64_t ldo, hipStream_t stream) {
    constexpr int LDS = 2 * 2 * TILE_ROWS * HD * 2;
    static bool flags[2][64] = {};
    const float c = 1.4426950408889634f / sqrtf((float)HD);
    const dim3 grid((N + 127) / 128, B * H);
    if (N % TILE_ROWS) {
        if (int e = set_lds(attn_fwd_kernel<HD, true>, LDS, flags[1], "va_attn_fwd")) return e;
        hipLaunchKernelGGL((attn_fwd_kernel<HD, true>), grid, dim3(256), LDS, stream, q, k, v, o, lse2, N, H, ld, ldo, c);
    } else {
        if (int e = set_lds(attn_fwd_kernel<HD, false>, LDS, flags[0], "va_attn_fwd")) return e;
        hipLaunchKernelGGL((attn_fwd_kernel<HD, false>), grid, dim3(256), LDS, stream, q, k, v, o, lse2, N, H, ld, ldo, c);
    }
    VA_CHECK_LAUNCH("va_attn_fwd");
    return VA_OK;
}

template <int HD>
int bwd_launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* o, const bf16_t* d_o,
               const float* lse2, float* delta, bf16_t* dq, bf16_t* dk, bf16_t* dv, int B, int N, int H, int64_t ld,
               int64_t ldo, int64_t lddq, hipStream_t stream) {
    constexpr int LDS_DQ = 2 * 2 * TILE_ROWS * HD * 2;
    constexpr int LDS_DKV = 2 * (2 * TILE_ROWS * HD * 2 + 2 * TILE_ROWS * 4);
    static bool f1[2][64] = {}, f2[2][64] = {};
    const bool ragged = (N % TILE_ROWS) != 0;
    if (ragged) {
        if (int e = set_lds(attn_bwd_dq_kernel<HD, true>, LDS_DQ, f1[1], "va_attn_bwd")) return e;
        if (int e = set_lds(attn_bwd_dkv_kernel<HD, true>, LDS_DKV, f2[1], "va_attn_bwd")) return e;
    } else {
        if (int e = set_lds(attn_bwd_dq_kernel<HD, false>, LDS_DQ, f1[0], "va_attn_bwd")) return e;
        if (int e = set_lds(attn_bwd_dkv_kernel<HD, false>, LDS_DKV, f2[0], "va_attn_bwd")) return e;
    }
    const float scale = 1.0f / sqrtf((float)HD);
    const float c = 1.4426950408889634f * scale;
    const int64_t rows = (int64_t)B * N;
    const int64_t threads = rows * H * (HD / 8);
    hipLaunchKernelGGL(attn_delta_kernel<HD>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, o, d_o,
                       delta, rows, N, H, ldo);
    VA_CHECK_LAUNCH("va_attn_bwd(delta)");
    const dim3 grid((N + 127) / 128, B * H);
    if (ragged)
        hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, true>), grid, dim3(256), LDS_DQ, stream, q, k, v, d_o, lse2, delta, dq,
                           N, H, ld, ldo, lddq, c, scale);
    else
        hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, false>), grid, dim3(256), LDS_DQ, stream, q, k, v, d_o, lse2, delta, dq,
                           N, H, ld, ldo, lddq, c, scale);
    VA_CHECK_LAUNCH("va_attn_bwd(dq)");
    if (ragged)
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, true>), grid, dim3(256), LDS_DKV, stream, q, k, v, d_o, lse2, delta,
                           dk, dv, N, H, ld, ldo, lddq, c, scale);
    else
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, false>), grid, dim3(256), LDS_DKV, stream, q, k, v, d_o, lse2, delta,
                           dk, dv, N, H, ld, ldo, lddq, c, scale);
    VA_CHECK_LAUNCH("va_attn_bwd(dkv)");
    return VA_OK;
}

}  // namespace

extern "C" int va_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B, int64_t N,
                           int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, hipStream_t stream) {
    VA_CHECK_ARG(q && k && v && o && lse2, "va_attn_fwd: null operand");
    if (int e = check_common("va_attn_fwd", B, N, H, hd, ldqkv, ldo)) return e;
    VA_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0, "va_attn_fwd: misaligned operand");
    if (hd == 64)
        return fwd_launch<64>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2, (int)B, (int)N,
                              (int)H, ldqkv, ldo, stream);
    return fwd_launch<128>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2, (int)B, (int)N, (int)H,
                           ldqkv, ldo, stream);
}

extern "C" int va_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                           const float* lse2, float* delta, void* dq, void* dk, void* dv, int64_t B, int64_t N,
                           int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, int64_t lddqkv, hipStream_t stream) {
    VA_CHECK_ARG(q && k && v && o && d_o && lse2 && delta && dq && dk && dv, "va_attn_bwd: null operand");
    if (int e = check_common("va_attn_bwd", B, N, H, hd, ldqkv, ldo)) return e;
    VA_CHECK_ARG(lddqkv >= H * hd && lddqkv % 8 == 0, "va_attn_bwd: bad gradient leading dimension");
    VA_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq |
                  (uintptr_t)dk | (uintptr_t)dv) % 16 == 0,
                 "va_attn_bwd: misaligned operand");
    if (hd == 64)
        return bwd_launch<64>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o,
                              lse2, delta, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, (int)B, (int)N, (int)H, ldqkv, ldo,
                              lddqkv, stream);
    return bwd_launch<128>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o,
                           lse2, delta, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, (int)B, (int)N, (int)H, ldqkv, ldo, lddqkv,
                           stream);
}
