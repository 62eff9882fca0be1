// Kernels of the predictor's gated delta-rule memory layer (inference form).
//
//  * gated_delta_kernel: the sequential recurrence of swm:523-555,
//        S <- alpha_t * S * (I - beta_t k_t k_t^T) + beta_t v_t k_t^T,      o_t = S q_t
//    evaluated in its rank-1 form  sk = S k;  S <- alpha*S + beta*(v - alpha*sk) k^T  (no hd^3 products).
//    One workgroup of 2*hd lanes per (batch, head), the hd x hd state in registers for the whole sequence.  Lane
//    l = 4*g + c holds rows 2g, 2g+1 of the columns {16j + 4c .. 16j + 4c + 3}: a step then needs only hd/4 values
//    of k and of q per lane (read from LDS as 16-byte pieces, the four lanes of a quad reading 64 consecutive
//    bytes) instead of all hd as broadcasts, and the 2*hd/64 waves of a head land on different SIMDs with a second
//    head's waves beside them to cover the step's dependent chain (dot product -> coefficient -> update).  The two
//    partial dot products of a lane are summed over its quad with two exchange steps that leave lanes 0,1 holding
//    the total of row 2g and lanes 2,3 that of row 2g+1; those lanes compute the row's update coefficient, which
//    goes back to the quad with two quad broadcasts, and one of them writes the row's output.  k/q/v/alpha/beta
//    of 16 steps at a time are staged in LDS (double-buffered; fetched into registers a chunk ahead), and a step's
//    operands are read from LDS while the step before it computes.
//  * gate_ln_kernel: the layer's output stage, (o * gate) * silu(g) then LayerNorm (swm:511-519), as one pass.
//  * dwconv_silu_kernel: causal depthwise conv1d (4 taps, swm:452-457 with the [..., :L] cut of swm:484-486)
//    + bias + SiLU, optionally followed by the row-wise L2 normalisation of swm:489-490.
#include "common.h"

namespace {

constexpr int GD_T = 16;   // steps per staged chunk

template <bool V> struct flag { static constexpr bool value = V; };

// x of another lane of the same quad; CTRL is the quad permutation (two bits per destination lane).
template <int CTRL>
__device__ __forceinline__ float quad_lane(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// Two values per lane, summed over the four lanes of a quad: lanes 0 and 1 return the total of p0, lanes 2 and 3
// the total of p1.  First exchange with the lane two away (each keeps the value it will end up owning), then add the
// neighbour's.  hi2 = bit 1 of the lane's position in its quad.
__device__ __forceinline__ float quad_reduce_pairs(float p0, float p1, bool hi2) {
    const float k = (hi2 ? p1 : p0) + quad_lane<0x4E>(hi2 ? p0 : p1);       // 0x4E: lanes [2,3,0,1]
    return k + quad_lane<0xB1>(k);                                           // 0xB1: lanes [1,0,3,2]
}

template <int HD>
__global__ __launch_bounds__(2 * HD) __attribute__((amdgpu_waves_per_eu(2))) void gated_delta_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, const float* __restrict__ ALPHA,
                                                         const float* __restrict__ BETA,
                                                         const float* __restrict__ STATE_IN, float* __restrict__ OUT,
                                                         float* __restrict__ STATE_OUT, int L, int H) {
    constexpr int WAVES = 2 * HD / 64;
    constexpr int ROWS_PER_PASS = WAVES * 64 * 4 / HD;          // rows one pass of 16-byte lanes covers
    constexpr int PASSES = GD_T / ROWS_PER_PASS;
    __shared__ __attribute__((aligned(16))) float ks[2][GD_T][HD];
    __shared__ __attribute__((aligned(16))) float qs[2][GD_T][HD];
    __shared__ __attribute__((aligned(16))) float vs[2][GD_T][HD];
    __shared__ __attribute__((aligned(16))) float ab[2][2][GD_T];
    const int i = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t row_stride = (int64_t)H * HD;
    const float* qb = Q + (int64_t)b * L * row_stride + h * HD;
    const float* kb = K + (int64_t)b * L * row_stride + h * HD;
    const float* vb = V + (int64_t)b * L * row_stride + h * HD;
    const float* ab_src[2] = {ALPHA + (int64_t)b * L * H + h, BETA + (int64_t)b * L * H + h};

    // the state as float pairs: the step arithmetic is written on pairs so that it maps to packed f32
    // instructions on adjacent registers.  T[r][2j], T[r][2j+1] = row 2g + r, columns 16j + 4c .. + 3.
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    constexpr int NJ = HD / 16;
    const int cg = i & 3, row0 = (i >> 2) * 2;
    const bool hi2 = (cg & 2) != 0;
    const int own = row0 + (hi2 ? 1 : 0);          // the row whose totals this lane ends up with
    f32x2_t T[2][2 * NJ];
    if (STATE_IN != nullptr) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float* src = STATE_IN + ((int64_t)bh * HD + row0 + r) * HD + 4 * cg;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4_t v4 = *(const f32x4_t*)(src + 16 * j);
                T[r][2 * j] = (f32x2_t){v4[0], v4[1]};
                T[r][2 * j + 1] = (f32x2_t){v4[2], v4[3]};
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 2 * NJ; ++j) T[r][j] = (f32x2_t){0.f, 0.f};
    }

    // Staging of a chunk: direct-to-LDS loads issued at the start of the previous chunk and waited for at its end.
    // They are written as assembly because the compiler, not knowing which LDS bytes such a load writes, would
    // otherwise drain the vector-memory counter before every LDS read of the step loop.  m0 (the LDS destination
    // base of such a load) is the compiler's register and cannot be named as a clobber, so the statement saves it,
    // sets it, and restores it itself; the s_nop is the wait state between a scalar write of m0 and an instruction
    // that reads it.
    const int r_in_pass = i * 4 / HD, col = (i * 4) % HD;     // this lane's 16 bytes inside a pass
    const int wave = __builtin_amdgcn_readfirstlane(i >> 6);
    auto to_lds = [](const float* src, const float* lds_wave_base) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"((uint32_t)(uintptr_t)lds_wave_base) : "memory");
    };
    // The decay / write-strength pair of a chunk travels through a register instead: an ordinary load, which the
    // compiler counts.  Its wait would drain the direct-to-LDS loads above if it came after them, so the value for
    // chunk c + 1 is requested one chunk early (at the start of chunk c - 1, ahead of that chunk's staging loads) and
    // written to LDS at the start of chunk c, when the wait at the top of the chunk has already emptied the queue.
    auto ab_load = [&](int chunk) -> float {
        float x = 0.f;
        if (i < 2 * GD_T) {
            int t = chunk * GD_T + i % GD_T;
            t = t < L ? t : L - 1;
            x = ab_src[i / GD_T][(int64_t)t * H];
        }
        return x;
    };
    float ab_next = ab_load(0);
    auto stage = [&](int chunk, int buf) {
        if (i < 2 * GD_T) ab[buf][i / GD_T][i % GD_T] = ab_next;
        ab_next = ab_load(chunk + 1);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int t = chunk * GD_T + p * ROWS_PER_PASS + r_in_pass;
            t = t < L ? t : L - 1;
            const int64_t off = (int64_t)t * row_stride + col;
            const int dst = p * ROWS_PER_PASS * HD + wave * 256;     // wave base; lanes land 16 bytes apart
            to_lds(kb + off, &ks[buf][0][0] + dst);
            to_lds(qb + off, &qs[buf][0][0] + dst);
            to_lds(vb + off, &vs[buf][0][0] + dst);
        }
    };

    // what a step needs before its first instruction: k and the three scalars (q is read inside the step)
    struct Operands {
        f32x4_t k[NJ];
        float a, bt, v;
    };

    const int chunks = (L + GD_T - 1) / GD_T;
    stage(0, 0);
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        wait_vm0();
        __syncthreads();                       // chunk c is in LDS; everybody is done with chunk c-1
        if (c + 1 < chunks) stage(c + 1, buf ^ 1);
        const int steps = (L - c * GD_T) < GD_T ? (L - c * GD_T) : GD_T;
        auto load = [&](Operands& x, int t) {
            x.a = ab[buf][0][t];
            x.bt = ab[buf][1][t];
            x.v = vs[buf][t][own];
#pragma unroll
            for (int j = 0; j < NJ; ++j) x.k[j] = *(const f32x4_t*)&ks[buf][t][16 * j + 4 * cg];
        };
        // One step on operands x, reading those of the next step into y half-way (its q at the start: both reads
        // have a phase of arithmetic to land in).  FOLDED: the registers hold T = S / gam, gam the running product of
        // the chunk's decays (the same number in every lane), so alpha * S costs nothing -- gam absorbs it -- and the
        // rank-1 update is one fma per element, T += (cu / gam') k.  Otherwise the registers hold S itself and the
        // update is a S + cu k.
        auto step = [&](auto folded, const Operands& x, Operands& y, int t, float& gam) {
            constexpr bool FOLDED = decltype(folded)::value;
            f32x4_t q[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) q[j] = *(const f32x4_t*)&qs[buf][t][16 * j + 4 * cg];
            __builtin_amdgcn_sched_barrier(0);
            const float a = x.a;
            f32x2_t acc[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[r] = (f32x2_t){0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x2_t ka = {x.k[j][0], x.k[j][1]}, kb2 = {x.k[j][2], x.k[j][3]};
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    acc[r] = T[r][2 * j] * ka + acc[r];
                    acc[r] = T[r][2 * j + 1] * kb2 + acc[r];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < steps) load(y, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            float sk = quad_reduce_pairs(acc[0][0] + acc[0][1], acc[1][0] + acc[1][1], hi2);        // (S k)[own]
            if (FOLDED) sk *= gam;
            float cu = x.bt * (x.v - a * sk);
            if (FOLDED) {
                gam *= a;
                cu *= __builtin_amdgcn_rcpf(gam);
            }
            const float cr[2] = {quad_lane<0x00>(cu), quad_lane<0xAA>(cu)};
            const f32x2_t a2 = {a, a};
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[r] = (f32x2_t){0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x2_t ka = {x.k[j][0], x.k[j][1]}, kb2 = {x.k[j][2], x.k[j][3]};
                const f32x2_t qa = {q[j][0], q[j][1]}, qb2 = {q[j][2], q[j][3]};
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const f32x2_t c2 = {cr[r], cr[r]};
                    if (FOLDED) {
                        T[r][2 * j] = c2 * ka + T[r][2 * j];
                        T[r][2 * j + 1] = c2 * kb2 + T[r][2 * j + 1];
                    } else {
                        T[r][2 * j] = c2 * ka + a2 * T[r][2 * j];
                        T[r][2 * j + 1] = c2 * kb2 + a2 * T[r][2 * j + 1];
                    }
                    acc[r] = T[r][2 * j] * qa + acc[r];
                    acc[r] = T[r][2 * j + 1] * qb2 + acc[r];
                }
            }
            float o = quad_reduce_pairs(acc[0][0] + acc[0][1], acc[1][0] + acc[1][1], hi2);
            if (FOLDED) o *= gam;
            if ((cg & 1) == 0) OUT[((int64_t)b * L + c * GD_T + t) * row_stride + h * HD + own] = o;
        };
        auto run = [&](auto folded, float& gam) {
            Operands x0, x1;
            load(x0, 0);
            for (int t = 0; t < steps; t += 2) {
                step(folded, x0, x1, t, gam);
                if (t + 1 < steps) step(folded, x1, x0, t + 1, gam);
            }
        };
        // A chunk runs folded when the product of its decays stays far inside the float range (then no decay is
        // zero either); gam goes back into the registers at the end of the chunk.
        float prod = 1.0f;
#pragma unroll
        for (int t4 = 0; t4 < GD_T; t4 += 4) {
            const f32x4_t a4 = *(const f32x4_t*)&ab[buf][0][t4];
#pragma unroll
            for (int e = 0; e < 4; ++e) prod *= (t4 + e < steps) ? a4[e] : 1.0f;
        }
        if (prod > 1e-20f) {
            float gam = 1.0f;
            run(flag<true>{}, gam);
            const f32x2_t g2 = {gam, gam};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 2 * NJ; ++j) T[r][j] = g2 * T[r][j];
        } else {
            float unused = 1.0f;
            run(flag<false>{}, unused);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float* dst = STATE_OUT + ((int64_t)bh * HD + row0 + r) * HD + 4 * cg;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            *(f32x4_t*)(dst + 16 * j) = (f32x4_t){T[r][2 * j][0], T[r][2 * j][1], T[r][2 * j + 1][0], T[r][2 * j + 1][1]};
    }
}

// One wave per (batch, time) row of C channels; lane owns channels 4*lane + 256*m.
__global__ __launch_bounds__(256) void dwconv_silu_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                          const float* __restrict__ BIAS, float* __restrict__ OUT,
                                                          int64_t rows, int L, int C, int normalize) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int t = (int)(row % L);
    const float* x = X + row * C;
    float* out = OUT + row * C;
    float ss = 0.f;
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        f32x4_t acc = *(const f32x4_t*)(BIAS + c0);
        f32x4_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = *(const f32x4_t*)(W + (int64_t)(c0 + e) * 4);   // taps of channel c0+e
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            const int back = 3 - tap;                     // tap i multiplies x[t - 3 + i]
            if (t - back < 0) continue;
            const f32x4_t xv = *(const f32x4_t*)(x - (int64_t)back * C + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[e][tap], xv[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] = acc[e] / (1.0f + __expf(-acc[e]));     // SiLU
            ss = fmaf(acc[e], acc[e], ss);
        }
        *(f32x4_t*)(out + c0) = acc;
    }
    if (normalize) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize: x / max(||x||, eps)
        for (int c0 = lane * 4; c0 < C; c0 += 256) {
            f32x4_t v = *(const f32x4_t*)(out + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= inv;
            *(f32x4_t*)(out + c0) = v;
        }
    }
}

// The same for rows of C = 256 * NCH channels (NCH <= 8): one wave per RUN consecutive time steps of one batch
// item.  The inputs of a run are read once into a sliding window (RUN + 3 rows instead of 4 * RUN), and a row's
// results stay in registers until its norm is known, so the normalised form needs no second pass over the output.
template <int NCH, int RUN>
__global__ __launch_bounds__(256) void dwconv_silu_run_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                              const float* __restrict__ BIAS, float* __restrict__ OUT,
                                                              int64_t runs, int L, int normalize) {
    constexpr int C = 256 * NCH;
    const int lane = threadIdx.x & 63;
    const int64_t run = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (run >= runs) return;
    const int runs_per_seq = (L + RUN - 1) / RUN;
    const int64_t b = run / runs_per_seq;
    const int t0 = (int)(run % runs_per_seq) * RUN;
    const float* xb = X + b * (int64_t)L * C;
    float* ob = OUT + b * (int64_t)L * C;
    f32x4_t res[RUN][NCH];
    float ss[RUN];
#pragma unroll
    for (int r = 0; r < RUN; ++r) ss[r] = 0.f;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int c0 = lane * 4 + 256 * m;
        const f32x4_t bias = *(const f32x4_t*)(BIAS + c0);
        f32x4_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = *(const f32x4_t*)(W + (int64_t)(c0 + e) * 4);   // taps of channel c0+e
        f32x4_t win[RUN + 3];                        // rows t0-3 .. t0+RUN-1 (zeros before the sequence start)
#pragma unroll
        for (int i = 0; i < RUN + 3; ++i) {
            const int t = t0 - 3 + i;
            win[i] = (t >= 0 && t < L) ? *(const f32x4_t*)(xb + (int64_t)t * C + c0) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < RUN; ++r) {
            f32x4_t acc = bias;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {      // tap i multiplies x[t - 3 + i]; same order as the one-row kernel
                if (t0 + r - (3 - tap) < 0) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[e][tap], win[r + tap][e], acc[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[e] = acc[e] / (1.0f + __expf(-acc[e]));     // SiLU
                ss[r] = fmaf(acc[e], acc[e], ss[r]);
            }
            res[r][m] = acc;
        }
    }
#pragma unroll
    for (int r = 0; r < RUN; ++r) {
        if (t0 + r >= L) break;
        float inv = 1.0f;
        if (normalize) {
            float tot = ss[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
            inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);   // F.normalize: x / max(||x||, eps)
        }
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            f32x4_t v = res[r][m];
            if (normalize) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= inv;
            }
            *(f32x4_t*)(ob + (int64_t)(t0 + r) * C + lane * 4 + 256 * m) = v;
        }
    }
}

template <int NCH>
void launch_dwconv_run(const float* x, const float* w, const float* bias, float* out, int64_t B, int L, int normalize,
                       hipStream_t stream) {
    constexpr int RUN = NCH <= 2 ? 8 : 4;
    const int64_t runs = B * ((L + RUN - 1) / RUN);
    hipLaunchKernelGGL((dwconv_silu_run_kernel<NCH, RUN>), dim3((unsigned)((runs + 3) / 4)), dim3(256), 0, stream, x, w,
                       bias, out, runs, L, normalize);
}

// The same arithmetic for 1024 channels read as bf16 from rows `ldx` elements apart -- a column block of the output of
// ONE projection GEMM that serves q, k, v, the value gate and g together.  A lane owns 8 consecutive channels (one
// 16-byte load per row) in each half of the row; RUN time steps per wave through the same sliding window; f32 results.
template <int RUN>
__global__ __launch_bounds__(256) void dwconv_silu_bf16_kernel(const bf16_t* __restrict__ X, int64_t ldx,
                                                               const float* __restrict__ W, const float* __restrict__ BIAS,
                                                               float* __restrict__ OUT, int64_t runs, int L, int normalize) {
    constexpr int C = 1024;
    const int lane = threadIdx.x & 63;
    const int64_t run = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (run >= runs) return;
    const int runs_per_seq = (L + RUN - 1) / RUN;
    const int64_t b = run / runs_per_seq;
    const int t0 = (int)(run % runs_per_seq) * RUN;
    const bf16_t* xb = X + b * (int64_t)L * ldx;
    float* ob = OUT + b * (int64_t)L * C;
    float res[RUN][2][8];
    float ss[RUN];
#pragma unroll
    for (int r = 0; r < RUN; ++r) ss[r] = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int c0 = lane * 8 + 512 * m;
        float bias[8];
        *(f32x4_t*)bias = *(const f32x4_t*)(BIAS + c0);
        *(f32x4_t*)(bias + 4) = *(const f32x4_t*)(BIAS + c0 + 4);
        f32x4_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = *(const f32x4_t*)(W + (int64_t)(c0 + e) * 4);   // taps of channel c0+e
        u32x4_t win[RUN + 3];                        // rows t0-3 .. t0+RUN-1 (zeros before the sequence start)
#pragma unroll
        for (int i = 0; i < RUN + 3; ++i) {
            const int t = t0 - 3 + i;
            win[i] = (t >= 0 && t < L) ? *(const u32x4_t*)(xb + (int64_t)t * ldx + c0) : (u32x4_t){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int r = 0; r < RUN; ++r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc = bias[e];
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {      // tap i multiplies x[t - 3 + i]; same order as the f32 kernels
                    if (t0 + r - (3 - tap) < 0) continue;
                    const uint32_t pair = win[r + tap][e >> 1];
                    const float xv = __uint_as_float((e & 1) ? (pair & 0xffff0000u) : (pair << 16));
                    acc = fmaf(w[e][tap], xv, acc);
                }
                acc = acc / (1.0f + __expf(-acc));     // SiLU
                ss[r] = fmaf(acc, acc, ss[r]);
                res[r][m][e] = acc;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RUN; ++r) {
        if (t0 + r >= L) break;
        float inv = 1.0f;
        if (normalize) {
            float tot = ss[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
            inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);   // F.normalize: x / max(||x||, eps)
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float* dst = ob + (int64_t)(t0 + r) * C + lane * 8 + 512 * m;
            f32x4_t lo = {res[r][m][0], res[r][m][1], res[r][m][2], res[r][m][3]};
            f32x4_t hi = {res[r][m][4], res[r][m][5], res[r][m][6], res[r][m][7]};
            if (normalize) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] *= inv; hi[e] *= inv; }
            }
            *(f32x4_t*)dst = lo;
            *(f32x4_t*)(dst + 4) = hi;
        }
    }
}

// Output stage of the memory layer (swm:511-519 up to the projection): row = (o * gate) * silu(g), then
// LayerNorm over the 1024-wide row, written as the bf16 operand of the output projection.  One wave per row,
// 16 elements per lane.
constexpr int GLN_D = 1024;
// G16: g is bf16 with rows `ldg` elements apart (a column block of the merged projection GEMM's output)
template <bool G16>
__global__ __launch_bounds__(256) void gate_ln_kernel(const float* __restrict__ o, const float* __restrict__ gate,
                                                      const void* __restrict__ g_any, int64_t ldg,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ y, int64_t M,
                                                      float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4_t v[4];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t at = row * GLN_D + q * 256 + lane * 4;
        const f32x4_t ov = *(const f32x4_t*)(o + at), gv = *(const f32x4_t*)(gate + at);
        f32x4_t xv;
        if (G16) {
            const u32x2_t pk = *(const u32x2_t*)((const bf16_t*)g_any + row * ldg + q * 256 + lane * 4);
            xv = (f32x4_t){__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u), __uint_as_float(pk[1] << 16),
                           __uint_as_float(pk[1] & 0xffff0000u)};
        } else {
            xv = *(const f32x4_t*)((const float*)g_any + at);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float silu = xv[j] / (1.0f + __expf(-xv[j]));
            v[q][j] = (ov[j] * gv[j]) * silu;
            s += v[q][j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mu = s * (1.0f / GLN_D);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[q][j] - mu;
            ss += d * d;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rs = rsqrtf(ss * (1.0f / GLN_D) + eps);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = q * 256 + lane * 4;
        const f32x4_t gm = *(const f32x4_t*)(gamma + c), bt = *(const f32x4_t*)(beta + c);
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (v[q][j] - mu) * rs * gm[j] + bt[j];
        *(u32x2_t*)(y + row * GLN_D + c) = (u32x2_t){pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
    }
}

// The memory layer's two per-head gate projections (swm:494-495: alpha = sigmoid(a_proj(x)), beta = sigmoid(b_proj(x)),
// no bias): x f32 [M, 1024]; WA, WB f32 [H, 1024], H <= 8.  The 2H weight rows live in LDS (rows beyond 2H are zero);
// a wave takes 4 rows of x at a time, every lane forming its 16-column share of the 4 x 16 dot products, and a
// halving exchange (32 + 16 + ... + 1 = 63 cross-lane moves instead of 64 six-step butterflies) leaves lane l with the
// complete sum of row l / 16, output l % 16.  The summation order of a row does not depend on where the row lies.
constexpr int ABG_D = 1024, ABG_ROWS = 4;
__global__ __launch_bounds__(256) void ab_gate_kernel(const float* __restrict__ X, const float* __restrict__ WA,
                                                      const float* __restrict__ WB, float* __restrict__ ALPHA,
                                                      float* __restrict__ BETA, bf16_t* __restrict__ XB, int64_t M, int H) {
    extern __shared__ __align__(16) float abw[];          // [16][1024]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * ABG_D / 4; i += 256) {
        const int o = i / (ABG_D / 4), c = (i % (ABG_D / 4)) * 4;
        f32x4_t w = {0.f, 0.f, 0.f, 0.f};
        if (o < H) w = *(const f32x4_t*)(WA + (int64_t)o * ABG_D + c);
        else if (o < 2 * H) w = *(const f32x4_t*)(WB + (int64_t)(o - H) * ABG_D + c);
        *(f32x4_t*)(abw + o * ABG_D + c) = w;
    }
    __syncthreads();
    const int64_t groups = (M + ABG_ROWS - 1) / ABG_ROWS;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t row0 = grp * ABG_ROWS;
        float acc[ABG_ROWS * 16];
#pragma unroll
        for (int i = 0; i < ABG_ROWS * 16; ++i) acc[i] = 0.f;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            f32x4_t xv[ABG_ROWS];
#pragma unroll
            for (int r = 0; r < ABG_ROWS; ++r) {
                const int64_t row = row0 + r < M ? row0 + r : M - 1;
                xv[r] = *(const f32x4_t*)(X + row * ABG_D + q * 256 + lane * 4);
                // the bf16 copy of x that the projection GEMM takes, made from the values this pass reads anyway
                if (XB != nullptr && row0 + r < M)
                    *(u32x2_t*)(XB + row * ABG_D + q * 256 + lane * 4) = (u32x2_t){pack_bf16x2(xv[r][0], xv[r][1]), pack_bf16x2(xv[r][2], xv[r][3])};
            }
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                const f32x4_t w = *(const f32x4_t*)(abw + o * ABG_D + q * 256 + lane * 4);
#pragma unroll
                for (int r = 0; r < ABG_ROWS; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r * 16 + o] = fmaf(xv[r][e], w[e], acc[r * 16 + o]);
            }
        }
        // halving exchange: after the step with offset `off` a lane keeps the half of its values whose index has
        // the same `off`-bit as the lane id
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const bool upper = (lane & off) != 0;
#pragma unroll
            for (int i = 0; i < off; ++i) {
                const float keep = upper ? acc[i + off] : acc[i];
                const float send = upper ? acc[i] : acc[i + off];
                acc[i] = keep + __shfl_xor(send, off);
            }
        }
        const int r = lane >> 4, o = lane & 15;
        const float y = 1.0f / (1.0f + __expf(-acc[0]));
        if (row0 + r < M) {
            if (o < H) ALPHA[(row0 + r) * H + o] = y;
            else if (o < 2 * H) BETA[(row0 + r) * H + (o - H)] = y;
        }
    }
}

}  // namespace

extern "C" int va_delta_gates_fwd(const float* x, const float* wa, const float* wb, float* alpha, float* beta, void* x_bf16,
                                  int64_t M, int64_t D, int64_t H, hipStream_t stream) {
    VA_CHECK_ARG(x && wa && wb && alpha && beta, "va_delta_gates_fwd: null operand");
    VA_CHECK_ARG(D == ABG_D, "va_delta_gates_fwd: width %lld not supported (1024)", (long long)D);
    VA_CHECK_ARG(H >= 1 && H <= 8, "va_delta_gates_fwd: %lld heads not supported (1..8)", (long long)H);
    VA_CHECK_ARG(M > 0 && M < (1ll << 33), "va_delta_gates_fwd: row count out of range");
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)wa | (uintptr_t)wb) % 16 == 0 && (uintptr_t)x_bf16 % 8 == 0,
                 "va_delta_gates_fwd: misaligned operand");
    static bool lds_ok = false;
    constexpr int lds = 16 * ABG_D * 4;
    if (!lds_ok) {
        if (hipFuncSetAttribute((const void*)ab_gate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            va_set_error("va_delta_gates_fwd: cannot reserve %d bytes of LDS", lds);
            return VA_ERR_LAUNCH;
        }
        lds_ok = true;
    }
    const int64_t blocks_wanted = (M + 4 * ABG_ROWS - 1) / (4 * ABG_ROWS);
    const unsigned grid = (unsigned)(blocks_wanted < 512 ? blocks_wanted : 512);
    hipLaunchKernelGGL(ab_gate_kernel, dim3(grid), dim3(256), lds, stream, x, wa, wb, alpha, beta, (bf16_t*)x_bf16, M, (int)H);
    VA_CHECK_LAUNCH("va_delta_gates_fwd");
    return VA_OK;
}

extern "C" int va_gated_delta_seq(const float* q, const float* k, const float* v, const float* alpha, const float* beta,
                                  const float* state_in, float* out, float* state_out, int64_t B, int64_t L, int64_t H,
                                  int64_t hd, hipStream_t stream) {
    VA_CHECK_ARG(q && k && v && alpha && beta && out && state_out, "va_gated_delta_seq: null operand");
    VA_CHECK_ARG(B > 0 && L > 0 && H > 0, "va_gated_delta_seq: empty problem");
    VA_CHECK_ARG(hd == 64 || hd == 128, "va_gated_delta_seq: head dim %lld not supported (64 or 128)", (long long)hd);
    VA_CHECK_ARG(B * H < (1ll << 31) && L < (1ll << 31), "va_gated_delta_seq: problem too large");
    VA_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)state_out | (uintptr_t)state_in) % 16 == 0,
                 "va_gated_delta_seq: misaligned operand");
    if (hd == 128)
        hipLaunchKernelGGL(gated_delta_kernel<128>, dim3((unsigned)(B * H)), dim3(256), 0, stream, q, k, v, alpha, beta,
                           state_in, out, state_out, (int)L, (int)H);
    else
        hipLaunchKernelGGL(gated_delta_kernel<64>, dim3((unsigned)(B * H)), dim3(128), 0, stream, q, k, v, alpha, beta,
                           state_in, out, state_out, (int)L, (int)H);
    VA_CHECK_LAUNCH("va_gated_delta_seq");
    return VA_OK;
}

extern "C" int va_dwconv1d_causal_silu(const float* x, const float* w, const float* bias, float* out, int64_t B,
                                       int64_t L, int64_t C, int normalize, hipStream_t stream) {
    VA_CHECK_ARG(x && w && bias && out, "va_dwconv1d_causal_silu: null operand");
    VA_CHECK_ARG(B > 0 && L > 0 && C > 0 && C % 4 == 0, "va_dwconv1d_causal_silu: C=%lld must be a positive multiple of 4",
                 (long long)C);
    VA_CHECK_ARG(x != out, "va_dwconv1d_causal_silu: cannot run in place (a row reads its three predecessors)");
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out) % 16 == 0,
                 "va_dwconv1d_causal_silu: misaligned operand");
    const int64_t rows = B * L;
    VA_CHECK_ARG(rows < (1ll << 33) && L < (1ll << 31) && C < (1ll << 31), "va_dwconv1d_causal_silu: problem too large");
    switch (C % 256 == 0 ? C / 256 : 0) {   // whole multiples of 256 channels up to 2048: the sliding-window kernel
        case 1: launch_dwconv_run<1>(x, w, bias, out, B, (int)L, normalize, stream); break;
        case 2: launch_dwconv_run<2>(x, w, bias, out, B, (int)L, normalize, stream); break;
        case 4: launch_dwconv_run<4>(x, w, bias, out, B, (int)L, normalize, stream); break;
        case 8: launch_dwconv_run<8>(x, w, bias, out, B, (int)L, normalize, stream); break;
        default:
            hipLaunchKernelGGL(dwconv_silu_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, w, bias, out,
                               rows, (int)L, (int)C, normalize);
    }
    VA_CHECK_LAUNCH("va_dwconv1d_causal_silu");
    return VA_OK;
}

extern "C" int va_gate_layernorm_fwd(const float* o, const float* gate, const float* g, const float* gamma,
                                     const float* beta, void* y_bf16, int64_t M, int64_t D, float eps, hipStream_t stream) {
    VA_CHECK_ARG(o && gate && g && gamma && beta && y_bf16, "va_gate_layernorm_fwd: null operand");
    VA_CHECK_ARG(D == GLN_D, "va_gate_layernorm_fwd: width %lld not supported (1024)", (long long)D);
    VA_CHECK_ARG(M > 0 && M < (1ll << 33), "va_gate_layernorm_fwd: row count out of range");
    VA_CHECK_ARG(((uintptr_t)o | (uintptr_t)gate | (uintptr_t)g | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y_bf16) % 16 == 0,
                 "va_gate_layernorm_fwd: misaligned operand");
    hipLaunchKernelGGL(gate_ln_kernel<false>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, o, gate, (const void*)g,
                       (int64_t)GLN_D, gamma, beta, (bf16_t*)y_bf16, M, eps);
    VA_CHECK_LAUNCH("va_gate_layernorm_fwd");
    return VA_OK;
}

extern "C" int va_gate_layernorm_fwd_g16(const float* o, const float* gate, const void* g_bf16, int64_t ldg,
                                         const float* gamma, const float* beta, void* y_bf16, int64_t M, int64_t D, float eps,
                                         hipStream_t stream) {
    VA_CHECK_ARG(o && gate && g_bf16 && gamma && beta && y_bf16, "va_gate_layernorm_fwd_g16: null operand");
    VA_CHECK_ARG(D == GLN_D, "va_gate_layernorm_fwd_g16: width %lld not supported (1024)", (long long)D);
    VA_CHECK_ARG(M > 0 && M < (1ll << 33), "va_gate_layernorm_fwd_g16: row count out of range");
    VA_CHECK_ARG(ldg >= D && ldg % 4 == 0, "va_gate_layernorm_fwd_g16: row stride of g must be >= D and keep 8-byte alignment");
    VA_CHECK_ARG(((uintptr_t)o | (uintptr_t)gate | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y_bf16) % 16 == 0 &&
                     (uintptr_t)g_bf16 % 8 == 0,
                 "va_gate_layernorm_fwd_g16: misaligned operand");
    hipLaunchKernelGGL(gate_ln_kernel<true>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, o, gate, g_bf16, ldg, gamma,
                       beta, (bf16_t*)y_bf16, M, eps);
    VA_CHECK_LAUNCH("va_gate_layernorm_fwd_g16");
    return VA_OK;
}

extern "C" int va_dwconv1d_causal_silu_bf16(const void* x_bf16, int64_t ldx, const float* w, const float* bias, float* out,
                                            int64_t B, int64_t L, int64_t C, int normalize, hipStream_t stream) {
    VA_CHECK_ARG(x_bf16 && w && bias && out, "va_dwconv1d_causal_silu_bf16: null operand");
    VA_CHECK_ARG(B > 0 && L > 0, "va_dwconv1d_causal_silu_bf16: empty problem");
    VA_CHECK_ARG(C == 1024, "va_dwconv1d_causal_silu_bf16: C=%lld not supported (1024)", (long long)C);
    VA_CHECK_ARG(ldx >= C && ldx % 8 == 0, "va_dwconv1d_causal_silu_bf16: row stride must be >= C and keep 16-byte alignment");
    VA_CHECK_ARG(((uintptr_t)x_bf16 | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out) % 16 == 0,
                 "va_dwconv1d_causal_silu_bf16: misaligned operand");
    VA_CHECK_ARG(B * L < (1ll << 33) && L < (1ll << 31), "va_dwconv1d_causal_silu_bf16: problem too large");
    constexpr int RUN = 8;      // measured at 64 x 2451 rows: 8 steps per wave 260 us, 4: 293, 2: 363, 12 / 16: 303-307
    const int64_t runs = B * ((L + RUN - 1) / RUN);
    hipLaunchKernelGGL((dwconv_silu_bf16_kernel<RUN>), dim3((unsigned)((runs + 3) / 4)), dim3(256), 0, stream,
                       (const bf16_t*)x_bf16, ldx, w, bias, out, runs, (int)L, normalize);
    VA_CHECK_LAUNCH("va_dwconv1d_causal_silu_bf16");
    return VA_OK;
}
