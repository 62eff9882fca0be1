// Flash-style multi-head self-attention for the encoder (no mask, no dropout): forward and backward.
//
// Layout: element (b, n, h, d) of q/k/v lives at ((b*N + n)*ld + h*HD + d) from the respective base, so
// the three thirds of the packed [M, 3*D] QKV-projection output are used in place; O / dO are [M, D].
//
// All three kernels share one skeleton (gfx950, wave64, MFMA 32x32x16 bf16):
//   * a workgroup is 4 waves; each wave owns 32 "stationary" rows (queries in forward / dQ, keys in
//     dK/dV) whose operand fragments stay in registers; the other side streams through LDS in tiles of
//     64 rows (a ring of two or three slots, 16-byte direct-to-LDS loads, one barrier per tile).
//   * scores are produced with the streamed side as the MFMA A operand, so a lane holds a COLUMN of the
//     score tile (one stationary row) and softmax statistics are per-lane scalars (one cross-half
//     exchange per reduction).  The accumulator tile then feeds the next MFMA directly as its B operand
//     (contraction over the streamed rows, which sit in the registers); the matching A operand, a
//     transposed view of the streamed tile, comes from ds_read_b64_tr_b16.
//   * one LDS image serves both the row reads (ds_read_b128) and the transposed reads: 16-byte chunk
//     index XOR f(row), f chosen so both patterns hit all 64 banks (see swz()).
//   * dQ and dK/dV are separate passes (scores recomputed in each): no atomics, bitwise reproducible.
//   * the softmax bookkeeping is kept off the VALU, which is the co-limiting pipe at head dim 64: scale
//     folded into the stationary operand, running max / log-sum-exp / delta entering as accumulator start
//     values or through one extra MFMA slice (see the kernels).
#include "common.h"
#include <type_traits>

namespace {

constexpr int TILE_ROWS = 64;

template <int HD>
VA_DEVINL int swz(int row) {
    if (HD == 32) return (row >> 2) & 3;                                // 4 chunks per 64-byte row
    if (HD == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);   // 8 chunks per 128-byte row
    return ((row & 3) << 2) | ((row >> 2) & 3);                          // 16 chunks per 256-byte row
}

// One BYTES-wide piece per lane, global -> LDS, through a buffer descriptor over [base, base + span) (base wave-uniform,
// `voff` the lane's byte offset).  The buffer form rather than the flat-family global_load_lds: behind the latter the
// compiler drains every outstanding LDS read before the next use of any; with this one its LDS wait counts are exact.
template <int BYTES>
VA_DEVINL void buf_to_lds(const void* base, int span, int voff, void* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)   // the descriptor type exists for the device target only
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, span, 0x00020000);
    if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_wave_base, 16, voff, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_wave_base, 4, voff, 0, 0, 0);
#endif
}

// Stage a [ROWS x HD] bf16 tile: rows row0..row0+ROWS-1 (clamped to n_rows-1) of `src` (row stride ld).
template <int HD, int ROWS = TILE_ROWS>
VA_DEVINL void stage_tile(const bf16_t* src, int64_t ld, int row0, int n_rows, char* dst, int tid,
                          int wave) {
    constexpr int CPR = HD / 8;                 // chunks per row
    constexpr int PER_THREAD = ROWS * CPR / 256;
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
        const int q = i * 256 + tid;
        const int row = q / CPR;
        const int c = (q % CPR) ^ swz<HD>(row);
        int g = row0 + row;
        g = g < n_rows ? g : n_rows - 1;
        buf_to_lds<16>(src + (int64_t)row0 * ld, (int)((TILE_ROWS - 1) * ld * 2 + HD * 2), (int)(((g - row0) * ld + c * 8) * 2),
                       dst + (i * 256 + wave * 64) * 16);
    }
}

// Per-thread byte offsets of its chunks inside a 64-row tile, computed once; whole tiles are then staged
// from a wave-uniform tile base plus these 32-bit offsets (no per-tile address arithmetic on the VALU).
template <int HD>
struct TileSrc {
    static constexpr int CPR = HD / 8, PER_THREAD = TILE_ROWS * CPR / 256;
    uint32_t off[PER_THREAD];
    VA_DEVINL void init(int64_t ld, int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int q = i * 256 + tid, row = q / CPR;
            off[i] = (uint32_t)(row * ld * 2) + (((q % CPR) ^ swz<HD>(row)) << 4);
        }
    }
};
template <int HD, bool RAGGED>
VA_DEVINL void stage_tile_at(const TileSrc<HD>& ts, const bf16_t* src, int64_t ld, int row0, int n_rows,
                             char* dst, int tid, int wave) {
    if (RAGGED && row0 + TILE_ROWS > n_rows) {
        stage_tile<HD>(src, ld, row0, n_rows, dst, tid, wave);
        return;
    }
    // one descriptor over the tile's rows (scalar registers; wave-uniform base)
#pragma unroll
    for (int i = 0; i < TileSrc<HD>::PER_THREAD; ++i)
        buf_to_lds<16>(src + (int64_t)row0 * ld, (int)((TILE_ROWS - 1) * ld * 2 + HD * 2), (int)ts.off[i],
                       dst + (i * 256 + wave * 64) * 16);
}

// Row-read fragment (MFMA operand with the tile's rows as the M/N index): lane (r = lane&31, h = lane>>5)
// gets tile[row_base + r][16*ks + 8h .. +7].
template <int HD>
VA_DEVINL bf16x8_t row_frag(const char* __restrict__ tile, int row_base, int ks, int lane) {
    const int row = row_base + (lane & 31);
    const int chunk = (2 * ks + (lane >> 5)) ^ swz<HD>(row);
    return *(const bf16x8_t*)(tile + row * (HD * 2) + (chunk << 4));
}

// Transposed fragment: the A operand X^T[c][k'] of a product that contracts over 16 tile rows
// k' = row_base + {0..15} for the 32 columns cb*32 .. cb*32+31, in the k-order of an accumulator tile used
// as the B operand: element j of lane half h is row  row_base + 8*(j>>2) + 4h + (j&3).
template <int HD>
VA_DEVINL bf16x8_t tr_frag(const char* __restrict__ tile, int row_base, int cb, int lane) {
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

// A bf16 fragment times a scalar, rounded back to bf16.
VA_DEVINL bf16x8_t scale_frag(bf16x8_t raw, float c) {
    u32x4_t pk;
#pragma unroll
    for (int j = 0; j < 4; ++j) pk[j] = pack_bf16x2((float)raw[2 * j] * c, (float)raw[2 * j + 1] * c);
    return __builtin_bit_cast(bf16x8_t, pk);
}

// max of three without the NaN-quieting moves fmaxf adds in front of every MFMA result
VA_DEVINL float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
VA_DEVINL float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }
// Same through the half-wave swap instruction (no LDS round trip, no NaN quieting).
VA_DEVINL float xhalf_max_fast(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return max3(__uint_as_float(r[0]), __uint_as_float(r[1]), v);
}
VA_DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Store a transposed 32-wide accumulator set acc[cb][i] = T[col = cb*32 + acc32_row(i,h)][row = lane&31]
// as bf16 rows: out_row[col..col+3] for each group of 4 registers.
// Plain stores on purpose: with the non-temporal hint the forward kernel itself ran 5 % faster, but its 134-MB output no
// longer waited in the last-level cache for the projection GEMM that reads it next, and the training step as a whole
// was 1.5 ms slower (the backward kernels were slower already on their own).
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

// The same through a wave-private LDS area (32 rows x (2*HD + 16) bytes): every lane drops its 8-byte pieces there and
// picks up 16-byte chunks of whole rows, so that a store instruction covers 64 / (HD/8) rows x 2*HD contiguous bytes with
// neighbouring lanes on neighbouring bytes, instead of 32 rows x two 8-byte pieces from lanes 32 apart.  A store
// instruction of the scattered shape holds the wave about 290 cycles, one of the contiguous shape about 100 (measured
// on the GEMM epilogues, tools/quad_epi_prof.py).  out32: row 0 of the 32 rows; all 32 must exist.  LDS operations of one
// wave execute in order, so no wait is needed between the writes and the reads.
template <int HD>
VA_DEVINL void store_rows_lds(const f32x16_t (&acc)[HD / 32], float mul, bf16_t* out32, int64_t ld, char* area, int lane) {
    constexpr int RS = 2 * HD + 16, CH = HD / 8;          // row stride in the area; 16-byte chunks per row
    const int row = lane & 31, h = lane >> 5;
#pragma unroll
    for (int cb = 0; cb < HD / 32; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2_t pk = {pack_bf16x2(acc[cb][4 * g] * mul, acc[cb][4 * g + 1] * mul),
                          pack_bf16x2(acc[cb][4 * g + 2] * mul, acc[cb][4 * g + 3] * mul)};
            *(u32x2_t*)(area + row * RS + (cb * 32 + 8 * g + 4 * h) * 2) = pk;
        }
#pragma unroll
    for (int t = 0; t < CH / 2; ++t) {
        const int r = t * (64 / CH) + lane / CH, c = lane % CH;
        const u32x4_t v = *(const u32x4_t*)(area + r * RS + c * 16);
        *(u32x4_t*)(out32 + (int64_t)r * ld + c * 8) = v;
    }
}

// Workgroup -> (head, 128-row block).  Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest), so
// taken literally the blocks of one head would land on all 8 XCDs and each XCD's L2 would fetch that head's K / V
// (or Q / dO) for itself.  The remap hands every XCD a contiguous run of work ids instead: the blocks of a head run
// on one XCD and share the head's operands in its L2.
struct HeadBlock { int bh, blk; };
VA_DEVINL HeadBlock head_block() {
    const int nblk = gridDim.x;
    const int v = xcd_remap(blockIdx.y * nblk + blockIdx.x, nblk * gridDim.y);
    return {v / nblk, v % nblk};
}

// LDS ring depth of the forward kernel: three 16 KiB slots at head dim 64 (three workgroups per CU still
// fit), two 32 KiB slots at head dim 128.
constexpr int fwd_ring_slots(int hd) { return hd <= 64 ? 3 : 2; }

// ------------------------------------------------------------------------------------------------
// forward.  The loop is bound by vector-instruction issue (32 exponentials + the bookkeeping per lane and
// 64-key tile against 16 MFMAs), so the bookkeeping is kept off the VALU:
//  * Q is scaled by log2(e)/sqrt(hd) once, and the score accumulators START at -m (a register block that
//    only changes when the maximum moves), so the MFMA chain itself delivers s*c - m: no per-score fma;
//  * the running maximum m is only raised when some row's tile maximum exceeds it by more than 2^THR
//    (probabilities then stay below 2^THR, harmless in bf16 x f32), so rescaling O is a rare,
//    wave-uniform branch taken before any score of the tile is exponentiated.
// ------------------------------------------------------------------------------------------------
// MODE: which keys a query sees.  0: all, N a multiple of 64.  1: all, ragged N.  2: block-causal, a query of
// timestep s = n / tpt sees the keys of timesteps <= s (the predictor's mask, swm:1008-1028); N % tpt == 0.
// Modes 1 and 2 are one mechanism: a per-lane key limit, applied only in tiles that reach past it.
// DROP: dropout on the attention probabilities (swm:326, training mode): the probabilities that enter P*V are
// masked and rescaled, the softmax normaliser is not.  Element (q, k) of head-batch bh takes its decision from
// the word drop_mix(drop_weyl(key + bh * DROP_HEAD, q * (NS / 2) + k / 2)), field k & 1, NS = N rounded up to
// even.  The rescale 1 / (1 - p) never touches an element: it is folded into the factor each kernel applies to
// its accumulators on the way out (and the backward's delta is pre-divided by it).
struct DropArgs {
    uint32_t key, thresh16;
    float scale;
};
VA_DEVINL uint32_t drop_half_stride(int N) { return ((uint32_t)N + 1u) >> 1; }   // NS / 2

// Explicit boolean mask (swm:316-323, True = do not attend), bit-packed once per call: `by_q` [.., N, W] words with
// W = 2 * ceil(N / 64), bit j of word w of row q set <=> key 32 w + j is masked for query q (keys >= N: set);
// `by_k` the transpose (row = key, bit = query; queries >= N: set), which the key-stationary backward pass reads.
// sb / sh: words from one batch entry / head to the next (0 = the mask is shared along that axis).
struct MaskArgs {
    const uint32_t* by_q;
    const uint32_t* by_k;
    int64_t sb, sh;
};
// In a 64-row tile a lane holds rows kb * 32 + acc32_row(i, h) = kb * 32 + 8 (i >> 2) + 4 h + (i & 3): with the tile's
// two words shifted right by 4 h once, register i tests the compile-time bit 8 (i >> 2) + (i & 3).
VA_DEVINL bool mask_bit(uint32_t shifted_word, int i) { return (shifted_word >> (8 * (i >> 2) + (i & 3))) & 1u; }

template <int HD, int MODE, bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                       const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                       float* __restrict__ LSE2, int N, int H, int64_t ld, int64_t ldo,
                                                       float c, int tpt, DropArgs drop, MaskArgs mask) {
    constexpr bool RAGGED = MODE != 0;
    constexpr bool BITS = MODE == 3;
    constexpr int KS = HD / 16, CB = HD / 32, TB = TILE_ROWS * HD * 2;
    constexpr int NBUF = fwd_ring_slots(HD), AHEAD = NBUF - 1, LOADS = 2 * TileSrc<HD>::PER_THREAD;
    constexpr float THR = 6.0f;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HeadBlock hb = head_block();
    const int bh = hb.bh, b = bh / H, hh = bh % H;
    const int q0 = hb.blk * 128 + wave * 32;
    const bf16_t* Qb = Q + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Kb = K + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Vb = V + (int64_t)b * N * ld + hh * HD;

    const int qrow = q0 + (lane & 31);
    const int qrow_c = qrow < N ? qrow : N - 1;
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t raw = *(const bf16x8_t*)(Qb + (int64_t)qrow_c * ld + ks * 16 + h * 8);
        u32x4_t pk;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            pk[j] = pack_bf16x2((float)raw[2 * j] * c, (float)raw[2 * j + 1] * c);
        qf[ks] = __builtin_bit_cast(bf16x8_t, pk);
    }

    f32x16_t o[CB], negm;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[cb][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;
    float m = 0.f, l = 0.f;
    // BITS: the first tile need not hold a visible key, so "first tile" becomes "nothing seen yet", per query
    bool seen = false;
    const uint32_t* mrow = nullptr;
    if (BITS) mrow = mask.by_q + b * mask.sb + hh * mask.sh + (int64_t)qrow_c * (2 * ((N + TILE_ROWS - 1) / TILE_ROWS));

    int klim = N;                                   // this lane's query sees keys [0, klim)
    int nt = (N + TILE_ROWS - 1) / TILE_ROWS;       // tiles the workgroup walks (uniform)
    if (MODE == 2) {
        klim = (qrow_c / tpt + 1) * tpt;
        int q_last = hb.blk * 128 + 127;
        q_last = q_last < N ? q_last : N - 1;
        nt = ((q_last / tpt + 1) * tpt + TILE_ROWS - 1) / TILE_ROWS;
    }
    // DROP: Weyl term of this lane's first key pair (keys 4h, 4h+1 of tile 0), and the threshold aligned to bit 16
    const uint32_t drop_w = drop_weyl(drop.key + (uint32_t)bh * DROP_HEAD,
                                      (uint32_t)qrow_c * drop_half_stride(N) + 2u * (uint32_t)h);
    const uint32_t drop_thi = drop.thresh16 << 16;
    TileSrc<HD> ts;
    ts.init(ld, tid);
    auto stage = [&](int tile, int slot) {
        char* nb = smem + slot * 2 * TB;
        stage_tile_at<HD, RAGGED>(ts, Kb, ld, tile * TILE_ROWS, N, nb, tid, wave);
        stage_tile_at<HD, RAGGED>(ts, Vb, ld, tile * TILE_ROWS, N, nb + TB, tid, wave);
    };
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (i < nt) stage(i, i);
    // One tile.  SLOT is a compile-time constant (the tile loop is unrolled by the ring depth), so every LDS
    // address is a loop-invariant register plus an immediate.  Tile t+AHEAD is issued in the middle of the
    // step, once all of this tile's LDS reads are on their way.
#ifdef VA_ATTN_PROFILE
    uint64_t pc[4] = {0, 0, 0, 0};
#define VA_STAMP(i) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); pc[i] += now_ - last_; last_ = now_; }
#else
#define VA_STAMP(i)
#endif
    auto tile_step = [&](int t, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
#ifdef VA_ATTN_PROFILE
        uint64_t last_ = __builtin_amdgcn_s_memtime();
#endif
        if (AHEAD > 1 && t + AHEAD - 1 < nt) wait_vm<LOADS*(AHEAD - 1)>();
        else wait_vm0();
        // raw barrier: __syncthreads() carries a fence that drains vmcnt to zero
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        VA_STAMP(0)
        const char* kt = smem + SLOT * 2 * TB;
        const char* vt = kt + TB;
        u32x2_t mw = {0u, 0u};
        if (BITS) mw = *(const u32x2_t*)(mrow + 2 * t);

        // hd 64: every LDS read of a phase is issued before its first MFMA (the fragments fit the register
        // file); hd 128: fragments are read next to their use.
        constexpr bool PRELOAD = HD <= 64;
        bf16x8_t kf[2][KS];
        if (PRELOAD) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = row_frag<HD>(kt, kb * 32, ks, lane);
        }
        f32x16_t st[2];   // scores of this tile relative to the running maximum, log2 units
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = mfma32(PRELOAD ? kf[kb][0] : row_frag<HD>(kt, kb * 32, 0, lane), qf[0], negm);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks)
                st[kb] = mfma32(PRELOAD ? kf[kb][ks] : row_frag<HD>(kt, kb * 32, ks, lane), qf[ks], st[kb]);
        }
        bf16x8_t vf[2][2][CB];
        if (PRELOAD) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) vf[kb][s][cb] = tr_frag<HD>(vt, kb * 32 + 16 * s, cb, lane);
        }
        if (t + AHEAD < nt) stage(t + AHEAD, (SLOT + AHEAD) % NBUF);
        if (BITS) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const uint32_t w = mw[kb] >> (4 * h);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (mask_bit(w, i)) st[kb][i] = -INFINITY;
            }
        } else if (MODE != 0 && __any((t + 1) * TILE_ROWS > klim)) {   // only builds with a key limit carry the selects
            const int rel = klim - t * TILE_ROWS;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (kb * 32 + acc32_row(i, h) >= rel) st[kb][i] = -INFINITY;
        }
        float mq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mq[g] = max3(st[0][4 * g], st[0][4 * g + 1], st[0][4 * g + 2]);
            mq[g] = max3(mq[g], st[0][4 * g + 3], st[1][4 * g]);
            mq[g] = max3(mq[g], st[1][4 * g + 1], st[1][4 * g + 2]);
        }
        float mx = max3(max3(mq[0], st[1][3], st[1][7]), max3(mq[1], st[1][11], st[1][15]), max3(mq[2], mq[3], mq[2]));
        mx = xhalf_max_fast(mx);
        VA_STAMP(1)
        // BITS: a query that has seen no key yet takes its first visible tile's maximum (either sign); a tile that is
        // masked out for it entirely (mx = -inf) moves nothing
        const bool fresh = BITS ? !seen : t == 0;
        const bool empty = BITS && mx == -INFINITY;
        if (BITS ? !__all(fresh ? empty : mx <= THR) : (t == 0 || !__all(mx <= THR))) {
            // move the maximum: by the tile's own maximum on the first tile (either sign), upwards only later
            const float delta = empty ? 0.f : fresh ? mx : fmaxf(mx, 0.f);
            const float alpha = fresh ? 1.f : fast_exp2(-delta);
            m += delta;
            l *= alpha;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[cb][i] *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                negm[i] = -m;
                st[0][i] -= delta;
                st[1][i] -= delta;
            }
        }
        if (BITS) seen = seen || !empty;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = fast_exp2(st[kb][i]);
                st[kb][i] = p;
                l += p;
            }
        if (DROP) {
            // registers i, i+1 (i even) hold keys t*64 + kb*32 + acc32_row(i, h) and the next one: one pair
            const uint32_t wt = drop_w + (uint32_t)t * ((TILE_ROWS / 2) * DROP_WEYL);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const uint32_t hsh = drop_mix(wt + (uint32_t)(kb * 16 + 4 * (i >> 2) + ((i & 3) >> 1)) * DROP_WEYL);
                    st[kb][i] = drop_keep_even(hsh, drop_thi) ? st[kb][i] : 0.f;
                    st[kb][i + 1] = drop_keep_odd(hsh, drop_thi) ? st[kb][i + 1] : 0.f;
                }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t pf = pack_acc(st[kb], s);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    o[cb] = mfma32(PRELOAD ? vf[kb][s][cb] : tr_frag<HD>(vt, kb * 32 + 16 * s, cb, lane), pf, o[cb]);
            }
        VA_STAMP(2)
    };
    for (int t0 = 0; t0 < nt; t0 += NBUF) {
        tile_step(t0, std::integral_constant<int, 0>{});
        if (t0 + 1 < nt) tile_step(t0 + 1, std::integral_constant<int, 1>{});
        if (NBUF > 2 && t0 + 2 < nt) tile_step(t0 + 2, std::integral_constant<int, (NBUF > 2 ? 2 : 0)>{});
    }
    l += __shfl_xor(l, 32);
    // the tiles are done with: the LDS becomes the waves' transposition areas for whole-row stores
    wait_vm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (q0 + 32 <= N)
        store_rows_lds<HD>(o, (DROP ? drop.scale : 1.0f) / l, O + ((int64_t)b * N + q0) * ldo + hh * HD, ldo,
                           smem + wave * (32 * (2 * HD + 16)), lane);
    else if (qrow < N)
        store_rows<HD>(o, (DROP ? drop.scale : 1.0f) / l, O + ((int64_t)b * N + qrow) * ldo + hh * HD, h);
    if (qrow < N && h == 0) LSE2[(int64_t)bh * N + qrow] = m + __log2f(l);
#ifdef VA_ATTN_PROFILE
    if (lane < 3) LSE2[(int64_t)bh * N + q0 + lane] = (float)pc[lane] / nt;
#endif
#undef VA_STAMP
}

// ------------------------------------------------------------------------------------------------
// x split exactly into three bf16 terms {hi, mid | lo, 0} (8+8+8 mantissa bits), laid out as the first two
// dwords of an MFMA operand fragment: how -lse2 and -delta of a query reach the dK/dV pass (see there).
// ------------------------------------------------------------------------------------------------
VA_DEVINL void split3_bf16(float x, uint32_t& d0, uint32_t& d1) {
    const uint32_t hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    const uint32_t mid = __float_as_uint(r) & 0xffff0000u;
    const float lo = r - __uint_as_float(mid);
    d0 = (hi >> 16) | mid;
    d1 = __float_as_uint(lo) >> 16;    // at most 8 significant bits are left: exact
}

// ------------------------------------------------------------------------------------------------
// dQ pass: query-stationary, streams K and V.  It also produces what both backward passes need per query:
// delta = sum_d dO * O (each lane already holds half of its query's dO row; the O row is read the same way), kept
// as -mul * delta (mul = 1, or 1 - p when the dropout rescale is folded into the output factors), written to
// NEG_DELTA and, together with -lse2, as the 16-byte record the dK/dV pass feeds to the matrix pipe.
// ------------------------------------------------------------------------------------------------
template <int HD, bool RAGGED, bool DROP = false, bool BITS = false>
__global__ __launch_bounds__(256, HD <= 64 ? 2 : 1) void attn_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ V, const bf16_t* __restrict__ O,
                                                          const bf16_t* __restrict__ DO,
                                                          const float* __restrict__ LSE2,
                                                          float* __restrict__ NEG_DELTA, u32x4_t* __restrict__ AUX,
                                                          bf16_t* __restrict__ DQ, int N, int H, int64_t ld,
                                                          int64_t ldo, int64_t lddq, float c, float scale, float mul,
                                                          DropArgs drop, MaskArgs mask) {
    constexpr int KS = HD / 16, CB = HD / 32, TB = TILE_ROWS * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HeadBlock hb = head_block();
    const int bh = hb.bh, b = bh / H, hh = bh % H;
    const int q0 = hb.blk * 128 + wave * 32;
    const bf16_t* Qb = Q + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Kb = K + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Vb = V + (int64_t)b * N * ld + hh * HD;
    const bf16_t* DOb = DO + (int64_t)b * N * ldo + hh * HD;

    const int qrow = q0 + (lane & 31);
    const int qrow_c = qrow < N ? qrow : N - 1;
    bf16x8_t qf[KS], dof[KS];   // qf carries the softmax scale (as in the forward kernel)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = scale_frag(*(const bf16x8_t*)(Qb + (int64_t)qrow_c * ld + ks * 16 + h * 8), c);
        dof[ks] = *(const bf16x8_t*)(DOb + (int64_t)qrow_c * ldo + ks * 16 + h * 8);
    }
    // A lane holds one query's column of every score tile, so -lse and -delta are per-lane constants: the
    // score / dP accumulators start from them and the MFMA chains deliver s*c - lse and dP - delta directly.
    const float lse = LSE2[(int64_t)bh * N + qrow_c];
    float dot = 0.f;
    {
        const bf16_t* orow = O + ((int64_t)b * N + qrow_c) * ldo + hh * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4_t ov = *(const u32x4_t*)(orow + ks * 16 + h * 8);
            const u32x4_t dv = __builtin_bit_cast(u32x4_t, dof[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dot += __uint_as_float(ov[j] << 16) * __uint_as_float(dv[j] << 16);
                dot += __uint_as_float(ov[j] & 0xffff0000u) * __uint_as_float(dv[j] & 0xffff0000u);
            }
        }
    }
    dot += __shfl_xor(dot, 32);                 // the other half of the row sits in the partner half-lane
    const float ndl = -(dot * mul);
    if (h == 0 && qrow < N) {
        const int64_t at = (int64_t)bh * N + qrow;
        NEG_DELTA[at] = ndl;
        u32x4_t rec;
        uint32_t d0, d1;
        split3_bf16(-lse, d0, d1);
        rec[0] = d0; rec[1] = d1;
        split3_bf16(ndl, d0, d1);
        rec[2] = d0; rec[3] = d1;
        AUX[at] = rec;
    }
    f32x16_t neg_lse, neg_dl;
#pragma unroll
    for (int i = 0; i < 16; ++i) { neg_lse[i] = -lse; neg_dl[i] = ndl; }

    f32x16_t dq[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[cb][i] = 0.f;

    const uint32_t drop_w = drop_weyl(drop.key + (uint32_t)bh * DROP_HEAD,
                                      (uint32_t)qrow_c * drop_half_stride(N) + 2u * (uint32_t)h);
    const uint32_t drop_thi = drop.thresh16 << 16;
    const int nt = (N + TILE_ROWS - 1) / TILE_ROWS;
    const uint32_t* mrow = nullptr;
    if (BITS) mrow = mask.by_q + b * mask.sb + hh * mask.sh + (int64_t)qrow_c * (2 * nt);
    TileSrc<HD> ts;
    ts.init(ld, tid);
    auto stage = [&](int tile, int slot) {
        char* nb = smem + slot * 2 * TB;
        stage_tile_at<HD, RAGGED>(ts, Kb, ld, tile * TILE_ROWS, N, nb, tid, wave);
        stage_tile_at<HD, RAGGED>(ts, Vb, ld, tile * TILE_ROWS, N, nb + TB, tid, wave);
    };
    stage(0, 0);
    auto tile_step = [&](int t, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        wait_vm0();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        u32x2_t mw = {0u, 0u};
        if (BITS) mw = *(const u32x2_t*)(mrow + 2 * t);
        if (t + 1 < nt) stage(t + 1, SLOT ^ 1);
        const char* kt = smem + SLOT * 2 * TB;
        const char* vt = kt + TB;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8_t kfr[KS], vfr[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kfr[ks] = row_frag<HD>(kt, kb * 32, ks, lane);
                vfr[ks] = row_frag<HD>(vt, kb * 32, ks, lane);
            }
            f32x16_t s = mfma32(kfr[0], qf[0], neg_lse);
            f32x16_t dp = mfma32(vfr[0], dof[0], neg_dl);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) {
                s = mfma32(kfr[ks], qf[ks], s);
                dp = mfma32(vfr[ks], dof[ks], dp);
            }
            bf16x8_t ktr[2][CB];
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) ktr[sidx][cb] = tr_frag<HD>(kt, kb * 32 + 16 * sidx, cb, lane);
            const bool ragged = RAGGED && !BITS && (t == nt - 1);      // BITS: keys past N are masked bits
            const uint32_t mword = mw[kb] >> (4 * h);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p = fast_exp2(s[i]);
                if (ragged && t * TILE_ROWS + kb * 32 + acc32_row(i, h) >= N) p = 0.f;
                if (BITS && mask_bit(mword, i)) p = 0.f;
                s[i] = DROP ? p : (BITS && p == 0.f) ? 0.f : p * dp[i];
            }
            if (DROP) {      // dS / scale = P * (keep ? dO V^T - delta/scale : -delta/scale); DELTA holds delta/scale
                const uint32_t wt = drop_w + (uint32_t)(t * (TILE_ROWS / 2) + kb * 16) * DROP_WEYL;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const uint32_t hsh = drop_mix(wt + (uint32_t)(4 * (i >> 2) + ((i & 3) >> 1)) * DROP_WEYL);
                    s[i] *= drop_keep_even(hsh, drop_thi) ? dp[i] : neg_dl[i];
                    s[i + 1] *= drop_keep_odd(hsh, drop_thi) ? dp[i + 1] : neg_dl[i + 1];
                }
            }
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const bf16x8_t dsf = pack_acc(s, sidx);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) dq[cb] = mfma32(ktr[sidx][cb], dsf, dq[cb]);
            }
        }
    };
    for (int t0 = 0; t0 < nt; t0 += 2) {
        tile_step(t0, std::integral_constant<int, 0>{});
        if (t0 + 1 < nt) tile_step(t0 + 1, std::integral_constant<int, 1>{});
    }
    wait_vm0();
    __builtin_amdgcn_s_barrier();      // the tiles are done with: the LDS becomes the waves' transposition areas
    asm volatile("" ::: "memory");
    if (q0 + 32 <= N)
        store_rows_lds<HD>(dq, DROP ? scale * drop.scale : scale, DQ + ((int64_t)b * N + q0) * lddq + hh * HD, lddq,
                           smem + wave * (32 * (2 * HD + 16)), lane);
    else if (qrow < N)
        store_rows<HD>(dq, DROP ? scale * drop.scale : scale, DQ + ((int64_t)b * N + qrow) * lddq + hh * HD, h);
}

// ------------------------------------------------------------------------------------------------
// dK/dV pass: key-stationary, streams Q and dO.  Here a lane holds one KEY's column of the score tile and
// the queries run over the accumulator registers, so -lse2 / -delta differ per register.  They enter
// through the matrix pipe instead of the VALU: one extra 16-deep MFMA slice per chain whose A operand
// carries the three bf16 terms of -lse2[query] (or -delta[query]) in k = 0..2 and whose B operand is 1.0
// there; the chains then deliver s*c - lse2 and dP - delta directly.  K carries the softmax scale.
// ------------------------------------------------------------------------------------------------
template <int HD, bool DROP>   // Q | dO | records | DROP: f32 -delta of the tile's queries
constexpr int dkv_stage_bytes() { return 2 * TILE_ROWS * HD * 2 + TILE_ROWS * 16 + (DROP ? TILE_ROWS * 4 : 0); }

template <int HD, bool RAGGED, bool DROP = false, bool BITS = false>
__global__ __launch_bounds__(256, HD <= 64 ? 2 : 1) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                           const bf16_t* __restrict__ V, const bf16_t* __restrict__ DO,
                                                           const u32x4_t* __restrict__ AUX,
                                                           const float* __restrict__ NEG_DELTA, bf16_t* __restrict__ DK,
                                                           bf16_t* __restrict__ DV, int N, int H, int64_t ld,
                                                           int64_t ldo, int64_t lddq, float c, float scale, DropArgs drop,
                                                           MaskArgs mask) {
    constexpr int KS = HD / 16, CB = HD / 32, TB = TILE_ROWS * HD * 2;
    constexpr int STAGE = dkv_stage_bytes<HD, DROP>();
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][STAGE] | 16 zero bytes
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HeadBlock hb = head_block();
    const int bh = hb.bh, b = bh / H, hh = bh % H;
    const int k0 = hb.blk * 128 + wave * 32;
    const bf16_t* Qb = Q + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Kb = K + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Vb = V + (int64_t)b * N * ld + hh * HD;
    const bf16_t* DOb = DO + (int64_t)b * N * ldo + hh * HD;
    const u32x4_t* aux_b = AUX + (int64_t)bh * N;

    const int krow = k0 + (lane & 31);
    const int krow_c = krow < N ? krow : N - 1;
    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = scale_frag(*(const bf16x8_t*)(Kb + (int64_t)krow_c * ld + ks * 16 + h * 8), c);
        vf[ks] = *(const bf16x8_t*)(Vb + (int64_t)krow_c * ld + ks * 16 + h * 8);
    }
    // B operand of the extra slice: 1.0 at k = 0..2 (lane half 0 holds k = 0..7), zero elsewhere
    const u32x4_t ones_u = {h == 0 ? 0x3f803f80u : 0u, h == 0 ? 0x00003f80u : 0u, 0u, 0u};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);
    if (tid < 4) ((uint32_t*)(smem + 2 * STAGE))[tid] = 0u;   // what lane half 1 reads as its (all-zero) record

    f32x16_t dk[CB], dv[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[cb][i] = 0.f; dv[cb][i] = 0.f; }

    // DROP: Weyl term of (query 4h, this lane's key), its step per query, and the shift that moves this key's
    // field (low half for an even key) to the top of the word
    const uint32_t drop_qstep = drop_half_stride(N) * DROP_WEYL;
    const uint32_t drop_w = drop_weyl(drop.key + (uint32_t)bh * DROP_HEAD, (uint32_t)krow_c >> 1) +
                            4u * (uint32_t)h * drop_qstep;
    const uint32_t drop_sh = (krow_c & 1) ? 0u : 16u;
    const uint32_t drop_thi = drop.thresh16 << 16;
    const int nt = (N + TILE_ROWS - 1) / TILE_ROWS;
    const uint32_t* mrow = nullptr;      // this lane's key: which queries do not see it
    if (BITS) mrow = mask.by_k + b * mask.sb + hh * mask.sh + (int64_t)krow_c * (2 * nt);
    TileSrc<HD> ts_q, ts_do;
    ts_q.init(ld, tid);
    ts_do.init(ldo, tid);
    auto stage = [&](int tile, int slot) {
        char* nb = smem + slot * STAGE;
        stage_tile_at<HD, RAGGED>(ts_q, Qb, ld, tile * TILE_ROWS, N, nb, tid, wave);
        stage_tile_at<HD, RAGGED>(ts_do, DOb, ldo, tile * TILE_ROWS, N, nb + TB, tid, wave);
        if (wave == 0) {   // 64 records of 16 bytes; rows past N are clamped here and masked in the compute
            int r = tile * TILE_ROWS + lane;
            r = r < N ? r : N - 1;
            buf_to_lds<16>(aux_b, N * 16, r * 16, nb + 2 * TB);
        }
        if (DROP && wave == 1) {   // the same queries' -delta as plain f32, read back per register in the mask step
            int r = tile * TILE_ROWS + lane;
            r = r < N ? r : N - 1;
            buf_to_lds<4>(NEG_DELTA + (int64_t)bh * N, N * 4, r * 4, nb + 2 * TB + TILE_ROWS * 16);
        }
    };
    stage(0, 0);
    auto tile_step = [&](int t, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        wait_vm0();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        u32x2_t mw = {0u, 0u};
        if (BITS) mw = *(const u32x2_t*)(mrow + 2 * t);
        if (t + 1 < nt) stage(t + 1, SLOT ^ 1);
        const char* qt = smem + SLOT * STAGE;
        const char* dot = qt + TB;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const char* rec_at = h == 0 ? qt + 2 * TB + (qb * 32 + (lane & 31)) * 16 : smem + 2 * STAGE;
            const u32x4_t rec = *(const u32x4_t*)rec_at;
            const u32x4_t a_lse = {rec[0], rec[1], 0u, 0u}, a_dl = {rec[2], rec[3], 0u, 0u};
            bf16x8_t qfr[KS], dofr[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qfr[ks] = row_frag<HD>(qt, qb * 32, ks, lane);
                dofr[ks] = row_frag<HD>(dot, qb * 32, ks, lane);
            }
            f32x16_t s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
            s = mfma32(__builtin_bit_cast(bf16x8_t, a_lse), ones, s);       // [query][key] = -lse2[query]
            dp = mfma32(__builtin_bit_cast(bf16x8_t, a_dl), ones, dp);      //               -delta[query]
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = mfma32(qfr[ks], kf[ks], s);
                dp = mfma32(dofr[ks], vf[ks], dp);
            }
            bf16x8_t dotr[2][CB], qtr[2][CB];
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    dotr[sidx][cb] = tr_frag<HD>(dot, qb * 32 + 16 * sidx, cb, lane);
                    qtr[sidx][cb] = tr_frag<HD>(qt, qb * 32 + 16 * sidx, cb, lane);
                }
            f32x16_t pr;
            const bool ragged = RAGGED && !BITS && (t == nt - 1);      // BITS: queries past N are masked bits
            const uint32_t mword = mw[qb] >> (4 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t nd;        // DROP: -delta of registers 4g..4g+3 (queries qb*32 + 8g + 4h + 0..3), what dP - delta
                                   // is for a dropped element
                if (DROP) nd = *(const f32x4_t*)(qt + 2 * TB + TILE_ROWS * 16 + (qb * 32 + 8 * g + 4 * h) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * g + j;
                    float p = fast_exp2(s[i]);
                    if (ragged && t * TILE_ROWS + qb * 32 + acc32_row(i, h) >= N) p = 0.f;
                    if (BITS && mask_bit(mword, i)) p = 0.f;
                    if (DROP) {      // one word per element here: the pair partner (key ^ 1) lives in the neighbouring lane
                        const uint32_t qi = (uint32_t)(t * TILE_ROWS + qb * 32 + 8 * g + j);
                        const uint32_t hsh = drop_mix(drop_w + qi * drop_qstep);
                        const bool keep = (hsh << drop_sh) >= drop_thi;
                        pr[i] = keep ? p : 0.f;
                        s[i] = p * (keep ? dp[i] : nd[j]);
                    } else {
                        pr[i] = p;
                        s[i] = (BITS && p == 0.f) ? 0.f : p * dp[i];     // a fully masked query's dP - delta is NaN
                    }
                }
                // keep the four groups' hash chains apart: interleaved, their temporaries push the kernel into spills
                if (DROP) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const bf16x8_t pf = pack_acc(pr, sidx);
                const bf16x8_t dsf = pack_acc(s, sidx);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    dv[cb] = mfma32(dotr[sidx][cb], pf, dv[cb]);
                    dk[cb] = mfma32(qtr[sidx][cb], dsf, dk[cb]);
                }
            }
        }
    };
    for (int t0 = 0; t0 < nt; t0 += 2) {
        tile_step(t0, std::integral_constant<int, 0>{});
        if (t0 + 1 < nt) tile_step(t0 + 1, std::integral_constant<int, 1>{});
    }
    wait_vm0();
    __builtin_amdgcn_s_barrier();      // the tiles are done with: the LDS becomes the waves' transposition areas
    asm volatile("" ::: "memory");
    if (k0 + 32 <= N) {
        char* area = smem + wave * (32 * (2 * HD + 16));
        store_rows_lds<HD>(dk, DROP ? scale * drop.scale : scale, DK + ((int64_t)b * N + k0) * lddq + hh * HD, lddq, area, lane);
        store_rows_lds<HD>(dv, DROP ? drop.scale : 1.0f, DV + ((int64_t)b * N + k0) * lddq + hh * HD, lddq, area, lane);
    } else if (krow < N) {
        store_rows<HD>(dk, DROP ? scale * drop.scale : scale, DK + ((int64_t)b * N + krow) * lddq + hh * HD, h);
        store_rows<HD>(dv, DROP ? drop.scale : 1.0f, DV + ((int64_t)b * N + krow) * lddq + hh * HD, h);
    }
}

// ------------------------------------------------------------------------------------------------
// dK/dV pass, head dim 64, whole tiles, no mask code: the same arithmetic as attn_bwd_dkv_kernel, software-pipelined.
//
// In the kernel above a wave issues a block's score MFMAs, waits for them, runs the block's exponentials with the
// matrix pipe idle, then issues the gradient MFMAs; a second wave on the SIMD covers little of that (each wave issues in
// order, and both want the same pipe at the same time).  Here ONE wave per SIMD owns 64 keys (two 32-key sub-blocks, so
// every query fragment read from LDS feeds twice the MFMAs) and walks (32-query, 32-key) blocks in steps that never
// leave the matrix pipe waiting for their own vector work: while the 10 MFMAs of the NEXT block's S and dP issue, the
// exp2 / multiply / convert instructions of the CURRENT block sit in the gaps between them, and so do the 8 MFMAs that
// add the PREVIOUS block into dV and dK and the LDS reads of later blocks.  S/dP, the packed operands and both kinds of
// streamed fragments are double-buffered in registers (the sub-block / query-block index is the parity), so nothing a
// step's MFMAs consume is produced inside that step and nothing is copied.
//
// A wave issues in order, so the schedule IS the instruction order.  Every MFMA and every vector instruction of the loop
// is a volatile asm statement (volatile statements keep their order), LDS reads are ordinary loads fenced by empty
// memory-clobbering statements (which pins them between the same MFMAs), and the order itself comes from
// tools/gen_attn_dkv.py (attn_dkv64_body.inc) together with the distances it keeps: the hardware does not interlock a
// vector read of an MFMA result, nor the use of a v_exp_f32 result by the next instruction.
// Accumulators (dK, dV: a[0:127], named literally), the stationary key / value fragments and the streamed row fragments
// of Q / dO (LDS reads land there directly) fill the accumulation half of the register file; scores, probabilities,
// packed operands and the transposed fragments live in the other half.
// Results are bit-identical to attn_bwd_dkv_kernel: every accumulation chain sees the same terms in the same order.
// ------------------------------------------------------------------------------------------------
#define DKV_PIN0 asm volatile("" ::: "memory")
#ifndef VA_DKV64_BODY            // a diagnostic build may name another (timing-only) instruction order
#define VA_DKV64_BODY "attn_dkv64_body.inc"
#endif
constexpr int DKV64_SLOTS = 4;
constexpr int DKV64_STAGE = 2 * TILE_ROWS * 64 * 2 + TILE_ROWS * 16;     // Q | dO | records of one 64-query tile

// dK and dV (8 accumulators of 16 registers) live in a[0:127], named literally in the instruction text.  They are not
// C++ values: left to the register allocator, a loop-carried accumulator that only asm statements touch is parked in
// vector registers between its uses and copied in and out around every MFMA.  EVERY MFMA statement of the kernel lists
// a0..a127 as clobbered, so the compiler never places a value that lives across one of them (the streamed fragments
// always do) into that block, and the kernel descriptor reserves it.
#define VA_ACC128 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
VA_DEVINL void mfz_vv(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b) {      // D = A * B (all in vector registers)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b) : VA_ACC128);
}
VA_DEVINL void mf_aa(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b) {       // D += A * B, A and B in accumulation registers
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b) : VA_ACC128);
}
template <int R>                                                                // accumulator R (a[16R : 16R+15]) += A * B
VA_DEVINL void mf_acc(const bf16x8_t& a, const bf16x8_t& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(16 * R), "n"(16 * R + 15) : VA_ACC128);
}
template <int R>                                                                // accumulator R = 0 (0 * 0 + 0)
VA_DEVINL void mf_acc_clear(const bf16x8_t& zero) {
    // s_nop: the compiler materialises `zero` with a move right in front of this statement and pads no wait states
    // between a vector write and an MFMA it cannot see inside an asm statement
    asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, %0, 0" ::"v"(zero), "n"(16 * R), "n"(16 * R + 15) : VA_ACC128);
}
template <int I>
VA_DEVINL float acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(I));
    return x;
}
template <int R>
VA_DEVINL f32x16_t acc_tile() {
    return (f32x16_t){acc_read<16 * R + 0>(), acc_read<16 * R + 1>(), acc_read<16 * R + 2>(), acc_read<16 * R + 3>(),
                      acc_read<16 * R + 4>(), acc_read<16 * R + 5>(), acc_read<16 * R + 6>(), acc_read<16 * R + 7>(),
                      acc_read<16 * R + 8>(), acc_read<16 * R + 9>(), acc_read<16 * R + 10>(), acc_read<16 * R + 11>(),
                      acc_read<16 * R + 12>(), acc_read<16 * R + 13>(), acc_read<16 * R + 14>(), acc_read<16 * R + 15>()};
}
VA_DEVINL float a_exp2(float x) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
VA_DEVINL float a_mul(float x, float y) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
VA_DEVINL uint32_t a_cvt_pk(float lo, float hi) {
    uint32_t r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// STAMPS (diagnostic library only, tools/attn_debug.hip): s_memtime stamps around the parts of a tile, summed per wave and
// written to `stamps` (8 floats per wave: tile-top waits + barrier + staging, the four steps, prologue, epilogue, tiles).
template <bool STAMPS = false>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ V, const bf16_t* __restrict__ DO,
                                                                const u32x4_t* __restrict__ AUX, bf16_t* __restrict__ DK,
                                                                bf16_t* __restrict__ DV, int N, int H, int64_t ld, int64_t ldo,
                                                                int64_t lddq, float c, float scale, float* __restrict__ stamps) {
    uint64_t tm[7] = {0, 0, 0, 0, 0, 0, 0}, t_last = 0;
    auto stamp = [&](int i) {
        if constexpr (STAMPS) {
            __builtin_amdgcn_sched_barrier(0);
            uint64_t now;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            tm[i] += now - t_last;
            t_last = now;
        }
    };
    if constexpr (STAMPS) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
    constexpr int HD = 64, KS = 4, TB = TILE_ROWS * HD * 2, STAGE = DKV64_STAGE, NSL = DKV64_SLOTS;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NSL][STAGE] | 16 zero bytes
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HeadBlock hb = head_block();
    const int bh = hb.bh, b = bh / H, hh = bh % H;
    const int k0 = hb.blk * 256 + wave * 64;
    const bf16_t* Qb = Q + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Kb = K + (int64_t)b * N * ld + hh * HD;
    const bf16_t* Vb = V + (int64_t)b * N * ld + hh * HD;
    const bf16_t* DOb = DO + (int64_t)b * N * ldo + hh * HD;
    const u32x4_t* aux_b = AUX + (int64_t)bh * N;

    // stationary operands as loaded: whole rows, 8 lanes per 128-byte row (piece t of sub-block kb: key 8t + lane/8,
    // 16-byte chunk lane%8); they are re-cut into MFMA fragments -- and land in accumulation registers -- further down,
    // once the first tiles are on their way
    u32x4_t kraw[2][4], vraw[2][4];
    bf16x8_t kf[2][KS], vf[2][KS];               // where the MFMAs take them from (the key fragments carry the softmax scale)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int krow = k0 + kb * 32 + 8 * t + (lane >> 3);
            const int krow_c = krow < N ? krow : N - 1;
            kraw[kb][t] = *(const u32x4_t*)(Kb + (int64_t)krow_c * ld + (lane & 7) * 8);
            vraw[kb][t] = *(const u32x4_t*)(Vb + (int64_t)krow_c * ld + (lane & 7) * 8);
        }
    // A query's record {-lse2 as three bf16 terms + 0 | -delta likewise} is the A operand of the first MFMA of BOTH
    // chains as it stands: the B operand selects the half that counts, 1.0 at k = 0..2 for the scores and at k = 4..6
    // for dP (lane half 0 holds k = 0..7), zero everywhere else.
    const u32x4_t ones_lse_u = {h == 0 ? 0x3f803f80u : 0u, h == 0 ? 0x00003f80u : 0u, 0u, 0u};
    const u32x4_t ones_dl_u = {0u, 0u, h == 0 ? 0x3f803f80u : 0u, h == 0 ? 0x00003f80u : 0u};
    const bf16x8_t ones_lse = __builtin_bit_cast(bf16x8_t, ones_lse_u), ones_dl = __builtin_bit_cast(bf16x8_t, ones_dl_u);
    char* const zero16 = smem + NSL * STAGE;
    if (tid < 4) ((uint32_t*)zero16)[tid] = 0u;       // what lane half 1 reads as its (all-zero) record

    {   // accumulator R = 4 * (key sub-block) + 2 * (0: dV, 1: dK) + (32-column half of the head dim)
        const u32x4_t z = {0u, 0u, 0u, 0u};
        const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, z);
        mf_acc_clear<0>(zero); mf_acc_clear<1>(zero); mf_acc_clear<2>(zero); mf_acc_clear<3>(zero);
        mf_acc_clear<4>(zero); mf_acc_clear<5>(zero); mf_acc_clear<6>(zero); mf_acc_clear<7>(zero);
    }

    const int nt = N / TILE_ROWS;
    TileSrc<HD> ts_q, ts_do;
    ts_q.init(ld, tid);
    ts_do.init(ldo, tid);
    auto stage = [&](int tile, int slot) {
        char* nb = smem + slot * STAGE;
        stage_tile_at<HD, false>(ts_q, Qb, ld, tile * TILE_ROWS, N, nb, tid, wave);
        stage_tile_at<HD, false>(ts_do, DOb, ldo, tile * TILE_ROWS, N, nb + TB, tid, wave);
        buf_to_lds<16>(aux_b + tile * TILE_ROWS, TILE_ROWS * 16, lane * 16, nb + 2 * TB);   // 64 records of 16 bytes (every wave: the same 1 KiB)
    };

    f32x16_t s[2], dp[2];                    // two sets, by key sub-block: one is consumed while the other is produced
    // streamed fragments, two sets each by query-block parity: one is read from LDS while the other feeds MFMAs
    bf16x8_t qfr[2][KS], dofr[2][KS], recf[2];        // row fragments of a query block (+ its records)
    bf16x8_t dotr[2][2][2], qtr[2][2][2];             // transposed fragments of a query block
    bf16x8_t pfr[2][2], dsfr[2][2];                   // packed P and dS, two sets by key sub-block: made in one step, consumed in the next
    // what the in-step loads of a tile need (set at the top of the tile): source row, destination slot
    int64_t ld_row0 = 0;
    char* ld_dst = smem;

#define DKV_PIN asm volatile("" ::: "memory")
#define MF_S0() mfz_vv(s[KBN], recf[QBN], ones_lse)
#define MF_S(ks) mf_aa(s[KBN], qfr[QBN][ks], kf[KBN][ks])
#define MF_P0() mfz_vv(dp[KBN], recf[QBN], ones_dl)
#define MF_P(ks) mf_aa(dp[KBN], dofr[QBN][ks], vf[KBN][ks])
#define MF_DV(sx, cb) mf_acc<4 * KBP + (cb)>(dotr[QBP][sx][cb], pfr[KBP][sx])
#define MF_DK(sx, cb) mf_acc<4 * KBP + 2 + (cb)>(qtr[QBP][sx][cb], dsfr[KBP][sx])
#define EXP(i) p[i] = a_exp2(s[KB][i])
#define MUL(i) d[i] = a_mul(p[i], dp[KB][i])
#define CVP(sx, q) pw[sx][q] = a_cvt_pk(p[8 * (sx) + 2 * (q)], p[8 * (sx) + 2 * (q) + 1])
#define CVD(sx, q) dw[sx][q] = a_cvt_pk(d[8 * (sx) + 2 * (q)], d[8 * (sx) + 2 * (q) + 1])
#define PACKP(sx) pfr[KB][sx] = __builtin_bit_cast(bf16x8_t, (u32x4_t){pw[sx][0], pw[sx][1], pw[sx][2], pw[sx][3]})
#define PACKD(sx) dsfr[KB][sx] = __builtin_bit_cast(bf16x8_t, (u32x4_t){dw[sx][0], dw[sx][1], dw[sx][2], dw[sx][3]})
    // reads of the NEXT query block NQ = QB ^ 1 (of this tile for QB = 0, the next tile's first one for QB = 1) into
    // fragment set NQ: its row fragments in the first key sub-block's step, its transposed fragments in the second's
#define RD_DOT(sx, cb) DKV_PIN; dotr[NQ][sx][cb] = tr_frag<HD>(nsrc + TB, NQ * 32 + 16 * (sx), cb, lane); DKV_PIN
#define RD_QT(sx, cb) DKV_PIN; qtr[NQ][sx][cb] = tr_frag<HD>(nsrc, NQ * 32 + 16 * (sx), cb, lane); DKV_PIN
#define RD_REC() DKV_PIN; recf[NQ] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(h == 0 ? nsrc + 2 * TB + (NQ * 32 + (lane & 31)) * 16 : zero16)); DKV_PIN
#define RD_Q(ks) DKV_PIN; qfr[NQ][ks] = row_frag<HD>(nsrc, NQ * 32, ks, lane); DKV_PIN
#define RD_DO(ks) DKV_PIN; dofr[NQ][ks] = row_frag<HD>(nsrc + TB, NQ * 32, ks, lane); DKV_PIN
    // the global -> LDS loads of the tile three ahead: 2 x Q chunks, 2 x dO chunks, the 64 records; one per step of a tile
    // (LD_STEP), the records with the last.  TILE_SYNC, inside the last step of tile t: this wave's loads of tile t + 2
    // have landed (those of tile t + 3, issued during this tile, may still be in flight), then the barrier: they have
    // landed everywhere, and nobody reads tile t any more (its slot is refilled from the next step on).
#define LD_TILE(i)                                                                                                       \
    DKV_PIN;                                                                                                             \
    if constexpr ((i) < 2) buf_to_lds<16>(Qb + ld_row0 * ld, (int)((TILE_ROWS - 1) * ld * 2 + HD * 2), (int)ts_q.off[(i) & 1], ld_dst + (((i) & 1) * 256 + wave * 64) * 16); \
    else if constexpr ((i) < 4) buf_to_lds<16>(DOb + ld_row0 * ldo, (int)((TILE_ROWS - 1) * ldo * 2 + HD * 2), (int)ts_do.off[(i) & 1], ld_dst + TB + (((i) & 1) * 256 + wave * 64) * 16); \
    else buf_to_lds<16>(aux_b + ld_row0, TILE_ROWS * 16, lane * 16, ld_dst + 2 * TB);     /* every wave: same 1 KiB, no branch in the step */        \
    DKV_PIN
#define LD_STEP()                                   \
    LD_TILE(2 * QB + KB);                           \
    if constexpr (2 * QB + KB == 3) { LD_TILE(4); }
#define TILE_SYNC()                                 \
    wait_vm<5>();                                   \
    asm volatile("" ::: "memory");                  \
    __builtin_amdgcn_s_barrier();                   \
    asm volatile("" ::: "memory")

    // One step: block (QB, KB) of the tile at LDS address `cur` is the current one; `nxt_tile` is the tile after it.
    // The S / dP of the next block are always produced and the later fragments always read -- also at the very end, where
    // they come from a slot that holds an older tile and are never used: the loop body stays one straight line of code
    // without branches.
    auto step = [&](auto qb_c, auto kb_c, const char* cur, const char* nxt_tile) {
        constexpr int QB = decltype(qb_c)::value, KB = decltype(kb_c)::value;
        constexpr int KBN = KB ^ 1;                                          // the next block: key sub-block ...
        constexpr int QBN = KB == 0 ? QB : QB ^ 1;                           // ... and query block (= its fragment set)
        constexpr int KBP = KB ^ 1, QBP = KB == 0 ? QB ^ 1 : QB;             // the previous block (its dV / dK MFMAs run now)
        constexpr int NQ = QB ^ 1;                                           // the query block whose fragments are read
        const char* nsrc = QB == 0 ? cur : nxt_tile;                         // ... and the tile it lies in
        float p[16], d[16];
        uint32_t pw[2][4], dw[2][4];
        (void)nsrc;
#include VA_DKV64_BODY
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    stage(0, 0);
    stage(nt > 1 ? 1 : 0, 1);
    stage(nt > 2 ? 2 : nt - 1, 2);
    // The stationary fragments belong in accumulation registers (the vector half is needed for scores and packed
    // operands), and a value only gets there without a copy in front of every use when the instruction that DEFINES it
    // writes there: an LDS read can, a conversion cannot.  So the rows loaded above go through this wave's 4 KiB of the
    // fourth ring slot (free until the first step's load), one 32-key sub-block of one operand at a time, while the
    // tiles are in flight: written as rows (chunk index XOR row index: no bank conflicts either way), read back as the
    // fragments -- key lane%32, chunk 2*ks + lane/32.  LDS operations of one wave execute in order.
    {
        char* area = smem + 3 * STAGE + wave * 4096;
        const int wr = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);            // + t * 1024
        const int rr = lane & 31;
        auto rd = [&](int ks) { return area + rr * 128 + ((((2 * ks + h) ^ rr) & 7) << 4); };
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *(bf16x8_t*)(area + t * 1024 + wr) = scale_frag(__builtin_bit_cast(bf16x8_t, kraw[kb][t]), c);
            DKV_PIN0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const bf16x8_t*)rd(ks);
            DKV_PIN0;
#pragma unroll
            for (int t = 0; t < 4; ++t) *(u32x4_t*)(area + t * 1024 + wr) = vraw[kb][t];
            DKV_PIN0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vf[kb][ks] = *(const bf16x8_t*)rd(ks);
            DKV_PIN0;
        }
    }
    wait_vm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {   // prologue: what the first step expects in place -- row and transposed fragments of query block 0 of tile 0
        // (set 0) -- and S / dP of block (0, 0) into set 0
        const char* nsrc = smem;
        constexpr int NQ = 0;
        RD_REC(); RD_Q(0); RD_Q(1); RD_Q(2); RD_Q(3); RD_DO(0); RD_DO(1); RD_DO(2); RD_DO(3);
        RD_DOT(0, 0); RD_DOT(0, 1); RD_QT(0, 0); RD_QT(0, 1); RD_DOT(1, 0); RD_DOT(1, 1); RD_QT(1, 0); RD_QT(1, 1);
        constexpr int QBN = 0, KBN = 0;
        MF_S0(); MF_S(0); MF_S(1); MF_S(2); MF_S(3);
        MF_P0(); MF_P(0); MF_P(1); MF_P(2); MF_P(3);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the first vector reads of these results follow closely
        // the first step adds a "previous block" that does not exist: all-zero operands (set 1 of both) make it add zero
        const u32x4_t z = {0u, 0u, 0u, 0u};
        const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, z);
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            pfr[1][sx] = zero; dsfr[1][sx] = zero;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) { dotr[1][sx][cb] = zero; qtr[1][sx][cb] = zero; }
        }
    }
    stamp(5);
    int slot = 0;
    for (int t = 0; t < nt; ++t) {
        const int slot1 = (slot + 1) & (NSL - 1), slot3 = (slot + 3) & (NSL - 1);
        // the steps of this tile load tile t + 3 into the slot tile t - 1 has left; past the end the last tile once more
        // (never read)
        ld_row0 = (int64_t)(t + 3 < nt ? t + 3 : nt - 1) * TILE_ROWS;
        ld_dst = smem + slot3 * STAGE;
        const char* cur = smem + slot * STAGE;
        const char* nxt_tile = smem + slot1 * STAGE;
        stamp(0);
        step(I0{}, I0{}, cur, nxt_tile);
        stamp(1);
        step(I0{}, I1{}, cur, nxt_tile);
        stamp(2);
        step(I1{}, I0{}, cur, nxt_tile);
        stamp(3);
        step(I1{}, I1{}, cur, nxt_tile);
        stamp(4);
        slot = slot1;
    }
    {   // the last block's dV / dK (its packed operands were made in the last step)
        constexpr int KBP = 1, QBP = 1;
        MF_DV(0, 0); MF_DV(0, 1); MF_DK(0, 0); MF_DK(0, 1); MF_DV(1, 0); MF_DV(1, 1); MF_DK(1, 0); MF_DK(1, 1);
    }
    wait_vm0();        // the trailing loads must not land in LDS after the workgroup has gone
#undef DKV_PIN
#undef MF_S0
#undef MF_S
#undef MF_P0
#undef MF_P
#undef MF_DV
#undef MF_DK
#undef EXP
#undef MUL
#undef CVP
#undef CVD
#undef PACKP
#undef PACKD
#undef RD_DOT
#undef RD_QT
#undef RD_REC
#undef RD_Q
#undef RD_DO
#undef LD_TILE
#undef LD_STEP
#undef TILE_SYNC
    // the asm MFMAs are opaque to the compiler's hazard tracking: let the last results retire before they are read
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_s_barrier();      // every wave is done with the tiles: the LDS becomes the waves' transposition areas
    asm volatile("" ::: "memory");
    char* area = smem + wave * (32 * (2 * HD + 16));
    auto store_sub = [&](auto kb_c) {
        constexpr int kb = decltype(kb_c)::value;
        const f32x16_t dvt[2] = {acc_tile<4 * kb + 0>(), acc_tile<4 * kb + 1>()};
        const f32x16_t dkt[2] = {acc_tile<4 * kb + 2>(), acc_tile<4 * kb + 3>()};
        // N is a whole number of 64-query tiles, not of this workgroup's 256 keys: a sub-block may lie (wholly) past N
        const int key0 = k0 + kb * 32, krow = key0 + (lane & 31);
        if (key0 + 32 <= N) {
            const int64_t row0 = (int64_t)b * N + key0;
            store_rows_lds<HD>(dkt, scale, DK + row0 * lddq + hh * HD, lddq, area, lane);
            store_rows_lds<HD>(dvt, 1.0f, DV + row0 * lddq + hh * HD, lddq, area, lane);
        } else if (krow < N) {
            store_rows<HD>(dkt, scale, DK + ((int64_t)b * N + krow) * lddq + hh * HD, h);
            store_rows<HD>(dvt, 1.0f, DV + ((int64_t)b * N + krow) * lddq + hh * HD, h);
        }
    };
    store_sub(std::integral_constant<int, 0>{});
    store_sub(std::integral_constant<int, 1>{});
    if constexpr (STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(6);
        if (lane == 0) {
            float* o = stamps + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 7; ++i) o[i] = (float)tm[i];
            o[7] = (float)nt;
        }
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
    VA_CHECK_ARG(hd == 32 || hd == 64 || hd == 128, "%s: head dim %lld not supported (32, 64 or 128)", who, (long long)hd);
    VA_CHECK_ARG(ldqkv >= H * hd && ldo >= H * hd, "%s: leading dimension smaller than H*hd", who);
    VA_CHECK_ARG(ldqkv % 8 == 0 && ldo % 8 == 0, "%s: leading dimensions must keep 16-byte alignment", who);
    VA_CHECK_ARG(B * H < 65536 && N < (1 << 24), "%s: B*H or N too large for the launch grid", who);
    // a 64-row tile is addressed through one buffer descriptor with 32-bit offsets
    VA_CHECK_ARG(ldqkv < (1 << 23) && ldo < (1 << 23), "%s: leading dimension too large", who);
    return VA_OK;
}

template <int HD, int MODE, bool DROP>
int fwd_launch_mode(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, float* lse2, int B, int N, int H,
                    int64_t ld, int64_t ldo, int tpt, DropArgs drop, hipStream_t stream, MaskArgs mask = MaskArgs{}) {
    constexpr int LDS = fwd_ring_slots(HD) * 2 * TILE_ROWS * HD * 2;
    static bool flags[64] = {};
    const float c = 1.4426950408889634f / sqrtf((float)HD);
    const dim3 grid((N + 127) / 128, B * H);
    if (int e = set_lds(attn_fwd_kernel<HD, MODE, DROP>, LDS, flags, "va_attn_fwd")) return e;
    hipLaunchKernelGGL((attn_fwd_kernel<HD, MODE, DROP>), grid, dim3(256), LDS, stream, q, k, v, o, lse2, N, H, ld, ldo,
                       c, tpt, drop, mask);
    VA_CHECK_LAUNCH("va_attn_fwd");
    return VA_OK;
}

template <int HD>
int fwd_launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, float* lse2, int B, int N, int H,
               int64_t ld, int64_t ldo, int tpt, DropArgs drop, hipStream_t stream, MaskArgs mask = MaskArgs{}) {
    if (mask.by_q) return fwd_launch_mode<HD, 3, false>(q, k, v, o, lse2, B, N, H, ld, ldo, 0, drop, stream, mask);
    if (drop.thresh16) {   // training-mode dropout: the encoder's attention only (no key limit other than N)
        if (N % TILE_ROWS) return fwd_launch_mode<HD, 1, true>(q, k, v, o, lse2, B, N, H, ld, ldo, 0, drop, stream);
        return fwd_launch_mode<HD, 0, true>(q, k, v, o, lse2, B, N, H, ld, ldo, 0, drop, stream);
    }
    if (tpt > 0) return fwd_launch_mode<HD, 2, false>(q, k, v, o, lse2, B, N, H, ld, ldo, tpt, drop, stream);
    if (N % TILE_ROWS) return fwd_launch_mode<HD, 1, false>(q, k, v, o, lse2, B, N, H, ld, ldo, 0, drop, stream);
    return fwd_launch_mode<HD, 0, false>(q, k, v, o, lse2, B, N, H, ld, ldo, 0, drop, stream);
}

// VA_ATTN_DKV_GENERIC=1 in the environment keeps every shape on the generic dK/dV kernel (A/B timing, cross-checks).
inline bool dkv_generic_only() {
    static const bool v = [] { const char* e = getenv("VA_ATTN_DKV_GENERIC"); return e && e[0] == '1'; }();
    return v;
}

template <int HD, bool RAGGED, bool DROP, bool BITS = false>
int bwd_launch_as(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* o, const bf16_t* d_o,
                  const float* lse2, float* workspace, bf16_t* dq, bf16_t* dk, bf16_t* dv, int B, int N, int H,
                  int64_t ld, int64_t ldo, int64_t lddq, DropArgs drop, hipStream_t stream, MaskArgs mask = MaskArgs{}) {
    constexpr int LDS_DQ = 2 * 2 * TILE_ROWS * HD * 2;
    constexpr int LDS_DKV = 2 * dkv_stage_bytes<HD, DROP>() + 16;
    static bool f1[64] = {}, f2[64] = {};
    if (int e = set_lds(attn_bwd_dq_kernel<HD, RAGGED, DROP, BITS>, LDS_DQ, f1, "va_attn_bwd")) return e;
    // With the mask code the whole-tile build of the dK/dV pass spills (its precomputed tile addresses stay live
    // across the hash temporaries); the build that clamps rows per tile does not and is correct for every N.
    constexpr bool RAGGED_DKV = RAGGED || DROP;
    if (int e = set_lds(attn_bwd_dkv_kernel<HD, RAGGED_DKV, DROP, BITS>, LDS_DKV, f2, "va_attn_bwd")) return e;
    const float scale = 1.0f / sqrtf((float)HD);
    const float c = 1.4426950408889634f * scale;
    const int64_t rows = (int64_t)B * N;
    // workspace: [B*H*N] 16-byte records for the dK/dV pass, then [B*H*N] f32 -delta; both written by the dQ pass
    u32x4_t* aux = (u32x4_t*)workspace;
    float* delta = workspace + 4 * rows * H;
    const dim3 grid((N + 127) / 128, B * H);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, RAGGED, DROP, BITS>), grid, dim3(256), LDS_DQ, stream, q, k, v, o, d_o, lse2,
                       delta, aux, dq, N, H, ld, ldo, lddq, c, scale, DROP ? 1.0f / drop.scale : 1.0f, drop, mask);
    VA_CHECK_LAUNCH("va_attn_bwd(dq)");
    if (HD == 64 && !RAGGED && !DROP && !BITS && !dkv_generic_only()) {
        constexpr int LDS_P = DKV64_SLOTS * DKV64_STAGE + 16;
        static bool f3[64] = {};
        if (int e = set_lds(attn_bwd_dkv64_kernel<false>, LDS_P, f3, "va_attn_bwd")) return e;
        const dim3 grid_p((N + 255) / 256, B * H);
        hipLaunchKernelGGL(attn_bwd_dkv64_kernel<false>, grid_p, dim3(256), LDS_P, stream, q, k, v, d_o, aux, dk, dv, N, H, ld, ldo,
                           lddq, c, scale, (float*)nullptr);
        VA_CHECK_LAUNCH("va_attn_bwd(dkv64)");
        return VA_OK;
    }
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, RAGGED_DKV, DROP, BITS>), grid, dim3(256), LDS_DKV, stream, q, k, v, d_o, aux, delta,
                       dk, dv, N, H, ld, ldo, lddq, c, scale, drop, mask);
    VA_CHECK_LAUNCH("va_attn_bwd(dkv)");
    return VA_OK;
}

template <int HD>
int bwd_launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* o, const bf16_t* d_o,
               const float* lse2, float* workspace, bf16_t* dq, bf16_t* dk, bf16_t* dv, int B, int N, int H, int64_t ld,
               int64_t ldo, int64_t lddq, DropArgs drop, hipStream_t stream, MaskArgs mask = MaskArgs{}) {
    if (mask.by_q)
        return bwd_launch_as<HD, true, false, true>(q, k, v, o, d_o, lse2, workspace, dq, dk, dv, B, N, H, ld, ldo, lddq, drop,
                                                    stream, mask);
    const bool ragged = (N % TILE_ROWS) != 0;
#define VA_BWD(R, D) bwd_launch_as<HD, R, D>(q, k, v, o, d_o, lse2, workspace, dq, dk, dv, B, N, H, ld, ldo, lddq, drop, stream)
    if (drop.thresh16) return ragged ? VA_BWD(true, true) : VA_BWD(false, true);
    return ragged ? VA_BWD(true, false) : VA_BWD(false, false);
#undef VA_BWD
}

}  // namespace

// p -> kernel arguments; p == 0 selects the kernels without any mask code
static int make_drop(const char* who, uint32_t key, float p, DropArgs* out) {
    VA_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability %g outside [0, 1)", who, (double)p);
    out->key = key;
    out->thresh16 = va_drop_thresh16(p);
    VA_CHECK_ARG(out->thresh16 < 65536u, "%s: dropout probability %g rounds to 1", who, (double)p);
    out->scale = va_drop_scale(out->thresh16);
    return VA_OK;
}

static int attn_fwd_entry(const char* who, const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B,
                          int64_t N, int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, int64_t tpt, DropArgs drop,
                          hipStream_t stream, MaskArgs mask = MaskArgs{}) {
    VA_CHECK_ARG(q && k && v && o && lse2, "%s: null operand", who);
    if (int e = check_common(who, B, N, H, hd, ldqkv, ldo)) return e;
    VA_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0, "%s: misaligned operand", who);
    if (hd == 32)
        return fwd_launch<32>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2, (int)B, (int)N,
                              (int)H, ldqkv, ldo, (int)tpt, drop, stream, mask);
    if (hd == 64)
        return fwd_launch<64>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2, (int)B, (int)N,
                              (int)H, ldqkv, ldo, (int)tpt, drop, stream, mask);
    return fwd_launch<128>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2, (int)B, (int)N, (int)H,
                           ldqkv, ldo, (int)tpt, drop, stream, mask);
}

extern "C" int va_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B, int64_t N,
                           int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, hipStream_t stream) {
    return attn_fwd_entry("va_attn_fwd", q, k, v, o, lse2, B, N, H, hd, ldqkv, ldo, 0, DropArgs{0u, 0u, 1.f}, stream);
}

extern "C" int va_attn_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B,
                                   int64_t N, int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, uint32_t key, float p,
                                   hipStream_t stream) {
    DropArgs drop;
    if (int e = make_drop("va_attn_fwd_dropout", key, p, &drop)) return e;
    return attn_fwd_entry("va_attn_fwd_dropout", q, k, v, o, lse2, B, N, H, hd, ldqkv, ldo, 0, drop, stream);
}

extern "C" int va_attn_fwd_block_causal(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B,
                                        int64_t N, int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo,
                                        int64_t tokens_per_timestep, hipStream_t stream) {
    VA_CHECK_ARG(tokens_per_timestep > 0 && N % tokens_per_timestep == 0,
                 "va_attn_fwd_block_causal: N=%lld must be a whole number of timesteps of %lld tokens", (long long)N,
                 (long long)tokens_per_timestep);
    return attn_fwd_entry("va_attn_fwd_block_causal", q, k, v, o, lse2, B, N, H, hd, ldqkv, ldo, tokens_per_timestep,
                          DropArgs{0u, 0u, 1.f}, stream);
}

extern "C" int64_t va_attn_bwd_workspace_bytes(int64_t B, int64_t N, int64_t H) {
    return (B > 0 && N > 0 && H > 0) ? 5 * B * H * N * (int64_t)sizeof(float) : 0;
}

static int attn_bwd_entry(const char* who, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                          const float* lse2, float* workspace, void* dq, void* dk, void* dv, int64_t B, int64_t N,
                          int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, int64_t lddqkv, DropArgs drop,
                          hipStream_t stream, MaskArgs mask = MaskArgs{}) {
    VA_CHECK_ARG(q && k && v && o && d_o && lse2 && workspace && dq && dk && dv, "%s: null operand", who);
    VA_CHECK_ARG((uintptr_t)workspace % 16 == 0, "%s: workspace must be 16-byte aligned", who);
    if (int e = check_common(who, B, N, H, hd, ldqkv, ldo)) return e;
    VA_CHECK_ARG(lddqkv >= H * hd && lddqkv % 8 == 0, "%s: bad gradient leading dimension", who);
    VA_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq |
                  (uintptr_t)dk | (uintptr_t)dv) % 16 == 0,
                 "%s: misaligned operand", who);
    if (hd == 32)
        return bwd_launch<32>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o,
                              lse2, workspace, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, (int)B, (int)N, (int)H, ldqkv, ldo,
                              lddqkv, drop, stream, mask);
    if (hd == 64)
        return bwd_launch<64>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o,
                              lse2, workspace, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, (int)B, (int)N, (int)H, ldqkv, ldo,
                              lddqkv, drop, stream, mask);
    return bwd_launch<128>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o,
                           lse2, workspace, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, (int)B, (int)N, (int)H, ldqkv, ldo, lddqkv,
                           drop, stream, mask);
}

extern "C" int va_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                           const float* lse2, float* workspace, void* dq, void* dk, void* dv, int64_t B, int64_t N,
                           int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, int64_t lddqkv, hipStream_t stream) {
    return attn_bwd_entry("va_attn_bwd", q, k, v, o, d_o, lse2, workspace, dq, dk, dv, B, N, H, hd, ldqkv, ldo, lddqkv,
                          DropArgs{0u, 0u, 1.f}, stream);
}

extern "C" int va_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                   const float* lse2, float* workspace, void* dq, void* dk, void* dv, int64_t B,
                                   int64_t N, int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, int64_t lddqkv,
                                   uint32_t key, float p, hipStream_t stream) {
    DropArgs drop;
    if (int e = make_drop("va_attn_bwd_dropout", key, p, &drop)) return e;
    return attn_bwd_entry("va_attn_bwd_dropout", q, k, v, o, d_o, lse2, workspace, dq, dk, dv, B, N, H, hd, ldqkv, ldo,
                          lddqkv, drop, stream);
}

// ------------------------------------------------------------------------------------------------
// Explicit boolean masks.  One thread per output word: `by_q` word (r, w) collects mask[r][32 w .. 32 w + 31] (32
// contiguous bytes), `by_k` word (c, w) collects mask[32 w .. 32 w + 31][c] (a column: neighbouring threads read
// neighbouring bytes).  Positions past N read as masked.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_mask_pack_kernel(const uint8_t* __restrict__ mask, int64_t batches, int N, int W,
                                                             uint32_t* __restrict__ by_q, uint32_t* __restrict__ by_k) {
    const int64_t per = (int64_t)N * W;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= batches * per) return;
    const int64_t bi = id / per;
    const int64_t rem = id - bi * per;
    const uint8_t* mb = mask + bi * (int64_t)N * N;
    {   // rem = r * W + w
        const int r = (int)(rem / W), w = (int)(rem - (int64_t)r * W);
        uint32_t bits = 0;
        for (int j = 0; j < 32; ++j) {
            const int c = 32 * w + j;
            bits |= (c >= N || mb[(int64_t)r * N + c] != 0) ? (1u << j) : 0u;
        }
        by_q[id] = bits;
    }
    {   // rem = w * N + c (column fastest across threads); stored at (c, w)
        const int w = (int)(rem / N), c = (int)(rem - (int64_t)w * N);
        uint32_t bits = 0;
        for (int j = 0; j < 32; ++j) {
            const int r = 32 * w + j;
            bits |= (r >= N || mb[(int64_t)r * N + c] != 0) ? (1u << j) : 0u;
        }
        by_k[bi * per + (int64_t)c * W + w] = bits;
    }
}

extern "C" int64_t va_attn_mask_words(int64_t N) { return N > 0 ? N * 2 * ((N + TILE_ROWS - 1) / TILE_ROWS) : 0; }

extern "C" int va_attn_mask_pack(const uint8_t* mask, int64_t batches, int64_t N, uint32_t* by_query, uint32_t* by_key,
                                 hipStream_t stream) {
    VA_CHECK_ARG(mask && by_query && by_key, "va_attn_mask_pack: null operand");
    VA_CHECK_ARG(batches > 0 && N > 0 && N < (1 << 24), "va_attn_mask_pack: bad shape");
    VA_CHECK_ARG(((uintptr_t)by_query | (uintptr_t)by_key) % 8 == 0, "va_attn_mask_pack: outputs must be 8-byte aligned");
    const int W = 2 * (int)((N + TILE_ROWS - 1) / TILE_ROWS);
    const int64_t total = batches * N * W;
    VA_CHECK_ARG((total + 255) / 256 < (1ll << 31), "va_attn_mask_pack: too many mask words");
    hipLaunchKernelGGL(attn_mask_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mask, batches, (int)N, W,
                       by_query, by_key);
    VA_CHECK_LAUNCH("va_attn_mask_pack");
    return VA_OK;
}

static int make_mask(const char* who, const uint32_t* by_query, const uint32_t* by_key, bool need_by_key, int64_t stride_b,
                     int64_t stride_h, MaskArgs* out) {
    VA_CHECK_ARG(by_query && (by_key || !need_by_key), "%s: null mask", who);
    VA_CHECK_ARG(stride_b >= 0 && stride_h >= 0 && stride_b % 2 == 0 && stride_h % 2 == 0,
                 "%s: mask strides must be even and non-negative", who);
    VA_CHECK_ARG(((uintptr_t)by_query | (uintptr_t)by_key) % 8 == 0, "%s: mask words must be 8-byte aligned", who);
    *out = MaskArgs{by_query, by_key, stride_b, stride_h};
    return VA_OK;
}

extern "C" int va_attn_fwd_masked(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B, int64_t N,
                                  int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, const uint32_t* mask_by_query,
                                  int64_t stride_b, int64_t stride_h, hipStream_t stream) {
    MaskArgs mask;
    if (int e = make_mask("va_attn_fwd_masked", mask_by_query, nullptr, false, stride_b, stride_h, &mask)) return e;
    return attn_fwd_entry("va_attn_fwd_masked", q, k, v, o, lse2, B, N, H, hd, ldqkv, ldo, 0, DropArgs{0u, 0u, 1.f}, stream,
                          mask);
}

extern "C" int va_attn_bwd_masked(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                  const float* lse2, float* workspace, void* dq, void* dk, void* dv, int64_t B, int64_t N,
                                  int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, int64_t lddqkv,
                                  const uint32_t* mask_by_query, const uint32_t* mask_by_key, int64_t stride_b,
                                  int64_t stride_h, hipStream_t stream) {
    MaskArgs mask;
    if (int e = make_mask("va_attn_bwd_masked", mask_by_query, mask_by_key, true, stride_b, stride_h, &mask)) return e;
    return attn_bwd_entry("va_attn_bwd_masked", q, k, v, o, d_o, lse2, workspace, dq, dk, dv, B, N, H, hd, ldqkv, ldo, lddqkv,
                          DropArgs{0u, 0u, 1.f}, stream, mask);
}
