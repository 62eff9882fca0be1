// Weight-gradient GEMM:  dW[N,K] (f32) += dY[M,N]^T * X[M,K],  bf16 operands, contraction over the
// token dimension M (tens of thousands), which is the SLOW index of both operands in memory.
//
// Common structure (gfx950):
//   * both operand tiles are staged exactly as they lie in memory ([rows of m][columns]) with 16-byte
//     direct-to-LDS loads; the MFMA fragments (8 consecutive m for one column) are produced by the hardware
//     transposing LDS read ds_read_b64_tr_b16, two per fragment.  The 16-byte chunk index is XORed with
//     ((row&3)<<2) on the load's source address and on the read, which spreads the four rows of one
//     transposed read over all 64 banks.
//   * M is split into slices; every slice adds its tile into dW with f32 atomics.  dY^T is the MFMA A
//     operand, so one accumulator register across a wave is two 128-byte row segments of dW, the shape at
//     which the memory-side atomic units run at full rate.
//   * a ragged last M-tile is zero-filled in LDS after it lands (sources are clamped to row M-1).
//   * the bias gradient (column sums of dY) is optionally fused: MFMAs against an all-ones operand reuse
//     the dY^T fragments that are in registers anyway, instead of a second pass over dY.
//
// Two kernels:
//   gemm_tn_ring_kernel  256(n) x 256(k) tile, 8 waves, 32 rows of M per stage through a 5-slot LDS ring with
//                        counted waits (as the forward GEMM's ring), fragments double-buffered in registers,
//                        slices and tiles laid out over XCD-aware work ids; the default for large problems.
//   gemm_tn_kernel       128 x 128 tile, 4 waves (2x2 of 64x64, MFMA 32x32x16), 64 rows of M per step through two
//                        buffers; small shapes, and shapes the ring kernel's constraints exclude.
#include "common.h"
#include <type_traits>

namespace {

constexpr int TBM = 64;    // contraction rows per step
constexpr int TBN = 128;   // tile edge (both n and k)
constexpr int TILE_BYTES = TBM * TBN * 2;     // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ DY, const bf16_t* __restrict__ X,
                                                      int M, int N, int K, int lddy, int ldx, float* __restrict__ DW,
                                                      int lddw, int m_per_slice, float* __restrict__ DBIAS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;

    const int tiles_k = K / TBN;
    const int tiles = (N / TBN) * tiles_k;
    const int id = xcd_remap(blockIdx.x, tiles);
    const int n0 = (id / tiles_k) * TBN;
    const int k0 = (id % tiles_k) * TBN;
    const int m_begin = blockIdx.y * m_per_slice;
    int m_end = m_begin + m_per_slice;
    m_end = m_end < M ? m_end : M;
    if (m_begin >= m_end) return;   // uniform for the whole workgroup
    const int nsteps = (m_end - m_begin + TBM - 1) / TBM;

    // staging: 1024 chunks of 16 B per operand tile, 4 per thread.  chunk q -> row q>>4, slot q&15.
    int srow[4], scol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 256 + tid;
        srow[i] = q >> 4;
        scol[i] = (((q & 15) ^ ((srow[i] & 3) << 2))) * 8;
    }
    auto stage = [&](int buf, int step) {
        char* yb = smem + buf * STAGE_BYTES;
        char* xb = yb + TILE_BYTES;
        const int mb = m_begin + step * TBM;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = mb + srow[i];
            m = m < M ? m : M - 1;
            glds16(DY + (size_t)m * lddy + n0 + scol[i], yb + (i * 256 + wave * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = mb + srow[i];
            m = m < M ? m : M - 1;
            glds16(X + (size_t)m * ldx + k0 + scol[i], xb + (i * 256 + wave * 64) * 16);
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // Optional fused bias gradient (column sums of dY): the workgroups of k-tile 0 multiply their dY^T
    // fragments with an all-ones B operand as well; the two k-waves of a row share the k-steps.
    const bool do_bias = (DBIAS != nullptr) && (id % tiles_k == 0);
    f32x16_t bacc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[a][i] = 0.f;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    // transposed-read lane roles: 16-lane group gi = lane>>4: column half cg = gi&1, k half h = gi>>1;
    // inside the group lane 4q+p addresses row q, columns 4p..4p+3.
    const int gi = lane >> 4, cg = gi & 1, h = gi >> 1, q = (lane & 15) >> 2, p = lane & 3;
    // byte offset inside a tile for k-step 0, column tile 0 of this wave; row r0 = 8h + q
    auto tr_off = [&](int colbase) {
        const int row = 8 * h + q;
        const int chunk = ((colbase + 16 * cg + 4 * p) >> 3) ^ ((row & 3) << 2);
        return row * 256 + (chunk << 4) + (p & 1) * 8;
    };
    int yoff[2], xoff[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        yoff[a] = tr_off(wn * 64 + a * 32);
        xoff[a] = tr_off(wk * 64 + a * 32);
    }

    auto compute = [&](int buf) {
        const char* yb = smem + buf * STAGE_BYTES;
        const char* xb = yb + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {   // 16 rows of m per MFMA k-step
            bf16x8_t af[2], bfr[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const char* pa = yb + yoff[a] + ks * 16 * 256;
                af[a] = join_tr(lds_read_tr16(pa), lds_read_tr16(pa + 4 * 256));
                const char* pb = xb + xoff[a] + ks * 16 * 256;
                bfr[a] = join_tr(lds_read_tr16(pb), lds_read_tr16(pb + 4 * 256));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(af[a], bfr[b], acc[a][b]);
            if (do_bias && (ks & 1) == wk) {
#pragma unroll
                for (int a = 0; a < 2; ++a) bacc[a] = mfma32(af[a], ones, bacc[a]);
            }
        }
    };

    // zero rows [valid, 64) of both operand tiles of `buf` (ragged tail of M)
    auto zero_tail = [&](int buf, int valid) {
        char* base = smem + buf * STAGE_BYTES;
        for (int c = tid; c < 2 * TBM * 16; c += 256) {
            const int t = c / (TBM * 16), r = (c / 16) % TBM;
            if (r >= valid) *(u32x4_t*)(base + t * TILE_BYTES + (c % (TBM * 16)) * 16) = (u32x4_t){0u, 0u, 0u, 0u};
        }
    };
    const int tail_valid = (m_end - m_begin) - (nsteps - 1) * TBM;   // rows valid in the last step (1..64)

    stage(0, 0);
    wait_vm0();
    __syncthreads();
    if (nsteps == 1 && tail_valid < TBM) {
        zero_tail(0, tail_valid);
        __syncthreads();
    }
    for (int s = 0; s < nsteps - 1; ++s) {
        stage((s + 1) & 1, s + 1);
        compute(s & 1);
        wait_vm0();
        __syncthreads();
        if (s + 1 == nsteps - 1 && tail_valid < TBM) {
            zero_tail((s + 1) & 1, tail_valid);
            __syncthreads();
        }
    }
    compute((nsteps - 1) & 1);

    // D[row = n][col = k]: lane column = lane&31, register i -> row acc32_row(i, lane>>5)
    const int hh = lane >> 5, col = lane & 31;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float* base = DW + (size_t)(n0 + wn * 64 + a * 32) * lddw + k0 + wk * 64 + b * 32 + col;
#pragma unroll
            for (int i = 0; i < 16; ++i) unsafeAtomicAdd(base + (size_t)acc32_row(i, hh) * lddw, acc[a][b][i]);
        }
    if (do_bias && col == 0) {   // every column of bacc holds the same sums: lane column 0 publishes them
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) unsafeAtomicAdd(DBIAS + n0 + wn * 64 + a * 32 + acc32_row(i, hh), bacc[a][i]);
    }
}

// ------------------------------------------------------------------------------------------------
// Large-problem variant: 256(n) x 256(k) tile, 8 waves (2 x 4, each 128 x 64 as 8 x 4 MFMA 16x16x32 tiles: under
// the chip's clock management this shape sustains about 1.12 x the rate of 32x32x16 at equal cycles per FLOP),
// contraction in 32-row stages (one MFMA k-step) through a 5-slot LDS ring (all 160 KiB) with counted vmcnt
// waits, and MFMA fragments read from LDS one stage ahead of their use (see gemm_nt.hip, ring loop, for the
// protocol).  Rows of a staged tile are 512 B (32 chunks); the chunk index is XORed with ((row&3)<<2) and with
// ((row>>3)&1)<<1, which keeps the 16 rows a transposed read touches (4 per 16-lane group) on distinct banks.
// One workgroup per CU, so the grid is sized to about one workgroup per CU by the M split.
// ------------------------------------------------------------------------------------------------
constexpr int RBM = 32;                       // contraction rows per stage
constexpr int ROP_BYTES = RBM * 256 * 2;      // 16 KiB per operand per stage
constexpr int RSTAGE_BYTES = 2 * ROP_BYTES;
constexpr int RRING = 5;

template <bool DO_BIAS>
VA_DEVINL void gemm_tn_ring_body(const bf16_t* __restrict__ DY, const bf16_t* __restrict__ X,
                                  int M, int N, int K, int lddy, int ldx, float* __restrict__ DW, int lddw,
                                  int m_per_slice, float* __restrict__ DBIAS, int work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;

    // Work ids are slice-major, and the workgroups that share an XCD take one contiguous run of them: the
    // ~32 workgroups resident on an XCD then sweep the SAME rows of dY and X (one M slice) for a compact
    // block of output tiles, so each staged row segment is fetched from HBM once and reused out of that
    // XCD's L2 by every tile that needs it.
    const int tiles_k = K / 256;
    const int tiles = (N / 256) * tiles_k;
    const int id = work % tiles;
    int nt_, kt_;
    grouped_tile(id, N / 256, tiles_k, nt_, kt_);     // compact blocks of tiles per XCD run, as in the NT kernels
    const int n0 = nt_ * 256;
    const int k0 = kt_ * 256;
    const int m_begin = (work / tiles) * m_per_slice;
    int m_end = m_begin + m_per_slice;
    m_end = m_end < M ? m_end : M;
    const int nk = (m_end - m_begin + RBM - 1) / RBM;            // >= 4 (host guarantees)
    const int tail_valid = (m_end - m_begin) - (nk - 1) * RBM;   // rows valid in the last stage (1..32)

    // staging: 1024 chunks per operand per stage, 2 per thread: chunk q -> row q>>5, slot q&31
    int srow[2], scol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        srow[i] = q >> 5;
        scol[i] = ((q & 31) ^ ((srow[i] & 3) << 2) ^ (((srow[i] >> 3) & 1) << 1)) * 8;
    }
    auto stage = [&](int kt, int slot) {
        char* yb = smem + slot * RSTAGE_BYTES;
        char* xb = yb + ROP_BYTES;
        const int mb = m_begin + kt * RBM;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = mb + srow[i];
            m = m < M ? m : M - 1;
            glds16(DY + (size_t)m * lddy + n0 + scol[i], yb + (i * 512 + wave * 64) * 16);
            glds16(X + (size_t)m * ldx + k0 + scol[i], xb + (i * 512 + wave * 64) * 16);
        }
    };

    // D[n][k]: dY^T is the A operand, X the B operand: lane holds rows n = 4*(lane>>4) + j, column k = lane & 15
    f32x4_t acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    constexpr bool do_bias = DO_BIAS;
    f32x4_t bacc[2];                // wave (wn, wk) sums the columns of n-blocks a = 2*wk, 2*wk + 1
    bacc[0] = bacc[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    // A 16x16x32 operand fragment: lane l holds 8 consecutive m (k-group l>>4) of column l & 15.  Its 16-lane group
    // reads rows 8*(l>>4) .. +3 (second read: +4) of 16 columns; lane 4q+p of the group addresses row q, columns
    // 4p .. 4p+3.
    const int gi = lane >> 4, q4 = (lane & 15) >> 2, p = lane & 3;
    auto tr_off = [&](int colbase) {
        const int row = 8 * gi + q4;
        const int chunk = ((colbase + 4 * p) >> 3) ^ ((row & 3) << 2) ^ (((row >> 3) & 1) << 1);
        return row * 512 + (chunk << 4) + (p & 1) * 8;
    };
    int yoff[8], xoff[4];
#pragma unroll
    for (int a = 0; a < 8; ++a) yoff[a] = tr_off(wn * 128 + a * 16);
#pragma unroll
    for (int b = 0; b < 4; ++b) xoff[b] = ROP_BYTES + tr_off(wk * 64 + b * 16);

    // fragment of the stage in `slot`
    auto frag = [&](int slot, int off) {
        const char* ptr = smem + slot * RSTAGE_BYTES + off;
        return join_tr(lds_read_tr16(ptr), lds_read_tr16(ptr + 4 * 512));
    };
    bf16x8_t fa[8], fb[4], fbn[4];
    int ymine[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) ymine[jj] = tr_off(wn * 128 + (2 * wk + jj) * 16);

    // zero rows [valid, 32) of both operand tiles of a slot (ragged end of M)
    auto zero_tail = [&](int slot, int valid) {
        char* base = smem + slot * RSTAGE_BYTES;
        for (int c = tid; c < 2 * RBM * 32; c += 512) {
            const int r = (c >> 5) & (RBM - 1);
            if (r >= valid) *(u32x4_t*)(base + c * 16) = (u32x4_t){0u, 0u, 0u, 0u};
        }
    };

    // one stage: 32 MFMAs; after the four MFMAs of n-block a the next stage's dY fragment a replaces the retired
    // one and (a < 4) X fragment a of the next stage is read into the spare set
    auto kstep = [&](auto prefetch, int cur_slot, int nslot, bool bias_now) {
        constexpr bool PREFETCH = decltype(prefetch)::value;
        if (DO_BIAS && bias_now) {
            // uniform branch: this workgroup's share of the M stages.  The wave's two n-blocks are a run-time choice
            // (by wave index), so their fragments are read again from the stage's slot rather than selected out of
            // the register set with 24 conditional moves.
            const bf16x8_t m0 = frag(cur_slot, ymine[0]), m1 = frag(cur_slot, ymine[1]);
            bacc[0] = mfma16(m0, ones, bacc[0]);
            bacc[1] = mfma16(m1, ones, bacc[1]);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
            if (PREFETCH) {
                fa[a] = frag(nslot, yoff[a]);
                if (a < 4) fbn[a] = frag(nslot, xoff[a]);
            }
        }
        if (PREFETCH) {
#define VA_GRP(n_reads)                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);       \
    __builtin_amdgcn_sched_group_barrier(0x100, n_reads, 0);
            VA_GRP(4) VA_GRP(4) VA_GRP(4) VA_GRP(4) VA_GRP(2) VA_GRP(2) VA_GRP(2) VA_GRP(2)
#undef VA_GRP
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = fbn[b];
        }
    };
    // the bias column sums are spread over the k-tile workgroups of an output row: each takes every
    // tiles_k-th stage, so no workgroup carries more than 1/tiles_k of the extra MFMAs
    const int my_ktile = kt_;
    auto bias_stage = [&](int stage_idx) { return DO_BIAS && (stage_idx % tiles_k) == my_ktile; };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    auto advance = [&](int sl) { return sl + 1 == RRING ? 0 : sl + 1; };

    stage(0, 0);
    stage(1, 1);
    stage(2, 2);
    stage(3, 3);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int a = 0; a < 8; ++a) fa[a] = frag(0, yoff[a]);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = frag(0, xoff[b]);

    int slot = 0;
    int kt = 0;
    for (; kt + 4 < nk; ++kt) {           // stages with three successors in flight
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(kt + 4, slot == 0 ? RRING - 1 : slot - 1);
        kstep(Yes{}, slot, advance(slot), bias_stage(kt));
        slot = advance(slot);
    }
    // tail: stages nk-4 .. nk-1
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    kstep(Yes{}, slot, advance(slot), bias_stage(kt));
    slot = advance(slot); ++kt;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    kstep(Yes{}, slot, advance(slot), bias_stage(kt));
    slot = advance(slot); ++kt;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tail_valid < RBM) {               // uniform: the last stage has landed everywhere, blank its unused rows
        zero_tail(advance(slot), tail_valid);
        __syncthreads();
    }
    kstep(Yes{}, slot, advance(slot), bias_stage(kt));
    slot = advance(slot); ++kt;
    kstep(No{}, slot, 0, bias_stage(kt));

    // one accumulator register across the wave: four 64-byte row segments of dW
    const int rg = lane >> 4, col = lane & 15;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float* base = DW + (size_t)(n0 + wn * 128 + a * 16 + 4 * rg) * lddw + k0 + wk * 64 + b * 16 + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(base + (size_t)j * lddw, acc[a][b][j]);
        }
    if (do_bias && col == 0) {   // every column of bacc holds the same sums: lane column 0 publishes them
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                unsafeAtomicAdd(DBIAS + n0 + wn * 128 + (2 * wk + jj) * 16 + 4 * rg + j, bacc[jj][j]);
    }
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(512) void gemm_tn_ring_kernel(const bf16_t* __restrict__ DY, const bf16_t* __restrict__ X,
                                                           int M, int N, int K, int lddy, int ldx,
                                                           float* __restrict__ DW, int lddw, int m_per_slice,
                                                           float* __restrict__ DBIAS) {
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_k = K / 256;
    const int tiles = (N / 256) * tiles_k;
    (void)tiles;
    gemm_tn_ring_body<HAS_BIAS>(DY, X, M, N, K, lddy, ldx, DW, lddw, m_per_slice, DBIAS, work);
}

}  // namespace

extern "C" int va_gemm_tn_accum(const void* dy, const void* x, int64_t M, int64_t N, int64_t K, int64_t lddy,
                                int64_t ldx, float* dw, int64_t lddw, float* dbias, int split_k,
                                hipStream_t stream) {
    VA_CHECK_ARG(dy && x && dw, "va_gemm_tn_accum: null operand");
    VA_CHECK_ARG(M > 0 && N > 0 && K > 0, "va_gemm_tn_accum: empty problem");
    VA_CHECK_ARG(N % 128 == 0 && K % 128 == 0, "va_gemm_tn_accum: N=%lld and K=%lld must be multiples of 128",
                 (long long)N, (long long)K);
    VA_CHECK_ARG(lddy >= N && ldx >= K && lddw >= K, "va_gemm_tn_accum: leading dimension smaller than the row length");
    VA_CHECK_ARG(lddy % 8 == 0 && ldx % 8 == 0, "va_gemm_tn_accum: leading dimensions must keep 16-byte alignment");
    VA_CHECK_ARG(((uintptr_t)dy | (uintptr_t)x) % 16 == 0 && (uintptr_t)dw % 4 == 0, "va_gemm_tn_accum: misaligned operand");
    VA_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "va_gemm_tn_accum: problem too large");
    // ---- large problems: 256 x 256 tiles, about one workgroup per CU
    if (N % 256 == 0 && K % 256 == 0 && split_k >= 0) {
        const int rtiles = (int)((N / 256) * (K / 256));
        const int64_t rstages = (M + RBM - 1) / RBM;
        int split = split_k;
        if (split == 0) {
            split = rtiles >= 256 ? 1 : 256 / rtiles;
            // every slice adds a whole f32 tile into dW: with few rows per slice that traffic outweighs the
            // contraction, so a short M is cut into fewer, longer slices (about 48 stages at least)
            const int by_len = (int)(rstages / 48);
            if (split > by_len) split = by_len;
            if (split < 1) split = 1;
        }
        int64_t per = (rstages + split - 1) / split;          // stages per slice
        if (per < 4) per = 4;
        const int64_t m_per = per * RBM;
        const int slices = (int)((M + m_per - 1) / m_per);
        const int64_t last = M - (int64_t)(slices - 1) * m_per;   // rows of the last slice
        if (last >= 4 * RBM - (RBM - 1) && (last + RBM - 1) / RBM >= 4) {
            static bool ring_attr[2][64] = {};
            int dev = 0;
            (void)hipGetDevice(&dev);
            const int hb = dbias != nullptr;
            const void* kern = hb ? (const void*)gemm_tn_ring_kernel<true> : (const void*)gemm_tn_ring_kernel<false>;
            if (dev >= 0 && dev < 64 && !ring_attr[hb][dev]) {
                if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, RRING * RSTAGE_BYTES) !=
                    hipSuccess) {
                    va_set_error("va_gemm_tn_accum: cannot reserve %d bytes of LDS", RRING * RSTAGE_BYTES);
                    return VA_ERR_LAUNCH;
                }
                ring_attr[hb][dev] = true;
            }
            if (hb)
                hipLaunchKernelGGL(gemm_tn_ring_kernel<true>, dim3(rtiles * slices), dim3(512), RRING * RSTAGE_BYTES,
                                   stream, (const bf16_t*)dy, (const bf16_t*)x, (int)M, (int)N, (int)K, (int)lddy,
                                   (int)ldx, dw, (int)lddw, (int)m_per, dbias);
            else
                hipLaunchKernelGGL(gemm_tn_ring_kernel<false>, dim3(rtiles * slices), dim3(512), RRING * RSTAGE_BYTES,
                                   stream, (const bf16_t*)dy, (const bf16_t*)x, (int)M, (int)N, (int)K, (int)lddy,
                                   (int)ldx, dw, (int)lddw, (int)m_per, dbias);
            VA_CHECK_LAUNCH("va_gemm_tn_accum");
            return VA_OK;
        }
    }
    if (split_k < 0) split_k = -split_k;      // negative: force the 128 x 128 kernel with |split_k| slices

    const int tiles = (int)((N / 128) * (K / 128));
    const int steps = (int)((M + TBM - 1) / TBM);
    if (split_k <= 0) {
        split_k = (1024 + tiles - 1) / tiles;          // aim at ~4 workgroups per CU
        const int max_split = (steps + 7) / 8;          // at least 8 steps per slice
        if (split_k > max_split) split_k = max_split;
        if (split_k < 1) split_k = 1;
    }
    if (split_k > steps) split_k = steps;
    const int steps_per = (steps + split_k - 1) / split_k;
    const int m_per_slice = steps_per * TBM;
    const int slices = (int)((M + m_per_slice - 1) / m_per_slice);

    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE_BYTES) != hipSuccess) {
            va_set_error("va_gemm_tn_accum: cannot reserve %d bytes of LDS", 2 * STAGE_BYTES);
            return VA_ERR_LAUNCH;
        }
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, slices), dim3(256), 2 * STAGE_BYTES, stream, (const bf16_t*)dy,
                       (const bf16_t*)x, (int)M, (int)N, (int)K, (int)lddy, (int)ldx, dw, (int)lddw, m_per_slice, dbias);
    VA_CHECK_LAUNCH("va_gemm_tn_accum");
    return VA_OK;
}
