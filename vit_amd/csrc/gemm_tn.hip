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

template <int N>
VA_DEVINL void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NW = 8: waves 2 x 4, each 128(n) x 64(k).  NW = 4: waves 2 x 2, each 128 x 128 with 256 accumulator registers (one
// wave per SIMD): a stage then costs 16 fragment reads per wave instead of 12 for twice the MFMAs, i.e. two thirds of the
// LDS read traffic per FLOP -- the LDS pipe (transposing reads plus the direct-to-LDS writes) is what the eight-wave
// form keeps busiest.
template <int NW, bool DO_BIAS>
VA_DEVINL void gemm_tn_ring_body(const bf16_t* __restrict__ DY, const bf16_t* __restrict__ X,
                                  int M, int N, int K, int lddy, int ldx, float* __restrict__ DW, int lddw,
                                  int m_per_slice, float* __restrict__ DBIAS, int work) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-descriptor type exists for the device target only
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int THREADS = NW * 64, WK = NW / 2, TB = 256 / WK / 16, CPT = 1024 / THREADS, NBIAS = 8 / WK;
    const int wn = wave / WK, wk = wave % WK;

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

    // staging: 1024 chunks per operand per stage, CPT per thread: chunk q -> row q>>5, slot q&31.  Sources go through
    // buffer descriptors based at the slice's first row and the tile's first column and sized to the rows that exist:
    // lanes whose row lies past M fetch nothing (their LDS bytes are blanked after the last stage has landed).  (The
    // buffer form also keeps the compiler's LDS wait counts exact: behind a flat-family load into LDS it waits for every outstanding
    // LDS read before the next use of any.)  The stage's row offset is part of the VECTOR offset: the range check
    // does not see the scalar offset.
    const int64_t ybytes = (int64_t)(M - m_begin) * lddy * 2 - (int64_t)n0 * 2;
    const int64_t xbytes = (int64_t)(M - m_begin) * ldx * 2 - (int64_t)k0 * 2;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(DY + (size_t)m_begin * lddy + n0), (short)0, (int)(ybytes < 0x7fffffff ? ybytes : 0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(X + (size_t)m_begin * ldx + k0), (short)0, (int)(xbytes < 0x7fffffff ? xbytes : 0x7fffffff), 0x00020000);
    int yvo[CPT], xvo[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int q = i * THREADS + tid, row = q >> 5;
        const int col = ((q & 31) ^ ((row & 3) << 2) ^ (((row >> 3) & 1) << 1)) * 8;
        yvo[i] = (row * lddy + col) * 2;
        xvo[i] = (row * ldx + col) * 2;
    }
    // load j (0 .. 2*CPT-1) of a stage: chunk j>>1 of dY (j even) or of X (j odd)
    auto stage_one = [&](int kt, int slot, int j) {
        char* yb = smem + slot * RSTAGE_BYTES;
        const int i = j >> 1;
        if ((j & 1) == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_ptr_t)(yb + (i * THREADS + wave * 64) * 16), 16,
                                                     yvo[i] + kt * RBM * lddy * 2, 0, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(yb + ROP_BYTES + (i * THREADS + wave * 64) * 16), 16,
                                                     xvo[i] + kt * RBM * ldx * 2, 0, 0, 0);
    };
    auto stage = [&](int kt, int slot) {
#pragma unroll
        for (int j = 0; j < 2 * CPT; ++j) stage_one(kt, slot, j);
    };

    // D[n][k]: dY^T is the A operand, X the B operand: lane holds rows n = 4*(lane>>4) + j, column k = lane & 15
    f32x4_t acc[8][TB];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    constexpr bool do_bias = DO_BIAS;
    f32x4_t bacc[NBIAS];            // wave (wn, wk) sums the columns of n-blocks a = NBIAS*wk .. NBIAS*wk + NBIAS-1
#pragma unroll
    for (int jj = 0; jj < NBIAS; ++jj) bacc[jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
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
    int yoff[8], xoff[TB];
#pragma unroll
    for (int a = 0; a < 8; ++a) yoff[a] = tr_off(wn * 128 + a * 16);
#pragma unroll
    for (int b = 0; b < TB; ++b) xoff[b] = ROP_BYTES + tr_off(wk * (TB * 16) + b * 16);

    // fragment of the stage in `slot`
    auto frag = [&](int slot, int off) {
        const char* ptr = smem + slot * RSTAGE_BYTES + off;
        return join_tr(lds_read_tr16(ptr), lds_read_tr16(ptr + 4 * 512));
    };
    bf16x8_t fa[8], fb[TB], fbn[TB];
    int ymine[NBIAS];
#pragma unroll
    for (int jj = 0; jj < NBIAS; ++jj) ymine[jj] = tr_off(wn * 128 + (NBIAS * wk + jj) * 16);

    // zero rows [valid, 32) of both operand tiles of a slot (ragged end of M)
    auto zero_tail = [&](int slot, int valid) {
        char* base = smem + slot * RSTAGE_BYTES;
        for (int c = tid; c < 2 * RBM * 32; c += THREADS) {
            const int r = (c >> 5) & (RBM - 1);
            if (r >= valid) *(u32x4_t*)(base + c * 16) = (u32x4_t){0u, 0u, 0u, 0u};
        }
    };

    // one stage: 8 * TB MFMAs against the X fragments in `fc`; after the TB MFMAs of n-block a the next stage's dY
    // fragment a replaces the retired one, and during the first four n-blocks the next stage's X fragments are read into
    // the other set `fx` -- the next stage's first MFMAs need all of them, so they go first; the two sets swap roles from
    // stage to stage (no register copies: with one wave per SIMD nothing would cover them)
    // Four waves: the 256 accumulators are pinned to the accumulation registers by issuing the MFMAs as volatile asm
    // with an accumulation-register operand (left to the register allocator such a set is shuffled between the two
    // register files inside the loop); volatile statements keep their order and the empty memory-clobbering statements
    // pin the LDS reads and the stage's global loads (`load_kt` >= 0: one per n-block) between them.
    auto mf = [](f32x4_t& c, const bf16x8_t& a, const bf16x8_t& b) {
        if constexpr (NW == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else c = mfma16(a, b, c);
    };
    auto kstep = [&](auto prefetch, bf16x8_t (&fc)[TB], bf16x8_t (&fx)[TB], int cur_slot, int nslot, bool bias_now,
                     int load_kt, int load_slot) {
        constexpr bool PREFETCH = decltype(prefetch)::value;
        if (DO_BIAS && bias_now) {
            // uniform branch: this workgroup's share of the M stages.  The wave's two n-blocks are a run-time choice
            // (by wave index), so their fragments are read again from the stage's slot rather than selected out of
            // the register set with 24 conditional moves.
#pragma unroll
            for (int jj = 0; jj < NBIAS; ++jj) {
                const bf16x8_t mine = frag(cur_slot, ymine[jj]);
                // four waves: every accumulation register belongs to the pinned tile accumulators; a compiler-issued
                // MFMA here would borrow some of them with copies whose hazards against the asm MFMAs it cannot see
                // (the s_nop: the compiler materialises the all-ones operand with moves right in front of the statement
                // and, not knowing it for an MFMA, leaves out the wait states between a vector write and an MFMA read)
                if constexpr (NW == 4)
                    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(bacc[jj]) : "v"(mine), "v"(ones));
                else bacc[jj] = mfma16(mine, ones, bacc[jj]);
            }
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#pragma unroll
            for (int b = 0; b < TB; ++b) mf(acc[a][b], fa[a], fc[b]);
            if (NW == 4) asm volatile("" ::: "memory");
            if (NW == 4 && load_kt >= 0) stage_one(load_kt, load_slot, a);
            if (PREFETCH) {
                fa[a] = frag(nslot, yoff[a]);
                if (a < 4) {
#pragma unroll
                    for (int h = 0; h < TB / 4; ++h) fx[a * (TB / 4) + h] = frag(nslot, xoff[a * (TB / 4) + h]);
                }
            }
            if (NW == 4) asm volatile("" ::: "memory");
        }
        if (PREFETCH && NW == 8) {
#define VA_GRP(n_reads)                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, TB, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, n_reads, 0);
            VA_GRP(4) VA_GRP(4) VA_GRP(4) VA_GRP(4) VA_GRP(2) VA_GRP(2) VA_GRP(2) VA_GRP(2)
#undef VA_GRP
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
    wait_vm<6 * CPT>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (X fragments first: inside the loop they are also the earlier reads, and the compiler's wait counts at the loop head
    // are the stricter of the two paths into it)
#pragma unroll
    for (int b = 0; b < TB; ++b) fb[b] = frag(0, xoff[b]);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int a = 0; a < 8; ++a) fa[a] = frag(0, yoff[a]);

    int slot = 0;
    int kt = 0;
    auto ring_prev = [&](int sl) { return sl == 0 ? RRING - 1 : sl - 1; };
    auto mid = [&](bf16x8_t (&fc)[TB], bf16x8_t (&fx)[TB]) {       // a stage with three successors in flight
        wait_vm<4 * CPT>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (NW == 8) stage(kt + 4, ring_prev(slot));
        kstep(Yes{}, fc, fx, slot, advance(slot), bias_stage(kt), kt + 4, ring_prev(slot));
        slot = advance(slot);
        ++kt;
    };
    while (kt + 5 < nk) {
        mid(fb, fbn);
        mid(fbn, fb);
    }
    if (kt + 4 < nk) {                    // an odd one left over: one copy per kernel puts the sets back in step
        mid(fb, fbn);
#pragma unroll
        for (int b2 = 0; b2 < TB; ++b2) fb[b2] = fbn[b2];
    }
    // tail: stages nk-4 .. nk-1
    wait_vm<4 * CPT>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    kstep(Yes{}, fb, fbn, slot, advance(slot), bias_stage(kt), -1, 0);
    slot = advance(slot); ++kt;
    wait_vm<2 * CPT>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    kstep(Yes{}, fbn, fb, slot, advance(slot), bias_stage(kt), -1, 0);
    slot = advance(slot); ++kt;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tail_valid < RBM) {               // uniform: the last stage has landed everywhere, blank its unused rows
        zero_tail(advance(slot), tail_valid);
        __syncthreads();
    }
    kstep(Yes{}, fb, fbn, slot, advance(slot), bias_stage(kt), -1, 0);
    slot = advance(slot); ++kt;
    kstep(No{}, fbn, fbn, slot, 0, bias_stage(kt), -1, 0);

    // one accumulator register across the wave: four 64-byte row segments of dW
    const int rg = lane >> 4, col = lane & 15;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            float* base = DW + (size_t)(n0 + wn * 128 + a * 16 + 4 * rg) * lddw + k0 + wk * (TB * 16) + b * 16 + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[a][b][j];
                // read out of the accumulation file at the point of use: hoisted, the 256 copies would spill
                if constexpr (NW == 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[a][b][j]));
                unsafeAtomicAdd(base + (size_t)j * lddw, v);
            }
        }
    if (do_bias && col == 0) {   // every column of bacc holds the same sums: lane column 0 publishes them
#pragma unroll
        for (int jj = 0; jj < NBIAS; ++jj)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                unsafeAtomicAdd(DBIAS + n0 + wn * 128 + (NBIAS * wk + jj) * 16 + 4 * rg + j, bacc[jj][j]);
    }
#endif
}

template <int NW, bool HAS_BIAS>
__global__ __launch_bounds__(NW * 64) void gemm_tn_ring_kernel(const bf16_t* __restrict__ DY, const bf16_t* __restrict__ X,
                                                           int M, int N, int K, int lddy, int ldx,
                                                           float* __restrict__ DW, int lddw, int m_per_slice,
                                                           float* __restrict__ DBIAS) {
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_k = K / 256;
    const int tiles = (N / 256) * tiles_k;
    (void)tiles;
    gemm_tn_ring_body<NW, HAS_BIAS>(DY, X, M, N, K, lddy, ldx, DW, lddw, m_per_slice, DBIAS, work);
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
        // (a slice's rows must lie within one buffer descriptor's 2 GiB)
        const bool fits = (m_per + RBM) * (lddy > ldx ? lddy : ldx) * 2 < 0x7fffffffll;
        if (fits && last >= 4 * RBM - (RBM - 1) && (last + RBM - 1) / RBM >= 4) {
            // four waves with 128 x 128 each unless VA_GEMM_TN_WAVES=8 asks for the eight-wave form (A/B timing)
            static const bool eight = [] {
                const char* e = getenv("VA_GEMM_TN_WAVES");
                return e && atoi(e) == 8;
            }();
            static bool ring_attr[4][64] = {};
            int dev = 0;
            (void)hipGetDevice(&dev);
            const int hb = dbias != nullptr, which = 2 * (int)eight + hb;
            using Kern = void (*)(const bf16_t*, const bf16_t*, int, int, int, int, int, float*, int, int, float*);
            const Kern kerns[4] = {gemm_tn_ring_kernel<4, false>, gemm_tn_ring_kernel<4, true>, gemm_tn_ring_kernel<8, false>,
                                   gemm_tn_ring_kernel<8, true>};
            const Kern kern = kerns[which];
            if (dev >= 0 && dev < 64 && !ring_attr[which][dev]) {
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RRING * RSTAGE_BYTES) !=
                    hipSuccess) {
                    va_set_error("va_gemm_tn_accum: cannot reserve %d bytes of LDS", RRING * RSTAGE_BYTES);
                    return VA_ERR_LAUNCH;
                }
                ring_attr[which][dev] = true;
            }
            hipLaunchKernelGGL(kern, dim3(rtiles * slices), dim3(eight ? 512 : 256), RRING * RSTAGE_BYTES, stream,
                               (const bf16_t*)dy, (const bf16_t*)x, (int)M, (int)N, (int)K, (int)lddy, (int)ldx, dw, (int)lddw,
                               (int)m_per, dbias);
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
