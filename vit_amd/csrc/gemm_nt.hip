// bf16 MFMA GEMM, both operands K-contiguous:  C[M,N] = X[M,K] * W[N,K]^T  (+ fused epilogue).
//
// This one kernel family carries every dense projection of the encoder: patch embedding (after the
// tubelet gather), QKV, attention out-projection, MLP fc1 / fc2, and all data-gradient GEMMs (which
// run against a transposed bf16 copy of the weight so that they have the same K-contiguous shape).
//
// Common structure (gfx950):
//   * operand tiles are staged with 16-byte direct-to-LDS loads (global_load_lds_dwordx4).  The LDS image
//     is lane-linear; bank conflicts of the ds_read_b128 fragment reads are removed by permuting the
//     per-lane SOURCE address (XOR on the 16-byte chunk index) and applying the same XOR when reading.
//   * each wave owns a (BM/WM) x (BN/WN) sub-tile as 16x16x32 bf16 MFMAs.  In the 128-wide "pair" kernel the
//     weight tile is the MFMA A operand, so an accumulator lane holds 4 consecutive output columns of one
//     output row; the 256-wide kernels put the activation there and stage the weight rows in a permuted order,
//     so that neighbouring lanes hold neighbouring columns of the same output row (see wide_wrow).
//   * launch-order block ids are remapped so that the blocks sharing an XCD walk neighbouring tiles.
//
// Three main loops:
//   quad  (256x256 tile, 4 waves of 128x128, persistent): the default for large problems; see the kernel.
//   ring  (256x256 tile, 8 waves of 64 x 128): K advances in 32-wide stages through a 5-slot LDS ring (5 x 32 KiB, the
//         whole LDS of a CU).  Default form ("paired"): loads are issued for two stages at a time so that
//         both 64-byte halves of every 128-byte line are requested back to back, one pair in flight while
//         two stages compute, counted s_waitcnt + raw s_barrier every two stages.  Alternative form: one
//         stage per barrier, three stages in flight.  MFMA fragments are read from LDS one stage ahead of
//         their use, interleaved with the MFMAs of the current stage, so the LDS pipe and the matrix pipe
//         overlap.  An epilogue store instruction writes 4 rows x 128 (2-byte outputs) or 256 contiguous bytes.
//   pair  (128x128 tile, 4 waves; also selectable at 256x256): K advances in 64-wide tiles through two
//         buffers, next tile issued before the current tile's MFMAs, vmcnt(0) + barrier per tile.  Used
//         for small M and short K.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <mutex>

namespace {

struct EpiArgs {
    void* out;            // bf16 or f32 [M, ldc]
    void* out2;           // bf16 [M, ldc]: GELU output (VA_EPI_BIAS_GELU, VA_EPI_BIAS_GELU_GRAD)
    const float* bias;    // [N] or null
    const float* resid;   // f32 [M, ldc] or null (VA_EPI_F32)
    const bf16_t* aux;    // bf16 [M, ldc]: saved pre-activation (VA_EPI_GELU_BWD) or factor (VA_EPI_MUL_AUX)
    int ldc;
    // Fused dropout (drop_thi != 0): applied to the GELU output (BIAS_GELU: out2), to acc + bias before the residual
    // (F32), to the product with the GELU derivative (GELU_BWD).  Decisions are those of va_dropout over a contiguous
    // [M, N] matrix: element index m * N + n, so the output must be stored with ldc == N.
    uint32_t drop_key, drop_thi;
    float drop_scale;
    int nt;               // store the outputs with the non-temporal hint (set by the host from their size)
    // QKV projection with the rotary embedding applied in the epilogue (EPI_QKV_ROPE, four-wave kernel only):
    const float* rope_tables;      // [cos | sin] tables of va_rope3d_qk_inplace
    const int32_t* rope_pos;       // per output row: t | h << 10 | w << 20
    int rope_tp, rope_hp, rope_wp;
};
constexpr int EPI_QKV_ROPE = 100;      // internal epilogue code: reached through va_gemm_nt_qkv_rope only

constexpr bool epi_reads_aux(int epi) { return epi == VA_EPI_GELU_BWD || epi == VA_EPI_MUL_AUX; }
constexpr bool epi_two_outputs(int epi) { return epi == VA_EPI_BIAS_GELU || epi == VA_EPI_BIAS_GELU_GRAD; }

// keep-or-zero and rescale of `cnt` (even) consecutive elements starting at element index e0 (even)
template <int CNT>
VA_DEVINL void drop_run(float* v, uint32_t e0, const EpiArgs& ep) {
    const uint32_t w0 = drop_weyl(ep.drop_key, e0 >> 1);
#pragma unroll
    for (int q = 0; q < CNT / 2; ++q) {
        const uint32_t hsh = drop_mix(w0 + (uint32_t)q * DROP_WEYL);
        v[2 * q] = drop_keep_even(hsh, ep.drop_thi) ? __fmul_rn(v[2 * q], ep.drop_scale) : 0.f;
        v[2 * q + 1] = drop_keep_odd(hsh, ep.drop_thi) ? __fmul_rn(v[2 * q + 1], ep.drop_scale) : 0.f;
    }
}

// Epilogue for a wave's accumulators acc[a][b] (a: 16-column block of n, b: 16-row block of m).
// Lane holds D rows n = nb..nb+3 (4 consecutive output columns) for output row m.
template <int EPI, int TM, int TN>
VA_DEVINL void store_tile(const f32x4_t (&acc)[TN][TM], const EpiArgs& ep, int M, int m_wave, int n_wave, int lane) {
    const int ldc = ep.ldc;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m_wave + b * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int nb = n_wave + a * 16 + (lane >> 4) * 4;
            const size_t o = (size_t)m * ldc + nb;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            if (!epi_reads_aux(EPI) && ep.bias != nullptr) {
                const f32x4_t bv = *(const f32x4_t*)(ep.bias + nb);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += bv[j];
            }
            if (EPI == VA_EPI_BF16) {
                u32x2_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2_t*)((bf16_t*)ep.out + o) = pk;
            } else if (EPI == VA_EPI_BIAS_GELU) {
                u32x2_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2_t*)((bf16_t*)ep.out + o) = pk;
                float gl[4] = {gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
                if (ep.drop_thi) drop_run<4>(gl, (uint32_t)o, ep);
                u32x2_t gk = {pack_bf16x2(gl[0], gl[1]), pack_bf16x2(gl[2], gl[3])};
                *(u32x2_t*)((bf16_t*)ep.out2 + o) = gk;
            } else if (EPI == VA_EPI_BIAS_GELU_GRAD) {
                float gl[4], gd[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) gelu_erf_both(v[j], gl[j], gd[j]);
                u32x2_t dk = {pack_bf16x2(gd[0], gd[1]), pack_bf16x2(gd[2], gd[3])};
                *(u32x2_t*)((bf16_t*)ep.out + o) = dk;
                if (ep.drop_thi) drop_run<4>(gl, (uint32_t)o, ep);
                u32x2_t gk = {pack_bf16x2(gl[0], gl[1]), pack_bf16x2(gl[2], gl[3])};
                *(u32x2_t*)((bf16_t*)ep.out2 + o) = gk;
            } else if (EPI == VA_EPI_GELU_ONLY) {
                float gl[4] = {gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
                if (ep.drop_thi) drop_run<4>(gl, (uint32_t)o, ep);
                u32x2_t gk = {pack_bf16x2(gl[0], gl[1]), pack_bf16x2(gl[2], gl[3])};
                *(u32x2_t*)((bf16_t*)ep.out + o) = gk;
            } else if (EPI == VA_EPI_F32) {
                if (ep.drop_thi) drop_run<4>(v, (uint32_t)o, ep);
                if (ep.resid != nullptr) {
                    const f32x4_t rv = *(const f32x4_t*)(ep.resid + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += rv[j];
                }
                *(f32x4_t*)((float*)ep.out + o) = (f32x4_t){v[0], v[1], v[2], v[3]};
            } else {  // VA_EPI_GELU_BWD, VA_EPI_MUL_AUX
                const u32x2_t pre = *(const u32x2_t*)(ep.aux + o);
                const float p0 = __uint_as_float(pre[0] << 16), p1 = __uint_as_float(pre[0] & 0xffff0000u);
                const float p2 = __uint_as_float(pre[1] << 16), p3 = __uint_as_float(pre[1] & 0xffff0000u);
                float r[4];
                if (EPI == VA_EPI_MUL_AUX) {
                    r[0] = v[0] * p0; r[1] = v[1] * p1; r[2] = v[2] * p2; r[3] = v[3] * p3;
                } else {
                    r[0] = v[0] * gelu_erf_grad(p0); r[1] = v[1] * gelu_erf_grad(p1);
                    r[2] = v[2] * gelu_erf_grad(p2); r[3] = v[3] * gelu_erf_grad(p3);
                }
                if (ep.drop_thi) drop_run<4>(r, (uint32_t)o, ep);
                u32x2_t pk = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
                *(u32x2_t*)((bf16_t*)ep.out + o) = pk;
            }
        }
    }
}

// The 256-wide kernels issue their MFMAs with the ACTIVATION fragment as the A operand and the weight fragment as B, so
// an accumulator register j of lane l belongs to output row 4*(l>>4) + j of its 16-row block and to the weight row that
// lane l & 15 of the B fragment carried.  A lane therefore owns, for each of its rows, one value per 16-row weight block
// of the wave (NV = WCOLS / 16 of them).  Which weight row LDS row `row` of the W image holds is chosen so that those NV
// values are CONSECUTIVE output columns and neighbouring lanes continue them:
//   2-byte outputs: lane r owns columns r*NV .. r*NV + NV-1 of the wave's WCOLS columns -- one 16-byte (NV = 8) or
//                   8-byte (NV = 4) store per lane and row, 16 lanes = 256 (128) contiguous bytes, 4 rows per instruction;
//   4-byte outputs (f32 + bias + residual) keep the weight as the A operand and the natural row order: register j of
//                   lane l is row l & 15, column 16*v + 4*(l>>4) + j.  That epilogue reads and writes 4 bytes per element
//                   and is bound by those bytes, not by store issue; the row-contiguous form measured 6 % slower on the
//                   K = 1024 shape and equal at K = 4096.
// What this buys: a store instruction whose CONSECUTIVE lanes write consecutive addresses.  Measured on the four-wave
// kernel (timing-only builds that store zeros, tools/quad_epi_prof.py): 32 stores of 16 rows x 64 bytes, the 64 bytes of a
// row spread over lanes 16 apart (the accumulator layout with the weight as the A operand), hold the wave for 8400
// cycles per tile; 8 rows x 128 bytes or 4 rows x 256 bytes from neighbouring lanes for 3400-3500.
template <int EPI, int WCOLS>
VA_DEVINL int wide_wrow(int row) {
    constexpr int NV = WCOLS / 16;
    const int r = row & 15, v = (row >> 4) % NV, base = row - (row % WCOLS);
    if (EPI == VA_EPI_F32) return row;
    return base + r * NV + v;
}
// Whether the weight fragment is the MFMA's A operand (the 4-byte epilogue) or its B operand (the 2-byte ones).
template <int EPI>
constexpr bool w_is_a() { return EPI == VA_EPI_F32; }

template <int N>
struct PackedBf16;
template <>
struct PackedBf16<8> { typedef u32x4_t type; };
template <>
struct PackedBf16<4> { typedef u32x2_t type; };

template <int NV>
VA_DEVINL typename PackedBf16<NV>::type pack_run(const float* v) {
    typename PackedBf16<NV>::type p;
#pragma unroll
    for (int q = 0; q < NV / 2; ++q) p[q] = pack_bf16x2(v[2 * q], v[2 * q + 1]);
    return p;
}
template <int NV>
VA_DEVINL void unpack_run(typename PackedBf16<NV>::type p, float* v) {
#pragma unroll
    for (int q = 0; q < NV / 2; ++q) {
        v[2 * q] = __uint_as_float(p[q] << 16);
        v[2 * q + 1] = __uint_as_float(p[q] & 0xffff0000u);
    }
}

// Epilogue of the 256-wide kernels for the layout above.  get(v, b, j): accumulator register j of 16-row block b and
// weight block v.  With ep.nt the stores carry the non-temporal hint: an output is hundreds of times the size of an L2,
// where it only evicts the operand panels that neighbouring tiles are about to re-read (the main loop of the four-wave
// kernel runs 9 % faster behind such an epilogue) -- but the hint also keeps it out of the last-level cache, where a
// consumer kernel that starts right after would have found an output of up to about that cache's size; the host sets
// the flag from the output's size.
// FULL_M: all of the wave's rows exist, so the row loop carries no branches (a skipped row would make the compiler drain
// the memory counter in front of every later row's stores).
// HAS_BIAS = false: the caller knows there is no bias and the adds are left out; true: a null bias pointer adds zeros.
// RESID_AHEAD: unused (kept so that call sites read the same); the f32 epilogue always requests one row block ahead.
// CAN_DROP = false compiles the fused-dropout branches out (the four-wave kernel has no registers to spare for them).
template <typename V>
VA_DEVINL void put_16(V v, V* p, bool nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Two 16-byte pieces per lane -- `lo` at p (row lane & 15 of a 16-row block, 64 contiguous bytes per row over the four
// lane groups) and `hi` 64 bytes further along the same row -- written as whole 128-byte lines: the upper half of each
// 16-lane row hands its `hi` to the lower half's rows and vice versa (one rotate-by-8 per register, merged by the DPP
// bank mask), so that one store instruction covers 8 rows x 128 bytes instead of 16 rows x 64.  All 16 rows must exist.
template <typename V>
VA_DEVINL void store_line_pairs(char* p, V lo, V hi, int64_t row_bytes, bool upper, bool nt) {
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    const i32x4_t l = __builtin_bit_cast(i32x4_t, lo), h = __builtin_bit_cast(i32x4_t, hi);
    i32x4_t a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = __builtin_amdgcn_update_dpp(l[j], h[j], 0x128, 0xf, 0xc, false);   // row_ror:8 into lanes 8..15 of each row
        b[j] = __builtin_amdgcn_update_dpp(l[j], h[j], 0x128, 0xf, 0x3, false);   // ... into lanes 0..7
    }
    char* pa = upper ? p + 64 - 8 * row_bytes : p;      // rows 0..7 of the block, both halves of their lines
    char* pb = upper ? p : p + 64 + 8 * row_bytes;      // rows 8..15
    if (nt) {
        __builtin_nontemporal_store(a, (i32x4_t*)pa);
        __builtin_nontemporal_store(b, (i32x4_t*)pb);
    } else {
        *(i32x4_t*)pa = a;
        *(i32x4_t*)pb = b;
    }
}

template <int EPI, int TM, int NV, bool FULL_M, bool HAS_BIAS, bool RESID_AHEAD, bool CAN_DROP, typename Get>
VA_DEVINL void store_tile_rows(Get get, const EpiArgs& ep, int M, int m_wave, int n_wave, int lane) {
    const int ldc = ep.ldc;
    const int r = lane & 15, g = lane >> 4;
    const bool nt = ep.nt != 0;
    typedef typename PackedBf16<NV>::type pk_t;
    if (EPI == VA_EPI_F32) {
        // weight as the A operand: register q of (v, b) is row b*16 + r, column 16*v + 4*g + q.  The residual values of
        // the NEXT row block are requested before this block's stores, so that waiting for them does not wait for the
        // stores (memory operations retire in issue order; the compiler may not move loads of `resid` above stores to
        // `out` on its own).  Safe for out == resid: a lane reads exactly the elements it later writes.
#pragma unroll
        for (int h = 0; h < NV / 4; ++h) {
            auto col = [&](int a) { return n_wave + (4 * h + a) * 16 + 4 * g; };
            f32x4_t bias[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bias[a] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if (HAS_BIAS && ep.bias != nullptr) bias[a] = *(const f32x4_t*)(ep.bias + col(a));
            }
            f32x4_t nxt[4];
            auto fetch = [&](int b) {
                const int m = m_wave + b * 16 + r;
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    nxt[a] = (ep.resid != nullptr && (FULL_M || m < M)) ? *(const f32x4_t*)(ep.resid + (size_t)m * ldc + col(a))
                                                                       : (f32x4_t){0.f, 0.f, 0.f, 0.f};
            };
            fetch(0);
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = m_wave + b * 16 + r;
                f32x4_t cur[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) cur[a] = nxt[a];
                if (b + 1 < TM) fetch(b + 1);
                if (!FULL_M && m >= M) continue;
#pragma unroll
                for (int a2 = 0; a2 < 4; a2 += 2) {
                    f32x4_t res[2];
#pragma unroll
                    for (int a = a2; a < a2 + 2; ++a) {
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = HAS_BIAS ? get(4 * h + a, b, q) + bias[a][q] : get(4 * h + a, b, q);
                        if (CAN_DROP && ep.drop_thi) drop_run<4>(v, (uint32_t)((size_t)m * ldc + col(a)), ep);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += cur[a][q];
                        res[a - a2] = (f32x4_t){v[0], v[1], v[2], v[3]};
                    }
                    float* o = (float*)ep.out + (size_t)m * ldc + col(a2);
                    if (FULL_M) {      // blocks a2, a2 + 1 are the two halves of a 128-byte line: whole lines, non-temporal
                        store_line_pairs((char*)o, res[0], res[1], (int64_t)ldc * 4, (lane & 8) != 0, nt);
                    } else {
                        *(f32x4_t*)o = res[0];
                        *(f32x4_t*)(o + 16) = res[1];
                    }
                }
            }
        }
        return;
    }
    const int col = n_wave + r * NV;
    float bias[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) bias[q] = 0.f;
    if (HAS_BIAS && !epi_reads_aux(EPI) && ep.bias != nullptr) {
#pragma unroll
        for (int q = 0; q < NV; q += 4) *(f32x4_t*)(bias + q) = *(const f32x4_t*)(ep.bias + col + q);
    }
    // x GELU': the saved pre-activation of the NEXT row is requested before this row's store, so that waiting for it
    // does not wait for the store (memory operations retire in issue order)
    pk_t aux_next = {};
    auto aux_row = [&](int i) {      // i = 4 * b + j
        const int m = m_wave + (i >> 2) * 16 + 4 * g + (i & 3);
        return (FULL_M || m < M) ? *(const pk_t*)(ep.aux + (size_t)m * ldc + col) : pk_t{};
    };
    if (epi_reads_aux(EPI)) aux_next = aux_row(0);
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m_wave + b * 16 + 4 * g + j;
            pk_t aux_cur = aux_next;
            if (epi_reads_aux(EPI) && 4 * b + j + 1 < 4 * TM) aux_next = aux_row(4 * b + j + 1);
            if (!FULL_M && m >= M) continue;
            const size_t o = (size_t)m * ldc + col;
            float v[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] = HAS_BIAS ? get(q, b, j) + bias[q] : get(q, b, j);
            if (EPI == VA_EPI_BF16) {
                put_16(pack_run<NV>(v), (pk_t*)((bf16_t*)ep.out + o), nt);
            } else if (EPI == VA_EPI_BIAS_GELU || EPI == VA_EPI_GELU_ONLY) {
                if (EPI == VA_EPI_BIAS_GELU) put_16(pack_run<NV>(v), (pk_t*)((bf16_t*)ep.out + o), nt);
                float gl[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) gl[q] = gelu_erf(v[q]);
                if (CAN_DROP && ep.drop_thi) drop_run<NV>(gl, (uint32_t)o, ep);
                put_16(pack_run<NV>(gl), (pk_t*)((bf16_t*)(EPI == VA_EPI_BIAS_GELU ? ep.out2 : ep.out) + o), nt);
            } else if (EPI == VA_EPI_BIAS_GELU_GRAD) {
                float gl[NV], gd[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) gelu_erf_both(v[q], gl[q], gd[q]);
                put_16(pack_run<NV>(gd), (pk_t*)((bf16_t*)ep.out + o), nt);
                if (CAN_DROP && ep.drop_thi) drop_run<NV>(gl, (uint32_t)o, ep);
                put_16(pack_run<NV>(gl), (pk_t*)((bf16_t*)ep.out2 + o), nt);
            } else {  // VA_EPI_GELU_BWD, VA_EPI_MUL_AUX
                float pre[NV], rr[NV];
                unpack_run<NV>(aux_cur, pre);
#pragma unroll
                for (int q = 0; q < NV; ++q) rr[q] = EPI == VA_EPI_MUL_AUX ? v[q] * pre[q] : v[q] * gelu_erf_grad(pre[q]);
                if (CAN_DROP && ep.drop_thi) drop_run<NV>(rr, (uint32_t)o, ep);
                put_16(pack_run<NV>(rr), (pk_t*)((bf16_t*)ep.out + o), nt);
            }
        }
}

// out = acc * aux for the four-wave kernel (a lane owns 8 consecutive columns of rows 4g + j of each 16-row block).  The
// factor pieces of ALL the wave's rows are already in registers: the kernel requests them -- 32 loads in flight together
// -- at the head of the tile's last stage, whose MFMAs cover most of the round trip (a load issued in the epilogue,
// behind a store, would wait for that store too: memory operations retire in order, and with one wave per SIMD nothing
// else covers the wait).  The next tile's first fragments are read after this epilogue, which frees the 128 registers.
template <int TM, bool FULL_M, typename Get>
VA_DEVINL void store_tile_mul_aux(Get get, const u32x4_t (&aux)[4 * TM], const EpiArgs& ep, int M, int m_wave, int n_wave,
                                  int lane) {
    const int ldc = ep.ldc;
    const int r = lane & 15, g = lane >> 4;
    const bool nt = ep.nt != 0;
    const int col = n_wave + r * 8;
#pragma unroll
    for (int i = 0; i < 4 * TM; ++i) {
        const int b = i >> 2, j = i & 3;
        const int m = m_wave + b * 16 + 4 * g + j;
        if (!FULL_M && m >= M) continue;
        float f[8], v[8];
        unpack_run<8>(aux[i], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = get(q, b, j) * f[q];
        put_16(pack_run<8>(v), (u32x4_t*)((bf16_t*)ep.out + (size_t)m * ldc + col), nt);
    }
}

// Epilogue of the fused QKV projection + 3-D rotary embedding (swm:265-281 on the output of swm:300-308).  The 3072 output
// columns are q | k | v; inside q and inside k, column c of the time segment [0, 256) rotates with c + 128, column 256 + i
// of the height segment [256, 640) with 448 + i, column 640 + i of the width segment [640, 1024) with 832 + i.  The
// kernel stages the weight rows so that a wave owns 64 such pairs -- 64 consecutive "first" columns from a0 on and their 64
// partners from b0 on -- and lane r the four pairs a0 + 4r .. + 3: its accumulator values 0..3 of an output row are the
// first columns, 4..7 the partners, and the rotation needs no other lane.  The value third of the output takes the same
// path with cos = 1, sin = 0 left out (plain stores of two 4-column pieces).  toff: the lane's offset into a position's
// table row, or -1 for the value third.  Rotation in f32 on the accumulators (the separate pass rotates the rounded
// bf16 product): y_a = a cos - b sin, y_b = b cos + a sin.
template <int TM, bool FULL_M, typename Get>
VA_DEVINL void store_tile_qkv_rope(Get get, const int (&code)[TM * 4], const EpiArgs& ep, int M, int m_wave, int col_a, int col_b,
                                   int seg, int seg_off, int lane) {
    const int ldc = ep.ldc;
    const int r = lane & 15, g = lane >> 4;
    const bool nt = ep.nt != 0;
    const int tab_sz = ep.rope_tp * 128 + (ep.rope_hp + ep.rope_wp) * 192;
    const bool rot = seg >= 0;
    // `code`: every row's packed position, requested by the kernel at the head of the tile's last stage (0 for rows past
    // M).  The table rows are requested DEPTH output rows ahead of their use: each is a cache round trip that nothing else
    // covers with one wave per SIMD, and a load issued behind a store waits for that store too (memory operations retire
    // in order), which the distance also relaxes.
    const int base = seg == 0 ? 0 : seg == 1 ? ep.rope_tp * 128 : ep.rope_tp * 128 + ep.rope_hp * 192;
    const int stride = seg == 0 ? 128 : 192, shift = seg == 0 ? 0 : seg == 1 ? 10 : 20;
    constexpr int DEPTH = 6;
    int toff[TM * 4];
#pragma unroll
    for (int i = 0; i < TM * 4; ++i) toff[i] = base + ((code[i] >> shift) & 1023) * stride + seg_off + 4 * r;
    f32x4_t cs_q[DEPTH], sn_q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        cs_q[d] = (f32x4_t){1.f, 1.f, 1.f, 1.f};
        sn_q[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (rot) {
            cs_q[d] = *(const f32x4_t*)(ep.rope_tables + toff[d]);
            sn_q[d] = *(const f32x4_t*)(ep.rope_tables + tab_sz + toff[d]);
        }
    }
#pragma unroll
    for (int i = 0; i < TM * 4; ++i) {
        const int b = i >> 2, j = i & 3;
        const int m = m_wave + b * 16 + 4 * g + j;
        const f32x4_t cs = cs_q[i % DEPTH], sn = sn_q[i % DEPTH];
        if (rot && i + DEPTH < TM * 4) {
            cs_q[i % DEPTH] = *(const f32x4_t*)(ep.rope_tables + toff[i + DEPTH]);
            sn_q[i % DEPTH] = *(const f32x4_t*)(ep.rope_tables + tab_sz + toff[i + DEPTH]);
        }
        if (!FULL_M && m >= M) continue;
        float a[4], bb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = get(q, b, j); bb[q] = get(4 + q, b, j); }
        if (rot) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float ya = a[q] * cs[q] - bb[q] * sn[q];
                const float yb = bb[q] * cs[q] + a[q] * sn[q];
                a[q] = ya;
                bb[q] = yb;
            }
        }
        bf16_t* o = (bf16_t*)ep.out + (size_t)m * ldc;
        put_16((u32x2_t){pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])}, (u32x2_t*)(o + col_a + 4 * r), nt);
        put_16((u32x2_t){pack_bf16x2(bb[0], bb[1]), pack_bf16x2(bb[2], bb[3])}, (u32x2_t*)(o + col_b + 4 * r), nt);
    }
}

// Where the 64 pairs of half-tile (nt, wn) of the fused QKV + rotary kernel lie: first columns from a0, partners from b0
// (columns of the 3072-wide output), segment 0 / 1 / 2 = time / height / width (-1: the value third, no rotation), and the
// offset of a0 inside its segment's table row.
VA_DEVINL void qkv_rope_cols(int nt, int wn, int& a0, int& b0, int& seg, int& seg_off) {
    const int part = nt >> 2, pair0 = (nt & 3) * 128 + wn * 64;     // q / k / v; first of the 64 pairs (of 512 per part)
    if (part == 2) {
        a0 = 2048 + (nt & 3) * 256 + wn * 128;
        b0 = a0 + 64;
        seg = -1;
        seg_off = 0;
    } else if (pair0 < 128) {
        a0 = part * 1024 + pair0; b0 = a0 + 128; seg = 0; seg_off = pair0;
    } else if (pair0 < 320) {
        a0 = part * 1024 + 256 + (pair0 - 128); b0 = a0 + 192; seg = 1; seg_off = pair0 - 128;
    } else {
        a0 = part * 1024 + 640 + (pair0 - 320); b0 = a0 + 192; seg = 2; seg_off = pair0 - 320;
    }
}

// ------------------------------------------------------------------------------------------------
// "pair" loop: BK = 64, two LDS buffers.  Rows are 128 B (8 chunks); chunk XOR ((row>>1)&7).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_pair_kernel(const bf16_t* __restrict__ X,
                                                                    const bf16_t* __restrict__ W, int M, int N, int K,
                                                                    int ldx, int ldw, EpiArgs ep) {
    constexpr int BK = 64;
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    constexpr int X_BYTES = BM * BK * 2;
    constexpr int W_BYTES = BN * BK * 2;
    constexpr int STAGE_BYTES = X_BYTES + W_BYTES;
    constexpr int XI = BM * 8 / NT;   // 16-byte chunks per thread per X tile
    constexpr int WI = BN * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = N / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (id / tiles_n) * BM;
    const int n0 = (id % tiles_n) * BN;

    const bf16_t* xsrc[XI];
    const bf16_t* wsrc[WI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int q = i * NT + tid, row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int grow = m0 + row;
        grow = grow < M ? grow : M - 1;
        xsrc[i] = X + (size_t)grow * ldx + c * 8;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int q = i * NT + tid, row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        wsrc[i] = W + (size_t)(n0 + row) * ldw + c * 8;
    }

    auto stage = [&](int buf, int k0) {
        char* xb = smem + buf * STAGE_BYTES;
        char* wb = xb + X_BYTES;
#pragma unroll
        for (int i = 0; i < XI; ++i) glds16(xsrc[i] + k0, xb + (i * NT + wave * 64) * 16);
#pragma unroll
        for (int i = 0; i < WI; ++i) glds16(wsrc[i] + k0, wb + (i * NT + wave * 64) * 16);
    };

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fchunk = lane >> 4;
    int woff[TN], xoff[TM];
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int row = wn * (BN / WN) + a * 16 + frow;
        woff[a] = row * 128 + ((fchunk ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int row = wm * (BM / WM) + b * 16 + frow;
        xoff[b] = row * 128 + ((fchunk ^ ((row >> 1) & 7)) << 4);
    }

    auto compute = [&](int buf) {
        const char* xb = smem + buf * STAGE_BYTES;
        const char* wb = xb + X_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {   // k-step 1 flips chunk bit 2
            bf16x8_t af[TN], bfr[TM];
#pragma unroll
            for (int a = 0; a < TN; ++a) af[a] = *(const bf16x8_t*)(wb + (woff[a] ^ (kk << 6)));
#pragma unroll
            for (int b = 0; b < TM; ++b) bfr[b] = *(const bf16x8_t*)(xb + (xoff[b] ^ (kk << 6)));
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) acc[a][b] = mfma16(af[a], bfr[b], acc[a][b]);
        }
    };

    const int nk = K / BK;
    stage(0, 0);
    wait_vm0();
    __syncthreads();
    for (int kt = 0; kt < nk - 1; ++kt) {
        stage((kt + 1) & 1, (kt + 1) * BK);
        compute(kt & 1);
        wait_vm0();
        __syncthreads();
    }
    compute((nk - 1) & 1);

    store_tile<EPI, TM, TN>(acc, ep, M, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

// ------------------------------------------------------------------------------------------------
// "ring" loop: 256 x 256 tile, 8 waves (4 x 2, 64 x 128 each), BK = 32 stages in a 5-slot ring (all 160 KiB of LDS).
// Rows are 64 B (4 chunks); chunk XOR (3 * ((row>>3)&1)) makes the 16x16x32 fragment reads conflict-free.
// Weight rows are staged in a permuted order (wide_wrow) so that the epilogue's lanes store contiguous row pieces.
// ------------------------------------------------------------------------------------------------
template <int EPI, bool PAIRED>
__global__ __launch_bounds__(512) void gemm_nt_ring_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                           int M, int N, int K, int ldx, int ldw, EpiArgs ep) {
    constexpr int BM = 256, BN = 256, BK = 32, WN = 2, NT = 512;
    constexpr int TM = 4, TN = 8;                 // per wave: 64 (m) x 128 (n): a lane of the epilogue owns 8 consecutive columns
    constexpr int OP_BYTES = 256 * BK * 2;        // 16 KiB per operand per stage
    constexpr int STAGE_BYTES = 2 * OP_BYTES;     // 32 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = N / BN;
    const int tiles_m = (M + BM - 1) / BM;
    int mt_, nt_;
    grouped_tile(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, mt_, nt_);
    const int m0 = mt_ * BM;
    const int n0 = nt_ * BN;

    // staging: 1024 chunks per operand per stage, 2 per thread: chunk q -> row q>>2, slot q&3
    const bf16_t* xsrc[2];
    const bf16_t* wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * NT + tid, row = q >> 2, c = (q & 3) ^ (((row >> 3) & 1) * 3);
        int grow = m0 + row;
        grow = grow < M ? grow : M - 1;
        xsrc[i] = X + (size_t)grow * ldx + c * 8;
        wsrc[i] = W + (size_t)(n0 + wide_wrow<EPI, 128>(row)) * ldw + c * 8;
    }
    auto stage = [&](int kt, int slot) {
        char* xb = smem + slot * STAGE_BYTES;
        char* wb = xb + OP_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(xsrc[i] + k0, xb + (i * NT + wave * 64) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(wsrc[i] + k0, wb + (i * NT + wave * 64) * 16);
    };

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment offset of this lane inside a 16-row block: row (lane&15), chunk (lane>>4) swizzled
    const int frow = lane & 15;
    const int foff = frow * 64 + (((lane >> 4) ^ (((frow >> 3) & 1) * 3)) << 4);
    const int woff = wn * 128 * 64 + foff;    // wave's first W row * 64 B
    const int xoff = wm * 64 * 64 + foff;

    // Fragments live in registers one stage ahead of the MFMAs that consume them: while stage kt's 32 MFMAs
    // issue, stage kt+1's fragments are read from LDS into the registers the retired ones leave behind
    // (the W fragment of column block a right after its 4 MFMAs; the X fragments into the other half of a
    // double buffer).  Matrix pipe and LDS pipe therefore overlap instead of alternating.
    bf16x8_t wf[TN];
    bf16x8_t xf[2][TM];
    auto frag_w = [&](int slot, int a) {
        return *(const bf16x8_t*)(smem + slot * STAGE_BYTES + OP_BYTES + woff + a * 16 * 64);
    };
    auto frag_x = [&](int slot, int b) { return *(const bf16x8_t*)(smem + slot * STAGE_BYTES + xoff + b * 16 * 64); };

    // One stage.  P = parity of kt (selects the W-fragment buffer); PREFETCH = stage kt+1 exists, its ring
    // slot is `next_slot`.
    auto body = [&](auto parity, auto prefetch, int next_slot) {
        constexpr int P = decltype(parity)::value;
        constexpr bool PREFETCH = decltype(prefetch)::value;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
#pragma unroll
            for (int b = 0; b < TM; ++b)
                acc[a][b] = w_is_a<EPI>() ? mfma16(wf[a], xf[P][b], acc[a][b]) : mfma16(xf[P][b], wf[a], acc[a][b]);
            if (PREFETCH) {
                wf[a] = frag_w(next_slot, a);
                if (a < TM) xf[P ^ 1][a] = frag_x(next_slot, a);
            }
        }
        if (PREFETCH) {
            // pin the interleave: 4 MFMAs, then the LDS reads whose registers they freed
#define VA_GRP(n_reads)                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);       \
    __builtin_amdgcn_sched_group_barrier(0x100, n_reads, 0);
            VA_GRP(2) VA_GRP(2) VA_GRP(2) VA_GRP(2) VA_GRP(1) VA_GRP(1) VA_GRP(1) VA_GRP(1)
#undef VA_GRP
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    constexpr int RING = 5;
    const int nk = K / BK;   // even and >= 4 (checked by the host)
    auto advance = [&](int sl) { return sl + 1 == RING ? 0 : sl + 1; };
    if (PAIRED) {
        // Loads are issued for TWO consecutive stages at a time, row by row, so that the two 64-byte halves
        // of every 128-byte line are requested back to back (one L2 request instead of two).  Pairs are
        // issued at the top of odd stages, into the two slots consumed last; the pair issued at stage j is
        // needed at stage j+2, so one pair (64 KiB) is in flight while two stages compute, and there is one
        // barrier per two stages.
        auto stage2 = [&](int kt, int slot_a, int slot_b) {
            char* xa = smem + slot_a * STAGE_BYTES;
            char* xb = smem + slot_b * STAGE_BYTES;
            const int k0 = kt * BK;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                glds16(xsrc[i] + k0, xa + (i * NT + wave * 64) * 16);
                glds16(xsrc[i] + k0 + BK, xb + (i * NT + wave * 64) * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                glds16(wsrc[i] + k0, xa + OP_BYTES + (i * NT + wave * 64) * 16);
                glds16(wsrc[i] + k0 + BK, xb + OP_BYTES + (i * NT + wave * 64) * 16);
            }
        };
        stage2(0, 0, 1);
        stage2(2, 2, 3);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // pair (0,1) landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = frag_w(0, a);
#pragma unroll
        for (int b = 0; b < TM; ++b) xf[0][b] = frag_x(0, b);
        body(P0{}, Yes{}, 1);                                 // stage 0
        int slot = 1;                                         // slot of stage j
        for (int j = 1; j + 2 < nk; j += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // pair (j+1, j+2) landed
            __builtin_amdgcn_s_barrier();                      // ... everywhere; stages j-2, j-1 are consumed
            asm volatile("" ::: "memory");
            const int s1 = advance(slot), s2 = advance(s1), s3 = advance(s2), s4 = advance(s3);
            if (j + 4 < nk) stage2(j + 3, s3, s4);
            body(P1{}, Yes{}, s1);                            // stage j   (prefetches j+1)
            body(P0{}, Yes{}, s2);                            // stage j+1 (prefetches j+2)
            slot = s2;
        }
        body(P1{}, No{}, 0);                                  // stage nk-1
    } else {
        stage(0, 0);
        stage(1, 1);
        stage(2, 2);
        stage(3, 3);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // stage 0 landed (own loads) ...
        __builtin_amdgcn_s_barrier();                          // ... and everybody else's
        asm volatile("" ::: "memory");
    #pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = frag_w(0, a);
    #pragma unroll
        for (int b = 0; b < TM; ++b) xf[0][b] = frag_x(0, b);

        // At the top of stage kt the loads of stages kt+1..kt+3 are outstanding; vmcnt(8) retires kt+1 (needed
        // by this stage's prefetch reads) and leaves two stages in flight across the barrier.  The barrier also
        // tells that every wave has consumed stage kt-1, whose ring slot stage kt+4 then refills.
        int slot = 0;                                   // ring slot of stage kt
        auto sync_and_refill = [&](int kt) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 4 < nk) stage(kt + 4, slot == 0 ? RING - 1 : slot - 1);
        };
        int kt = 0;
        for (; kt + 4 < nk; kt += 2) {     // stages 0 .. nk-5 in pairs (each has three successors in flight)
            sync_and_refill(kt);
            body(P0{}, Yes{}, advance(slot));
            slot = advance(slot);
            sync_and_refill(kt + 1);
            body(P1{}, Yes{}, advance(slot));
            slot = advance(slot);
        }
        // tail: stages nk-4 .. nk-1; nothing is issued any more, so the counts step down
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        body(P0{}, Yes{}, advance(slot));
        slot = advance(slot);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        body(P1{}, Yes{}, advance(slot));
        slot = advance(slot);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        body(P0{}, Yes{}, advance(slot));
        body(P1{}, No{}, 0);
    }

    auto get = [&](int v, int b, int j) { return acc[v][b][j]; };
    if (m0 + wm * 64 + 64 <= M)
        store_tile_rows<EPI, TM, TN, true, true, false, true>(get, ep, M, m0 + wm * 64, n0 + wn * 128, lane);
    else
        store_tile_rows<EPI, TM, TN, false, true, false, true>(get, ep, M, m0 + wm * 64, n0 + wn * 128, lane);
}

// ------------------------------------------------------------------------------------------------
// "quad" loop: 256 x 256 tile, FOUR waves (2 x 2), each owning 128 x 128 of the output: 256 accumulator
// registers per lane (the accumulation half of the register file at one wave per SIMD).  Compared with the
// eight-wave ring a stage needs a third fewer LDS fragment reads per MFMA (16 reads feed 64 MFMAs), and every
// wave carries a long, uniform instruction stream into which the next stage's fragment reads AND the global
// loads of the stage four ahead are spread evenly (one load per eight MFMAs), so the load path never sees a
// burst and the matrix pipe is never left waiting behind a queue of load issues.
// Same LDS image as the ring (BK = 32 stages, 5 slots, chunk XOR, permuted weight rows); one barrier per stage:
// at the top of stage kt the loads of kt+2 and kt+3 may still be in flight.
// The kernel is persistent: one workgroup per CU takes tile `block id` and then draws further tiles from shared
// counters until none is left.  The LDS ring never drains between
// tiles -- the last four stages of a tile issue the first four stages of the next one, so the only bubble at a
// tile boundary is the epilogue itself, not a load round trip.
// ------------------------------------------------------------------------------------------------
template <int EPI, bool HAS_BIAS, int DBG = 0>
__global__ __launch_bounds__(256) void gemm_nt_quad_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                           int M, int N, int K, int ldx, int ldw, EpiArgs ep,
                                                           int* __restrict__ tk) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-descriptor type exists for the device target only; the host pass needs just the stub
    constexpr int BM = 256, BN = 256, BK = 32, NT = 256;
    constexpr int TM = 8, TN = 8;                 // per wave: 128 (m) x 128 (n)
    constexpr int OP_BYTES = 256 * BK * 2;        // 16 KiB per operand per stage
    constexpr int STAGE_BYTES = 2 * OP_BYTES;     // 32 KiB
    constexpr int RING = 5, LPS = 8;              // loads per thread and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // diagnostic instantiations: 5 stamps only; 8 stamps + epilogue without its stores; 9 stamps + tiles assigned
    // statically (no ticket atomics); 10 stamps + epilogue that stores without reading the accumulators
    constexpr bool STAMP = DBG == 5 || (DBG >= 8 && DBG <= 17);
    const uint64_t t_entry = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = N / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles = tiles_m * tiles_n;

    // Sources are addressed through buffer descriptors: one per operand and tile (scalar registers, based at the
    // tile's first row, sized to the rows that exist so that rows past M read as zero), a per-thread byte offset
    // that is the same for every tile (1024 chunks per operand per stage, 4 per thread: chunk q -> row q>>2,
    // slot q&3), and the stage's k offset as the scalar offset.
    constexpr bool ROPE = EPI == EPI_QKV_ROPE;
    // epilogues that need the registers of the next tile's first fragments: the last stage reads none ahead, they are
    // read after the epilogue (needs nk >= 6: checked by the host)
    constexpr bool LATE_FRAGS = ROPE || EPI == VA_EPI_MUL_AUX;
    struct Tile {
        int m0, n0;
        __amdgpu_buffer_rsrc_t x, w;
        int sw[4];        // ROPE: byte offset of the weight rows behind load i (the tile's rows come from four ranges)
    };
    auto set_tile = [&](int v, Tile& t) {
        int mt, nt;
        grouped_tile(xcd_remap(v, tiles), tiles_m, tiles_n, mt, nt);
        t.m0 = mt * BM;
        t.n0 = nt * BN;
        const int64_t xb = (int64_t)(M - t.m0) * ldx * 2, wb = (int64_t)(N - t.n0) * ldw * 2;
        t.x = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)t.m0 * ldx), (short)0,
                                                (int)(xb < 0x7fffffff ? xb : 0x7fffffff), 0x00020000);
        if (ROPE) {
            // weight rows of load i: LDS rows [64 i, 64 i + 64) = wave column half i >> 1, first columns (i even) or
            // partners (i odd); the descriptor spans the whole matrix
            const int64_t all = (int64_t)N * ldw * 2;
            t.w = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, (int)(all < 0x7fffffff ? all : 0x7fffffff), 0x00020000);
#pragma unroll
            for (int hw = 0; hw < 2; ++hw) {
                int a0, b0, seg, so;
                qkv_rope_cols(nt, hw, a0, b0, seg, so);
                t.sw[2 * hw] = a0 * ldw * 2;
                t.sw[2 * hw + 1] = b0 * ldw * 2;
            }
        } else {
            t.w = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)t.n0 * ldw), (short)0,
                                                    (int)(wb < 0x7fffffff ? wb : 0x7fffffff), 0x00020000);
        }
    };
    int xvo[4], wvo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * NT + tid, row = q >> 2, c = (q & 3) ^ (((row >> 3) & 1) * 3);
        xvo[i] = (row * ldx + c * 8) * 2;
        // ROPE: LDS row (v, r) of a 64-row range holds weight row 4 r + (v & 3) of that range (see store_tile_qkv_rope)
        wvo[i] = ROPE ? ((4 * (row & 15) + ((row >> 4) & 3)) * ldw + c * 8) * 2 : (wide_wrow<EPI, 128>(row) * ldw + c * 8) * 2;
    }
    // load i (0..7) of stage kt of tile t into ring slot `slot`: X chunks first, then W chunks
    auto load_one = [&](const Tile& t, int kt, int slot, int i) {
        char* base = smem + slot * STAGE_BYTES + (i >> 2) * OP_BYTES + ((i & 3) * NT + wave * 64) * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds((i >> 2) ? t.w : t.x, (lds_ptr_t)base, 16,
                                                 (i >> 2) ? wvo[i & 3] : xvo[i & 3],
                                                 kt * BK * 2 + ((ROPE && (i >> 2)) ? t.sw[i & 3] : 0), 0, 0);
    };

    const int frow = lane & 15;
    const int foff = frow * 64 + (((lane >> 4) ^ (((frow >> 3) & 1) * 3)) << 4);
    const int woff = wn * 128 * 64 + foff;
    const int xoff = wm * 128 * 64 + foff;
    bf16x8_t wf[2][TN];
    bf16x8_t xf[TM];
    auto frag_w = [&](int slot, int a) {
        return *(const bf16x8_t*)(smem + slot * STAGE_BYTES + OP_BYTES + woff + a * 16 * 64);
    };
    auto frag_x = [&](int slot, int b) { return *(const bf16x8_t*)(smem + slot * STAGE_BYTES + xoff + b * 16 * 64); };

    f32x4_t acc[2][4][TM];                        // [64-column half][16-column block][16-row block]

    // The accumulators are pinned to the accumulation registers by issuing the MFMAs as volatile asm with an
    // accumulation-register operand: left to the register allocator, a 256-register accumulator set is shuffled
    // between the two register files inside the loop.  Volatile asm statements keep their order, and the empty
    // memory-clobbering statements pin the loads between them, so the instruction stream is the written one:
    // per 16-row block  4 MFMAs | W fragment read | 4 MFMAs | X fragment read + one global load.
    // (w, x) in; for the 2-byte epilogues issued as A = x, B = w: a lane's registers are then four output ROWS of one
    // weight row (see wide_wrow)
    auto mf = [](f32x4_t& c, const bf16x8_t& w, const bf16x8_t& x) {
        if (w_is_a<EPI>()) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(x));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0" : "+a"(c) : "v"(w), "v"(x));
    };
    auto mf_first = [](f32x4_t& c, const bf16x8_t& w, const bf16x8_t& x) {   // first stage of a tile: C = A*B + 0
        if (w_is_a<EPI>()) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(w), "v"(x));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, 0" : "=a"(c) : "v"(w), "v"(x));
    };
#define VA_PIN asm volatile("" ::: "memory")
    // One stage: 64 MFMAs; the next stage's fragments are read into the registers just retired and the eight
    // loads of the stage four ahead (stage kt_load of tile `lt`) go out one per row block.
    u32x4_t sink[2][8];          // DBG 6 / 7 only (timing-only instantiations of the diagnostic library)
    auto lds_off_dbg = [&](int sl) { return (uint32_t)(uintptr_t)(lds_ptr_t)(smem + sl * STAGE_BYTES); };
    if (DBG == 6 || DBG == 7) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) sink[i][j] = (u32x4_t){0u, 0u, 0u, 0u};
    }
    // VA_EPI_MUL_AUX: the current tile's factor pieces (see store_tile_mul_aux), this lane's pointer to the first of them
    // and the number of the wave's 128 rows that exist
    int rope_code[4 * TM];               // EPI_QKV_ROPE: packed positions of the wave's rows (see store_tile_qkv_rope)
    const int32_t* pos_tile = nullptr;   // wave-uniform: the first of them
    u32x4_t auxr[4 * TM];
    const char* aux_tile = nullptr;      // wave-uniform: first of the wave's rows, first of its columns
    const uint32_t aux_loff = (uint32_t)((4 * (lane >> 4) * ep.ldc + (lane & 15) * 8) * 2);   // the lane's byte offset from there
    int aux_rows = 0;
    auto body = [&](auto parity, auto first, int next_slot, const Tile& lt, int kt_load, int load_slot, auto prefetch) {
        constexpr int P = decltype(parity)::value;
        constexpr bool FIRST = decltype(first)::value;
        // PREFETCH false (the last stage of a tile in the fused QKV + rotary build): the next tile's first fragments are
        // read after the epilogue instead -- that epilogue needs the 128 registers they would occupy across it
        constexpr bool PREFETCH = decltype(prefetch)::value;
        if constexpr (!PREFETCH && ROPE) {
#pragma unroll
            for (int i = 0; i < 4 * TM; ++i) {
                const int urow = (i >> 2) * 16 + (i & 3);
                rope_code[i] = urow + 4 * (lane >> 4) < aux_rows ? pos_tile[urow + 4 * (lane >> 4)] : 0;
            }
            asm volatile("" ::: "memory");
        }
        if constexpr (!PREFETCH && EPI == VA_EPI_MUL_AUX) {
#pragma unroll
            for (int i = 0; i < 4 * TM; ++i) {
                const int urow = (i >> 2) * 16 + (i & 3);          // uniform part of the row: scalar base + 32-bit lane offset
                auxr[i] = urow + 4 * (lane >> 4) < aux_rows ? *(const u32x4_t*)(aux_tile + (size_t)urow * ep.ldc * 2 + aux_loff)
                                                            : (u32x4_t){0u, 0u, 0u, 0u};
            }
            asm volatile("" ::: "memory");
        }
        auto block = [&](auto bc) {
            constexpr int b = decltype(bc)::value;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (FIRST) mf_first(acc[0][a][b], wf[P][a], xf[b]);
                else mf(acc[0][a][b], wf[P][a], xf[b]);
            }
            VA_PIN;
            if (DBG != 2 && PREFETCH) wf[P ^ 1][b] = frag_w(next_slot, b);
            VA_PIN;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (FIRST) mf_first(acc[1][a][b], wf[P][4 + a], xf[b]);
                else mf(acc[1][a][b], wf[P][4 + a], xf[b]);
            }
            VA_PIN;
            if (DBG != 2 && PREFETCH) xf[b] = frag_x(next_slot, b);
            if (DBG == 6 || DBG == 7) {
                // timing only: the stage's bytes fetched as whole 128-byte lines into registers (8 lanes per row) instead
                // of 64-byte half lines into LDS; DBG 7 also writes the registers of the previous stage's loads into the slot
                const int ld_ = b < 4 ? ldx : ldw;
                const bf16_t* base = (b < 4 ? X + (size_t)lt.m0 * ldx : W + (size_t)lt.n0 * ldw) +
                                     (size_t)((kt_load & 1) * 128) * ld_ + (kt_load >> 1) * 64;      // wave-uniform
                const int q = (b & 3) * NT + tid;
                const uint32_t voff = (uint32_t)(((q >> 3) * ld_ + (q & 7) * 8) * 2);
                if (DBG == 7) {
                    const uint32_t dst = lds_off_dbg(load_slot) + (uint32_t)(b * NT + tid) * 16;
                    asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(sink[P ^ 1][b]) : "memory");
                }
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sink[P][b]) : "v"(voff), "s"(base) : "memory");
            } else if (DBG != 1) load_one(lt, kt_load, load_slot, b);
            VA_PIN;
        };
        block(std::integral_constant<int, 0>{}); block(std::integral_constant<int, 1>{});
        block(std::integral_constant<int, 2>{}); block(std::integral_constant<int, 3>{});
        block(std::integral_constant<int, 4>{}); block(std::integral_constant<int, 5>{});
        block(std::integral_constant<int, 6>{}); block(std::integral_constant<int, 7>{});
    };
#undef VA_PIN
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    const int nk = K / BK;   // even and >= 4 (checked by the host)
    auto advance = [&](int sl) { return sl + 1 == RING ? 0 : sl + 1; };
    auto before = [&](int sl) { return sl == 0 ? RING - 1 : sl - 1; };

    // Tiles after a workgroup's first are handed out at run time: a workgroup that starts late or runs slowly
    // (its CU shared with another stream's kernel, a collective for one) simply takes fewer.  One counter per
    // XCD class (block id mod 8) keeps a tile id congruent to its workgroup's class, which is what xcd_remap's
    // locality rests on.  Thread 0 draws a ticket one tile ahead -- right before the epilogue, whose duration
    // covers the atomic's round trip -- and passes it to the other waves through the one ring slot that is idle
    // at a tile boundary (the slot of the tile's last stage; refilled only after the next stage-0 barrier).
    const int xclass = blockIdx.x & 7;
    const int vbase = (int)gridDim.x + ((xclass - (int)gridDim.x) & 7);    // first id of this class past the static round
    auto draw = [&]() { return tid == 0 ? atomicAdd(tk + xclass, 1) : 0; };
    auto lds_off = [&](int sl) { return (uint32_t)(uintptr_t)(lds_ptr_t)(smem + sl * STAGE_BYTES); };
    auto post = [&](int sl, int ticket) {
        if (tid == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_off(sl)), "v"(ticket) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto collect = [&](int sl) {      // after the barrier that follows post()
        int t;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lds_off(sl)) : "memory");
        return vbase + 8 * __builtin_amdgcn_readfirstlane(t);
    };

    int v = blockIdx.x;
    const int first_ticket = draw();
    Tile cur;
    set_tile(v, cur);
    auto aim_aux = [&](const Tile& t) {
        if constexpr (EPI == VA_EPI_MUL_AUX) {
            aux_tile = (const char*)(ep.aux + (size_t)(t.m0 + wm * 128) * ep.ldc + t.n0 + wn * 128);
            aux_rows = M - (t.m0 + wm * 128);
        }
        if constexpr (ROPE) {
            pos_tile = ep.rope_pos + (t.m0 + wm * 128);
            aux_rows = M - (t.m0 + wm * 128);
        }
    };
    aim_aux(cur);
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int i = 0; i < LPS; ++i) load_one(cur, st, st, i);
    post(RING - 1, first_ticket);
    wait_vm<3 * LPS>();                                   // stage 0 landed (own loads) ...
    __builtin_amdgcn_s_barrier();                          // ... and everybody else's
    asm volatile("" ::: "memory");
    int vn = collect(RING - 1);                            // the tile after this one (>= tiles: none)
    if (DBG == 9) vn = (int)blockIdx.x + (int)gridDim.x;
#pragma unroll
    for (int a = 0; a < TN; ++a) wf[0][a] = frag_w(0, a);
#pragma unroll
    for (int b = 0; b < TM; ++b) xf[b] = frag_x(0, b);

    // Top of a stage: the loads of the next three stages are outstanding; waiting down to two stages retires the
    // next one (read by this stage's prefetch) and the barrier tells that every wave has taken the current stage
    // into registers and is done with the previous one, whose slot the stage four ahead refills during this body.
    int slot = 0;                                          // ring slot of the current stage
    uint64_t t_sync = 0, t_body = 0, t_epi = 0, t_bound = 0, t_last = 0, t_first = 0, w_first = 0;   // stamped builds: cycles in waits+barriers / stage bodies / epilogues / tile hand-over
    // `relaxed`: the first three stages after the epilogue of a tile whose rows all exist.  Memory operations
    // retire in issue order and the loads of this tile's stages 1..3 went out before the epilogue's stores, so the
    // loads a stage needs are followed by two stages' loads AND those stores (32 instructions per wave for one
    // 2-byte output, 64 otherwise): that many may stay in flight, and the stores need not drain before the matrix
    // work resumes.  (An edge tile skips the stores of missing row blocks, so it gets the plain count.)
    // (the epilogues that read `aux` issue a load per store: 64 as well)
    constexpr int STORES = (EPI == VA_EPI_BF16 || EPI == VA_EPI_GELU_ONLY) ? 32 : 64;
    constexpr int RELAXED = 2 * LPS + STORES < 63 ? 2 * LPS + STORES : 63;
    auto sync = [&](bool relaxed) {
        if (STAMP) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_body += now - t_last; t_last = now; }
        if (relaxed) wait_vm<RELAXED>();
        else wait_vm<2 * LPS>();
        if (DBG != 3) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (STAMP) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_sync += now - t_last; t_last = now; }
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    if (STAMP) { t_last = t_first = __builtin_amdgcn_s_memtime(); w_first = __builtin_amdgcn_s_memrealtime(); }
    bool after_epilogue = false;
    for (;;) {
        // the tile whose first four stages ride in this tile's last four bodies; after the last tile the same
        // loads are issued once more (valid addresses, never read) so that the counted waits stay uniform
        Tile nxt;
        set_tile(vn < tiles ? vn : v, nxt);
        if (STAMP) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_bound += now - t_last; t_last = now; }
        // stage kt issues the loads of stage kt+4: of this tile, or of the next one from its stage 0 on
        auto stage_one = [&](auto parity, auto first, int kt, bool relaxed, auto prefetch) {
            const bool in_cur = kt + 4 < nk;
            sync(relaxed);
            body(parity, first, advance(slot), in_cur ? cur : nxt, in_cur ? kt + 4 : kt + 4 - nk, before(slot), prefetch);
            slot = advance(slot);
        };
        stage_one(P0{}, Yes{}, 0, after_epilogue, Yes{});
        stage_one(P1{}, No{}, 1, after_epilogue, Yes{});
        stage_one(P0{}, No{}, 2, after_epilogue, Yes{});
        stage_one(P1{}, No{}, 3, false, Yes{});
        if constexpr (LATE_FRAGS) {      // nk >= 6 (checked by the host): the last stage is peeled off and reads no fragments ahead
            for (int kt = 4; kt + 2 < nk; kt += 2) {
                stage_one(P0{}, No{}, kt, false, Yes{});
                stage_one(P1{}, No{}, kt + 1, false, Yes{});
            }
            stage_one(P0{}, No{}, nk - 2, false, Yes{});
            stage_one(P1{}, No{}, nk - 1, false, No{});
        } else {
            for (int kt = 4; kt < nk; kt += 2) {
                stage_one(P0{}, No{}, kt, false, Yes{});
                stage_one(P1{}, No{}, kt + 1, false, Yes{});
            }
        }

        // the asm MFMAs are opaque to the compiler's hazard tracking: let the last results retire before they are read
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        if (STAMP) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_body += now - t_last; t_last = now; }
        const bool more = vn < tiles;
        const int ticket = (more && DBG != 9) ? draw() : 0;
        const bool full = cur.m0 + wm * 128 + 128 <= M;
        if (DBG >= 11 && DBG <= 17) {
            // timing only: the same 32 KiB per wave stored from zeroed registers in other shapes -- 11: 8 rows x 128 B per
            // instruction, 12: 4 rows x 256 B, 13: as 10 with the non-temporal hint, 14: as 12 with it, 15: 16 rows x 64 B
            // but the two halves of a line 16 instructions apart, 16: 8-byte stores (16 rows x 32 B, 64 instructions)
            u32x4_t z = {0u, 0u, 0u, 0u};
            asm volatile("" : "+v"(z));
            bf16_t* base = (bf16_t*)ep.out + (size_t)(cur.m0 + wm * 128) * ep.ldc + cur.n0 + wn * 128;
            if (cur.m0 + wm * 128 + 128 <= M) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    if (DBG == 11) *(u32x4_t*)(base + (size_t)((i >> 1) * 8 + (lane >> 3)) * ep.ldc + (i & 1) * 64 + (lane & 7) * 8) = z;
                    if (DBG == 12) *(u32x4_t*)(base + (size_t)(i * 4 + (lane >> 4)) * ep.ldc + (lane & 15) * 8) = z;
                    if (DBG == 13) __builtin_nontemporal_store(z, (u32x4_t*)(base + (size_t)((i >> 2) * 16 + (lane & 15)) * ep.ldc + (i & 3) * 32 + (lane >> 4) * 8));
                    if (DBG == 14) __builtin_nontemporal_store(z, (u32x4_t*)(base + (size_t)(i * 4 + (lane >> 4)) * ep.ldc + (lane & 15) * 8));
                    if (DBG == 15) *(u32x4_t*)(base + (size_t)((i & 7) * 16 + (lane & 15)) * ep.ldc + (i >> 3) * 32 + (lane >> 4) * 8) = z;
                    if (DBG == 17) {      // 8-byte stores, neighbouring lanes on neighbouring bytes: 4 rows x 128 B per instruction, 64 of them
                        *(u32x2_t*)(base + (size_t)(i * 4 + (lane >> 4)) * ep.ldc + (lane & 15) * 4) = (u32x2_t){z[0], z[1]};
                        *(u32x2_t*)(base + (size_t)(i * 4 + (lane >> 4)) * ep.ldc + 64 + (lane & 15) * 4) = (u32x2_t){z[2], z[3]};
                    }
                    if (DBG == 16) {
                        *(u32x2_t*)(base + (size_t)((i >> 2) * 16 + (lane & 15)) * ep.ldc + (i & 3) * 32 + (lane >> 4) * 4) = (u32x2_t){z[0], z[1]};
                        *(u32x2_t*)(base + (size_t)((i >> 2) * 16 + (lane & 15)) * ep.ldc + (i & 3) * 32 + 16 + (lane >> 4) * 4) = (u32x2_t){z[2], z[3]};
                    }
                }
            }
        } else if (DBG == 8 || DBG == 10) {
            // timing only: the 2-byte epilogue's two halves apart -- 8 reads and converts every accumulator and drops the
            // result, 10 issues the 32 stores of a full tile from registers that hold zeros
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    u32x4_t lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
                    if (DBG == 8) {
                        lo = (u32x4_t){pack_bf16x2(acc[hh][0][b][0], acc[hh][0][b][1]), pack_bf16x2(acc[hh][0][b][2], acc[hh][0][b][3]),
                                       pack_bf16x2(acc[hh][1][b][0], acc[hh][1][b][1]), pack_bf16x2(acc[hh][1][b][2], acc[hh][1][b][3])};
                        hi = (u32x4_t){pack_bf16x2(acc[hh][2][b][0], acc[hh][2][b][1]), pack_bf16x2(acc[hh][2][b][2], acc[hh][2][b][3]),
                                       pack_bf16x2(acc[hh][3][b][0], acc[hh][3][b][1]), pack_bf16x2(acc[hh][3][b][2], acc[hh][3][b][3])};
                        asm volatile("" ::"v"(lo), "v"(hi));
                    } else {
                        asm volatile("" : "+v"(lo), "+v"(hi));
                        const int m = cur.m0 + wm * 128 + b * 16 + (lane & 15);
                        if (m < M) {
                            bf16_t* o = (bf16_t*)ep.out + (size_t)m * ep.ldc + cur.n0 + wn * 128 + hh * 64 + (lane >> 4) * 8;
                            *(u32x4_t*)o = lo;
                            *(u32x4_t*)(o + 32) = hi;
                        }
                    }
                }
        } else
        {
            auto get = [&](int v, int b, int j) { return acc[v >> 2][v & 3][b][j]; };
            // accumulators are copied out where they are used, by statements the compiler keeps in order: left to
            // itself it copies all 256 out first and spills forty of them around a register-hungry epilogue
            auto get_here = [&](int v, int b, int j) {
                float x;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[v >> 2][v & 3][b][j]));
                return x;
            };
            if constexpr (EPI == VA_EPI_MUL_AUX) {
                if (full) store_tile_mul_aux<TM, true>(get_here, auxr, ep, M, cur.m0 + wm * 128, cur.n0 + wn * 128, lane);
                else store_tile_mul_aux<TM, false>(get_here, auxr, ep, M, cur.m0 + wm * 128, cur.n0 + wn * 128, lane);
            } else if constexpr (ROPE) {
                int a0, b0, seg, so;
                qkv_rope_cols(cur.n0 / BN, wn, a0, b0, seg, so);
                if (full) store_tile_qkv_rope<TM, true>(get_here, rope_code, ep, M, cur.m0 + wm * 128, a0, b0, seg, so, lane);
                else store_tile_qkv_rope<TM, false>(get_here, rope_code, ep, M, cur.m0 + wm * 128, a0, b0, seg, so, lane);
            } else {
                if (full) store_tile_rows<EPI, TM, 8, true, HAS_BIAS, true, false>(get, ep, M, cur.m0 + wm * 128, cur.n0 + wn * 128, lane);
                else store_tile_rows<EPI, TM, 8, false, HAS_BIAS, false, false>(get, ep, M, cur.m0 + wm * 128, cur.n0 + wn * 128, lane);
            }
        }
        if (STAMP) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_epi += now - t_last; t_last = now; }
        if (!more) break;
        if (DBG == 9) {       // static hand-out: the next tile of this workgroup's class
            __builtin_amdgcn_s_barrier();
            v = vn;
            vn = vn + (int)gridDim.x;
        } else {
        post(before(slot), ticket);
        __builtin_amdgcn_s_barrier();
        v = vn;
        vn = collect(before(slot));
        }
        cur = nxt;
        aim_aux(cur);
        after_epilogue = full;
        if constexpr (LATE_FRAGS) {      // the fragments of the next tile's first stage (landed since the last stage's wait + barrier)
#pragma unroll
            for (int a = 0; a < TN; ++a) wf[0][a] = frag_w(slot, a);
#pragma unroll
            for (int b = 0; b < TM; ++b) xf[b] = frag_x(slot, b);
        }
    }
    wait_vm0();   // the trailing loads must not land in LDS after the workgroup has gone
    if (DBG == 6 || DBG == 7) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(sink[i][j]));
    }
    // the last workgroup out rewinds the counters for the next launch that is given this set
    if (tid == 0 && atomicAdd(tk + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tk[i] = 0;
    }
    if (STAMP && lane == 0) {     // diagnostic instantiations only: the stamps leave through a buffer of their own
        float* o = (float*)ep.out2 + ((size_t)blockIdx.x * 4 + wave) * 8;
        const uint64_t now = __builtin_amdgcn_s_memtime(), wall = __builtin_amdgcn_s_memrealtime();
        o[0] = (float)t_sync; o[1] = (float)t_body; o[2] = (float)(now - t_last); o[3] = (float)t_epi;
        o[4] = (float)(now - t_first);          // shader cycles from the first stage to here
        o[5] = (float)(wall - w_first);         // the same span in ticks of the constant 100 MHz clock
        o[6] = (float)(t_first - t_entry);      // prologue: kernel entry to the first stage (cycles)
        o[7] = (float)t_bound;                  // tile hand-over: ticket exchange, barrier, next tile's descriptors
    }
#endif
}

int ensure_lds(const void* kern, int bytes, bool* flags) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !flags[dev]) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            va_set_error("va_gemm_nt: cannot reserve %d bytes of LDS", bytes);
            return VA_ERR_LAUNCH;
        }
        flags[dev] = true;
    }
    return VA_OK;
}

template <int BM, int BN, int WM, int WN, int EPI>
int launch_pair(const bf16_t* X, const bf16_t* W, int M, int N, int K, int ldx, int ldw, const EpiArgs& ep,
                hipStream_t stream) {
    constexpr int LDS = 2 * (BM + BN) * 64 * 2;
    auto kern = gemm_nt_pair_kernel<BM, BN, WM, WN, EPI>;
    static bool flags[64] = {};
    if (int e = ensure_lds((const void*)kern, LDS, flags)) return e;
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), LDS, stream, X, W, M, N, K, ldx, ldw, ep);
    VA_CHECK_LAUNCH("va_gemm_nt");
    return VA_OK;
}

template <int EPI, bool PAIRED>
int launch_ring(const bf16_t* X, const bf16_t* W, int M, int N, int K, int ldx, int ldw, const EpiArgs& ep,
                hipStream_t stream) {
    constexpr int LDS = 5 * 2 * 256 * 32 * 2;     // the whole 160 KiB of a CU
    auto kern = gemm_nt_ring_kernel<EPI, PAIRED>;
    static bool flags[64] = {};
    if (int e = ensure_lds((const void*)kern, LDS, flags)) return e;
    const int tiles = ((M + 255) / 256) * (N / 256);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, stream, X, W, M, N, K, ldx, ldw, ep);
    VA_CHECK_LAUNCH("va_gemm_nt");
    return VA_OK;
}

// Ticket counters of the persistent kernel: sets of 16 ints (8 per-XCD-class counters, one exit counter), all
// zero between launches (the kernel rewinds its set on the way out).  A set belongs to a (device, stream) pair:
// launches on one stream run in order and may share a set, launches on different streams may run at the same
// time and must not.  A device serves QUAD_SETS distinct streams this way; a launch on a further stream gets
// no set (*none = true) and the caller uses the non-persistent kernel instead.  (A captured graph holds the set
// of the stream it was captured on: replaying it on several streams at once is not supported.)
constexpr int QUAD_SETS = 64;
__device__ int quad_tickets[QUAD_SETS * 16];
int* quad_ticket_set(int dev, hipStream_t stream, bool* none) {
    struct PerDevice {
        int* base = nullptr;
        int used = 0;
        hipStream_t owner[QUAD_SETS];
    };
    static PerDevice table[64];
    static std::mutex lock;
    *none = false;
    if (dev < 0 || dev >= 64) { va_set_error("va_gemm_nt: device index %d out of range", dev); return nullptr; }
    std::lock_guard<std::mutex> guard(lock);
    PerDevice& t = table[dev];
    if (!t.base) {
        int* p = nullptr;
        if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(quad_tickets)) != hipSuccess || !p) {
            va_set_error("va_gemm_nt: cannot resolve the ticket counters");
            return nullptr;
        }
        t.base = p;
    }
    for (int i = 0; i < t.used; ++i)
        if (t.owner[i] == stream) return t.base + i * 16;
    if (t.used == QUAD_SETS) { *none = true; return nullptr; }
    t.owner[t.used] = stream;
    return t.base + (t.used++) * 16;
}

// DBG != 0 and grid_limit are for the diagnostic library built from tools/gemm_nt_debug.hip; this file instantiates
// DBG = 0 only.
template <int EPI, bool HAS_BIAS, int DBG = 0>
int launch_quad(const bf16_t* X, const bf16_t* W, int M, int N, int K, int ldx, int ldw, const EpiArgs& ep,
                hipStream_t stream, int grid_limit = 0) {
    constexpr int LDS = 5 * 2 * 256 * 32 * 2;
    auto kern = gemm_nt_quad_kernel<EPI, HAS_BIAS, DBG>;
    static bool flags[64] = {};
    if (int e = ensure_lds((const void*)kern, LDS, flags)) return e;
    const int tiles = ((M + 255) / 256) * (N / 256);
    static int cus[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && cus[dev] == 0) {
        if (hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus[dev] <= 0)
            cus[dev] = 256;
    }
    int grid = tiles < cus[dev & 63] ? tiles : cus[dev & 63];
    if (grid_limit > 0 && grid_limit < grid) grid = grid_limit;
    bool none = false;
    int* tk = quad_ticket_set(dev, stream, &none);
    if (none) {
        if constexpr (EPI == EPI_QKV_ROPE) {      // only the four-wave kernel has this epilogue: the caller takes the two-pass route
            va_set_error("va_gemm_nt_qkv_rope: no ticket set is left for this stream");
            return VA_ERR_UNSUPPORTED;
        } else {
            return launch_ring<EPI, true>(X, W, M, N, K, ldx, ldw, ep, stream);
        }
    }
    if (!tk) return VA_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, X, W, M, N, K, ldx, ldw, ep, tk);
    VA_CHECK_LAUNCH("va_gemm_nt");
    return VA_OK;
}

template <int EPI>
int launch_epi(int tile, const bf16_t* X, const bf16_t* W, int M, int N, int K, int ldx, int ldw, const EpiArgs& ep,
               hipStream_t stream) {
    if (tile == 256) return launch_ring<EPI, true>(X, W, M, N, K, ldx, ldw, ep, stream);
    if (tile == 2256) return launch_ring<EPI, false>(X, W, M, N, K, ldx, ldw, ep, stream);
    if (tile == 4256) {
        // The bias adds are compiled out only for the plain 2-byte output without a bias (the f32 epilogue keeps
        // them: its bias-free build schedules worse and measures slower than adding zeros).
        if (EPI == VA_EPI_BF16 && ep.bias == nullptr) return launch_quad<EPI, EPI != VA_EPI_BF16>(X, W, M, N, K, ldx, ldw, ep, stream);
        return launch_quad<EPI, !epi_reads_aux(EPI)>(X, W, M, N, K, ldx, ldw, ep, stream);
    }
    if (tile == 1256) return launch_pair<256, 256, 2, 4, EPI>(X, W, M, N, K, ldx, ldw, ep, stream);
    return launch_pair<128, 128, 2, 2, EPI>(X, W, M, N, K, ldx, ldw, ep, stream);
}

}  // namespace

static int gemm_nt_entry(const void* x, const void* w, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                         int epilogue, void* out, void* out2, const float* bias, const float* resid,
                         const void* aux, int64_t ldc, int tile, uint32_t drop_key, uint32_t drop_thresh16,
                         hipStream_t stream) {
    VA_CHECK_ARG(x && w && out, "va_gemm_nt: null operand");
    VA_CHECK_ARG(M > 0 && N > 0 && K > 0, "va_gemm_nt: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N,
                 (long long)K);
    VA_CHECK_ARG(K % 64 == 0, "va_gemm_nt: K=%lld must be a multiple of 64", (long long)K);
    VA_CHECK_ARG(N % 128 == 0, "va_gemm_nt: N=%lld must be a multiple of 128", (long long)N);
    VA_CHECK_ARG(ldx >= K && ldw >= K && ldc >= N, "va_gemm_nt: leading dimension smaller than the row length");
    VA_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "va_gemm_nt: leading dimensions must keep 16-byte alignment");
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) % 16 == 0, "va_gemm_nt: operands must be 16-byte aligned");
    VA_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "va_gemm_nt: problem too large");
    if (tile == 0) {
        // Large problems: the four-wave persistent kernel for the plain 2-byte and the f32 outputs and for the forward
        // GELU epilogues (with one wave per SIMD nothing overlaps their transcendental math, but its main loop and
        // row-contiguous stores still win: fc1 + bias + GELU 0.576 against 0.597 ms, GELU only 0.537 against 0.566 at
        // M = 65536).  The GELU-derivative epilogue (a load per element on top of the math) measures equal on both
        // and stays on the eight-wave ring, as do the dropout epilogues (the four-wave kernel has none).
        // ... and only when the 256-wide tiles fill most of the chip: a problem with few of them (small batches)
        // spreads over four times as many workgroups with the 128-wide tile.
        const int64_t tiles256 = ((M + 255) / 256) * (N / 256);
        const bool big = N % 256 == 0 && M >= 2048 && K >= 128 && tiles256 >= 150;
        const bool quad = epilogue == VA_EPI_BF16 ||
                          ((epilogue == VA_EPI_F32 || epilogue == VA_EPI_BIAS_GELU || epilogue == VA_EPI_GELU_ONLY ||
                            epilogue == VA_EPI_BIAS_GELU_GRAD || (epilogue == VA_EPI_MUL_AUX && K >= 192)) && drop_thresh16 == 0);
        tile = !big ? 128 : quad ? 4256 : 256;
    }
    VA_CHECK_ARG(tile == 128 || ((tile == 256 || tile == 1256 || tile == 2256 || tile == 4256) && N % 256 == 0 && K >= 128),
                 "va_gemm_nt: tile %d not usable with N=%lld K=%lld", tile, (long long)N, (long long)K);
    VA_CHECK_ARG(drop_thresh16 == 0 || tile < 4256, "va_gemm_nt_dropout: the four-wave kernel (tile %d) has no dropout epilogue", tile);
    VA_CHECK_ARG(!(tile == 4256 && epilogue == VA_EPI_MUL_AUX && K < 192), "va_gemm_nt: the four-wave kernel's multiply epilogue needs K >= 192");
    // non-temporal stores for outputs beyond the last-level cache's reach (256 MB; VA_GEMM_NT_MIN_MB overrides, 0 = always)
    static const long nt_min_mb = [] {
        const char* e = getenv("VA_GEMM_NT_MIN_MB");
        return e ? atol(e) : 300L;
    }();
    const int64_t out_bytes = M * N * (epilogue == VA_EPI_F32 ? 4 : 2) * (epi_two_outputs(epilogue) ? 2 : 1);
    EpiArgs ep{out, out2, bias, resid, (const bf16_t*)aux, (int)ldc, drop_key, drop_thresh16 << 16,
               va_drop_scale(drop_thresh16), out_bytes >= nt_min_mb * (1ll << 20) ? 1 : 0, nullptr, nullptr, 0, 0, 0};
    const bf16_t* X = (const bf16_t*)x;
    const bf16_t* W = (const bf16_t*)w;
    switch (epilogue) {
        case VA_EPI_BF16:
            return launch_epi<VA_EPI_BF16>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_BIAS_GELU:
            VA_CHECK_ARG(out2 && bias, "va_gemm_nt: bias+GELU epilogue needs out2 and bias");
            return launch_epi<VA_EPI_BIAS_GELU>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_GELU_ONLY:
            VA_CHECK_ARG(bias, "va_gemm_nt: the GELU epilogue needs a bias");
            return launch_epi<VA_EPI_GELU_ONLY>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_F32:
            return launch_epi<VA_EPI_F32>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_GELU_BWD:
            VA_CHECK_ARG(aux, "va_gemm_nt: GELU-backward epilogue needs the saved pre-activation");
            return launch_epi<VA_EPI_GELU_BWD>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_BIAS_GELU_GRAD:
            VA_CHECK_ARG(out2 && bias, "va_gemm_nt: bias+GELU epilogue needs out2 and bias");
            return launch_epi<VA_EPI_BIAS_GELU_GRAD>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_MUL_AUX:
            VA_CHECK_ARG(aux, "va_gemm_nt: the multiply epilogue needs its factor matrix");
            return launch_epi<VA_EPI_MUL_AUX>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        default:
            va_set_error("va_gemm_nt: unknown epilogue %d", epilogue);
            return VA_ERR_INVALID;
    }
}

extern "C" int va_gemm_nt(const void* x, const void* w, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                          int epilogue, void* out, void* out2, const float* bias, const float* resid,
                          const void* aux, int64_t ldc, int tile, hipStream_t stream) {
    return gemm_nt_entry(x, w, M, N, K, ldx, ldw, epilogue, out, out2, bias, resid, aux, ldc, tile, 0u, 0u, stream);
}

extern "C" int va_gemm_nt_dropout(const void* x, const void* w, int64_t M, int64_t N, int64_t K, int64_t ldx,
                                  int64_t ldw, int epilogue, void* out, void* out2, const float* bias,
                                  const float* resid, const void* aux, int64_t ldc, int tile, uint32_t key, float p,
                                  hipStream_t stream) {
    VA_CHECK_ARG(p >= 0.f && p < 1.f, "va_gemm_nt_dropout: dropout probability %g outside [0, 1)", (double)p);
    const uint32_t thresh = va_drop_thresh16(p);
    VA_CHECK_ARG(thresh < 65536u, "va_gemm_nt_dropout: dropout probability %g rounds to 1", (double)p);
    if (thresh != 0) {
        VA_CHECK_ARG(epilogue != VA_EPI_BF16, "va_gemm_nt_dropout: the plain bf16 epilogue has no dropout site");
        VA_CHECK_ARG(ldc == N, "va_gemm_nt_dropout: the output must be contiguous (ldc == N): mask positions are m * N + n");
        VA_CHECK_ARG(M * N < (1ll << 32), "va_gemm_nt_dropout: more than 2^32 output elements");
    }
    return gemm_nt_entry(x, w, M, N, K, ldx, ldw, epilogue, out, out2, bias, resid, aux, ldc, tile, key, thresh, stream);
}

extern "C" int va_gemm_nt_qkv_rope(const void* x, const void* w, int64_t M, int64_t K, int64_t ldx, int64_t ldw, void* out,
                                   int64_t ldc, const float* tables, const int32_t* pos_code, int64_t Tp, int64_t Hp, int64_t Wp,
                                   hipStream_t stream) {
    constexpr int64_t N = 3072;
    VA_CHECK_ARG(x && w && out && tables && pos_code, "va_gemm_nt_qkv_rope: null operand");
    VA_CHECK_ARG(M >= 256 && M < (1ll << 31) && K >= 192 && K % 64 == 0 && K < (1ll << 20),
                 "va_gemm_nt_qkv_rope: M=%lld (>= 256) or K=%lld (multiple of 64, >= 192) out of range", (long long)M, (long long)K);
    VA_CHECK_ARG(ldx >= K && ldw >= K && ldc >= N && ldx % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0,
                 "va_gemm_nt_qkv_rope: bad leading dimension");
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)tables) % 16 == 0 && (uintptr_t)pos_code % 4 == 0,
                 "va_gemm_nt_qkv_rope: misaligned operand");
    VA_CHECK_ARG(Tp > 0 && Hp > 0 && Wp > 0 && Tp < 1024 && Hp < 1024 && Wp < 1024, "va_gemm_nt_qkv_rope: grid out of range");
    VA_CHECK_ARG(N * ldw * 2 < (1ll << 31), "va_gemm_nt_qkv_rope: weight matrix too large for one descriptor");
    static const long nt_min_mb = [] {
        const char* e = getenv("VA_GEMM_NT_MIN_MB");
        return e ? atol(e) : 300L;
    }();
    EpiArgs ep{out, nullptr, nullptr, nullptr, nullptr, (int)ldc, 0u, 0u, 1.f, M * N * 2 >= nt_min_mb * (1ll << 20) ? 1 : 0,
               tables, pos_code, (int)Tp, (int)Hp, (int)Wp};
    return launch_quad<EPI_QKV_ROPE, false>((const bf16_t*)x, (const bf16_t*)w, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
}
