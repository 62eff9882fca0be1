// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libvit_amd.
// Wavefront = 64 lanes.  bf16 data is carried as raw 16-bit words (`bf16_t`).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vit_amd.h"

typedef uint16_t bf16_t;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;     // one ds_read_b64_tr_b16 result
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define VA_DEVINL __device__ __forceinline__

// ---- bf16 <-> f32 -------------------------------------------------------
VA_DEVINL float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Round-to-nearest-even through the compiler's own conversion (keeps NaN a NaN).
VA_DEVINL bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// Two f32 -> one dword of two bf16 (lo in bits 0-15): a single v_cvt_pk_bf16_f32.
VA_DEVINL uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// ---- LDS / global helpers ----------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Asynchronous 16-byte-per-lane global -> LDS copy.  `lds_wave_base` must be wave-uniform; lane l
// lands at lds_wave_base + 16*l.  The global source address is per lane.
VA_DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gsrc, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// The 4-byte form: lane l lands at lds_wave_base + 4*l.
VA_DEVINL void glds4(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gsrc, (lds_ptr_t)lds_wave_base, 4, 0, 0);
}

// Transposed LDS read: within each group of 16 lanes a 4-row x 16-column block of 16-bit words is
// delivered column-major (lane i of the group receives column i, row q in element q).  Lane 4q+p of the
// group supplies the address of row q, columns 4p..4p+3 (8-byte aligned).  EXEC must be all ones.
VA_DEVINL s16x4_t lds_read_tr16(const void* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)lds_addr);
}

VA_DEVINL bf16x8_t join_tr(s16x4_t lo, s16x4_t hi) {
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

VA_DEVINL f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
VA_DEVINL f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

VA_DEVINL void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
VA_DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Row of accumulator register i in lane-half h for a 32x32 MFMA result.
VA_DEVINL int acc32_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// Exact (erf-form) GELU and its derivative.  erf comes from the Abramowitz-Stegun 7.1.26 rational
// form (|error| <= 1.5e-7, below f32 rounding of the result for the bf16 outputs produced here): one
// v_rcp_f32, one v_exp_f32 and five FMAs instead of the branchy library erff, because the GEMM epilogues
// that call this run on a CU whose matrix pipe idles meanwhile.  exp(-x^2/2) is shared between the
// normal CDF and the normal PDF of the derivative.
// half_erfc(|x|/sqrt2) and exp(-x^2/2) with the constants folded: one fma into the rcp, x^2 and one multiply into
// the exp2, four fmas (the 0.5 is in the coefficients), two multiplies.
VA_DEVINL void gelu_parts(float x, float& half_erfc, float& gauss) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));
    gauss = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.4426950408889634f));  // exp(-x^2/2)
    float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    poly = fmaf(poly, t, 0.5f * 1.421413741f);
    poly = fmaf(poly, t, 0.5f * -0.284496736f);
    poly = fmaf(poly, t, 0.5f * 0.254829592f);
    half_erfc = (poly * t) * gauss;                                            // 0.5 * erfc(|x|/sqrt2)
}
// x * Phi(x) = max(x, 0) - |x| * half_erfc(|x|): no compare / select
VA_DEVINL float gelu_erf(float x) {
    float h, g;
    gelu_parts(x, h, g);
    return fmaf(-fabsf(x), h, fmaxf(x, 0.f));
}
// Phi(x) + x * phi(x)
VA_DEVINL float gelu_erf_grad(float x) {
    float h, g;
    gelu_parts(x, h, g);
    const float cdf = x >= 0.f ? 1.0f - h : h;
    return fmaf(x * 0.39894228040143268f, g, cdf);
}

// Both at once (the forward epilogue that keeps the derivative, not the pre-activation, for the backward pass)
VA_DEVINL void gelu_erf_both(float x, float& gelu, float& grad) {
    float h, g;
    gelu_parts(x, h, g);
    gelu = fmaf(-fabsf(x), h, fmaxf(x, 0.f));
    const float cdf = x >= 0.f ? 1.0f - h : h;
    grad = fmaf(x * 0.39894228040143268f, g, cdf);
}

// Map a launch-order block id onto a work id so that the blocks that share an XCD (ids equal mod 8)
// receive one contiguous run of work ids: neighbouring tiles then reuse each other's operand panels
// in that XCD's L2.  Bijective for every block count.
VA_DEVINL int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// Work id -> (m-tile, n-tile) of a GEMM.  Ids run over groups of G m-panels, m fastest inside a group: the ~32
// workgroups an XCD runs at a time (consecutive ids after xcd_remap) then cover G activation panels x 32/G weight
// panels instead of 2 x 16: fewer distinct operand panels go through that XCD's L2 per round.  G = 4 measured best
// on the whole training step (2, 3, 6, 8 within 1 %; 16 -- activation panels alone overflow the L2 -- 5 % worse).
VA_DEVINL void grouped_tile(int id, int tiles_m, int tiles_n, int& mt, int& nt) {
    constexpr int G = 4;
    const int per = G * tiles_n;
    const int grp = id / per, within = id - grp * per;
    const int left = tiles_m - grp * G;
    const int gm = left < G ? left : G;
    nt = within / gm;
    mt = grp * G + within % gm;
}

// ---- dropout masks ------------------------------------------------------
// Counter-based keep/drop decisions.  Elements 2*pair and 2*pair+1 share one 32-bit word
//     h = mix(pair * DROP_WEYL + key),   mix(w): w ^= w >> 16; w *= DROP_MUL; w ^= w >> 15
// and are kept when their 16-bit field (low half for the even element, high half for the odd one) is
// >= thresh16 = round(p * 65536).  Any kernel can therefore regenerate the decision of any element from
// (key, element index) alone, and a kernel that walks consecutive pairs advances the Weyl term by additions.
// The compares work on the fields in place: with thi = thresh16 << 16, "high field >= thresh16" is h >= thi and
// "low field >= thresh16" is (h << 16) >= thi.
constexpr uint32_t DROP_WEYL = 0x9E3779B1u, DROP_MUL = 0x21F0AAADu, DROP_HEAD = 0x85EBCA6Bu;
VA_DEVINL uint32_t drop_weyl(uint32_t key, uint32_t pair) { return pair * DROP_WEYL + key; }
VA_DEVINL uint32_t drop_mix(uint32_t w) {
    w ^= w >> 16;
    w *= DROP_MUL;
    w ^= w >> 15;
    return w;
}
VA_DEVINL bool drop_keep_even(uint32_t h, uint32_t thi) { return (h << 16) >= thi; }
VA_DEVINL bool drop_keep_odd(uint32_t h, uint32_t thi) { return h >= thi; }

// ---- host side ------------------------------------------------------------
#define VA_OK 0
#define VA_ERR_INVALID VA_ERR_INVALID_ARG
#define VA_ERR_LAUNCH VA_ERR_LAUNCH_FAILED
#define VA_ERR_UNSUPPORTED VA_ERR_UNSUPPORTED_SHAPE
#ifdef __cplusplus
extern "C" void va_set_error(const char* fmt, ...);
#endif
// p -> 16-bit threshold, and the rescale 1 / (1 - thresh/65536) of the kept elements
static inline uint32_t va_drop_thresh16(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
static inline float va_drop_scale(uint32_t thresh16) { return 65536.0f / (float)(65536u - thresh16); }

#define VA_CHECK_ARG(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            va_set_error(__VA_ARGS__);         \
            return VA_ERR_INVALID;             \
        }                                      \
    } while (0)

#define VA_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            va_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return VA_ERR_LAUNCH;                                                 \
        }                                                                         \
    } while (0)

