// bf16 MFMA GEMM, both operands K-contiguous:  C[M,N] = X[M,K] * W[N,K]^T  (+ fused epilogue).
//
// This one kernel family carries every dense projection of the encoder: patch embedding (after the
// tubelet gather), QKV, attention out-projection, MLP fc1 / fc2, and all data-gradient GEMMs (which
// run against a transposed bf16 copy of the weight so that they have the same K-contiguous shape).
//
// Structure (gfx950):
//   * workgroup tile BM x BN x 64, double-buffered in LDS, staged with 16-byte direct-to-LDS loads
//     (global_load_lds_dwordx4).  The LDS image is lane-linear; bank conflicts of the ds_read_b128
//     fragment reads are removed by permuting the per-lane SOURCE address (16-byte chunk index XOR
//     ((row>>1)&7)) and applying the same XOR when reading.
//   * each wave owns a (BM/WM) x (BN/WN) sub-tile as 16x16x32 bf16 MFMAs.  The weight tile is the MFMA
//     A operand and the activation tile the B operand, so an accumulator lane holds 4 consecutive
//     output columns of one output row: stores are 8-byte (bf16) / 16-byte (f32) vectors.
//   * one barrier per K-tile: the next tile's loads are issued before the current tile's MFMAs.
//   * launch-order block ids are remapped so that the blocks sharing an XCD walk neighbouring tiles.
#include "common.h"

namespace {

struct EpiArgs {
    void* out;            // bf16 or f32 [M, ldc]
    void* out2;           // bf16 [M, ldc]: GELU output (VA_EPI_BIAS_GELU)
    const float* bias;    // [N] or null
    const float* resid;   // f32 [M, ldc] or null (VA_EPI_F32)
    const bf16_t* aux;    // bf16 [M, ldc]: saved pre-activation (VA_EPI_GELU_BWD)
    int ldc;
};

constexpr int BK = 64;

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(const bf16_t* __restrict__ X,
                                                               const bf16_t* __restrict__ W, int M, int N, int K,
                                                               int ldx, int ldw, EpiArgs ep) {
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

    // per-thread source pointers (row and swizzled chunk are fixed; only k advances)
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

    // fragment read offsets inside a tile (bytes), for k-step 0; k-step 1 flips chunk bit 2
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
        for (int kk = 0; kk < 2; ++kk) {
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

    // ---- epilogue: lane holds rows n = nb..nb+3 of D for column m -> 4 consecutive output columns
    const int ldc = ep.ldc;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * (BM / WM) + b * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int nb = n0 + wn * (BN / WN) + a * 16 + (lane >> 4) * 4;
            const size_t o = (size_t)m * ldc + nb;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            if (EPI != VA_EPI_GELU_BWD && ep.bias != nullptr) {
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
                u32x2_t gk = {pack_bf16x2(gelu_erf(v[0]), gelu_erf(v[1])), pack_bf16x2(gelu_erf(v[2]), gelu_erf(v[3]))};
                *(u32x2_t*)((bf16_t*)ep.out2 + o) = gk;
            } else if (EPI == VA_EPI_F32) {
                if (ep.resid != nullptr) {
                    const f32x4_t rv = *(const f32x4_t*)(ep.resid + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += rv[j];
                }
                *(f32x4_t*)((float*)ep.out + o) = (f32x4_t){v[0], v[1], v[2], v[3]};
            } else {  // VA_EPI_GELU_BWD
                const u32x2_t pre = *(const u32x2_t*)(ep.aux + o);
                const float p0 = __uint_as_float(pre[0] << 16), p1 = __uint_as_float(pre[0] & 0xffff0000u);
                const float p2 = __uint_as_float(pre[1] << 16), p3 = __uint_as_float(pre[1] & 0xffff0000u);
                u32x2_t pk = {pack_bf16x2(v[0] * gelu_erf_grad(p0), v[1] * gelu_erf_grad(p1)),
                              pack_bf16x2(v[2] * gelu_erf_grad(p2), v[3] * gelu_erf_grad(p3))};
                *(u32x2_t*)((bf16_t*)ep.out + o) = pk;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
int launch_cfg(const bf16_t* X, const bf16_t* W, int M, int N, int K, int ldx, int ldw, const EpiArgs& ep,
               hipStream_t stream) {
    constexpr int LDS = 2 * (BM + BN) * BK * 2;
    auto kern = gemm_nt_kernel<BM, BN, WM, WN, EPI>;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            va_set_error("va_gemm_nt: cannot reserve %d bytes of LDS", LDS);
            return VA_ERR_LAUNCH;
        }
        attr_set[dev] = true;
    }
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), LDS, stream, X, W, M, N, K, ldx, ldw, ep);
    VA_CHECK_LAUNCH("va_gemm_nt");
    return VA_OK;
}

template <int EPI>
int launch_epi(int tile, const bf16_t* X, const bf16_t* W, int M, int N, int K, int ldx, int ldw, const EpiArgs& ep,
               hipStream_t stream) {
    if (tile == 256) return launch_cfg<256, 256, 2, 4, EPI>(X, W, M, N, K, ldx, ldw, ep, stream);
    return launch_cfg<128, 128, 2, 2, EPI>(X, W, M, N, K, ldx, ldw, ep, stream);
}

}  // namespace

extern "C" int va_gemm_nt(const void* x, const void* w, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                          int epilogue, void* out, void* out2, const float* bias, const float* resid,
                          const void* aux, int64_t ldc, int tile, hipStream_t stream) {
    VA_CHECK_ARG(x && w && out, "va_gemm_nt: null operand");
    VA_CHECK_ARG(M > 0 && N > 0 && K > 0, "va_gemm_nt: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N,
                 (long long)K);
    VA_CHECK_ARG(K % 64 == 0, "va_gemm_nt: K=%lld must be a multiple of 64", (long long)K);
    VA_CHECK_ARG(N % 128 == 0, "va_gemm_nt: N=%lld must be a multiple of 128", (long long)N);
    VA_CHECK_ARG(ldx >= K && ldw >= K && ldc >= N, "va_gemm_nt: leading dimension smaller than the row length");
    VA_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "va_gemm_nt: leading dimensions must keep 16-byte alignment");
    VA_CHECK_ARG(((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) % 16 == 0, "va_gemm_nt: operands must be 16-byte aligned");
    VA_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "va_gemm_nt: problem too large");
    if (tile == 0) tile = (N % 256 == 0 && M >= 2048) ? 256 : 128;
    VA_CHECK_ARG(tile == 128 || (tile == 256 && N % 256 == 0), "va_gemm_nt: tile %d not usable with N=%lld", tile,
                 (long long)N);
    EpiArgs ep{out, out2, bias, resid, (const bf16_t*)aux, (int)ldc};
    const bf16_t* X = (const bf16_t*)x;
    const bf16_t* W = (const bf16_t*)w;
    switch (epilogue) {
        case VA_EPI_BF16:
            return launch_epi<VA_EPI_BF16>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_BIAS_GELU:
            VA_CHECK_ARG(out2 && bias, "va_gemm_nt: bias+GELU epilogue needs out2 and bias");
            return launch_epi<VA_EPI_BIAS_GELU>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_F32:
            return launch_epi<VA_EPI_F32>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        case VA_EPI_GELU_BWD:
            VA_CHECK_ARG(aux, "va_gemm_nt: GELU-backward epilogue needs the saved pre-activation");
            return launch_epi<VA_EPI_GELU_BWD>(tile, X, W, (int)M, (int)N, (int)K, (int)ldx, (int)ldw, ep, stream);
        default:
            va_set_error("va_gemm_nt: unknown epilogue %d", epilogue);
            return VA_ERR_INVALID;
    }
}
