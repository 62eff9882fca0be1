/* libvit_amd: C ABI of the MI355X (gfx950) kernels behind the video-ViT encoder hot path.
 *
 * Conventions (all entry points):
 *   - plain device pointers and sizes; no framework types.  Pointers are borrowed for the duration of
 *     the call's enqueued work; nothing is retained, allocated, freed or synchronised by the library.
 *   - all work is enqueued on `stream` (a hipStream_t; NULL = the default stream).
 *   - return 0 on success, a negative VA_ERR_* code otherwise; va_last_error() then describes it
 *     (thread-local string).  Nothing throws across the boundary.
 *   - bf16 tensors are raw 16-bit words; "f32" is IEEE binary32.  Row-major; `ld*` are leading
 *     dimensions in ELEMENTS.  16-byte alignment of every tensor base is required.
 *
 * Each declaration names the reference code it stands in for.  Citations are into the reference
 * repository (HananKibria/VIT): swm = surgical_world_model.py, s1 = stage1_surgvu_pretraining.py.
 */
#ifndef VIT_AMD_H
#define VIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* va_stream_t; /* == hipStream_t */

#define VA_ERR_INVALID_ARG (-1)
#define VA_ERR_LAUNCH_FAILED (-2)
#define VA_ERR_UNSUPPORTED_SHAPE (-3)

/* GEMM epilogues */
#define VA_EPI_BF16 0      /* out(bf16) = acc [+ bias]                                   */
#define VA_EPI_BIAS_GELU 1 /* out(bf16) = acc + bias ; out2(bf16) = gelu_erf(acc + bias)  */
#define VA_EPI_F32 2       /* out(f32)  = acc [+ bias] [+ resid(f32)]                     */
#define VA_EPI_GELU_BWD 3  /* out(bf16) = acc * gelu_erf'(aux(bf16))                      */
#define VA_EPI_GELU_ONLY 4 /* out(bf16) = gelu_erf(acc + bias): the forward of a pass that keeps nothing for a
                              backward (half the output bytes of VA_EPI_BIAS_GELU) */
#define VA_EPI_BIAS_GELU_GRAD 5 /* out(bf16) = gelu_erf'(acc + bias) ; out2(bf16) = gelu_erf(acc + bias): the forward that
                                   keeps the activation's derivative instead of the pre-activation (same bytes), so that
                                   the backward epilogue is a multiplication (VA_EPI_MUL_AUX) with no transcendental math */
#define VA_EPI_MUL_AUX 6        /* out(bf16) = acc * aux(bf16)                                   */

int va_version(void);
const char* va_last_error(void);

/* ---- dense projections ------------------------------------------------------------------------
 * out[M,N] = x[M,K] * w[N,K]^T with a fused epilogue; bf16 operands, f32 accumulation on MFMA.
 * Replaces nn.Linear forward for qkv_proj / out_proj (swm:296, 334), the MLP (swm:698-701), the
 * Conv3d patch projection once patches are gathered (swm:102), and the data-gradient half of their
 * autograd backward (with w = a transposed copy of the weight).  K % 64 == 0, N % 128 == 0.
 * `tile`: 0 = choose, 128 or 256 = force the workgroup tile edge. */
int va_gemm_nt(const void* x, const void* w, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
               int epilogue, void* out, void* out2, const float* bias, const float* resid, const void* aux,
               int64_t ldc, int tile, va_stream_t stream);

/* The same with a training-mode dropout (nn.Dropout of swm:700, 702 and its backward) folded into the epilogue,
 * with the decisions va_dropout(key, p) makes over the contiguous [M,N] output (ldc == N, M*N < 2^32):
 *   VA_EPI_BIAS_GELU: out2 = drop(gelu(acc + bias));  VA_EPI_F32: out = drop(acc + bias) + resid;
 *   VA_EPI_GELU_ONLY: out = drop(gelu(acc + bias));
 *   VA_EPI_GELU_BWD:  out = drop(acc * gelu'(aux));  VA_EPI_MUL_AUX: out = drop(acc * aux);
 *   VA_EPI_BIAS_GELU_GRAD: out2 = drop(gelu(acc + bias)), out = gelu'(acc + bias) untouched.
 *   p == 0 is va_gemm_nt.  Not with tile >= 4256. */
int va_gemm_nt_dropout(const void* x, const void* w, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                       int epilogue, void* out, void* out2, const float* bias, const float* resid, const void* aux,
                       int64_t ldc, int tile, uint32_t key, float p, va_stream_t stream);

/* The QKV projection of an attention block with the 3-D rotary embedding applied in the GEMM's epilogue
 * (RoPEMultiheadAttention.forward, swm:300-308 followed by apply_rope_to_qk, swm:265-281), for the encoder's regular token
 * grid:  out (bf16 [M, 3072], q | k | v) = rope(x[M, K] w[3072, K]^T), the rotation as va_rope3d_qk_inplace applies it
 * to the q and k thirds, but on the f32 accumulators (the two-pass route rotates the bf16-rounded product).
 * tables: the [cos | sin] tables of va_rope3d_qk_inplace for the grid (Tp, Hp, Wp); pos_code int32 [M]: per output row
 * t | h << 10 | w << 20, its token's grid coordinates (each < 1024).  M >= 256, K a multiple of 64 and >= 192.
 * Runs on the persistent four-wave kernel only: VA_ERR_UNSUPPORTED_SHAPE when that kernel has no ticket set left for the
 * stream (callers then take va_gemm_nt + va_rope3d_qk_inplace). */
int va_gemm_nt_qkv_rope(const void* x, const void* w, int64_t M, int64_t K, int64_t ldx, int64_t ldw, void* out, int64_t ldc,
                        const float* tables, const int32_t* pos_code, int64_t Tp, int64_t Hp, int64_t Wp, va_stream_t stream);

/* dw[N,K] (f32) += dy[M,N]^T * x[M,K]: the weight-gradient half of nn.Linear's backward (autograd of
 * swm:296, 334, 698-701, 102).  bf16 operands, split over M in `split_k` slices (0 = choose; negative =
 * |split_k| slices on the small-tile kernel) that are
 * combined with f32 atomic adds, so `dw` must hold the value to accumulate onto (zeros for a fresh
 * gradient).  If dbias != NULL, dbias[N] (f32) += column sums of dy (the bias gradient) in the same pass.
 * N % 128 == 0, K % 128 == 0; any M. */
int va_gemm_tn_accum(const void* dy, const void* x, int64_t M, int64_t N, int64_t K, int64_t lddy, int64_t ldx,
                     float* dw, int64_t lddw, float* dbias, int split_k, va_stream_t stream);

/* out[N] (f32) += column sums of y[M,N] (bf16): bias gradients (autograd of swm:698, 701, 78). */
int va_colsum_bf16_accum(const void* y, int64_t M, int64_t N, int64_t ldy, float* out, va_stream_t stream);

/* ---- tubelet patch gather -----------------------------------------------------------------------
 * video (B,C,T,H,W) f32 -> patches [B*Tp*Hp*Wp, C*tt*th*tw] bf16, row = token (t slowest, w fastest),
 * columns ordered (c, dt, dh, dw) like a flattened Conv3d filter: the im2col of the non-overlapping
 * Conv3d at swm:78-84 with the flatten/transpose of swm:108 folded in.  tw % 8 == 0. */
int va_tubelet_gather(const float* video, void* patches, int64_t B, int64_t C, int64_t T, int64_t H, int64_t W,
                      int64_t tt, int64_t th, int64_t tw, va_stream_t stream);

/* The adjoint placement: patches [tokens, C*tt*th*tw] f32 (gradient of the patch matrix) -> video gradient
 * (B,C,T,H,W) f32; pixels outside every tubelet (T, H or W not a multiple of the tubelet) are left untouched,
 * so the caller zero-fills those.  The input-gradient half of the Conv3d backward (autograd of swm:78-84, 99).
 * tw % 4 == 0. */
int va_tubelet_scatter(const float* patches, float* video, int64_t B, int64_t C, int64_t T, int64_t H, int64_t W,
                       int64_t tt, int64_t th, int64_t tw, va_stream_t stream);

/* ---- LayerNorm (nn.LayerNorm, eps 1e-6; swm:686, 695, 757) ---------------------------------------
 * forward: x f32 [M,D] -> y (bf16 if out_bf16 else f32) [M,D]; also stores mean/rstd [M] for backward (both NULL: a
 * forward-only call that keeps no statistics). */
int va_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int out_bf16, float* mean,
                     float* rstd, int64_t M, int64_t D, float eps, va_stream_t stream);
/* backward: dy (bf16 if dy_bf16 else f32) [M,D]; dx_out(f32) = (resid_grad ? resid_grad : 0) + dLN/dx;
 * dx_bf16 (optional) receives a bf16 copy of dx_out; partial[(2*nblk), D] f32 is workspace of
 * va_layernorm_bwd_partial_rows(M) * 2 * D floats; dgamma/dbeta [D] f32 are overwritten. */
int64_t va_layernorm_bwd_partial_rows(int64_t M);
int va_layernorm_bwd(const void* dy, int dy_bf16, const float* x, const float* gamma, const float* mean,
                     const float* rstd, const float* resid_grad, float* dx_out, void* dx_bf16, float* partial,
                     float* dgamma, float* dbeta, int64_t M, int64_t D, va_stream_t stream);
/* The same with two optional extras for the bf16 copy, which is the operand of the gradient GEMMs of the linear layer
 * above (swm:701-702): p > 0 makes it va_dropout(dx_out; key, p) instead of a plain copy (dx_out itself stays
 * undropped), and copy_colsum [D] f32 (nullable) receives its column sums = that layer's bias gradient.  With
 * copy_colsum the workspace is va_layernorm_bwd_partial_rows(M) * 3 * D floats. */
int va_layernorm_bwd_dropout(const void* dy, int dy_bf16, const float* x, const float* gamma, const float* mean,
                             const float* rstd, const float* resid_grad, float* dx_out, void* dx_bf16, float* partial,
                             float* dgamma, float* dbeta, int64_t M, int64_t D, uint32_t key, float p,
                             float* copy_colsum, va_stream_t stream);

/* va_layernorm_bwd_dropout without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE:
 * it exists for callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  The partial sums live in a static device array of the library (6 MiB per GPU,
 * held until the process ends; rows beyond VA_LN_NOALLOC_MAX_ROWS are rejected with VA_ERR_INVALID_ARG).  Results are
 * bit-identical to va_layernorm_bwd_dropout.  The library calls no hipMalloc / hipFree here; the first call on each
 * device resolves that device's array address (make that call outside graph capture).  All calls on one GPU share
 * the array: calls that overlap on one GPU (different streams, different threads) race -- serialise them. */
#define VA_LN_NOALLOC_MAX_ROWS (1ll << 20)
int va_layernorm_bwd_noalloc(const void* dy, int dy_bf16, const float* x, const float* gamma, const float* mean,
                             const float* rstd, const float* resid_grad, float* dx_out, void* dx_bf16, float* dgamma,
                             float* dbeta, int64_t M, int64_t D, uint32_t key, float p, float* copy_colsum,
                             va_stream_t stream);

/* ---- 3D rotary embedding (swm:33-51, 265-281) ------------------------------------------------------
 * In-place rotation of the Q and K thirds of a packed qkv buffer [M, 3*D] bf16 (D = 1024, column
 * segments t:0-255, h:256-639, w:640-1023; partner of column i inside a segment of width d is i+d/2).
 * Token n = m % (Tp*Hp*Wp) has coordinates (n/(Hp*Wp), (n/Wp)%Hp, n%Wp) (swm:760-770).  `tables` holds
 * the half-width cos then sin tables [(Tp x 128) | (Hp x 192) | (Wp x 192)] cos, then the same for sin
 * (values of swm:192-222).  inverse != 0 rotates by the negative angle (the backward of the rotation). */
int va_rope3d_qk_inplace(void* qkv, const float* tables, int64_t M, int64_t Tp, int64_t Hp, int64_t Wp,
                         int inverse, va_stream_t stream);

/* The general form of the same rotation, for RoPEMultiheadAttention.forward called on its own with any cache and any
 * token positions (swm:265-281, 300-303): cos_x / sin_x are the full-width f32 tables VideoRoPE3D.forward returns
 * ([len_t, 256], [len_h, 384], [len_w, 384]; the two halves of a row need not be equal) and pos is an int32 [3, N]
 * array of per-token (t, h, w) table rows, the same for every sequence of N tokens in the M rows.  The caller
 * guarantees 0 <= pos < len.  transpose != 0 applies the transposed rotation (the backward). */
int va_rope3d_qk_indexed(void* qkv, const float* cos_t, const float* sin_t, const float* cos_h, const float* sin_h,
                         const float* cos_w, const float* sin_w, const int32_t* pos, int64_t M, int64_t N,
                         int transpose, va_stream_t stream);

/* ---- flash attention (swm:306-331: softmax(q k^T * hd^-0.5) v, no mask, eval-mode dropout) ---------
 * q,k,v: bf16, element (b, n, h, d) at ((b*N + n)*ld + h*hd + d) of the respective base pointer (so the
 * three thirds of one packed qkv buffer can be passed with ld = 3*D).  o likewise with ldo.
 * lse2[b*H + h][n] f32 receives log2-sum-exp of the scaled scores (needed by the backward).
 * hd in {32, 64, 128} (32, 16 or 8 heads at width 1024). */
int va_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B, int64_t N, int64_t H,
                int64_t hd, int64_t ldqkv, int64_t ldo, va_stream_t stream);
/* Same with the predictor's block-causal mask (BlockCausalMask.create_mask, swm:1008-1028, as used by
 * HybridPredictorBlock swm:975-997): a query of timestep n / tokens_per_timestep attends to every key of the
 * same and of earlier timesteps.  N must be a whole number of timesteps.  Forward only (inference). */
int va_attn_fwd_block_causal(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B, int64_t N,
                             int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, int64_t tokens_per_timestep,
                             va_stream_t stream);
/* backward of the same: given o, do (bf16, layout of o), lse2, writes dq, dk, dv (layout of q/k/v, ld =
 * lddqkv).  workspace: va_attn_bwd_workspace_bytes(B, N, H) bytes (5*B*H*N floats), 16-byte aligned (per-query
 * softmax statistics in the two forms the dQ and dK/dV passes consume). */
int64_t va_attn_bwd_workspace_bytes(int64_t B, int64_t N, int64_t H);
int va_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2,
                float* workspace, void* dq, void* dk, void* dv, int64_t B, int64_t N, int64_t H, int64_t hd,
                int64_t ldqkv, int64_t ldo, int64_t lddqkv, va_stream_t stream);

/* Same with an explicit boolean mask (the `attn_mask` argument of RoPEMultiheadAttention.forward, swm:316-323: True =
 * do not attend, any pattern; 2-D masks are shared by all batch entries and heads, 3-D ones by all heads).  The mask is
 * handed over bit-packed, which va_attn_mask_pack does once per mask: `by_query` [batches, N, W] 32-bit words with
 * W = 2 * ceil(N / 64) (va_attn_mask_words(N) = N * W words per [N, N] mask), bit j of word w of row q set <=> key
 * 32 w + j is masked for query q (keys past N: set); `by_key` is the transpose (row = key, bits = queries), which the
 * key-stationary backward pass reads.  stride_b / stride_h: words from one batch entry / head to the next, 0 when
 * the mask is shared along that axis.  A query with every key masked gets NaN output, as in the reference (softmax of
 * a row of -inf); its gradient contributions are then undefined.  No dropout variant (eval mode or p = 0). */
int64_t va_attn_mask_words(int64_t N);
int va_attn_mask_pack(const uint8_t* mask /* [batches, N, N] bytes, non-zero = masked */, int64_t batches, int64_t N,
                      uint32_t* by_query, uint32_t* by_key, va_stream_t stream);
int va_attn_fwd_masked(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B, int64_t N, int64_t H,
                       int64_t hd, int64_t ldqkv, int64_t ldo, const uint32_t* mask_by_query, int64_t stride_b,
                       int64_t stride_h, va_stream_t stream);
int va_attn_bwd_masked(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2,
                       float* workspace, void* dq, void* dk, void* dv, int64_t B, int64_t N, int64_t H, int64_t hd,
                       int64_t ldqkv, int64_t ldo, int64_t lddqkv, const uint32_t* mask_by_query,
                       const uint32_t* mask_by_key, int64_t stride_b, int64_t stride_h, va_stream_t stream);

/* ---- training-mode dropout ---------------------------------------------------------------------------
 * The reference's encoder block applies nn.Dropout to the attention probabilities (swm:326, inside
 * scaled_dot_product_attention when the module is in training mode) and after both MLP linears (swm:700-702).
 * Keep/drop decisions here are a pure function of (key, element position), so the backward kernels regenerate
 * the mask of the forward instead of reading a stored one.  Two neighbouring elements share one 32-bit word
 *     h = mix(pair * 0x9E3779B1 + key),   mix(w): w ^= w >> 16; w *= 0x21F0AAAD; w ^= w >> 15   (mod 2^32);
 * the even element of the pair is kept when h & 0xffff >= T, the odd one when h >> 16 >= T, T = round(p * 65536),
 * and kept values are multiplied by 65536 / (65536 - T).  The random STREAM differs from torch's Philox
 * generator; the distribution (independent Bernoulli(1 - p) per element) is the same.
 *
 * va_attn_fwd_dropout / va_attn_bwd_dropout: as va_attn_fwd / va_attn_bwd with dropout of probability p on the
 * probabilities that multiply V.  Element (q, k) of head-batch bh = b * H + h uses the key key + bh * 0x85EBCA6B
 * and is field k & 1 of pair q * ceil(N / 2) + k / 2.  The backward must be given the forward's key and p.
 * p == 0 runs the plain kernels. */
int va_attn_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* lse2, int64_t B, int64_t N,
                        int64_t H, int64_t hd, int64_t ldqkv, int64_t ldo, uint32_t key, float p, va_stream_t stream);
int va_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2,
                        float* workspace, void* dq, void* dk, void* dv, int64_t B, int64_t N, int64_t H, int64_t hd,
                        int64_t ldqkv, int64_t ldo, int64_t lddqkv, uint32_t key, float p, va_stream_t stream);
/* out[e] = (resid ? resid[e] : 0) + keep(e) * in[e] * scale over n contiguous elements (n % 8 == 0), element e
 * being field e & 1 of pair e / 2; in / out are f32 or bf16 as the flags say, resid is f32 or NULL; in == out is
 * allowed.  The same call with the same key applied to a gradient is the backward. */
int va_dropout(const void* in, int in_bf16, void* out, int out_bf16, const float* resid, int64_t n, uint32_t key,
               float p, va_stream_t stream);

/* ---- parameter casts -------------------------------------------------------------------------------
 * f32 -> bf16 copy of `n` elements (master weights -> MFMA operands). */
int va_cast_f32_bf16(const float* src, void* dst, int64_t n, va_stream_t stream);
/* w f32 [R,C] -> wt bf16 [C,R] (transposed operand for the data-gradient GEMMs). R,C % 32 == 0. */
int va_cast_transpose_f32_bf16(const float* w, void* wt, int64_t R, int64_t C, va_stream_t stream);
/* Both copies of many matrices in one launch (all weights of a model after an optimizer step).  `jobs` is a DEVICE
 * array; job i covers tiles [first_tile, first_tile + rows/64 * cols/64) of the launch, first_tile ascending from 0;
 * rows, cols % 64 == 0; dst (bf16 [rows, cols]) and dst_t (bf16 [cols, rows]) may each be null.  The caller
 * validates the jobs: the library only sees the device copy. */
typedef struct va_cast_job {
    const float* src;
    void* dst;
    void* dst_t;
    int32_t rows, cols;
    int64_t first_tile;
} va_cast_job;
int va_multi_cast_f32_bf16(const va_cast_job* jobs, int64_t njobs, int64_t total_tiles, va_stream_t stream);

/* ---- token masking ------------------------------------------------------------------------------------
 * Deterministic future masking (s1:300-333): rows n >= cutoff*(N/Tp) of every batch item are zeroed,
 * cutoff = (int)(Tp*(1-ratio)) computed by the caller; out [B,N,D] (elem_bytes 2 or 4), mask [B,N] u8. */
int va_future_mask(const void* x, void* out, uint8_t* mask, int64_t B, int64_t N, int64_t D, int64_t elem_bytes,
                   int64_t first_masked_row, va_stream_t stream);
/* Row gather out[b,j,:] = x[b, idx[b,j], :] and its inverse scatter (out pre-zeroed by the call).
 * No reference symbol: these serve JEPA-style context/target token selection. row_bytes % 16 == 0.
 * An index outside [0, N) touches no memory: the gather writes a zero row for it, the scatter skips it. */
int va_token_gather(const void* x, const int32_t* idx, void* out, int64_t B, int64_t N_src, int64_t N_keep,
                    int64_t row_bytes, va_stream_t stream);
int va_token_scatter(const void* src, const int32_t* idx, void* out, int64_t B, int64_t N_dst, int64_t N_keep,
                     int64_t row_bytes, va_stream_t stream);

/* ---- token pooling of the stage-1 heads (stage1_surgvu_pretraining.py:426, 431-432 `encode_video(...).mean(dim=1)`;
 * :452-460 the mean over the tokens future_masking leaves unmasked) ------------------------------------------------
 * out[b][d] = scale * sum over tokens n < n_first of x[b][n][d];  x f32 [B, N, D], out f32 [B, D], D % 4 == 0.
 * n_first = N, scale = 1/N is the plain mean; n_first = cutoff * tokens_per_timestep, scale = 1/max(n_first, 1) is the
 * masked mean (future_masking masks a suffix of the token axis), computed without materialising the zero-filled copy.
 * workspace: va_token_pool_workspace_floats(B, D) floats.  Fixed summation order (bitwise reproducible).
 * The backward writes dx[b][n][:] = n < n_first ? scale * g[b][:] : 0. */
int64_t va_token_pool_workspace_floats(int64_t B, int64_t D);
int va_token_pool_fwd(const float* x, float* out, float* workspace, int64_t B, int64_t N, int64_t D, int64_t n_first,
                      float scale, va_stream_t stream);
int va_token_pool_bwd(const float* g, float* dx, int64_t B, int64_t N, int64_t D, int64_t n_first, float scale,
                      va_stream_t stream);

/* ---- EMA target update and L1 latent loss (no reference symbol; JEPA objective pieces) -----------------
 * target <- m*target + (1-m)*online over n f32 elements; optional bf16 shadow of the new target. */
int va_ema_update(float* target, const float* online, void* target_bf16, int64_t n, float momentum,
                  va_stream_t stream);
/* loss_sum[0] (f32, pre-zeroed by the call) = sum |pred - target|; grad (f32 or bf16 per grad_bf16, optional)
 * = sign(pred - target) * grad_scale.  pred/target f32 or bf16 per in_bf16. */
int va_l1_loss(const void* pred, const void* target, int in_bf16, float* loss_sum, void* grad, int grad_bf16,
               float grad_scale, int64_t n, va_stream_t stream);

/* ---- loss heads of the stage-1 step (stage1_surgvu_pretraining.py:426-462): value AND gradient per call ----------
 * All f32, fixed summation orders (bitwise reproducible), gradients for an upstream gradient of 1.
 * va_infonce_sym (s1:434-446): z = rows of p1 / p2 divided by max(|row|, 1e-12) when normalize != 0 (F.normalize), else
 *   the rows as they are; logits = z1 z2^T / temperature;
 *   loss[0] = 0.5 * (cross_entropy(logits, arange(B)) + cross_entropy(logits^T, arange(B)));  g1, g2 [B, D] its gradient
 *   with respect to p1, p2 (through the normalisation when normalize != 0).  workspace: va_infonce_workspace_floats(B, D).
 * va_normalize_rows_fwd / _bwd (s1:434-435, F.normalize(., dim=1)): z = p / max(|p|, 1e-12) per row and inv = that factor;
 *   dp = (dz - z (z . dz)) * inv.  The same arithmetic as va_infonce_sym's built-in normalisation: a data-parallel step
 *   normalises locally, gathers z over the ranks and calls va_infonce_sym with normalize = 0 on the global batch.
 * va_cross_entropy (s1:428): logits f32 [B, C], labels int64 [B]; loss[0] = mean_b (logsumexp - logit[label]);
 *   dlogits [B, C] = (softmax - onehot) * grad_scale: 1/B is the gradient of the mean; a caller that will multiply by an
 *   upstream factor g passes 1 and applies g/B itself -- one rounding, whatever B and g are.  workspace: B floats.
 * va_mse_loss (s1:459): loss[0] = mean((x - y)^2) over n elements; dx = (x - y) * grad_scale (2/n for the gradient of the
 *   mean; y is a target: no gradient).  workspace: va_mse_workspace_floats(n) floats. */
int va_normalize_rows_fwd(const float* p, float* z, float* inv, int64_t B, int64_t D, va_stream_t stream);
int va_normalize_rows_bwd(const float* dz, const float* z, const float* inv, float* dp, int64_t B, int64_t D,
                          va_stream_t stream);
int64_t va_infonce_workspace_floats(int64_t B, int64_t D);
int va_infonce_sym(const float* p1, const float* p2, float* loss, float* g1, float* g2, float* workspace, int64_t B,
                   int64_t D, float temperature, int normalize, va_stream_t stream);
int va_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits, float* workspace, int64_t B,
                     int64_t C, float grad_scale, va_stream_t stream);
int64_t va_mse_workspace_floats(int64_t n);
int va_mse_loss(const float* x, const float* y, float* loss, float* dx, float* workspace, int64_t n, float grad_scale,
                va_stream_t stream);

/* ---- predictor: gated delta-rule memory (GatedDeltaLayer, swm:431-555), inference form -----------------
 * Sequential recurrence of _sequential_gated_delta (swm:523-555) for every (batch, head):
 *   S <- alpha_t * S * (I - beta_t k_t k_t^T) + beta_t v_t k_t^T ;  out_t = S q_t
 * q,k,v,out: f32 [B, L, H, hd] contiguous; alpha,beta: f32 [B, L, H]; state_in (may be NULL = zeros) and
 * state_out: f32 [B, H, hd, hd].  hd in {64, 128}. */
int va_gated_delta_seq(const float* q, const float* k, const float* v, const float* alpha, const float* beta,
                       const float* state_in, float* out, float* state_out, int64_t B, int64_t L, int64_t H, int64_t hd,
                       va_stream_t stream);
/* Causal depthwise conv1d with 4 taps + bias + SiLU over x f32 [B, L, C] (nn.Conv1d(C, C, 4, padding=3,
 * groups=C) cut to the first L outputs, swm:452-457 / 484-486, then swm:489-491): out[b,t,c] =
 * silu(bias[c] + sum_i w[c][i] * x[b, t-3+i, c]); with normalize != 0 each row is then divided by
 * max(||row||_2, 1e-12) (F.normalize over all C channels).  w: f32 [C, 4].  Not in place. */
int va_dwconv1d_causal_silu(const float* x, const float* w, const float* bias, float* out, int64_t B, int64_t L,
                            int64_t C, int normalize, va_stream_t stream);

/* Output stage of the layer between the recurrence and the output projection (swm:511-519):
 * y = LayerNorm((o * gate) * silu(g); gamma, beta, eps) as bf16.  o (the recurrence output), gate (the second half
 * of the convolved v projection, swm:500) and g (the raw g_proj output, swm:515): f32 [M, D] contiguous; D = 1024. */
int va_gate_layernorm_fwd(const float* o, const float* gate, const float* g, const float* gamma, const float* beta,
                          void* y_bf16, int64_t M, int64_t D, float eps, va_stream_t stream);

/* The two kernels above for operands that are column blocks of ONE bf16 projection output: the layer's q, k, v and g
 * projections (swm:484-500, 515) computed as a single GEMM against the row-wise concatenation of their weights give a
 * bf16 matrix [B*L, 5 * 1024]; these entry points read a 1024-wide block of it in place.
 * va_dwconv1d_causal_silu_bf16: x bf16, row (b, t) at x + (b*L + t) * ldx, C = 1024 channels; f32 out [B, L, C];
 * otherwise as va_dwconv1d_causal_silu.  va_gate_layernorm_fwd_g16: g bf16 with rows ldg elements apart; o and gate
 * f32 [M, D] contiguous; otherwise as va_gate_layernorm_fwd. */
int va_dwconv1d_causal_silu_bf16(const void* x_bf16, int64_t ldx, const float* w, const float* bias, float* out, int64_t B,
                                 int64_t L, int64_t C, int normalize, va_stream_t stream);
int va_gate_layernorm_fwd_g16(const float* o, const float* gate, const void* g_bf16, int64_t ldg, const float* gamma,
                              const float* beta, void* y_bf16, int64_t M, int64_t D, float eps, va_stream_t stream);

/* The layer's two per-head gates, alpha = sigmoid(x @ wa^T) and beta = sigmoid(x @ wb^T) (swm:448-449, 494-495; the
 * projections have no bias): x f32 [M, D] contiguous, D = 1024; wa, wb f32 [H, D], H <= 8; alpha, beta f32 [M, H].
 * f32 products and sums; a row's summation order does not depend on its position in the batch.  x_bf16 (optional, may be
 * NULL): bf16 [M, D] copy of x, written by the same pass (the operand of the layer's projection GEMM). */
int va_delta_gates_fwd(const float* x, const float* wa, const float* wb, float* alpha, float* beta, void* x_bf16, int64_t M,
                       int64_t D, int64_t H, va_stream_t stream);

/* ---- input pipeline: the tensor side of the reference's per-frame transform (ToTensor + Normalize,
 * stage1_surgvu_pretraining.py:241-244, 279-286) ---------------------------------------------------------
 * frames: uint8 [B, T, H, W, 3] on the device -> out: f32 [B, 3, T, H, W], (u / 255 - mean[c]) / std[c]
 * (bit-identical to the torch f32 arithmetic).  mean3 / std3: three HOST floats each.  H*W % 4 == 0. */
int va_normalize_u8(const uint8_t* frames, float* out, int64_t B, int64_t T, int64_t H, int64_t W, const float* mean3,
                    const float* std3, va_stream_t stream);

/* ---- optimizer step of the training loops (stage1_surgvu_pretraining.py:390 AdamW, :469
 * clip_grad_norm_(max_norm=1.0), :470 optimizer.step()) as multi-tensor kernels ----------------------------
 * A parameter set is a DEVICE array of va_opt_tensor records, f32 pointers with n elements each, plus a DEVICE table of
 * int32 pairs (tensor index, chunk index), one pair per VA_OPT_CHUNK elements of every tensor. */
#define VA_OPT_CHUNK 65536
typedef struct {
    float* p;        /* parameter                 */
    const float* g;  /* gradient                  */
    float* m;        /* first moment  (exp_avg)   */
    float* v;        /* second moment (exp_avg_sq)*/
    int64_t n;
    float inv_bc1;       /* 1 / (1 - beta1^step) of THIS tensor (torch counts steps per parameter) */
    float inv_sqrt_bc2;  /* 1 / sqrt(1 - beta2^step)                                                */
} va_opt_tensor;
/* total[0] = sum over all tensors of g^2 (the squared global L2 norm); partial: workspace of num_chunks
 * floats.  Fixed summation order (bitwise reproducible). */
int va_multi_sumsq(const va_opt_tensor* tensors, const int32_t* chunk_table, int64_t num_chunks, float* partial,
                   float* total, va_stream_t stream);
/* One AdamW step (torch.optim.AdamW arithmetic, decoupled weight decay, bias corrections from the tensor
 * records).  If
 * total_sumsq is non-NULL and max_norm > 0 the gradients are first scaled by
 * min(1, max_norm / (sqrt(total_sumsq[0]) + 1e-6)) as clip_grad_norm_ does; the gradient buffers themselves
 * are left unscaled. */
int va_multi_adamw(const va_opt_tensor* tensors, const int32_t* chunk_table, int64_t num_chunks, float lr, float beta1,
                   float beta2, float eps, float weight_decay, const float* total_sumsq, float max_norm,
                   va_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VIT_AMD_H */
