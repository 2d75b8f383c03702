/*
 * ivit_hip.h -- C ABI of libivit_hip.so: the MI355X (gfx950) integer-only ViT
 * operators that replace the hot path of lionnus/I-ViT
 * (models/quantization_utils/{quant_modules,ivit_modules,quant_utils}.py).
 *
 * The reference has no native interface at all (it is pure PyTorch, float32
 * "fake-quant" emulation), so each entry point below cites the Python code it
 * replaces; INTEGRATION.md shows the ctypes stub a maintainer of the reference
 * would add.  Conventions:
 *   - every function returns 0 on success, a negative IVIT_ERR_* code otherwise;
 *     ivit_last_error_string() describes the last failure on the calling thread;
 *   - all pointers are DEVICE pointers (hipMalloc / torch tensor .data_ptr())
 *     unless a parameter is documented as host scalar; the library allocates
 *     nothing and keeps no state: the caller owns inputs, outputs and tables;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream); functions are re-entrant;
 *   - integer tensors are row-major; `ld*` are leading dimensions in ELEMENTS;
 *   - a dyadic requantiser is the pair (m, e) of the reference's batch_frexp
 *     (quant_utils.py:151-175): out = RNE(z * m / 2^e), 2^30 <= m <= 2^31,
 *     e = 31 - exponent.  Per-channel tables are `const uint32_t* m,
 *     const int32_t* e`; per-tensor requantisers are passed by value.
 */
#ifndef IVIT_HIP_H
#define IVIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVIT_OK 0
#define IVIT_ERR_INVALID (-1)     /* bad argument (shape, alignment, NULL)        */
#define IVIT_ERR_UNSUPPORTED (-2) /* valid request this build has no kernel for  */
#define IVIT_ERR_LAUNCH (-3)      /* HIP reported an error at launch              */

typedef void* ivit_stream_t; /* hipStream_t */

int ivit_version(void);
const char* ivit_last_error_string(void);

/* ---- calibration statistics -----------------------------------------------------------------
 * QuantAct running_stat mode (quant_modules.py:310-349) observes x.min() / x.max() of the float view it is handed.
 * out_min_max[0] = min, out_min_max[1] = max over x[0..n) (device float[2]; NaNs are skipped). */
int ivit_minmax_f32(const float* x, int64_t n, float* out_min_max, ivit_stream_t stream);

/* ---- input quantisation ---------------------------------------------------------------
 * QuantAct input mode = SymmetricQuantFunction.forward (quant_utils.py:79-97,
 * linear_quantize :13-49):  q = clamp(round(inv_scale * x), -128, 127), inv_scale = fl(1/s)
 * computed by the caller in float32. */
int ivit_quantize_input_f32_i8(const float* x, int8_t* out, int64_t n, float inv_scale, ivit_stream_t stream);
/* the same at any width `bits` <= 32 (e.g. the 16-bit position embedding of pos_encoding_bw = 16), int32 out */
int ivit_quantize_input_f32_i32(const float* x, int32_t* out, int64_t n, float inv_scale, int bits, ivit_stream_t stream);

/* The same fused with the im2col of PatchEmbed's strided convolution
 * (layers_quant.py:197-198, quant_modules.py:506-511):
 *   img [B, chans, hw, hw] float32 -> A [B*(hw/patch)^2, chans*patch*patch] int8,
 *   column order (c, kh, kw) = the flattening of the conv weight [out, c, kh, kw]. */
int ivit_quantize_patchify_f32_i8(const float* img, int8_t* A, int batch, int chans, int hw, int patch,
                                  float inv_scale, ivit_stream_t stream);
/* as above with an explicit row stride lda >= chans*patch*patch (Swin: patch 4 -> K = 48, padded to the GEMM's
 * K granularity of 64); columns [K, lda) are not written: the caller zero-fills them once. */
int ivit_quantize_patchify_ld_f32_i8(const float* img, int8_t* A, int64_t lda, int batch, int chans, int hw, int patch,
                                     float inv_scale, ivit_stream_t stream);
/* The same from uint8 pixels [batch, chans, hw, hw]: lut[c * 256 + v] is the int8 the float pipeline in front of the model gives
 * pixel value v of channel c -- ToTensor (v / 255), Normalize ((x - mean[c]) / std[c]) and the input QuantAct
 * (clamp(round(x / s)), quant_utils.py:79-97), evaluated by the caller in float32 in that order (ivit_amd.prepare.input_lut_u8):
 * the results are identical to the float path on the same pixels and the input is a quarter of the bytes. */
int ivit_quantize_patchify_u8_i8(const uint8_t* img, int8_t* A, int64_t lda, int batch, int chans, int hw, int patch,
                                 const int8_t* lut, ivit_stream_t stream);

/* ---- INT8 GEMM on v_mfma_i32_32x32x32_i8 with fused epilogues -------------------------------
 * QuantLinear.forward / QuantConv2d.forward (quant_modules.py:186-226, 478-511) followed by
 * the QuantAct that always consumes them (quant_modules.py:302-387 -> fixedpoint_mul,
 * quant_utils.py:193-253).
 *   acc[t][n] = sum_k A[t][k] * W[n][k] + bias[n]          (exact int32)
 * A [M, K] int8 (lda), W [N, K] int8 (ldw), bias [N] int32 or NULL.  K % 64 == 0.
 * Contract for the requantising forms: e[n] >= 31 (multiplier <= 1: int32 accumulators -> 8 bits).
 */

/* ---- operand layouts of the GEMMs.  IVIT_LAYOUT_ROWS: row-major with a leading dimension (the default of every
 * entry point).  IVIT_LAYOUT_BLOCKS: 1 KB blocks of 16 rows x 64 bytes, block (row / 16, k / 64) at
 * ((row / 16) * (K / 64) + k / 64) * 1024, and inside a block the 16-byte chunk (r, c) at position 4r + (c ^ ((r >> 2) & 3))
 * -- the order in which the GEMM's LDS-DMA lays a piece into its LDS stage, so that one instruction reads 1 KB
 * contiguous.  Rows padded to a multiple of 16 (ceil(rows / 16) * 16 * K bytes), K % 64 == 0.  The `_ex` forms take
 * `layouts` = IVIT_A_BLOCKS | IVIT_W_BLOCKS; block operands need M >= 2048 and N >= 128.  Producers that can write the
 * block layout directly: ivit_layernorm_i8_ex, ivit_attention_fused_i8_ex, ivit_shiftgelu_lut_i8_ex. */
#define IVIT_A_BLOCKS 1
#define IVIT_W_BLOCKS 2
#define IVIT_OUT_BLOCKS 4   /* ivit_gemm_i8_requant_ex only: the int8 output in the block layout (ldo == N, N % 64 == 0) */
/* IVIT_W_FRAGS: `W` is the copy made by ivit_pack_weight_frags_i8 -- the weights in the order the MFMA consumes them, so that
 * a wave loads its fragments straight into registers (1 KB contiguous per instruction) and only the token tile goes through
 * the LDS: channel n, byte k at ((n / 64) * (K / 64) + k / 64) * 4096 + (((n / 32) % 2 * 2 + (k / 32) % 2) * 2 + (k / 16) % 2) * 512
 * + (n % 32) * 16 + k % 16, channels padded with zero rows to a multiple of 64 (ceil(N / 64) * 64 * K bytes).  Needs M >= 2048,
 * N >= 128, N % 64 == 0, K % 192 == 0; combines with IVIT_A_BLOCKS / IVIT_OUT_BLOCKS, not with IVIT_W_BLOCKS.  `ldw` is
 * ignored.  Results are identical to every other form. */
#define IVIT_W_FRAGS 8
int ivit_pack_weight_frags_i8(const int8_t* W, int64_t ldw, int N, int K, int8_t* dst, ivit_stream_t stream);
/* IVIT_W_FRAGS16: the same idea for v_mfma_i32_16x16x64_i8 (the chip holds a higher clock on that shape under the power limit:
 * DESIGN.md section 5): channel n, byte k at ((n / 64) * (K / 64) + k / 64) * 4096 + ((n / 16) % 4) * 1024 + ((k / 16) % 4) * 256
 * + (n % 16) * 16 + k % 16 (copy made by ivit_pack_weight_frags16_i8; same size, same constraints as IVIT_W_FRAGS; the two bits
 * exclude each other).  Accepted by the int8-output `_ex` forms (requant, requant_lut, residual, residual_i16, qkv). */
#define IVIT_W_FRAGS16 16
int ivit_pack_weight_frags16_i8(const int8_t* W, int64_t ldw, int N, int K, int8_t* dst, ivit_stream_t stream);
int ivit_tile_operand_i8(const int8_t* src, int64_t ld, int64_t rows, int K, int8_t* dst, ivit_stream_t stream);
int ivit_untile_operand_i8(const int8_t* src, int64_t rows, int K, int8_t* dst, int64_t ld, ivit_stream_t stream);

/* out[t][n] = clamp8(RNE(acc * m[n] / 2^e[n]));  N % 16 == 0 */
int ivit_gemm_i8_requant(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                         const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo,
                         int M, int N, int K, ivit_stream_t stream);

int ivit_gemm_i8_requant_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                            const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo,
                            int M, int N, int K, int layouts, ivit_stream_t stream);

/* ivit_gemm_i8_requant_ex followed by an elementwise int8 -> int8 map on every output: out = lut[k + 128], lut int8[256] -- an
 * operator behind the QuantAct that depends on the value alone (I-BERT GELU + mlp.qact1: ivit_ibert_gelu_build_lut row 0),
 * applied in the epilogue instead of by a kernel of its own.  Needs IVIT_W_FRAGS (the weights-in-registers kernel). */
int ivit_gemm_i8_requant_lut_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                const uint32_t* m, const int32_t* e, const int8_t* lut, int8_t* out, int64_t ldo,
                                int M, int N, int K, int layouts, ivit_stream_t stream);

/* as above, then the two-operand QuantAct of the residual connection
 * (vit_quant.py:147,153; quant_utils.py:232-245):
 *   k = clamp8(RNE(acc * m[n] / 2^e[n]))
 *   out = clamp8(RNE(k * m_main / 2^e_main) + RNE(res[t][n] * m_res / 2^e_res))
 * `out` may be `res` itself (in place: one thread reads a residual chunk and writes the same chunk). */
int ivit_gemm_i8_requant_residual(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                  const int32_t* bias, const uint32_t* m, const int32_t* e,
                                  const int8_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                  uint32_t m_res, int32_t e_res, int8_t* out, int64_t ldo,
                                  int M, int N, int K, ivit_stream_t stream);

int ivit_gemm_i8_requant_residual_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                     const int32_t* bias, const uint32_t* m, const int32_t* e,
                                     const int8_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                     uint32_t m_res, int32_t e_res, int8_t* out, int64_t ldo,
                                     int M, int N, int K, int layouts, ivit_stream_t stream);

/* Swin form of the above: the residual stream is 16 bits wide (swin_quant.py:299, mlp.fc2 + shortcut):
 *   k = clamp8(RNE(acc * m[n] / 2^e[n]))                       (mlp.qact2)
 *   out = clamp16(RNE(k * m_main / 2^e_main) + RNE(res[t][n] * m_res / 2^e_res))   (qact4, 16 bit)
 * res / out: int16 rows, ldr / ldo in elements and multiples of 8.  Replaces ivit_gemm_i8_requant followed by
 * ivit_residual_requant_i16(a_bits = 8). */
int ivit_gemm_i8_requant_residual_i16(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                      const int32_t* bias, const uint32_t* m, const int32_t* e,
                                      const int16_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                      uint32_t m_res, int32_t e_res, int16_t* out, int64_t ldo,
                                      int M, int N, int K, ivit_stream_t stream);
int ivit_gemm_i8_requant_residual_i16_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                         const int32_t* bias, const uint32_t* m, const int32_t* e,
                                         const int16_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                         uint32_t m_res, int32_t e_res, int16_t* out, int64_t ldo,
                                         int M, int N, int K, int layouts, ivit_stream_t stream);

/* as ivit_gemm_i8_requant but the output is written head-major for the attention kernel:
 * N = 3 * heads * head_dim, row t = b * tokens + tok  ->
 *   qkv[which][b][h][tok][d],  n = which*heads*head_dim + h*head_dim + d
 * (the reshape/permute of vit_quant.py:65-71).  head_dim % 16 == 0. */
int ivit_gemm_i8_requant_qkv(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                             const uint32_t* m, const int32_t* e, int8_t* qkv, int tokens, int heads,
                             int head_dim, int M, int N, int K, ivit_stream_t stream);
int ivit_gemm_i8_requant_qkv_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                const uint32_t* m, const int32_t* e, int8_t* qkv, int tokens, int heads,
                                int head_dim, int M, int N, int K, int layouts, ivit_stream_t stream);

/* 16-bit per-channel QuantAct of the accumulators (Swin attn.proj + attn.qact4, swin_quant.py:164-166):
 *   out[t][n] = clamp16(RNE(acc * m[n] / 2^e[n])),  out int16 [M, N] (ldo in elements), N % 4 == 0, ldo % 4 == 0.
 * Followed by ivit_residual_requant_i16(a_bits = 16) it replaces ivit_gemm_i8_i32 + ivit_residual_requant_i16(a_bits = 32)
 * with half the intermediate bytes. */
int ivit_gemm_i8_requant_i16(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                             const uint32_t* m, const int32_t* e, int16_t* out, int64_t ldo, int M, int N, int K,
                             ivit_stream_t stream);

/* the two calls above in one (ViT attn.proj + attn.qact3 at 16 bits + qact2, mlp.fc2 + mlp.qact2 at 16 bits + qact4,
 * vit_quant.py:131-150 with attention_out_bw = mlp_out_bw = norm2_in_bw = att_block_out_bw = 16):
 *   k16 = clamp16(RNE(acc * m[n] / 2^e[n])),  out = clamp16(RNE(k16 * M_main) + RNE(res * M_res)),  res / out int16 rows.
 * `layouts`: IVIT_W_FRAGS (optionally | IVIT_A_BLOCKS) where that form applies -- the epilogue is then transposed through LDS --
 * or 0: any shape, the 128 x 128-tile kernel with the epilogue straight from its registers (Swin attn.proj at C = 96 .. 384).
 * N % 8 == 0, ldr / ldo multiples of 8, res / out 16-byte aligned; out may alias res. */
int ivit_gemm_i8_requant_i16_residual_i16_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                             const uint32_t* m, const int32_t* e, const int16_t* res, int64_t ldr,
                                             uint32_t m_main, int32_t e_main, uint32_t m_res, int32_t e_res, int16_t* out,
                                             int64_t ldo, int M, int N, int K, int layouts, ivit_stream_t stream);

/* raw accumulators (classifier head; module-level QuantLinear): out int32 [M, N], N % 4 == 0 */
int ivit_gemm_i8_i32(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                     int32_t* out, int64_t ldo, int M, int N, int K, ivit_stream_t stream);

/* ---- fused attention core -------------------------------------------------------------------
 * vit_quant.py:72-85: matmul_1 (q.k^T) -> qact_attn1 -> IVITIntSoftmax (Shiftmax,
 * ivit_modules.py:150-179) -> matmul_2 (P.v) -> qact2, per (image, head), never
 * materialising the [B,H,T,T] score tensor.
 *   qkv   [3][B][H][T][head_dim] int8 (as written by ivit_gemm_i8_requant_qkv)
 *   out   [B*T, H*head_dim] int8, column h*head_dim + d (the transpose(1,2).reshape of :83)
 *   (m_s, e_s): requantiser of q.k^T into the Shiftmax input (8 bit)
 *   s_attn: float32 scale of the Shiftmax input (x0 = floor(-1/s_attn), n = 15)
 *   (m_o, e_o): requantiser of P.v into the 8-bit output.
 * Supported: head_dim 64, 193 <= tokens <= 208 (13 key tiles of 16). */
int ivit_attention_fused_i8(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim,
                            uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o, int32_t e_o,
                            ivit_stream_t stream);
/* out_blocks = 1: `out` ([batch*tokens, heads*head_dim]) is written in IVIT_LAYOUT_BLOCKS (the A operand of attn.proj) */
int ivit_attention_fused_i8_ex(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim,
                               uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o, int32_t e_o, int out_blocks,
                               ivit_stream_t stream);

/* Natural ("as calibrated", non power-of-two) Shiftmax input scale.  The reference's Shiftmax then runs its float32
 * sequence on phi(q) = fl(fl(q*s)/s) instead of q (ivit_modules.py:165-170: QuantAct returned q*s, quant_modules.py:387;
 * the .to(int32) of :166 is discarded), so exp_int depends on (row max, q) jointly.  exp2d: [256][256] uint32 on the
 * device, entry (qmax+128)*256 + (q+128) = exp_int of that pair, tabulated by the caller with the reference's float32
 * steps (i-vit_amd/prepare.py shiftexp2d); NULL = power-of-two scale (identical to ivit_attention_fused_i8_ex). */
int ivit_attention_fused_i8_compat(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim,
                                   uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o, int32_t e_o,
                                   const uint32_t* exp2d, int out_blocks, ivit_stream_t stream);

/* The same with the table also in BAND form, staged per query tile in LDS (the fast path; exp2d may then be NULL):
 * band[(qmax+128)*band_w + j] = exp_int of (qmax, q = qmax - j) for j < band_w, band_w a multiple of 16 in [16, 256] (4 x 16 x (band_w + 4) x 4 bytes of dynamic LDS on top of 31 KiB static: 97 KiB at 256, within the 160 KiB of a CU; launches above 64 KiB are covered by test_attention_fused_compat) chosen
 * by the caller such that entry band_w - 1 is already the saturated value (the exponent's argument is clamped at n*x0,
 * ivit_modules.py:155, so every larger distance gives the same entry); i-vit_amd/prepare.py shiftexp_band. */
int ivit_attention_fused_i8_compat_band(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim,
                                        uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o, int32_t e_o,
                                        const uint32_t* exp2d, const uint32_t* band, int band_w, int out_blocks,
                                        ivit_stream_t stream);
/* the same with the softmax output width as a parameter (softmax_bw, vit_quant.py:184): softmax_bits = 8 or 16.  16:
 * p = floor(fl32(e * factor) / 2^16) <= 2^15 at scale 2^-15 (ivit_modules.py:175-176), carried into P.V as three 7-bit planes;
 * (m_o, e_o) then is the requantiser of 2^-15 * s_v. */
int ivit_attention_fused_i8_wide(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim, uint32_t m_s,
                                 int32_t e_s, float s_attn, uint32_t m_o, int32_t e_o, const uint32_t* exp2d,
                                 const uint32_t* band, int band_w, int softmax_bits, int out_blocks, ivit_stream_t stream);

/* ---- I-LayerNorm + the QuantAct behind it ---------------------------------------------------
 * IVITIntLayerNorm.forward (ivit_modules.py:30-65) then QuantAct (fixedpoint_mul).
 *   x [rows, C] int8 (ldx), per channel: bias_int[c] = floor((beta/gamma)/(sqrt(C)/2^30)),
 *   s_ln[c] = (sqrt(C)/2^30)*gamma[c] (both float32, prepared by the caller as the reference
 *   computes them, :53-62), (m[c], e[c]) requantiser s_ln[c] -> output scale; contract e[c] >= 40
 *   (multiplier <= 2^-9: I-LayerNorm outputs are ~30-bit integers requantised to 8 bits).
 *   out [rows, C] int8 (ldo).  C % 4 == 0, C <= 4096. */
int ivit_layernorm_i8(const int8_t* x, int64_t ldx, int rows, int C, const float* bias_int, const float* s_ln,
                      const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, ivit_stream_t stream);
/* out_blocks = 1: `out` is written in IVIT_LAYOUT_BLOCKS (C % 64 == 0, ldo == C, rows padded to 16) */
int ivit_layernorm_i8_ex(const int8_t* x, int64_t ldx, int rows, int C, const float* bias_int, const float* s_ln,
                         const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, int out_blocks,
                         ivit_stream_t stream);

/* The same for an input carried at a natural (non power-of-two) scale s: IVITIntLayerNorm sees phi(q) = fl(fl(q*s)/s)
 * (ivit_modules.py:36).  remap[q+128] = trunc(phi(q)) (int8, the `.to(int32)` of :38) and phi[q+128] (float32; the mean of
 * :37 is taken over these: equal to RNE(sum q / C) except on rows whose integer sum is an exact .5 tie, where torch's CPU
 * float32 reduction order decides -- restated in the kernel).  Both tables on the device, prepared by the caller
 * (i-vit_amd/prepare.py phi_tables).  C % 8 == 0, C >= 32. */
int ivit_layernorm_i8_compat(const int8_t* x, int64_t ldx, int rows, int C, const float* bias_int, const float* s_ln,
                             const uint32_t* m, const int32_t* e, const int8_t* remap /* [256] */,
                             const float* phi /* [256] */, int8_t* out, int64_t ldo, int flags,
                             ivit_stream_t stream);
#define IVIT_LN_OUT_BLOCKS 1              /* `flags` of ivit_layernorm_i8_compat: `out` in IVIT_LAYOUT_BLOCKS */
#define IVIT_LN_OUTER_MEAN(L) ((L) << 8)  /* tie rows in torch's outer-reduction order for a transposed view of contiguous extent L
                                           * (see ivit_layernorm_f32_f32_ex): the Swin patch embedding */

/* module-level form: int32 input (8 or 16 bit values), float32 output y*s_ln (ivit_modules.py:63) */
int ivit_layernorm_i32_f32(const int32_t* x, int64_t ldx, int rows, int C, const float* bias_int,
                           const float* s_ln, float* out, int64_t ldo, ivit_stream_t stream);

/* module-level LITERAL form, any input scale: x is the float view q*s the reference's module receives, s_in its scale
 * (n_s = 1 per tensor, or C per channel).  Every float32 step of ivit_modules.py:36-63 as written, including the mean over
 * x/s in torch's CPU reduction order (decides rows whose mean is an exact .5 tie when s is not a power of two) and the
 * truncating .to(int32).  Equals ivit_layernorm_i32_f32 on round(x/s) when s is a power of two. */
int ivit_layernorm_f32_f32(const float* x, int64_t ldx, int rows, int C, const float* s_in, int n_s,
                           const float* bias_int, const float* s_ln, float* out, int64_t ldo, ivit_stream_t stream);
/* outer_mean = L > 0: the caller's tensor reaches the reference's `x_int.mean(axis=2)` as a TRANSPOSED view (the reduced dimension
 * is not the contiguous one: the Swin patch embedding, layers_quant.py:198-201, whose elementwise QuantAct keeps the strides of
 * `x.flatten(2).transpose(1, 2)`); L = extent of the contiguous dimension (tokens per image), rows % L == 0, row = image * L +
 * column.  ATen then sums every row with its outer-reduction order (csrc/rowsum.h: one cascade over the reduced index for
 * columns below 32 * (L / 32), four interleaved cascades for the last L % 32 columns) instead of the 32-partial-sum order of a
 * contiguous row; rows whose mean is an exact .5 tie come out differently.  `x` itself is still passed row-contiguous. */
int ivit_layernorm_f32_f32_ex(const float* x, int64_t ldx, int rows, int C, const float* s_in, int n_s,
                              const float* bias_int, const float* s_ln, float* out, int64_t ldo, int outer_mean,
                              ivit_stream_t stream);

/* ---- ShiftGELU -----------------------------------------------------------------------------
 * IVITIntGELU.forward (ivit_modules.py:89-126, n = 23, output_bit = 8) on rows of L int8 values
 * with input scale s, followed by the per-tensor QuantAct (m, e) -> int8.
 * Direct arithmetic form: */
int ivit_shiftgelu_i8(const int8_t* x, int64_t ldx, int rows, int L, float s, uint32_t m, int32_t e,
                      int8_t* out, int64_t ldo, ivit_stream_t stream);
/* module-level form: int32 output k*sigmoid_int (ivit_modules.py:123), no requant */
int ivit_shiftgelu_i8_i32(const int8_t* x, int64_t ldx, int rows, int L, float s, int32_t* out, int64_t ldo,
                          ivit_stream_t stream);
/* Table form used by the engine: for a fixed (s, m, e) the result depends only on (row max, k):
 * lut[(kmax+128)*256 + (k+128)] = int8 result.  Build once per layer at load time ... */
int ivit_shiftgelu_build_lut(float s, uint32_t m, int32_t e, int8_t* lut /* [256*256] */, ivit_stream_t stream);
/* natural input scale: remap[q+128] = trunc(phi(q)) (device, [256] int8; ivit_modules.py:106-107) is applied to k and to the
 * row maximum before the arithmetic; remap == NULL is ivit_shiftgelu_build_lut */
int ivit_shiftgelu_build_lut_ex(float s, uint32_t m, int32_t e, const int8_t* remap, int8_t* lut /* [256*256] */,
                                ivit_stream_t stream);
/* ... then per call: wave-per-row max reduction + LDS-staged table row + byte gather.  L % 4 == 0. */
int ivit_shiftgelu_lut_i8(const int8_t* x, int64_t ldx, int rows, int L, const int8_t* lut, int8_t* out,
                          int64_t ldo, ivit_stream_t stream);
/* layouts bit 0: `out` is written in IVIT_LAYOUT_BLOCKS (L % 64 == 0, ldo == L, rows padded to 16); bit 1: `x` is read in
 * it (ldx == L).  In place (out == x) is allowed when both sides use the same layout. */
int ivit_shiftgelu_lut_i8_ex(const int8_t* x, int64_t ldx, int rows, int L, const int8_t* lut, int8_t* out,
                             int64_t ldo, int layouts, ivit_stream_t stream);

/* ---- stand-alone Shiftmax (module-level IVITIntSoftmax, ivit_modules.py:164-179) -----------------
 * x [rows, L] int8 with scale s -> out [rows, L] int8 in [0, 127] (scale 2^-7). L <= 1024. */
int ivit_shiftmax_i8(const int8_t* x, int64_t ldx, int rows, int L, float s, int8_t* out, int64_t ldo,
                     ivit_stream_t stream);
/* module-level LITERAL form, any input scale: x is the float view the reference's module receives (q*s, plus Swin's float
 * mask); every float32 step of ivit_modules.py:150-176 on x/s as written (the reference discards its .to(int32), :166),
 * the row sum in torch's CPU reduction order. */
int ivit_shiftmax_f32_i8(const float* x, int64_t ldx, int rows, int L, float s, int8_t* out, int64_t ldo,
                         ivit_stream_t stream);
/* the same with a wider output (IVITIntSoftmax(output_bit), the reference's softmax_bw knob, vit_quant.py:184): int16 values in
 * [0, 2^(output_bit-1) - 1], scale 2^-(output_bit-1) */
int ivit_shiftmax_f32_i16(const float* x, int64_t ldx, int rows, int L, float s, int output_bit, int16_t* out, int64_t ldo,
                          ivit_stream_t stream);
/* the same on int32 inputs, |x| < 2^28: Swin adds the shift mask (-100/s, beyond 8 bits) to the scores in front of
 * the softmax (swin_quant.py:151-156) */
int ivit_shiftmax_i32_i8(const int32_t* x, int64_t ldx, int rows, int L, float s, int8_t* out, int64_t ldo,
                         ivit_stream_t stream);

/* ---- generic QuantAct on integers (fixedpoint_mul, quant_utils.py:193-253) ---------------------
 * z [rows, C] int32; (m,e) per channel (n_me == C) or per tensor (n_me == 1); optional identity
 * branch z2 with (m2,e2) (n_me2 in {0,1,C}); bits in {8,16,32}; out int32 [rows, C]. */
int ivit_requant_i32(const int32_t* z, int64_t rows, int C, const uint32_t* m, const int32_t* e, int n_me,
                     const int32_t* z2, const uint32_t* m2, const int32_t* e2, int n_me2, int bits,
                     int32_t* out, ivit_stream_t stream);

/* int8 two-operand form (Block.qact2 / qact4 as a stand-alone op) */
int ivit_residual_requant_i8(const int8_t* a, uint32_t m_a, int32_t e_a, const int8_t* b, uint32_t m_b,
                             int32_t e_b, int8_t* out, int64_t n, ivit_stream_t stream);

/* ---- cls token + position embedding (vit_quant.py:290-296) -------------------------------------
 * patch [B*(T-1), C] int8 (PatchEmbed output); out [B*T, C] int8:
 *   out[b][0][:]   = cls_row[:]                    (constant, prepared by the caller)
 *   out[b][1+p][c] = clamp8(RNE(patch*m/2^e) + pos_add[1+p][c])
 * pos_add [T, C] int16 = RNE(qact_pos(pos_embed) * m_pos / 2^e_pos) prepared by the caller. */
int ivit_embed_assemble_i8(const int8_t* patch, const int16_t* pos_add, const int8_t* cls_row, uint32_t m,
                           int32_t e, int8_t* out, int batch, int tokens, int C, ivit_stream_t stream);

/* the same for a 16-bit patch embedding and block input (patch_embed_bw = block_input_bw = 16, vit_quant.py:180-187):
 * out16 = clamp16(RNE(patch16 * m / 2^e) + pos_add[tok][c]) for tok >= 1, out16[b][0] = cls_row; pos_add int32 [tokens, C]. */
int ivit_embed_assemble_i16(const int16_t* patch, const int32_t* pos_add, const int16_t* cls_row, uint32_t m,
                            int32_t e, int16_t* out, int batch, int tokens, int C, ivit_stream_t stream);

/* ---- classifier output (quant_modules.py:225-226; scripts/inference.py:249-250) ----------------
 * logits_f32[b][n] = fl(float(acc[b][n]) * s_acc[n]); top1[b] = argmax_n logits_f32 (first max).
 * logits_f32 may be NULL. */
int ivit_head_argmax(const int32_t* acc, const float* s_acc, int batch, int N, float* logits_f32,
                     int32_t* top1, ivit_stream_t stream);

/* ---- module-level QuantMatMul (quant_modules.py:404-409): batched int8 matmul -> int32 ----------
 * qk: S[b][i][j] = sum_d Q[b][i][d] * K[b][j][d];  pv: O[b][i][d] = sum_j P[b][i][j] * V[b][j][d].
 * All operands dense row-major per batch entry. */
int ivit_bgemm_qk_i8(const int8_t* Q, const int8_t* K, int32_t* S, int batch, int Tq, int Tk, int D,
                     ivit_stream_t stream);
int ivit_bgemm_pv_i8(const int8_t* P, const int8_t* V, int32_t* O, int batch, int Tq, int Tk, int D,
                     ivit_stream_t stream);
/* P of more than 8 bits.  CONTRACT: P is a softmax output -- every row of P sums to at most 2^15 + Tk (Shiftmax with output_bit
 * up to 16), so |acc| <= 128 * (2^15 + Tk) < 2^23; arbitrary int16 P would overflow the int32 accumulator from Tk = 512 on and is
 * not checked (for wider or unbounded P use ivit_bgemm_pv_i32_i8, which takes and checks a bound) */
int ivit_bgemm_pv_i16_i8(const int16_t* P, const int8_t* V, int32_t* O, int batch, int Tq, int Tk, int D,
                         ivit_stream_t stream);
/* P as int32 with |P| <= p_absmax (I-BERT's softmax with output_bit = 16 reaches 2^15 on a one-hot row, ibert_modules.py:314);
 * refused when p_absmax * 128 * Tk does not fit int32 */
int ivit_bgemm_pv_i32_i8(const int32_t* P, const int8_t* V, int32_t* O, int batch, int Tq, int Tk, int D, int64_t p_absmax,
                         ivit_stream_t stream);

/* ---- float <-> integer views at module edges (quant_utils.py:220; quant_modules.py:223,385-387) --
 * z = round(x / s[c]) (mode 0) or trunc(x / s[c]) (mode 1, the `.to(int32)` of ivit_modules.py:38,107);
 * y = float(z) * s[c].  n_s in {1, C}. */
int ivit_f32_to_i32(const float* x, int64_t rows, int C, const float* s, int n_s, int mode, int32_t* z,
                    ivit_stream_t stream);
int ivit_i32_to_f32(const int32_t* z, int64_t rows, int C, const float* s, int n_s, float* y,
                    ivit_stream_t stream);

/* int32 integer view -> int8 operand of the GEMM / matmul kernels (module-level path).  Values outside
 * [-128, 127] saturate and set *overflow_flag (device int32, may be NULL) to 1. */
int ivit_narrow_i32_i8(const int32_t* z, int8_t* out, int64_t n, int32_t* overflow_flag, ivit_stream_t stream);

/* =================================================================================================
 * Swin (models/swin_quant.py): 16-bit residual stream, windowed attention, patch merging, pooling.
 * The window partition, cyclic shift and their inverses (swin_quant.py:18-50, 258-271, 278-289) are row
 * permutations of a [B, H*W, C] tensor; kernels that take (H, W, ws, shift) apply
 *   win_row(b, y, x) = window-major index of token (b, (y - shift) mod H, (x - shift) mod W)
 * on the fly (ws == 0: identity), so no partitioned copy of the activations is ever materialised.
 * ================================================================================================= */

/* 8 -> 16 bit QuantAct (SwinTransformer.qact1, swin_quant.py:546): out = clamp16(RNE(x * m / 2^e)).
 * With (m, e) = (2^30, 30) it is the exact int8 -> int16 widening used behind PatchMerging. */
int ivit_requant_i8_i16(const int8_t* x, uint32_t m, int32_t e, int16_t* out, int64_t n, ivit_stream_t stream);

/* Two-operand 16-bit QuantAct of the residual connections (SwinTransformerBlock.qact2 / qact4,
 * swin_quant.py:293,299; quant_utils.py:232-245):
 *   out[r][c] = clamp16(RNE(k[win_row(r)][c] * m_a / 2^e_a) + RNE(res[r][c] * m_r / 2^e_r))
 * a_bits = 8 / 16: k = a (int8 / int16 [rows, C]);
 * a_bits = 32: a holds raw int32 accumulators of attn.proj and k = clamp16(RNE(a * m_pre[c] / 2^e_pre[c]))
 *   is the 16-bit attn.qact4 (swin_quant.py:166) fused in; (m_pre, e_pre) must be NULL otherwise.
 * The window map applies to `a` only (window_reverse + un-shift, swin_quant.py:278-289). C % 4 == 0. */
int ivit_residual_requant_i16(const void* a, int a_bits, const uint32_t* m_pre, const int32_t* e_pre, uint32_t m_a,
                              int32_t e_a, const int16_t* res, uint32_t m_r, int32_t e_r, int16_t* out, int64_t rows,
                              int C, int H, int W, int ws, int shift, ivit_stream_t stream);

/* I-LayerNorm on the 16-bit stream + 8-bit QuantAct (ivit_modules.py:30-65 with the Newton iteration on
 * float32(var), as the reference runs it); row r of x is written to row win_row(r) of out (stride ldo >= C):
 * norm1 writes straight into window order (swin_quant.py:258-271).  Same constants and contract as
 * ivit_layernorm_i8. */
int ivit_layernorm_i16_i8(const int16_t* x, int rows, int C, const float* bias_int, const float* s_ln,
                          const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, int H, int W, int ws, int shift,
                          ivit_stream_t stream);
/* the same for a 16-bit input carried at a natural scale s_in: the reference's LayerNorm sees fl(fl(q*s_in)/s_in) (float32 mean
 * over those values in torch's CPU reduction order, truncating .to(int32), ivit_modules.py:36-38).  fast_division = 1: the
 * caller has verified (exhaustively over the 65 536 inputs, prepare.markstein_division_ok) that the 3-instruction quotient by
 * the invariant s_in -- q0 = x*r, e = fma(-s, q0, x), fma(e, r, q0) with r = RN(1/s_in) -- is the correctly rounded one for this
 * s_in, which enables the tiled kernel; 0: the literal one-wave-per-row kernel with IEEE divisions.  fast_division |
 * IVIT_LN_OUTER_MEAN(L): the reference takes this mean over a transposed view of contiguous extent L (see
 * ivit_layernorm_f32_f32_ex: Swin's first norm1, whose input still carries the patch embedding's layout). */
int ivit_layernorm_i16_i8_compat(const int16_t* x, int rows, int C, float s_in, int fast_division, const float* bias_int,
                                 const float* s_ln, const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, int H, int W,
                                 int ws, int shift, ivit_stream_t stream);

/* PatchMerging gather (swin_quant.py:337-344): x [B, H*W, C] int16 -> out [B, (H/2)*(W/2), 4C],
 * channel blocks (even y, even x), (odd y, even x), (even y, odd x), (odd y, odd x). */
int ivit_patch_merge_i16(const int16_t* x, int16_t* out, int batch, int H, int W, int C, ivit_stream_t stream);

/* AdaptiveAvgPool1d over the tokens + qact3 (swin_quant.py:554-555):
 *   out[b][c] = clamp8(RNE(round(fl32(sum_t x[b][t][c] / tokens)) * m / 2^e)) */
int ivit_avgpool_requant_i8(const int8_t* x, int8_t* out, int batch, int tokens, int C, uint32_t m, int32_t e,
                            ivit_stream_t stream);

/* WindowAttention core (swin_quant.py:137-161): matmul_1 -> qact_attn1 -> + relative position bias through the
 * two-operand qact2 -> + shift mask -> Shiftmax -> matmul_2 -> qact3, one wave per (window, head).
 *   qkv      [3][windows][heads][tokens][32] int8 (ivit_gemm_i8_requant_qkv on window-ordered rows)
 *   out      [windows*tokens, ldo] int8, column h*32 + d
 *   bias_add [heads][tokens][64] int16 = RNE(qact_table(table)[index] * m2 / 2^e2) (key index padded to 64): the
 *            identity operand of qact2, a load-time constant
 *   mask_region [windows_per_image][64] uint8 or NULL (un-shifted block): region id of every token of a window of
 *            the rolled image (the img_mask of swin_quant.py:223-243); scores of (query, key) pairs from different
 *            regions get mask_value = -100 / s_attn (an integer in the supported regime) added after qact2's clamp,
 *            as the reference adds the float mask to the fake-quantised scores (:149-155, 243-246)
 *   (m_s,e_s): q.k^T -> qact_attn1;  (m_b,e_b): qact_attn1 -> qact2;  s_attn: scale of qact2 (Shiftmax input);
 *   (m_o,e_o): P.v -> qact3.
 * Supported: head_dim 32, 2 <= tokens <= 64. */
int ivit_window_attention_i8(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add,
                             const uint8_t* mask_region, int mask_value, int windows, int windows_per_image, int heads,
                             int tokens, int head_dim, uint32_t m_s, int32_t e_s, uint32_t m_b, int32_t e_b,
                             float s_attn, uint32_t m_o, int32_t e_o, ivit_stream_t stream);
/* Natural (non power-of-two) scale s_attn of the Shiftmax input: phi[q+128] = fl(fl(q*s)/s) is what the reference's Shiftmax
 * sees of an unmasked score q, phi_masked[q+128] = fl(fl(fl(q*s) - 100)/s) of a score under the shift mask
 * (swin_quant.py:151-156 adds float -100 to q*s; ivit_modules.py:165 divides by s); the kernel runs the float32 sequence of
 * ivit_modules.py:150-170 on those values per score.  Both tables float32 [256] on the device (prepare.py); both NULL =
 * ivit_window_attention_i8. */
int ivit_window_attention_i8_compat(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add,
                                    const uint8_t* mask_region, int mask_value, int windows, int windows_per_image, int heads,
                                    int tokens, int head_dim, uint32_t m_s, int32_t e_s, uint32_t m_b, int32_t e_b,
                                    float s_attn, uint32_t m_o, int32_t e_o, const float* phi, const float* phi_masked,
                                    ivit_stream_t stream);
/* Natural scale, table form (round 4): Shiftmax's exp_int of a score q under the row maximum qmax is read from
 * band[(qmax + 128) * band_w + min(qmax - q, band_w - 1)] (uint32 [256][band_w] on the device, band_w a multiple of 16 in [16, 192],
 * 16-byte aligned; prepare.shiftexp_band: the float32 sequence of ivit_modules.py:150-170 evaluated for every pair on the host, as
 * ivit_attention_fused_i8_compat_band does for ViT).  Scores under the shift mask take the saturated entry band_w - 1: the caller
 * hands the table over only when that is what the reference computes for every masked score and no masked score can be a row
 * maximum (prepare.window_shiftexp_band checks both; otherwise the literal form above runs).  band_rows = 256, or 1 when the rows
 * of the table do not depend on the maximum (the host compares them): `band` is then that one row [band_w] and the kernel runs
 * exactly as at a power-of-two scale, on these values.  ws != 0: output rows at their image positions as in
 * ivit_window_attention_i8_unwindow (H, W, ws, shift); ws == 0: window order. */
int ivit_window_attention_i8_band(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add, const uint8_t* mask_region,
                                  int windows, int windows_per_image, int heads, int tokens, int head_dim, uint32_t m_s, int32_t e_s,
                                  uint32_t m_b, int32_t e_b, float s_attn, uint32_t m_o, int32_t e_o, const uint32_t* band,
                                  int band_w, int band_rows, int H, int W, int ws, int shift, ivit_stream_t stream);
/* The same with the output rows at their IMAGE positions: window_reverse and the roll back (swin_quant.py:278-287) applied to
 * the row index here, `out` [batch * H * W, heads * head_dim].  attn.proj is row-wise, so it and the residual QuantAct behind it
 * then work on image-ordered rows without a map (ivit_gemm_i8_requant_i16_residual_i16_ex).  tokens == ws * ws, windows_per_image
 * == (H / ws) * (W / ws), 0 <= shift < ws. */
int ivit_window_attention_i8_unwindow(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add,
                                      const uint8_t* mask_region, int mask_value, int windows, int windows_per_image, int heads,
                                      int tokens, int head_dim, uint32_t m_s, int32_t e_s, uint32_t m_b, int32_t e_b,
                                      float s_attn, uint32_t m_o, int32_t e_o, const float* phi, const float* phi_masked,
                                      int H, int W, int ws, int shift, ivit_stream_t stream);

/* =================================================================================================
 * I-BERT operator family (models/quantization_utils/ibert_modules.py; registry key 'ibert', the fork's default,
 * vit_quant.py:188-190).  Module-level kernels: integer activations in, the module's output out.  The scalar
 * constants are the float32 values the reference computes on the host side of every call (floor(coef / scale) ...);
 * the caller passes them by value.  Each kernel performs the reference's float32 operations in the reference's
 * order (see csrc/ibert.hip).
 * ================================================================================================= */

/* IBERTIntGELU.forward (:220-235) on integers k (= x / scaling_factor):
 *   sigmoid_int = floor(sign(k) * ((min(|k|, -b_int) + b_int)^2 + c_int) / 2^6);  out = k * (sigmoid_int + shift_int)
 * out holds the integer x_int of :231; the module's float output is out * s_out (s_out computed by the caller, :232). */
int ivit_ibert_gelu_i32(const int32_t* k, int64_t n, float b_int, float c_int, float shift_int, int32_t* out,
                        ivit_stream_t stream);

/* IBERTIntSoftmax.forward (:297-319) on rows of L integers: int_exp (:285-295, n = 30) with x0_int, b_int, c_int and
 * scale exp_sf; the module's internal QuantAct(16) (self.act, :260,308) as the dyadic (m_act, e_act) = batch_frexp(
 * exp_sf / act_sf) with the fake-quant round trip through act_sf; row sum; factor = floor(2^32 / sum);
 * out = floor(exp_int * factor / 2^(32 - output_bit + 1)) in [0, 2^(output_bit-1)], scale 2 / 2^output_bit.
 * If exp_out != NULL only exp_int (float, [rows, L] dense) is produced: the statistics pass of the internal QuantAct
 * in calibration mode (out may then be NULL). */
int ivit_ibert_softmax_i32(const int32_t* k, int64_t ldx, int rows, int L, float x0_int, float b_int, float c_int,
                           float exp_sf, float act_sf, uint32_t m_act, int32_t e_act, int output_bit, int32_t* out,
                           int64_t ldo, float* exp_out, ivit_stream_t stream);

/* IBERTIntLayerNorm.forward (:112-158, use_int_sqrt = False): mean_int = round(sum / C); y = k - mean_int;
 * var = sum floor(y / 2^shift)^2; std = floor(sqrt(var)) * 2^shift; factor = floor(2^31 / std);
 * out[c] = (floor(y * factor / 2) + bias_int[c]) * s_out[c]  (float32, the module's output; s_out = sqrt(C)/2^30 * gamma). */
int ivit_ibert_layernorm_i32_f32(const int32_t* k, int64_t ldx, int rows, int C, const float* bias_int,
                                 const float* s_out, float shift_pow2, float* out, int64_t ldo, ivit_stream_t stream);

/* LITERAL forms of the three I-BERT operators: the float view x = q*s and its scale in, the reference's float32 sequence on
 * x / s itself (ibert_modules.py:126, 226, 303) -- any scale, not only the power-of-two ones for which x / s is the integer q;
 * row sums in torch's CPU reduction order.  Outputs are the modules' float outputs (GELU :234, Softmax :319, LayerNorm :153). */
int ivit_ibert_gelu_f32_f32(const float* x, int64_t n, float s, float b_int, float c_int, float shift_int, float s_out,
                            float* out, ivit_stream_t stream);
int ivit_ibert_softmax_f32_f32(const float* x, int64_t ldx, int rows, int L, float s, float x0_int, float b_int, float c_int,
                               float exp_sf, float act_sf, uint32_t m_act, int32_t e_act, int output_bit, float* out,
                               int64_t ldo, float* exp_out, ivit_stream_t stream);
int ivit_ibert_layernorm_f32_f32(const float* x, int64_t ldx, int rows, int C, const float* s_in, int n_s,
                                 const float* bias_int, const float* s_out, float shift_pow2, float* out, int64_t ldo,
                                 ivit_stream_t stream);

/* ---- the I-BERT family inside the fused int8 engine (engine.py, family "ibert"; ibert_modules.py:12-319) ---------------
 * int8 activations in and out; every operator runs the reference's float32 sequence on fl(q * s) (what its float tensors hold)
 * and fuses the QuantAct behind it, so power-of-two scales and scales as calibrated are covered alike.
 *  - ivit_ibert_gelu_build_lut: IBERTIntGELU (:203-235) + the 8-bit QuantAct `mlp.qact1` as a function of q, written as all
 *    256 rows of a (row max, q) table: ivit_shiftgelu_lut_i8(_ex) applies it.  (m_q, e_q) = dyadic(s_out / s_next).
 *  - ivit_ibert_softmax_build_table: exp_int after the softmax's internal 16-bit QuantAct (:303-310) as the float32 the
 *    reference sums and multiplies, for every (row max qm, q <= qm): table[(qm + 128) * 256 + q + 128], 65 536 floats.
 *  - ivit_attention_fused_i8_ibert: ivit_attention_fused_i8 with that softmax: row sum in float32 in torch's CPU reduction
 *    order, factor = floor(2^32 / sum), p = floor(fl(e * factor) / 2^25) in [0, 128] (output_bit 8, scale 2^-7).  tokens 193..207.
 *    band / band_w: the table in the band form of ivit_attention_fused_i8_compat_band (band[(qm + 128) * band_w + j] = entry of
 *    q = qm - j, entry band_w - 1 already the saturated value), staged in LDS per query tile; band_w = 0: gather from `table`.
 *  - ivit_ibert_layernorm_i8: IBERTIntLayerNorm (:126-153; mean and variance sums in torch's order) + the QuantAct behind it.
 *    bias_int / s_out / (m, e) as for ivit_layernorm_i8; shift_pow2 = 2^shift (the module's overflow buffer). */
int ivit_ibert_gelu_build_lut(float s, float b_int, float c_int, float shift_int, float s_out, uint32_t m_q, int32_t e_q,
                              int8_t* lut, ivit_stream_t stream);
int ivit_ibert_softmax_build_table(float s, float x0_int, float b_int, float c_int, float exp_sf, float act_sf, uint32_t m_act,
                                   int32_t e_act, float* table, ivit_stream_t stream);
int ivit_attention_fused_i8_ibert(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim, uint32_t m_s,
                                  int32_t e_s, uint32_t m_o, int32_t e_o, const float* table, const float* band, int band_w,
                                  int out_blocks, ivit_stream_t stream);
/* softmax_bits = 16: p = floor(fl32(e * factor) / 2^17) <= 2^15 at scale 2 / 2^16 (ibert_modules.py:314-317) */
int ivit_attention_fused_i8_ibert_wide(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim, uint32_t m_s,
                                       int32_t e_s, uint32_t m_o, int32_t e_o, const float* table, const float* band, int band_w,
                                       int softmax_bits, int out_blocks, ivit_stream_t stream);
int ivit_ibert_layernorm_i8(const int8_t* x, int64_t ldx, int rows, int C, float s_in, const float* bias_int, const float* s_out,
                            float shift_pow2, const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, int out_blocks,
                            ivit_stream_t stream);
/* the same on an int16 row (norm2_in_bw / att_block_out_bw = 16: the 16-bit residual stream); literal form, row-major output */
int ivit_ibert_layernorm_i16_i8(const int16_t* x, int64_t ldx, int rows, int C, float s_in, const float* bias_int,
                                const float* s_out, float shift_pow2, const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo,
                                ivit_stream_t stream);
/* fast_division = 1: the caller has checked (for all 65536 inputs at this s_in) that the three-instruction quotient by the
 * invariant s_in -- q0 = x * r, e = fma(-s, q0, x), fma(e, r, q0), r = RN(1 / s_in) -- is the correctly rounded x / s_in */
int ivit_ibert_layernorm_i16_i8_ex(const int16_t* x, int64_t ldx, int rows, int C, float s_in, const float* bias_int,
                                   const float* s_out, float shift_pow2, const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo,
                                   int fast_division, ivit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IVIT_HIP_H */
