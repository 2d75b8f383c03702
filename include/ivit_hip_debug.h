/*
 * ivit_hip_debug.h -- test and measurement hooks of libivit_hip.so.  NOT part of the drop-in boundary
 * (include/ivit_hip.h): process-wide, not thread-safe, for tests/ and scripts/ only.
 */
#ifndef IVIT_HIP_DEBUG_H
#define IVIT_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: ivit_gemm_i8_* pick between two kernels by problem size (a 256x128-tile LDS-DMA kernel
 * for M >= 2048, N >= 128; a 128x128-tile kernel otherwise).  on != 0 forces the small-tile kernel so
 * tests can cover both on the same inputs.  Process-wide, not thread-safe; not for production use. */
int ivit_debug_force_small_gemm(int on);
/* Perf-ablation hook for scripts/gemm_ablate.py (bit 0: skip the in-loop DMA, bit 1: skip the MFMAs,
 * bit 2: skip the epilogue, ...); results are WRONG whenever flags & 1023 != 0.  Bits that keep results correct
 * (A/B timing): 64 no start stagger, 1024 the relaunch-per-tile form instead of the persistent one, 2048 split a sparse last
 * round of tiles into half tiles, 4096 one workgroup per CU, 32768 per-CU turn-taking of the main loops, bits 16-21 start delay of the second co-resident workgroup in
 * ~1K-cycle units, bit 26 a 256-workgroup grid of the persistent kernel without the LDS blocker (two-stream probe). */
int ivit_debug_set_gemm_flags(int flags);
/* Second flag word: A/B of cache policies in the GEMM epilogue (results stay correct).  Bits 0-1: policy of the int8 output
 * stores (0 plain, 1 nt, 2 sc1, 3 sc0 sc1); bit 2: the residual operand is read with nt loads; bit 13: narrow (128 x 128) work items of
 * the weights-in-registers kernel (16x16x64 form; measured slower than its 128 x 256 tiles, scripts/gemm_narrow_ab.py); bit 15: the
 * wave-pipelined form of csrc/gemm_wp.h (one workgroup of eight waves per CU, a tile's epilogue inside the next tile's main loop;
 * plain and head-major int8 epilogues, K >= 768; exact, measured slower), bits 16-17 its timing ablations (results WRONG: 1 no
 * epilogue work inside the loop, 2 no barrier per K step); scripts/gemm_ab.py --frags16 ... -- 0:0 0:32768 0:98304; bit 20: no
 * skinny-K form (K <= 128, N <= 320, M >= 8192: the tile kernels instead; A/B and parity of both), bit 21: the skinny-K form's
 * run-time-K instantiation also for K = 64 / 128 (the form before the compile-time ones) */
int ivit_debug_set_gemm_flags2(int flags);
/* Diagnostic timeline buffer (8 x uint64 per workgroup) for the stamped build (flags = 512); scripts/gemm_timeline.py */
int ivit_debug_set_stamp_buffer(void* buf);

/* ivit_layernorm_i8 kernel form: 0 = automatic (the streaming kernel of ln_stream.h for C = 192 / 384 / 512 / 768 / 1024,
 * else half a wave per row for C <= 384, the grouped kernel up to 1024, a wave per row above), 1 = always a wave per row,
 * 2 = half a wave per row wherever it exists (C <= 1536), 3 = the automatic choice without the streaming kernel (rounds 2-3),
 * 4 = the streaming kernel wherever it applies, whatever the size: parity tests of every form, A/B timing */
int ivit_debug_ln_wave_per_row(int on);

/* streaming LayerNorm kernel, A/B timing (results stay correct): bits 0-3 ring depth (2, 3, 6; else the default 4; C = 768
 * only), bits 4-7 workgroups per CU (1-4; 0 = the default 2); scripts/ln_ablate.py */
int ivit_debug_ln_stream_cfg(int cfg);
/* wave timeline of the streaming LayerNorm kernel: 8 x uint64 per wave (s_memrealtime, 100 MHz: entry, table ready, slots 0-2 of
 * the first round computed, -, all stores done, groups of the wave); NULL = off; scripts/ln_timeline.py */
int ivit_debug_ln_stamp_buffer(void* buf);

/* timing ablations of the default int8 LayerNorm kernel (results WRONG when non-zero): 1 no element chain, 2 no row
 * statistics, 4 no stores, 8 no per-workgroup table build; correct results: bits 4-5 = 1 / 2 / 3 force groups of 8 rows (oversubscribed
 * grid) / 16 rows (one resident set of workgroups) / 4 rows, bit 6 odd waves start with half a group, bit 7 + bits 8-11 delayed start of every
 * other workgroup; bits 16-19 workgroup cap of the tiled 16-bit LayerNorm (x 256), bit 20 natural-scale 16-bit LayerNorm with its
 * row sums through LDS (the round-3 form; results stay correct), bits 21-22 = 1 / 2 / 3 the half-wave int8 kernel with 4 / 2 / 1 row
 * pairs per wave whatever the row count, bit 23 the Swin window attention requantises its scores in float64 also where the float32
 * form is exact (power-of-two multipliers), bit 24 the ShiftGELU table pass takes a whole wave per row also for rows of at most 384
 * bytes, bit 25 rows of at most 128 channels on the two-dword half-wave kernel (the form before round 4's one-dword one), bit 26
 * the one-dword kernel with 4 row pairs per wave also from 64 K rows, bit 27 the half-wave kernel's row sums by ds_bpermute butterflies
 * (the form before the DPP / v_permlane16_swap one), bit 28 the short-row ShiftGELU table pass with a prefetch of the next
 * iteration's rows (measured: no gain); scripts/ln_ablate.py, scripts/ln_ab.py --small, scripts/time_swin_kernels.py */
int ivit_debug_ln_ablate(int bits);

/* (lab library only since round 4: the engines never called it and it is slower than the pair it replaces)
 * EXPERIMENTAL -- measured SLOWER than the two launches it replaces ("ShiftGELU stays a pass of its own"); kept,
 * bit-exact and tested, as the record of that experiment; the engines do not call it.
 * mlp.fc1 + mlp.qact_gelu + ShiftGELU + mlp.qact1 in ONE launch (layers_quant.py:141-146; ivit_modules.py:105-126):
 *   k   = clamp8(RNE(acc * m[n] / 2^e[n]))                  as ivit_gemm_i8_requant_ex
 *   out = gelu_lut[(max_n k[t][:] + 128) * 256 + (k + 128)]   the table of ivit_shiftgelu_build_lut(_ex): ShiftGELU with the row
 *                                                             maximum over all N outputs of token t, requantised by mlp.qact1
 * = ivit_gemm_i8_requant_ex followed by ivit_shiftgelu_lut_i8_ex in place, bit for bit: every channel tile counts itself in at
 * its 128-token panel and the workgroup that completes a panel reads it back, takes the row maxima and maps it
 * (gemm_common.h: gelu_panel_phase).  Needs the weights-in-registers kernel: layouts contains IVIT_W_FRAGS16 (M >= 2048,
 * K % 192 == 0, N % 64 == 0, 128 <= N <= 4096), optionally IVIT_A_BLOCKS / IVIT_OUT_BLOCKS.
 * `workspace`: ivit_gemm_gelu_workspace_bytes(M) bytes = 4 * ceil(M / 128), owned by the caller, ZERO before the first launch;
 * every launch leaves it zero again (launches that share a workspace must not overlap). */
int ivit_gemm_gelu_workspace_bytes(int M, int64_t* bytes);
int ivit_gemm_i8_requant_gelu_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                 const uint32_t* m, const int32_t* e, const int8_t* gelu_lut, void* workspace,
                                 int8_t* out, int64_t ldo, int M, int N, int K, int layouts, ivit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IVIT_HIP_DEBUG_H */
