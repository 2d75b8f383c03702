// gemm_common.h -- shared pieces of the INT8 GEMM kernels: tile constants, argument block, LDS swizzle, the requantising
// int8 epilogue and the persistent-kernel helpers.  Included by gemm.hip (the product kernels) and gemm_lab.hip (kernel
// forms kept for A/B measurement and ablation).
#pragma once
#include <type_traits>

#include "common.h"

// Two builds of the same sources (csrc/Makefile):
//   libivit_hip.so      IVIT_LAB = 0: the product.  No process-wide state -- the knobs below are compile-time constants, the
//                       lab kernels (gemm_lab.hip) and the ivit_debug_* hooks are not in it;
//   libivit_hip_lab.so  IVIT_LAB = 1: the same entry points plus include/ivit_hip_debug.h (kernel-form A/B, ablations,
//                       time stamps), for tests/ and scripts/ only.
#if IVIT_LAB
// test / measurement state set through include/ivit_hip_debug.h (defined in gemm_lab.hip)
extern int g_kernel_choice;   // 0 = automatic, 1 = never the 256x256 kernel
extern bool g_force_small;    // route every problem through the small-tile kernel
extern void* g_stamp_buf;     // timeline buffer of the stamped builds
extern int g_debug_flags;     // see ivit_debug_set_gemm_flags
extern int g_debug_flags2;    // see ivit_debug_set_gemm_flags2 (cache-policy A/B of the weights-in-registers kernel)
#else
constexpr int g_kernel_choice = 0;
constexpr bool g_force_small = false;
constexpr int g_debug_flags = 0;
constexpr int g_debug_flags2 = 0;
#endif
#ifndef IVIT_STORE_POLICY
#define IVIT_STORE_POLICY 0    // product default of store16_sel (see there)
#endif

namespace {

constexpr int BM = 128;  // tokens per block
constexpr int BN = 128;  // channels per block
constexpr int BK = 64;   // K bytes per stage
constexpr int NT = 256;
constexpr int STAGE_BYTES = (BM + BN) * BK;  // 16 KiB
constexpr int W_OFF = BM * BK;               // weight tile behind the token tile
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;  // 32 KiB >= 128 * 132

// EPI_RESID16 (Swin mlp.fc2 + the 16-bit residual QuantAct): int8 requant as EPI_RQ, then
//   out16 = clamp16(RNE(k8 * M_main) + RNE(res16 * M_res)) with res / out int16 (res, out of GemmArgs reinterpreted)
// EPI_RQ16 (Swin attn.proj + attn.qact4): out16 = clamp16(RNE(acc * M[n])), int16 rows, straight from the accumulator registers
// (small-tile kernel only)
// EPI_RQ16_RES16 (ViT attn.proj / mlp.fc2 on a 16-bit residual stream: attn.qact3 / mlp.qact2 at 16 bits + the block's residual
//   QuantAct): k16 = clamp16(RNE(acc * M[n])), out16 = clamp16(RNE(k16 * M_main) + RNE(res16 * M_res)), straight from the
//   accumulator registers (weights-in-registers kernel only)
enum { EPI_RQ = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_I32 = 3, EPI_RESID16 = 4, EPI_RQ16 = 5, EPI_RQ16_RES16 = 6 };


struct GemmArgs {
    const int8_t* A;
    int64_t lda;
    const int8_t* W;
    int64_t ldw;
    const int32_t* bias;
    const uint32_t* m;
    const int32_t* e;
    void* out;
    int64_t ldo;
    const int8_t* res;
    int64_t ldr;
    double M_main, M_res;
    float Mf_main, Mf_res;   // EPI_RESID: float32 images of M_main / M_res, and
    int res_f32;             // 1: RNE(k * Mf) == RNE(k * M) for all 256 int8 k, both multipliers (verified by the launcher): the
                             //    residual QuantAct runs on two float32 fmas per output instead of two float64 ones
    int M, N, K;
    int tokens, heads, head_dim;
    int tiles_m, tiles_n;
    int flags;
    int stagger;  // number of first-generation blocks subject to the start stagger (0 = off)
    int cu_turns;       // persistent kernel: 1 = co-resident workgroups alternate main loops through the per-CU token
    int stagger_units;  // persistent kernel: start delay of the second co-resident workgroup, in s_sleep(16) (~1K cycle) units
    int split_from;  // persistent kernel: tiles [split_from, tiles_m*tiles_n) are processed as two half tiles each
    int narrow;      // weights-in-registers kernel (16x16x64 form): 1 = 128 token x 128 channel work items (gemm.hip wr_tile)
    int a_blocks, w_blocks;   // operand in the block layout (common.h: ivit_block_offset); persistent kernel only
    int out_blocks;           // EPI_RQ: the int8 output in the block layout (row length N): it is the next GEMM's A operand
    int w_frags;              // W is the MFMA-fragment copy (ivit_pack_weight_frags_i8): the weights-in-registers kernel
    unsigned long long* stamp;   // lab build only: timeline buffer of the stamped kernel forms (ivit_debug_set_stamp_buffer)
    int flags2;               // lab build only (ivit_debug_set_gemm_flags2): cache policies of the epilogue's stores / residual loads
    const int8_t* lut;        // EPI_RQ, weights-in-registers kernel: out = lut[q + 128] applied to every requantised byte (an
                              // elementwise int8 -> int8 operator behind the QuantAct, e.g. I-BERT GELU + mlp.qact1), or NULL
    // EPI_RQ, 16x16x64 weights-in-registers kernel, ABL bit 15 (ivit_gemm_i8_requant_gelu_ex, EXPERIMENTAL): ShiftGELU + mlp.qact1
    // behind the QuantAct, applied by the workgroup that completes a token panel (gelu_panel_phase)
    const int8_t* gelu_lut;   // [256][256] (row max + 128, k + 128) -> int8: the table of ivit_shiftgelu_build_lut_ex
    int* gelu_ws;             // caller-owned arrival counters of the 128-token panels [tiles_m]: zero before the first use, left
                              // zero by every launch
    unsigned tokens_magic;    // EPI_QKV, wave-pipelined kernel (gemm_wp.h): floor(2^32 / tokens) + 1 -- t / tokens == umulhi(t, magic) for
                              // every row index (the launcher checks M * tokens < 2^32): no per-lane division inside the main loop
};

IVIT_DEV int nk_of(const GemmArgs& g) { return g.K / 64; }

// byte offset of 16-byte chunk c (0..3) of tile row r; rows are 64 B, four rows per 256-B bank row.
IVIT_DEV int swz(int r, int c) { return r * BK + ((c ^ ((r >> 2) & 3)) << 4); }

// LDS reads the compiler does not see.  A kernel that has an LDS-DMA (global_load_lds) in flight pays `s_waitcnt vmcnt(0)` in
// front of every LDS read the compiler knows about (it cannot prove that the read does not alias the DMA's destination), i.e.
// the whole latency of a prefetch issued just before.  These reads are for data that the kernel's own barriers already
// ordered; completion is waited for with lds_wait(), whose "+v" operands tie every later use behind the wait.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
IVIT_DEV unsigned lds_addr(const void* p) { return (unsigned)(__UINTPTR_TYPE__)(lds_ptr_t)p; }
IVIT_DEV void lds_read16_async(v4f& d, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); }
IVIT_DEV void lds_read16_async(v4i& d, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); }
template <int OFF> IVIT_DEV void lds_read16_async_off(v4i& d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
template <int OFF> IVIT_DEV void lds_read16_async_off(v4f& d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
IVIT_DEV void lds_read8x2_async(v2i& d, unsigned a) { asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(d) : "v"(a)); }
IVIT_DEV void lds_wait(v4f& a, v4f& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
IVIT_DEV void lds_wait(v4i& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
IVIT_DEV void lds_wait(v2i& a, v2i& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }

// clamp16(RNE(p)) for ANY double p: the magic-number conversion alone reads the low 32 bits and wraps for |p| >= 2^31 (a per-channel
// multiplier above what a 16-bit output can hold); the reference saturates (torch.clamp on the float64 value, quant_utils.py:249)
IVIT_DEV int rne_clamp16(double p)
{
    p = __builtin_fmin(__builtin_fmax(p, -32768.0), 32767.0);
    return (int)(unsigned)__double_as_longlong(p + IVIT_MAGIC);
}

IVIT_DEV int pack4_i8(int a, int b, int c, int d)
{
    return (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | ((d & 0xff) << 24);
}


// Cache policy of the epilogue's 16-byte output stores (MI355X_MICROARCH.md, "stores of each flavour": plain / nt keep the line in
// the XCD's L2, sc1 / sc0 sc1 write through and drop it).  The output of a GEMM is consumed by the NEXT kernel, never by this one:
// keeping it in L2 only displaces the weight and token panels the other workgroups of the XCD are re-reading.
// POL: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1.  The asm stores are invisible to the compiler's vmcnt bookkeeping, which can only make its
// own counted waits wait for more (loads and stores retire in issue order).
template <int POL>
IVIT_DEV void store16_pol(void* p, int4 v)
{
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const v4i_ d = {v.x, v.y, v.z, v.w};
    // (lab A/B of cache policies only.  The s_nop: the data comes straight out of v_permlane16_swap / v_permlane32_swap in
    // epilogue_direct_16; the compiler's hazard recognizer puts wait states in front of its own VMEM instructions but does not look
    // into inline asm.)
    if constexpr (POL == 0) *reinterpret_cast<int4*>(p) = v;
    else if constexpr (POL == 1) asm volatile("s_nop 7\n\tglobal_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(d) : "memory");
    else if constexpr (POL == 2) asm volatile("s_nop 7\n\tglobal_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
    else asm volatile("s_nop 7\n\tglobal_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(d) : "memory");
}
IVIT_DEV void store16_sel(void* p, int4 v, int pol)
{
#if IVIT_LAB
    switch (pol & 3) {     // uniform
        case 1: store16_pol<1>(p, v); return;
        case 2: store16_pol<2>(p, v); return;
        case 3: store16_pol<3>(p, v); return;
        default: break;
    }
#endif
    store16_pol<IVIT_STORE_POLICY>(p, v);
}
// the residual operand is read exactly once: nt keeps it from displacing L2 lines that are re-read
IVIT_DEV int4 load16_sel(const void* p, int pol)
{
#if IVIT_LAB
    if (pol & 4) {
        typedef int v4i_ __attribute__((ext_vector_type(4)));
        const v4i_ d = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(p));
        return make_int4(d.x, d.y, d.z, d.w);
    }
#endif
    return *reinterpret_cast<const int4*>(p);
}

// ---- shared int8 epilogue --------------------------------------------------------------------
// acc[TI][TJ]: TI channel sub-tiles x TJ token sub-tiles of 32x32 owned by this wave, channel origin
// `wch`, token origin `wtok` inside a block tile of TOK tokens x 128 channels.
// Phase 1: per-channel requant -> int8, 4 channels per dword -> LDS tile Cs[token][channel].
// Phase 2: 16-byte row-contiguous chunks: optional residual QuantAct, optional head-major remap, store.
// Per-block table of the float32 neighbours (lo, hi) of each channel's requant multiplier, written once at
// kernel start (one thread per channel); visible to the epilogue through the main loop's barriers.
IVIT_DEV void fill_rq_table(const GemmArgs& g, char* rq_lds, int n0, int nch, int tid)
{
    if (tid < nch) {
        float2 lh = make_float2(0.f, 0.f);
        const int c = n0 + tid;
        if (c < g.N) {
            const double M = dyadic_mult(g.m[c], g.e[c]);
            const float mf = (float)M;
            const double back = (double)mf;
            const int bits = __float_as_int(mf);
            lh.x = (back > M) ? __int_as_float(bits - 1) : mf;  // largest float32 <= M
            lh.y = (back < M) ? __int_as_float(bits + 1) : mf;  // smallest float32 >= M
        }
        reinterpret_cast<float2*>(rq_lds)[tid] = lh;
    }
}

struct NoHook {
    IVIT_DEV void issue() const {}
    IVIT_DEV void consume() const {}
};

// Phase 2 of the int8 epilogues: the staged tile Cs[token][channel] (row stride CH + 4) -> 16-byte row-contiguous chunks: optional
// residual QuantAct, optional head-major remap or byte map, store.  Called by every thread right after its phase-1 LDS writes.
template <int EPI, int TOK, int NTHREADS, int ABL, int CH, typename Hook>
IVIT_DEV void epilogue_phase2(const GemmArgs& g, char* smem, int m0, int n0, int tid, const Hook& hook, const unsigned char* lut_lds,
                              unsigned long long* st = nullptr)
{
    constexpr int CSS = CH + 4;
    constexpr int CPR = CH / 16;
    unsigned long long t_p1 = 0, t_sync = 0;
    if constexpr (ABL & (512 | 2048)) t_p1 = __builtin_amdgcn_s_memtime();
    // Phase 2 work items of this thread: NIT chunks (token row tl, 16-byte column chunk cc).  The residual loads go out BEFORE the
    // barrier (the accumulators are dead, their registers free), so their latency runs under the barrier wait and the LDS reads.
    constexpr int NIT = TOK * CPR / NTHREADS;
    int4 rv[NIT];
    int4 rw[EPI == EPI_RESID16 ? NIT : 1][2];   // 16 int16 residual values per chunk
    // A tile that lies inside the matrix (all but the last token panel / channel tile) needs no per-chunk clamping or bounds test,
    // and its chunk addresses are one base plus a uniform step per `it` (the thread's rows are NTHREADS / CPR apart): the address
    // arithmetic and the exec-mask juggling were ~20 of the ~30 instructions a chunk of the plain int8 epilogue costs, at one
    // VALU instruction per ~8 cycles beside the co-resident workgroup's MFMA stream.
    const bool interior = (m0 + TOK <= g.M) && (n0 + CH <= g.N);     // uniform
    constexpr int RPI = NTHREADS / CPR;                              // rows between a thread's consecutive chunks
    static_assert(NTHREADS % CPR == 0, "a thread keeps its chunk column");
    if constexpr (EPI == EPI_RESID || EPI == EPI_RESID16) {
        if constexpr (!(ABL & 16)) {
            const int8_t* rbase = g.res + (int64_t)(m0 + tid / CPR) * g.ldr + (n0 + 16 * (tid % CPR));
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = tid + NTHREADS * it;
                const int tl = q / CPR, cc = q % CPR;
                const int t = min(m0 + tl, g.M - 1), cn = min(n0 + 16 * cc, g.N - 16);
                if constexpr (EPI == EPI_RESID) {
                    rv[it] = load16_sel(interior ? rbase + (int64_t)(it * RPI) * g.ldr : g.res + (int64_t)t * g.ldr + cn, g.flags2);
                } else {
                    const int4* rp = reinterpret_cast<const int4*>(reinterpret_cast<const int16_t*>(g.res) + (int64_t)t * g.ldr + cn);
                    rw[it][0] = rp[0];
                    rw[it][1] = rp[1];
                }
            }
        }
    }
    // barrier without the vmcnt(0) drain of __syncthreads(): only this wave's LDS writes have to be complete
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (ABL & 512) {
        t_sync = __builtin_amdgcn_s_memtime();
        if (tid == 0 && g.res != nullptr) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<int8_t*>(g.res)) + 8ull * blockIdx.x;
            d[4] = t_p1; d[5] = t_sync;
        }
    }
    if constexpr ((ABL & 16) && !(ABL & 2048)) return;
    if constexpr (ABL & 2048) { st[4] = t_p1; st[5] = __builtin_amdgcn_s_memtime(); }

    int8_t* out = reinterpret_cast<int8_t*>(g.out);
    int v[NIT][4];
    v2i vv[NIT][2];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = tid + NTHREADS * it;
        const int tl = q / CPR, cc = q % CPR;
        lds_read8x2_async(vv[it][0], lds_addr(smem) + (unsigned)(tl * CSS + 16 * cc));
        lds_read8x2_async(vv[it][1], lds_addr(smem) + (unsigned)(tl * CSS + 16 * cc + 8));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        lds_wait(vv[it][0], vv[it][1]);
        v[it][0] = vv[it][0].x; v[it][1] = vv[it][0].y; v[it][2] = vv[it][1].x; v[it][3] = vv[it][1].y;
    }
    hook.consume();
    if constexpr (ABL & 2048) st[6] = __builtin_amdgcn_s_memtime();
    const bool oblk = (EPI == EPI_RQ) && g.out_blocks;
    const int64_t off0 = oblk ? (int64_t)block_off(block_row(m0 + tid / CPR, g.N), block_col(n0 + 16 * (tid % CPR)))
                              : (int64_t)(m0 + tid / CPR) * g.ldo + (n0 + 16 * (tid % CPR));
    const int64_t ostep = oblk ? (int64_t)(RPI / 16) * (g.N >> 6) * 1024 : (int64_t)RPI * g.ldo;     // uniform
    float magic_v = 12582912.0f;      // 1.5 * 2^23 in a VGPR (the residual form's fmas take their multiplier from an SGPR)
    asm volatile("" : "+v"(magic_v));
    // EPI_QKV addressing state (see below)
    constexpr int qkv_rows_per_it = NTHREADS / CPR;
    int qkv_b = 0, qkv_tok = 0;
    int qkv_col = 0;
    if constexpr (EPI == EPI_QKV) {
        const int cn0 = min(n0 + 16 * (tid % CPR), g.N - 16);
        const int cdim = g.heads * g.head_dim;
        const int which = cn0 / cdim, rem = cn0 - which * cdim;
        const int hh = rem / g.head_dim, d0 = rem - hh * g.head_dim;
        const int nb = g.M / g.tokens;
        qkv_col = ((which * nb * g.heads + hh) * g.tokens) * g.head_dim + d0;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = tid + NTHREADS * it;
        const int tl = q / CPR, cc = q % CPR;
        const int t = m0 + tl, cn = n0 + 16 * cc;
        if constexpr (EPI == EPI_QKV) {
            if (it > 0 && qkv_rows_per_it < g.tokens) {   // the row advanced by qkv_rows_per_it: at most one image boundary
                qkv_tok += qkv_rows_per_it;
                if (qkv_tok >= g.tokens) {
                    qkv_tok -= g.tokens;
                    ++qkv_b;
                }
            }
        }
        if (!interior && (t >= g.M || cn >= g.N)) continue;
        if constexpr (EPI == EPI_RESID) {
            const int rr[4] = {rv[it].x, rv[it].y, rv[it].z, rv[it].w};
            if constexpr (ABL & 4096) {   // kernel form chosen by the launcher when g.res_f32.  RNE(k * M) as ONE float32 fma against 1.5 * 2^23 per product: the float's low bits are the
                               // integer, and the launcher has checked all 256 int8 inputs of both multipliers against the
                               // float64 evaluation (the epilogue's VALU instructions issue at ~1 per 8 cycles beside the
                               // co-resident workgroup's MFMA stream: 11 -> 7 instructions per output byte)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned o[4];
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const float kf = (float)(int)(int8_t)(v[it][d] >> (8 * bb));
                        const float xf = (float)(int)(int8_t)(rr[d] >> (8 * bb));
                        // plain v_fma_f32 through asm: left to itself the compiler packs the pair into v_pk_fma_f32, which costs more
                        // than two scalar fmas beside an MFMA stream (cdna guide, 'packed f32 VALU')
                        float f1, f2;
                        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(kf), "s"(g.Mf_main), "v"(magic_v));
                        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(f2) : "v"(xf), "s"(g.Mf_res), "v"(magic_v));
                        // quant_utils.py:229-245: two rounded products, then the sum; + 128 so that the clamp leaves an unsigned byte
                        o[bb] = (unsigned)clamp_i32((int)((unsigned)__float_as_int(f1) + (unsigned)__float_as_int(f2) - 2u * 0x4B400000u), -128, 127);
                    }
                    const unsigned w01 = __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0400u);
                    const unsigned w23 = __builtin_amdgcn_perm(o[3], o[2], 0x04000c0cu);
                    v[it][d] = (int)(w01 | w23);
                }
            } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int o[4];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    int k3 = (int)(int8_t)(v[it][d] >> (8 * bb));
                    int xr = (int)(int8_t)(rr[d] >> (8 * bb));
                    // quant_utils.py:229-245: two independently rounded products, then the sum
                    int sres = requant_exact(k3, g.M_main) + requant_exact(xr, g.M_res);
                    o[bb] = clamp_i32(sres, -128, 127);
                }
                v[it][d] = pack4_i8(o[0], o[1], o[2], o[3]);
            }
            }
        }
        if constexpr (EPI == EPI_RESID16) {
            // quant_utils.py:232-245 with a 16-bit output range (swin_quant.py:299): two independently rounded products
            const int rr[8] = {rw[it][0].x, rw[it][0].y, rw[it][0].z, rw[it][0].w, rw[it][1].x, rw[it][1].y, rw[it][1].z, rw[it][1].w};
            int ow[8];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int o[4];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int idx = 4 * d + bb;                                   // channel within the chunk
                    const int k8 = (int)(int8_t)(v[it][d] >> (8 * bb));
                    const int xr = (idx & 1) ? (rr[idx >> 1] >> 16) : (int)(int16_t)rr[idx >> 1];
                    o[bb] = clamp_i32(requant_exact(k8, g.M_main) + requant_exact(xr, g.M_res), -32768, 32767);
                }
                ow[2 * d] = (o[0] & 0xffff) | (o[1] << 16);
                ow[2 * d + 1] = (o[2] & 0xffff) | (o[3] << 16);
            }
            int4* op = reinterpret_cast<int4*>(reinterpret_cast<int16_t*>(g.out) + (int64_t)t * g.ldo + cn);
            op[0] = make_int4(ow[0], ow[1], ow[2], ow[3]);
            op[1] = make_int4(ow[4], ow[5], ow[6], ow[7]);
            continue;
        }
        if constexpr (EPI == EPI_RQ) {
            if (lut_lds) {      // uniform: a 256-entry int8 -> int8 map (LDS) over the 16 bytes of the chunk
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned u = (unsigned)v[it][d] ^ 0x80808080u;      // q + 128 per byte
                    v[it][d] = (int)((unsigned)lut_lds[u & 255] | ((unsigned)lut_lds[(u >> 8) & 255] << 8) |
                                     ((unsigned)lut_lds[(u >> 16) & 255] << 16) | ((unsigned)lut_lds[u >> 24] << 24));
                }
            }
        }
        int64_t off;
        if constexpr (EPI == EPI_QKV) {
            // (which, head, d0) depend on the thread's chunk column only (q % CPR == tid % CPR for every it) and the image /
            // token of a row advances by NTHREADS / CPR rows per it: integer divisions once per tile, not once per chunk
            // (they were ~100 VALU instructions per chunk, as much as phase 1)
            if (it == 0 || qkv_rows_per_it >= g.tokens) {
                qkv_b = t / g.tokens;
                qkv_tok = t - qkv_b * g.tokens;
            }
            off = (int64_t)(unsigned)(qkv_col + ((qkv_b * g.heads * g.tokens + qkv_tok) * g.head_dim));   // 32-bit: the launcher checks 3*M*heads*head_dim < 2^31
        } else if (interior && RPI % 16 == 0) {     // base + uniform step (a block row is 16 token rows: RPI / 16 block rows per `it`)
            off = off0 + (int64_t)it * ostep;
        } else {
            off = (EPI == EPI_RQ && g.out_blocks) ? (int64_t)block_off(block_row(t, g.N), block_col(cn)) : (int64_t)t * g.ldo + cn;
        }
        store16_sel(out + off, make_int4(v[it][0], v[it][1], v[it][2], v[it][3]), g.flags2);
    }
}

template <int EPI, int TI, int TJ, int TOK, int NTHREADS, int ABL = 0, int CH = 128, typename Hook = NoHook>
IVIT_DEV void epilogue_i8(v16i (&acc)[TI][TJ], const GemmArgs& g, char* smem, const char* rq_lds, int m0, int n0,
                          int wch, int wtok, int tid, int h, int l31, const Hook& hook = Hook(),
                          const unsigned char* lut_lds = nullptr)
{
    // The epilogue is a short VALU burst next to the co-resident workgroup's MFMA stream: give it issue priority
    // so its dependent chains do not wait behind queued MFMAs (which run in the matrix pipe once issued).
    __builtin_amdgcn_s_setprio(2);
    hook.issue();    // persistent kernels: the next tile's table loads have all of phase 1 to land
    constexpr int CSS = CH + 4;       // LDS row stride: (CH/4 + 1) dwords, odd -> conflict-free dword writes
    constexpr int CPR = CH / 16;      // 16-byte chunks per row
    // Phase 1.  out = clamp8(RNE(acc * M)), M = m * 2^-e, must equal the reference's float64
    // evaluation (quant_utils.py:229-230) bit for bit.  Fast path on the ordinary float32 VALU (the
    // float64 ops contend with the MFMA pipe): with lo <= M <= hi the two float32 neighbours of M,
    //   t_lo = fma(acc, lo, 1.5*2^23), t_hi = fma(acc, hi, 1.5*2^23)
    // are RNE(acc*lo) and RNE(acc*hi) exactly (one rounding, ulp 1), and RNE is monotone, so
    // t_lo == t_hi certifies RNE(acc*M) -- including exact ties, which straddle and fail the test.
    // Valid while acc is exact in float32 and |acc*hi| < 2^22 (M <= 1 is part of the contract), i.e.
    // |acc| < 2^22; anything else, and any failed certificate, takes the float64 path for that quad.
    // (lo, hi) of the quad's four channels: read one batch ahead (double-buffered), invisible to the compiler (see lds_read16_async)
    const unsigned rq_a = lds_addr(rq_lds) + 8u * (unsigned)(wch + 4 * h);
    v4f lhbuf[2][2];
    lds_read16_async(lhbuf[0][0], rq_a);
    lds_read16_async(lhbuf[0][1], rq_a + 16u);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = wch + 32 * i + 8 * q + 4 * h;  // local channel of the quad
            v4f& lh01 = lhbuf[(4 * i + q) & 1][0];      // lo0 hi0 lo1 hi1
            v4f& lh23 = lhbuf[(4 * i + q) & 1][1];      // lo2 hi2 lo3 hi3
            lds_wait(lh01, lh23);
            if (4 * i + q + 1 < 4 * TI) {
                const int nb = 4 * i + q + 1;
                lds_read16_async(lhbuf[nb & 1][0], rq_a + 8u * (unsigned)(32 * (nb >> 2) + 8 * (nb & 3)));
                lds_read16_async(lhbuf[nb & 1][1], rq_a + 8u * (unsigned)(32 * (nb >> 2) + 8 * (nb & 3)) + 16u);
            }
            const float lo[4] = {lh01.x, lh01.z, lh23.x, lh23.z};
            const float hi[4] = {lh01.y, lh01.w, lh23.y, lh23.w};
            // one branch-free batch of TJ*4 independent chains (instruction-level parallelism: the wave that
            // runs this shares its SIMD with a main-loop wave, so there is no second VALU wave to hide latency)
            int b[TJ][4];
            unsigned unc = 0;      // OR of (t_lo ^ t_hi): non-zero <=> some certificate failed
            float amax = 0.0f;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if constexpr (ABL & 8) {
                        b[j][jj] = acc[i][j][4 * q + jj] + (int)lo[jj];
                    } else {
                        const float a = (float)acc[i][j][4 * q + jj];
                        int tl, th;
                        if constexpr (ABL & 64) {   // A/B: both brackets in one packed fma (lo, hi are adjacent table entries)
                            typedef float v2f_ __attribute__((ext_vector_type(2)));
                            const v2f_ r = __builtin_elementwise_fma((v2f_){a, a}, (v2f_){lo[jj], hi[jj]}, (v2f_){12582912.0f, 12582912.0f});
                            tl = __float_as_int(r.x);
                            th = __float_as_int(r.y);
                        } else {
                            tl = __float_as_int(__builtin_fmaf(a, lo[jj], 12582912.0f));
                            th = __float_as_int(__builtin_fmaf(a, hi[jj], 12582912.0f));
                        }
                        // unc += |tl - th| in ONE instruction (v_sad_u32): zero iff every certificate of the batch holds.
                        // tl, th are bit patterns of floats next to 1.5 * 2^23, their differences are tiny: no wrap-around.
                        if constexpr (ABL & 32) {   // A/B: the former two-instruction form
                            unc |= (unsigned)(tl ^ th);
                            asm volatile("" : "+v"(unc));
                        } else {
                            asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                        }
                        amax = fmaxf(amax, fabsf(a));
                        b[j][jj] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);  // low byte = int8 result
                    }
                }
            if constexpr (!(ABL & 8)) {
                const bool bad = (unc != 0) | (amax >= 4194304.0f);
                if (__builtin_amdgcn_ballot_w64(bad) != 0) {  // rare: exact float64 evaluation of the batch
                    const int c0 = min(n0 + cl, g.N - 4);
                    const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
                    const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
                    const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z),
                                          dyadic_mult(m4.w, e4.w)};
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            // quant_utils.py:229-230: float64 product (53-bit rounding), /2^e, round-half-even
                            double p = (double)acc[i][j][4 * q + jj] * Mc[jj];
                            double t = p + IVIT_MAGIC;
                            b[j][jj] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int tl_ = wtok + 32 * j + l31;
                const unsigned w01 = __builtin_amdgcn_perm((unsigned)b[j][1], (unsigned)b[j][0], 0x0c0c0400u);
                const unsigned w23 = __builtin_amdgcn_perm((unsigned)b[j][3], (unsigned)b[j][2], 0x04000c0cu);
                *reinterpret_cast<unsigned*>(smem + tl_ * CSS + cl) = w01 | w23;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    epilogue_phase2<EPI, TOK, NTHREADS, ABL, CH, Hook>(g, smem, m0, n0, tid, hook, lut_lds);
}

// ---- the int8 epilogue for accumulators of v_mfma_i32_16x16x64_i8 (the weights-in-registers kernel's S16 form).
// acc[i][j]: channel sub-tile i (16 channels) x token sub-tile j (16 tokens); lane (g4 = lane >> 4, l15 = lane & 15) holds token
// 16 j + l15 and the four consecutive channels wch + 16 i + 4 g4 + r of register r -- again one dword of four channels per
// token, so phase 1 is the arithmetic of epilogue_i8 with other loop bounds (batches of 16 outputs per certificate ballot) and
// phase 2 is shared.
template <int EPI, int NJ, int NTHREADS, int ABL, int CH, typename Hook>
IVIT_DEV void epilogue_i8_16(v4i (&acc)[4][NJ], const GemmArgs& g, char* smem, const char* rq_lds, int m0, int n0, int wch, int tid,
                             int g4, int l15, const Hook& hook, const unsigned char* lut_lds, unsigned long long* st = nullptr)
{
    static_assert(NJ % 4 == 0, "batches of four token sub-tiles");
    __builtin_amdgcn_s_setprio(2);
    hook.issue();
    constexpr int CSS = CH + 4;
    const unsigned rq_a = lds_addr(rq_lds) + 8u * (unsigned)(wch + 4 * g4);
    v4f lhbuf[2][2];
    lds_read16_async(lhbuf[0][0], rq_a);
    lds_read16_async(lhbuf[0][1], rq_a + 16u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cl = wch + 16 * i + 4 * g4;   // local channel of the quad
        v4f& lh01 = lhbuf[i & 1][0];            // lo0 hi0 lo1 hi1
        v4f& lh23 = lhbuf[i & 1][1];            // lo2 hi2 lo3 hi3
        lds_wait(lh01, lh23);
        if (i + 1 < 4) {
            lds_read16_async(lhbuf[(i + 1) & 1][0], rq_a + 8u * (unsigned)(16 * (i + 1)));
            lds_read16_async(lhbuf[(i + 1) & 1][1], rq_a + 8u * (unsigned)(16 * (i + 1)) + 16u);
        }
        const float lo[4] = {lh01.x, lh01.z, lh23.x, lh23.z};
        const float hi[4] = {lh01.y, lh01.w, lh23.y, lh23.w};
#pragma unroll
        for (int jb = 0; jb < NJ; jb += 4) {
            // The table holds brackets WIDENED by two float32 steps on either side of M (table_write of the S16 kernel): then
            // a' * lo <= acc * M <= a' * hi also for the a' = fl(acc) of an accumulator beyond 2^24 ((1 - 2^-24)(1 + 2^-23) > 1), and
            // no range test is needed: products beyond the int8 range saturate the clamp on either side whatever their rounding
            // (the bit pattern of t is monotone in acc * M over the whole int32 range), so a certificate that fails there only
            // sends the batch to the exact path.
            int b[4][4];
            unsigned unc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = (float)acc[i][jb + j][r];
                    const int tl = __float_as_int(__builtin_fmaf(a, lo[r], 12582912.0f));
                    const int th = __float_as_int(__builtin_fmaf(a, hi[r], 12582912.0f));
                    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                    b[j][r] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);  // low byte = int8 result
                }
            const bool bad = unc != 0;
            if (__builtin_amdgcn_ballot_w64(bad) != 0) {  // rare: exact float64 evaluation of the batch (quant_utils.py:229-230)
                const int c0 = min(n0 + cl, g.N - 4);
                const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
                const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
                const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z), dyadic_mult(m4.w, e4.w)};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double t = (double)acc[i][jb + j][r] * Mc[r] + IVIT_MAGIC;
                        b[j][r] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                    }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tl_ = 16 * (jb + j) + l15;
                const unsigned w01 = __builtin_amdgcn_perm((unsigned)b[j][1], (unsigned)b[j][0], 0x0c0c0400u);
                const unsigned w23 = __builtin_amdgcn_perm((unsigned)b[j][3], (unsigned)b[j][2], 0x04000c0cu);
                *reinterpret_cast<unsigned*>(smem + tl_ * CSS + cl) = w01 | w23;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    epilogue_phase2<EPI, 16 * NJ, NTHREADS, ABL, CH, Hook>(g, smem, m0, n0, tid, hook, lut_lds, st);
}

// ---- the same epilogue WITHOUT the LDS round trip (EPI_RQ row-major or block layout, EPI_RESID, EPI_QKV).
// After phase 1 a lane (g4, l15) holds, for token 16 j + l15, one dword per channel sub-tile i: bytes 16 i + 4 g4 .. + 3 of the
// wave's 64 channels.  The four lanes of a token (g4 = 0..3) hold a 4 x 4 matrix of dwords [g4][i]; transposed across those lanes --
// two v_permlane32_swap, two v_permlane16_swap, as in attention.hip's output path -- lane g4 ends with the 16 CONTIGUOUS bytes
// 16 g4 .. 16 g4 + 15 of the token's 64-byte row segment: one 16-byte store per lane and token sub-tile, a wave instruction covers
// 16 rows x 64 bytes (one whole 1 KB block of the block layout; one (image, head) row group of the head-major q/k/v layout).
// The staged form spent 32 ds_write_b32, 16 ds_read2_b32, a workgroup barrier (1.5-2 K cycles of skew in the residual form) and
// per-chunk address arithmetic on the same bytes; its VALU and LDS instructions issue at one per ~8 cycles beside the co-resident
// workgroup's MFMA stream (profiles/r03g_wreg16_timeline.txt, r03i_epilogue_probe.txt), so what is not issued is what is saved.
// The residual QuantAct works on the transposed chunk against a 16-byte residual load of the same row segment.
IVIT_DEV void residual_chunk_f32(int (&v)[4], const int4& res, float Mf_main, float Mf_res, float magic_v)
{
    const int rr[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned o[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float kf = (float)(int)(int8_t)(v[d] >> (8 * bb));
            const float xf = (float)(int)(int8_t)(rr[d] >> (8 * bb));
            float f1, f2;      // plain v_fma_f32 through asm: see epilogue_phase2
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(kf), "s"(Mf_main), "v"(magic_v));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(f2) : "v"(xf), "s"(Mf_res), "v"(magic_v));
            o[bb] = (unsigned)clamp_i32((int)((unsigned)__float_as_int(f1) + (unsigned)__float_as_int(f2) - 2u * 0x4B400000u), -128, 127);
        }
        v[d] = (int)(__builtin_amdgcn_perm(o[1], o[0], 0x0c0c0400u) | __builtin_amdgcn_perm(o[3], o[2], 0x04000c0cu));
    }
}

IVIT_DEV void residual_chunk_f64(int (&v)[4], const int4& res, double M_main, double M_res)
{
    const int rr[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int o[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const int k3 = (int)(int8_t)(v[d] >> (8 * bb));
            const int xr = (int)(int8_t)(rr[d] >> (8 * bb));
            o[bb] = clamp_i32(requant_exact(k3, M_main) + requant_exact(xr, M_res), -128, 127);   // quant_utils.py:229-245
        }
        v[d] = pack4_i8(o[0], o[1], o[2], o[3]);
    }
}

template <int EPI, int NJ, int ABL, typename Hook>
IVIT_DEV void epilogue_direct_16(v4i (&acc)[4][NJ], const GemmArgs& g, const char* rq_lds, int m0, int n0, int wch, int g4, int l15,
                                 const Hook& hook, const unsigned char* lut_lds, unsigned long long* st = nullptr)
{
    static_assert(NJ % 4 == 0 && (EPI == EPI_RQ || EPI == EPI_RESID || EPI == EPI_QKV), "direct epilogue: int8 outputs");
    __builtin_amdgcn_s_setprio(2);
    hook.issue();
    const unsigned rq_a = lds_addr(rq_lds) + 8u * (unsigned)(wch + 4 * g4);
    const int ncol = n0 + wch;                      // first of this wave's 64 channels (uniform)
    const bool col_ok = ncol < g.N;                 // N % 64 == 0: a wave is in or out as a whole
    const int c = ncol + 16 * g4;                   // this lane's 16-byte chunk column
    // Phase A: per batch of four token sub-tiles, requantise (all four channel sub-tiles) and pack; the batch's accumulators are
    // dead then, and its residual chunks are requested into the registers they leave.  Phase B: transpose, residual QuantAct,
    // store -- batch 0 runs under the flight of batch 1's residual loads; the next work item's table (hook.consume: vmcnt(0) for its
    // loads, issued before this epilogue) is written between the two batches.
    constexpr int NB = NJ / 4;
    constexpr bool GELU = (ABL & 32768) != 0;     // the tile is read back by the workgroup that completes its panel (gelu_panel_phase)
    static_assert(!GELU || EPI == EPI_RQ, "the fused ShiftGELU follows a plain requantising epilogue");
    unsigned D[NB][4][4];     // [batch][i][j]: bytes 16 i + 4 g4 .. + 3 of token 16 (4 batch + j) + l15
    int4 rv[EPI == EPI_RESID ? NB : 1][4];
    v4f lhbuf[2][2];
    lds_read16_async(lhbuf[0][0], rq_a);
    lds_read16_async(lhbuf[0][1], rq_a + 16u);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        const int jb = 4 * bi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sidx = jb + i;                        // position in the sequence of table reads
            v4f& lh01 = lhbuf[sidx & 1][0];                 // lo0 hi0 lo1 hi1
            v4f& lh23 = lhbuf[sidx & 1][1];                 // lo2 hi2 lo3 hi3
            lds_wait(lh01, lh23);
            if (sidx + 1 < NJ) {
                const unsigned nxt = rq_a + 8u * (unsigned)(16 * ((i + 1) & 3));
                lds_read16_async(lhbuf[(sidx + 1) & 1][0], nxt);
                lds_read16_async(lhbuf[(sidx + 1) & 1][1], nxt + 16u);
            }
            const float lo[4] = {lh01.x, lh01.z, lh23.x, lh23.z};
            const float hi[4] = {lh01.y, lh01.w, lh23.y, lh23.w};
            int b[4][4];
            unsigned unc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = (float)acc[i][jb + j][r];
                    const int tl = __float_as_int(__builtin_fmaf(a, lo[r], 12582912.0f));
                    const int th = __float_as_int(__builtin_fmaf(a, hi[r], 12582912.0f));
                    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                    b[j][r] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);  // low byte = int8 result
                }
            if (__builtin_amdgcn_ballot_w64(unc != 0) != 0) {  // rare: exact float64 evaluation of the batch (quant_utils.py:229-230)
                const int c0 = min(ncol + 16 * i + 4 * g4, g.N - 4);
                const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
                const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
                const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z), dyadic_mult(m4.w, e4.w)};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double t = (double)acc[i][jb + j][r] * Mc[r] + IVIT_MAGIC;
                        b[j][r] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                    }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                D[bi][i][j] = __builtin_amdgcn_perm((unsigned)b[j][1], (unsigned)b[j][0], 0x0c0c0400u) |
                              __builtin_amdgcn_perm((unsigned)b[j][3], (unsigned)b[j][2], 0x04000c0cu);
        }
        if constexpr (EPI == EPI_RESID) {
            // ALL residual chunks are requested as soon as the first batch's accumulators are dead (64 registers free, 16 of them
            // for its packed results): requested batch by batch, the last batch's loads were only one batch of phase B old when
            // hook.consume() drained vmcnt -- 3.4 K cycles of a 15 K cycle epilogue spent waiting for them (stamped timeline)
            const bool batchwise = IVIT_LAB && (g.flags2 & 1024);      // lab A/B: the former order
#pragma unroll
            for (int b2 = 0; b2 < NB; ++b2) {
                if (batchwise ? b2 != bi : bi != 0) continue;         // uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = min(m0 + 16 * (4 * b2 + j) + l15, g.M - 1);
                    rv[b2][j] = load16_sel(g.res + ((unsigned)t * (unsigned)g.ldr + (unsigned)min(c, g.N - 16)), g.flags2);   // 32-bit offsets: launcher
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // (derived only now: during phase A every register is taken)
    int8_t* const out = reinterpret_cast<int8_t*>(g.out);
    float magic_v = 12582912.0f;
    asm volatile("" : "+v"(magic_v));
    // ---- store addressing: offset of (row m0 + l15, column c), and what 16 more rows add
    unsigned off0, ostep;                           // 32-bit byte offsets (the launcher checks that the operands stay below 4 GiB)
    int q_b = 0, q_tok = 0;                         // EPI_QKV: image / token of row m0 + l15
    if constexpr (EPI == EPI_QKV) {
        const int cc = min(c, g.N - 16);
        const int cdim = g.heads * g.head_dim;
        const int which = cc / cdim, rem = cc - which * cdim;
        const int hh = rem / g.head_dim, d0 = rem - hh * g.head_dim;
        const int nb = g.M / g.tokens;
        off0 = (unsigned)(((which * nb * g.heads + hh) * g.tokens) * g.head_dim + d0);   // 32-bit range checked by the launcher
        ostep = 0;
        const int t0 = min(m0 + l15, g.M - 1);
        q_b = t0 / g.tokens;
        q_tok = t0 - q_b * g.tokens;
    } else if (EPI == EPI_RQ && g.out_blocks) {
        off0 = block_off(block_row(m0 + l15, g.N), block_col(min(c, g.N - 16)));
        ostep = (unsigned)(g.N >> 6) * 1024u;
    } else {
        off0 = (unsigned)(m0 + l15) * (unsigned)g.ldo + (unsigned)c;
        ostep = 16u * (unsigned)g.ldo;
    }
    if constexpr (ABL & 2048) st[4] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        const int jb = 4 * bi;
        if (bi == NB - 1) {
            if constexpr (ABL & 2048) st[5] = __builtin_amdgcn_s_memtime();
            hook.consume();
            if constexpr (ABL & 2048) st[6] = __builtin_amdgcn_s_memtime();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            // 4 x 4 dword transpose over the token's four lanes: lane g4 <- dwords [0..3][g4]
            const v2u ab = __builtin_amdgcn_permlane32_swap(D[bi][0][j], D[bi][2][j], false, false);
            const v2u cd = __builtin_amdgcn_permlane32_swap(D[bi][1][j], D[bi][3][j], false, false);
            const v2u ac = __builtin_amdgcn_permlane16_swap(ab.x, cd.x, false, false);
            const v2u bd = __builtin_amdgcn_permlane16_swap(ab.y, cd.y, false, false);
            int v[4] = {(int)ac.x, (int)ac.y, (int)bd.x, (int)bd.y};
            if constexpr (EPI == EPI_RESID) {
                if constexpr (ABL & 4096) residual_chunk_f32(v, rv[bi][j], g.Mf_main, g.Mf_res, magic_v);
                else residual_chunk_f64(v, rv[bi][j], g.M_main, g.M_res);
            }
            if constexpr (EPI == EPI_RQ) {
                if (lut_lds) {      // uniform: a 256-entry int8 -> int8 map (LDS) over the 16 bytes of the chunk
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned u = (unsigned)v[d] ^ 0x80808080u;      // q + 128 per byte
                        v[d] = (int)((unsigned)lut_lds[u & 255] | ((unsigned)lut_lds[(u >> 8) & 255] << 8) |
                                     ((unsigned)lut_lds[(u >> 16) & 255] << 16) | ((unsigned)lut_lds[u >> 24] << 24));
                    }
                }
            }
            const int t = m0 + 16 * (jb + j) + l15;
            unsigned off;
            if constexpr (EPI == EPI_QKV) {
                if (jb + j > 0) {       // 16 rows further: at most one image boundary when an image has at least 16 tokens
                    if (g.tokens >= 16) {
                        q_tok += 16;
                        if (q_tok >= g.tokens) { q_tok -= g.tokens; ++q_b; }
                    } else {
                        const int tt = min(t, g.M - 1);
                        q_b = tt / g.tokens;
                        q_tok = tt - q_b * g.tokens;
                    }
                }
                off = off0 + (unsigned)((q_b * g.heads * g.tokens + q_tok) * g.head_dim);
            } else {
                off = off0 + (unsigned)(jb + j) * ostep;
            }
            if constexpr (GELU) {
                // written through to memory (sc1 = device scope): read back by whichever workgroup completes the panel, possibly on
                // another XCD.  Two 64-bit atomic stores, not an inline-asm 128-bit one: compiler-visible, so its hazard and
                // s_waitcnt bookkeeping cover them
                if (col_ok && t < g.M) {
                    long long* q = reinterpret_cast<long long*>(out + off);
                    __hip_atomic_store(q, (long long)(((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(q + 1, (long long)(((unsigned long long)(unsigned)v[3] << 32) | (unsigned)v[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (col_ok && t < g.M) store16_sel(out + off, make_int4(v[0], v[1], v[2], v[3]), g.flags2);
            }
        }
    }
}

// ---- EXPERIMENTAL (measured slower than GEMM + table pass; no engine uses it): ShiftGELU + mlp.qact1 behind
// mlp.fc1 + mlp.qact_gelu inside the GEMM that produces its input (ivit_modules.py:105-126).
// The map of a byte depends on the maximum over ALL N outputs of its token (the exponent's argument is k - max), and a token's N
// channels are N / 256 tiles of N / 256 different workgroups.  So: every tile writes its requantised bytes through to memory and
// counts itself in at its 128-token panel; the workgroup whose arrival completes the panel reads the panel's 128 x N bytes back,
// takes the row maxima, and maps the rows in place.  Nobody waits for anybody: a workgroup either finds the panel complete or
// leaves.  Visibility across XCDs (one L2 each): sc1 stores, all acknowledged (s_waitcnt vmcnt(0)) and written back (release)
// before the arrival is counted; the completing workgroup reads with sc1 loads.  The counters are left as they were found (zero).
template <int NTHREADS>
IVIT_DEV void gelu_panel_phase(const GemmArgs& g, char* lds, int m0, int half, int tid)
{
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    constexpr int RG = 4;                          // rows per wave and step: RG x N / 1024 16-byte chunks in flight per lane
    constexpr int MAXC = 4;                        // 16-byte chunks per lane and row: N <= 4096
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's tile stores have been acknowledged
    __syncthreads();
    int* flag = reinterpret_cast<int*>(lds);
    const int panel = m0 >> 7, mp = panel << 7;
    int* cnt = g.gelu_ws + panel;
    if (tid == 0) {
        const int inc = half ? 1 : 2;
        const int old = __hip_atomic_fetch_add(cnt, inc, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);   // acquire too: the completing workgroup reads the other tiles next
        flag[0] = (old + inc == 2 * g.tiles_n) ? 1 : 0;
    }
    __syncthreads();
    if (flag[0] == 0) return;                      // uniform: the panel is not complete (or another workgroup completes it)
#if IVIT_LAB
    if (g.flags2 & 0x1000) {      // timing ablation: count, but leave the panel unmapped (results wrong)
        if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#endif
    const int lane = tid & 63, wave = tid >> 6;
    unsigned char* tab = reinterpret_cast<unsigned char*>(lds) + 64 + wave * (RG * 256);
    int8_t* const out = reinterpret_cast<int8_t*>(g.out);
    const int nch = g.N >> 4;                      // 16-byte chunks per row (N % 64 == 0)
    constexpr int RPW = 128 / (NTHREADS / 64);     // rows per wave
    for (int rs = 0; rs < RPW; rs += RG) {
        const int r0 = mp + wave * RPW + rs;
        if (r0 >= g.M) break;                      // uniform
        v4i_ w[RG][MAXC];
        unsigned off[RG][MAXC];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int row = min(r0 + r, g.M - 1);
            const BlockRow br = block_row(row, g.N);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                w[r][c] = (v4i_){0, 0, 0, 0};
                const int ch = min(lane + 64 * c, nch - 1);
                off[r][c] = g.out_blocks ? block_off(br, block_col(16 * ch)) : (unsigned)row * (unsigned)g.ldo + 16u * (unsigned)ch;
                if (64 * c < nch) {                // uniform.  (An inline-asm 128-bit sc1 load here returned wrong first dwords now and
                                                   // then: the compiler reuses these registers for the mapped output and knows nothing of
                                                   // a load it cannot see.  Compiler-visible atomic loads, twice as many, are exact.)
                    const long long* q = reinterpret_cast<const long long*>(out + off[r][c]);
                    const long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    w[r][c] = (v4i_){(int)lo, (int)(lo >> 32), (int)hi, (int)(hi >> 32)};
                }
            }
        }
        // the rows' maxima from the bytes themselves, then their table slices
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            int km = -128;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (64 * c >= nch) continue;
                if (lane + 64 * c < nch) {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        km = max(max(km, (int)(int8_t)(w[r][c][d])), max((int)(int8_t)(w[r][c][d] >> 8), max((int)(int8_t)(w[r][c][d] >> 16), w[r][c][d] >> 24)));
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) km = max(km, __shfl_xor(km, o));
            reinterpret_cast<int*>(tab + 256 * r)[lane] = reinterpret_cast<const int*>(g.gelu_lut + (int64_t)(km + 128) * 256)[lane];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const unsigned char* tb = tab + 256 * r;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (64 * c >= nch) continue;       // uniform
                int o[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned u = (unsigned)w[r][c][d] ^ 0x80808080u;      // k + 128 per byte
                    o[d] = (int)((unsigned)tb[u & 255] | ((unsigned)tb[(u >> 8) & 255] << 8) | ((unsigned)tb[(u >> 16) & 255] << 16) |
                                 ((unsigned)tb[u >> 24] << 24));
                }
                if (lane + 64 * c < nch && r0 + r < g.M) *reinterpret_cast<int4*>(out + off[r][c]) = make_int4(o[0], o[1], o[2], o[3]);
            }
        }
        __builtin_amdgcn_wave_barrier();           // every lane has read its slices before the next step overwrites them
    }
    if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- 16-bit epilogue of the weights-in-registers kernel (EPI_RQ16_RES16).  A wave owns 64 of the tile's 256 channels x 32 TJ
// tokens; a lane holds, per accumulator quad, 4 consecutive channels of one token.  Straight from the registers that is one
// 8-byte access per lane into 32 different rows per instruction -- measured ~75 us per launch over the int8 epilogue, the
// texture path serialising on the rows.  So, like the int8 epilogue, the tile is transposed through LDS -- 64 tokens at a time:
// 64 rows x 256 int16 is exactly the staging region of the int8 tile (128 rows x 256 bytes) -- and phase B moves 16-byte chunks,
// 32 consecutive threads per 512-byte row, for the residual load and the store alike.
// A 16-bit result is too wide for the float32 bracket certificate of the int8 epilogue (an undecided bracket every ~100
// outputs), so the arithmetic is the reference's float64 (quant_utils.py:229-230: product rounded to 53 bits, then RNE; the two
// 16-bit operands of the residual QuantAct have exact products: one fused multiply-add each).
template <int TJ, int NTHREADS, typename Hook>
IVIT_DEV void epilogue_rq16_res16(v16i (&acc)[2][TJ], const GemmArgs& g, char* cs, int m0, int n0, int wch, int tid, int h, int l31,
                                  const Hook& hook)
{
    constexpr int CH = 256;
    constexpr int RS = CH * 2 + 8;            // LDS row stride: 130 dwords, consecutive rows two banks apart
    constexpr int CPR = CH / 8;               // 16-byte chunks (8 channels) per row
    constexpr int NIT = 64 * CPR / NTHREADS;  // chunks per thread and round
    static_assert(TJ % 2 == 0 && 64 * CPR % NTHREADS == 0, "rounds of 64 tokens");
    __builtin_amdgcn_s_setprio(2);
    hook.issue();
    const int16_t* res = reinterpret_cast<const int16_t*>(g.res);
    int16_t* out = reinterpret_cast<int16_t*>(g.out);
#pragma unroll
    for (int rd = 0; rd < TJ / 2; ++rd) {
        // ---- phase A: k16 = clamp16(RNE(acc * M[n])) of token sub-tiles 2 rd, 2 rd + 1 -> cs[token][channel]
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            const int i = grp >> 2, q = grp & 3;
            const int cl = wch + 32 * i + 8 * q + 4 * h;
            const int c0 = min(n0 + cl, g.N - 4);
            const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
            const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
            const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z), dyadic_mult(m4.w, e4.w)};
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                int o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double p = (double)acc[i][2 * rd + jj][4 * q + r] * Mc[r];
                    o[r] = rne_clamp16(p);
                }
                v2i ow;
                ow.x = (o[0] & 0xffff) | (o[1] << 16);
                ow.y = (o[2] & 0xffff) | (o[3] << 16);
                *reinterpret_cast<v2i*>(cs + (32 * jj + l31) * RS + 2 * cl) = ow;
            }
        }
        // ---- phase B work items: the residual loads go out before the barrier
        v4i rr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int qd = tid + NTHREADS * it;
            const int tl = qd / CPR, cc = qd % CPR;
            const int t = min(m0 + 64 * rd + tl, g.M - 1), cn = min(n0 + 8 * cc, g.N - 8);
            rr[it] = *reinterpret_cast<const v4i*>(res + (int64_t)t * g.ldr + cn);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // two batches of NIT / 2 chunks: 16 registers of staged values live instead of 32 (the accumulators of the next round are)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            constexpr int NH = NIT / 2;
            v4i kk[NH];
#pragma unroll
            for (int u = 0; u < NH; ++u) {
                const int qd = tid + NTHREADS * (NH * hb + u);
                lds_read16_async(kk[u], lds_addr(cs) + (unsigned)((qd / CPR) * RS + 16 * (qd % CPR)));
            }
#pragma unroll
            for (int u = 0; u < NH; ++u) lds_wait(kk[u]);
            if (hb == 1 && rd + 1 < TJ / 2) {      // the next round's phase A overwrites the staging rows
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (rd == 0 && hb == 0) hook.consume();   // vmcnt(0) while no store of this epilogue is in flight
#pragma unroll
            for (int u = 0; u < NH; ++u) {
                const int it = NH * hb + u;
                const int qd = tid + NTHREADS * it;
                const int tl = qd / CPR, cc = qd % CPR;
                const int t = m0 + 64 * rd + tl, cn = n0 + 8 * cc;
                v4i ov;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int kw = kk[u][d], rw = rr[it][d];
                    const int lo = clamp_i32(requant_exact((int)(int16_t)kw, g.M_main) + requant_exact((int)(int16_t)rw, g.M_res), -32768, 32767);
                    const int hi = clamp_i32(requant_exact(kw >> 16, g.M_main) + requant_exact(rw >> 16, g.M_res), -32768, 32767);
                    ov[d] = (lo & 0xffff) | (hi << 16);
                }
                if (t < g.M && cn < g.N) *reinterpret_cast<v4i*>(out + (int64_t)t * g.ldo + cn) = ov;
            }
        }
    }
}

// The same for the 128 x 128-tile kernel (wave (wm, wn): tokens 64 wm + 32 j + l31, channels 64 wn + 32 i + 8 q + 4 h): the whole
// tile of 16-bit intermediates is staged at once -- 128 rows x (256 + 8) bytes is exactly the kernel's LDS (two main-loop stages
// + the requantiser table, all dead here) -- then 16-byte chunks, 16 consecutive threads per 256-byte row.
constexpr int RQ16S_RS = 128 * 2 + 8;
IVIT_DEV void epilogue_rq16_res16_small(v16i (&acc)[2][2], const GemmArgs& g, char* cs, int m0, int n0, int wm, int wn, int tid,
                                        int h, int l31)
{
    constexpr int RS = RQ16S_RS, CPR = 16, NTH = 256, NIT = 128 * CPR / NTH;
    const int16_t* res = reinterpret_cast<const int16_t*>(g.res);
    int16_t* out = reinterpret_cast<int16_t*>(g.out);
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        const int i = grp >> 2, q = grp & 3;
        const int cl = 64 * wn + 32 * i + 8 * q + 4 * h;
        const int c0 = min(n0 + cl, g.N - 4);
        const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
        const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
        const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z), dyadic_mult(m4.w, e4.w)};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double p = (double)acc[i][j][4 * q + r] * Mc[r];
                o[r] = rne_clamp16(p);
            }
            v2i ow;
            ow.x = (o[0] & 0xffff) | (o[1] << 16);
            ow.y = (o[2] & 0xffff) | (o[3] << 16);
            *reinterpret_cast<v2i*>(cs + (64 * wm + 32 * j + l31) * RS + 2 * cl) = ow;
        }
    }
    v4i rr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int qd = tid + NTH * it;
        const int tl = qd / CPR, cc = qd % CPR;
        const int t = min(m0 + tl, g.M - 1), cn = min(n0 + 8 * cc, g.N - 8);
        rr[it] = *reinterpret_cast<const v4i*>(res + (int64_t)t * g.ldr + cn);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int qd = tid + NTH * it;
        const int tl = qd / CPR, cc = qd % CPR;
        const int t = m0 + tl, cn = n0 + 8 * cc;
        const v4i kk = *reinterpret_cast<const v4i*>(cs + tl * RS + 16 * cc);
        v4i ov;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int kw = kk[d], rw = rr[it][d];
            const int lo = clamp_i32(requant_exact((int)(int16_t)kw, g.M_main) + requant_exact((int)(int16_t)rw, g.M_res), -32768, 32767);
            const int hi = clamp_i32(requant_exact(kw >> 16, g.M_main) + requant_exact(rw >> 16, g.M_res), -32768, 32767);
            ov[d] = (lo & 0xffff) | (hi << 16);
        }
        if (t < g.M && cn < g.N) *reinterpret_cast<v4i*>(out + (int64_t)t * g.ldo + cn) = ov;
    }
}

// ---- 256 x 128 LDS-DMA tiles (persistent kernel, relaunch form, deep-ring form)
constexpr int BTOK = 256, BCH = 128, BIG_NT = 256, BIG_STAGES = 3;
constexpr int BIG_A_BYTES = BTOK * BK;                 // 16 KiB
constexpr int BIG_STAGE = (BTOK + BCH) * BK;           // 24 KiB
constexpr int BIG_SMEM = BIG_STAGES * BIG_STAGE;       // 72 KiB  (>= 256 * 132 epilogue tile)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---- persistent-kernel helpers
constexpr int PT_OFF = BIG_SMEM;          // tables: 2 x { float2 lohi[128]; int bias[128] }
constexpr int PT_BYTES = BCH * 12;
constexpr int PERS_SMEM = BIG_SMEM + 2 * PT_BYTES;

struct PersTile {
    int m0, n0;
};

IVIT_DEV PersTile pers_tile(const GemmArgs& g, int t)
{
    const int nblk = g.tiles_m * g.tiles_n;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = t & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (t >> 3);
    const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
    return PersTile{tm * BTOK, tn * BCH};
}

// loads of one channel's table entry (issued early, consumed later)
struct PersTableLoad {
    unsigned m;
    int e, bias;
    bool valid;
};

IVIT_DEV PersTableLoad pers_table_issue(const GemmArgs& g, int n0, int tid)
{
    PersTableLoad r{0u, 0, 0, false};
    const int c = n0 + tid;
    if (tid < BCH && c < g.N) {
        r.m = g.m[c];
        r.e = g.e[c];
        r.bias = g.bias ? g.bias[c] : 0;
        r.valid = true;
    }
    return r;
}

IVIT_DEV void pers_table_write(const PersTableLoad& r, char* tab, int tid)
{
    if (tid < BCH) {
        float2 lh = make_float2(0.f, 0.f);
        if (r.valid) {
            const double M = dyadic_mult(r.m, r.e);
            const float mf = (float)M;
            const int bits = __float_as_int(mf);
            lh.x = ((double)mf > M) ? __int_as_float(bits - 1) : mf;
            lh.y = ((double)mf < M) ? __int_as_float(bits + 1) : mf;
        }
        reinterpret_cast<float2*>(tab)[tid] = lh;
        reinterpret_cast<int*>(tab + BCH * 8)[tid] = r.bias;
    }
}

// Work items of one workgroup.  The launch has G workgroups (2 per CU); tile t < split_from belongs to workgroup
// t % G.  If the last round of full tiles would be at most half full (R = F mod G tiles, 2R <= G), those R tiles are
// split into 2R half tiles of 128 tokens, one per workgroup 0 .. 2R-1, so the tail costs half a tile time instead of a
// whole one (DeiT-B, N = 768: 1182 tiles on 512 workgroups = 2.31 rounds -> 2.5 instead of 3).

}  // namespace

// kernel forms of gemm_lab.hip: returns 1 if `g_debug_flags` selected one of them and it was launched (status in *rc)
#if IVIT_LAB
int ivit_gemm_lab_launch(int epi, void* gemm_args, const char* name, ivit_stream_t stream, int* rc);
#endif
