// gemm.hip -- INT8 GEMM on v_mfma_i32_32x32x32_i8 with the QuantAct requantiser fused in the
// epilogue.  Replaces QuantLinear.forward / QuantConv2d.forward + the QuantAct that follows
// (/root/reference/models/quantization_utils/quant_modules.py:186-226, 302-387, 478-511;
//  fixedpoint_mul quant_utils.py:193-253).
//
// Formulation: out^T[n][t] = W[n][:] . A[t][:]  -- the weight rows are the MFMA "A" operand and the
// activation rows the "B" operand (both K-contiguous, so both fragments are one 16-byte LDS read).
// The 32x32 accumulator tile then has the TOKEN on the lane (col = lane & 31) and 4 consecutive
// CHANNELS in each register quad (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), which is what the
// int8 epilogue wants: 4 requantised channels pack into one dword.
//
// Block tile 128 tokens x 128 channels x 64 bytes of K, 4 waves (2 x 2, 64 x 64 each = 2 x 2 MFMA
// tiles), two LDS stages filled through registers (global_load_dwordx4 -> ds_write_b128), XOR
// swizzled so that every ds_read_b128 fragment read is bank-conflict free.  The epilogue stages the
// int8 tile through LDS so that global stores (and the residual loads) are 16 B per lane, row
// contiguous.  Block ids are remapped so that the 8 XCDs each walk a contiguous range of tiles
// (token panel reuse in the XCD-private L2).
#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 128;  // tokens per block
constexpr int BN = 128;  // channels per block
constexpr int BK = 64;   // K bytes per stage
constexpr int NT = 256;
constexpr int STAGE_BYTES = (BM + BN) * BK;  // 16 KiB
constexpr int W_OFF = BM * BK;               // weight tile behind the token tile
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;  // 32 KiB >= 128 * 132

enum { EPI_RQ = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_I32 = 3 };

int g_kernel_choice = 0;      // 0 = automatic, 1 = never the 256x256 kernel (tests / A-B timing)
bool g_force_small = false;  // tests: route every problem through the small-tile kernel
void* g_stamp_buf = nullptr;
int g_debug_flags = 0;       // perf ablations (scripts/gemm_ablate.py): 1 = skip DMA in the loop, 2 = skip MFMA, 4 = skip epilogue

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
    int M, N, K;
    int tokens, heads, head_dim;
    int tiles_m, tiles_n;
    int flags;
    int stagger;  // number of first-generation blocks subject to the start stagger (0 = off)
    int cu_turns;       // persistent kernel: 1 = co-resident workgroups alternate main loops through the per-CU token
    int stagger_units;  // persistent kernel: start delay of the second co-resident workgroup, in s_sleep(16) (~1K cycle) units
    int split_from;  // persistent kernel: tiles [split_from, tiles_m*tiles_n) are processed as two half tiles each
};

IVIT_DEV int nk_of(const GemmArgs& g) { return g.K / 64; }

// byte offset of 16-byte chunk c (0..3) of tile row r; rows are 64 B, four rows per 256-B bank row.
IVIT_DEV int swz(int r, int c) { return r * BK + ((c ^ ((r >> 2) & 3)) << 4); }

IVIT_DEV int pack4_i8(int a, int b, int c, int d)
{
    return (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | ((d & 0xff) << 24);
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

template <int EPI, int TI, int TJ, int TOK, int NTHREADS, int ABL = 0, int CH = 128, typename Hook = NoHook>
IVIT_DEV void epilogue_i8(v16i (&acc)[TI][TJ], const GemmArgs& g, char* smem, const char* rq_lds, int m0, int n0,
                          int wch, int wtok, int tid, int h, int l31, const Hook& hook = Hook())
{
    // The epilogue is a short VALU burst next to the co-resident workgroup's MFMA stream: give it issue priority
    // so its dependent chains do not wait behind queued MFMAs (which run in the matrix pipe once issued).
    __builtin_amdgcn_s_setprio(2);
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
    const float2* rq = reinterpret_cast<const float2*>(rq_lds);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = wch + 32 * i + 8 * q + 4 * h;  // local channel of the quad
            const float4 lh01 = *reinterpret_cast<const float4*>(rq + cl);      // lo0 hi0 lo1 hi1
            const float4 lh23 = *reinterpret_cast<const float4*>(rq + cl + 2);  // lo2 hi2 lo3 hi3
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
                        const int tl = __float_as_int(__builtin_fmaf(a, lo[jj], 12582912.0f));
                        const int th = __float_as_int(__builtin_fmaf(a, hi[jj], 12582912.0f));
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
    unsigned long long t_p1 = 0, t_sync = 0;
    if constexpr (ABL & 512) t_p1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if constexpr (ABL & 512) {
        t_sync = __builtin_amdgcn_s_memtime();
        if (tid == 0 && g.res != nullptr) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<int8_t*>(g.res)) + 8ull * blockIdx.x;
            d[4] = t_p1; d[5] = t_sync;
        }
    }
    if constexpr (ABL & 16) return;

    int8_t* out = reinterpret_cast<int8_t*>(g.out);
    constexpr int NIT = TOK * CPR / NTHREADS;
    int v[NIT][4];
    int4 rv[NIT];
    hook.issue();    // persistent kernel: next tile's table loads go out before this tile's stores
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = tid + NTHREADS * it;
        const int tl = q / CPR, cc = q % CPR;
        const int* src = reinterpret_cast<const int*>(smem + tl * CSS + 16 * cc);
        v[it][0] = src[0]; v[it][1] = src[1]; v[it][2] = src[2]; v[it][3] = src[3];
        if constexpr (EPI == EPI_RESID) {
            const int t = min(m0 + tl, g.M - 1), cn = min(n0 + 16 * cc, g.N - 16);
            rv[it] = *reinterpret_cast<const int4*>(g.res + (int64_t)t * g.ldr + cn);
        }
    }
    hook.consume();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = tid + NTHREADS * it;
        const int tl = q / CPR, cc = q % CPR;
        const int t = m0 + tl, cn = n0 + 16 * cc;
        if (t >= g.M || cn >= g.N) continue;
        if constexpr (EPI == EPI_RESID) {
            const int rr[4] = {rv[it].x, rv[it].y, rv[it].z, rv[it].w};
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
        int64_t off;
        if constexpr (EPI == EPI_QKV) {
            const int cdim = g.heads * g.head_dim;
            const int which = cn / cdim, rem = cn - which * cdim;
            const int hh = rem / g.head_dim, d0 = rem - hh * g.head_dim;
            const int b = t / g.tokens, tok = t - b * g.tokens;
            const int nb = g.M / g.tokens;
            off = ((((int64_t)which * nb + b) * g.heads + hh) * g.tokens + tok) * g.head_dim + d0;
        } else {
            off = (int64_t)t * g.ldo + cn;
        }
        *reinterpret_cast<int4*>(out + off) = make_int4(v[it][0], v[it][1], v[it][2], v[it][3]);
    }
}

template <int EPI>
__global__ __launch_bounds__(NT) void gemm_i8_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES + BN * 8];

    // ---- XCD-aware block -> tile map (bijective for any block count)
    const int nblk = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    if constexpr (EPI != EPI_I32) fill_rq_table(g, smem + SMEM_BYTES, n0, BN, tid);

    // ---- staging: thread moves chunks (row = tid/4 + 64 i, c = tid%4) of both tiles
    const int srow = tid >> 2, sc = tid & 3;
    const int8_t* ap[2];
    const int8_t* wp[2];
    int soff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = srow + 64 * i;
        int ar = min(m0 + r, g.M - 1);
        int wr = min(n0 + r, g.N - 1);
        ap[i] = g.A + (int64_t)ar * g.lda + 16 * sc;
        wp[i] = g.W + (int64_t)wr * g.ldw + 16 * sc;
        soff[i] = swz(r, sc);
    }

    // ---- accumulators start from the int32 bias (QuantLinear adds bias_integer to the product)
    v16i acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int cn = n0 + 64 * wn + 32 * i + 8 * (r >> 2) + 4 * h + (r & 3);
            int b = (g.bias != nullptr && cn < g.N) ? g.bias[cn] : 0;
            acc[i][0][r] = b;
            acc[i][1][r] = b;
        }

    v4i ra[2], rw[2];
    const int nk = g.K / BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ra[i] = *reinterpret_cast<const v4i*>(ap[i]);
        rw[i] = *reinterpret_cast<const v4i*>(wp[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<v4i*>(smem + soff[i]) = ra[i];
        *reinterpret_cast<v4i*>(smem + W_OFF + soff[i]) = rw[i];
    }
    __syncthreads();

    // fragment rows of this lane
    const int wrow0 = 64 * wn + l31, arow0 = 64 * wm + l31;

    for (int kt = 0; kt < nk; ++kt) {
        const int st = (kt & 1) * STAGE_BYTES;
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i] = *reinterpret_cast<const v4i*>(ap[i] + (int64_t)(kt + 1) * BK);
                rw[i] = *reinterpret_cast<const v4i*>(wp[i] + (int64_t)(kt + 1) * BK);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i wf[2], af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wf[i] = *reinterpret_cast<const v4i*>(smem + st + W_OFF + swz(wrow0 + 32 * i, 2 * ks + h));
                af[i] = *reinterpret_cast<const v4i*>(smem + st + swz(arow0 + 32 * i, 2 * ks + h));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            const int sn = ((kt + 1) & 1) * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<v4i*>(smem + sn + soff[i]) = ra[i];
                *reinterpret_cast<v4i*>(smem + sn + W_OFF + soff[i]) = rw[i];
            }
        }
        __syncthreads();
    }

    // ---- epilogue
    if constexpr (EPI == EPI_I32) {
        int32_t* out = reinterpret_cast<int32_t*>(g.out);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int t = m0 + 64 * wm + 32 * j + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int c0 = n0 + 64 * wn + 32 * i + 8 * q + 4 * h;
                    if (t < g.M && c0 < g.N) {
                        v4i v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<v4i*>(out + (int64_t)t * g.ldo + c0) = v;
                    }
                }
            }
        return;
    } else {
        epilogue_i8<EPI, 2, 2, BM, NT>(acc, g, smem, smem + SMEM_BYTES, m0, n0, 64 * wn, 64 * wm, tid, h, l31);
    }
}


// ================================================================================================
// Large-problem kernel: block tile 256 tokens x 128 channels x 64 K-bytes, 4 waves (2 x 2, each
// 64 channels x 128 tokens = 2 x 4 MFMA tiles, 128 accumulator registers), THREE LDS stages filled by
// LDS-DMA (global_load_lds_dwordx4: no staging registers), one raw s_barrier per K step with a
// counted vmcnt so the next stage's DMA stays in flight across it.  72 KiB LDS and <= 256 registers
// give two workgroups per CU: one block's requant epilogue (VALU/float64 pipe) overlaps the other's
// MFMA main loop.  LDS images are lane-linear per DMA instruction (16 rows x 64 B); the bank swizzle
// is applied on the per-lane SOURCE address and again on the fragment read.
// ================================================================================================
constexpr int BTOK = 256, BCH = 128, BIG_NT = 256, BIG_STAGES = 3;
constexpr int BIG_A_BYTES = BTOK * BK;                 // 16 KiB
constexpr int BIG_STAGE = (BTOK + BCH) * BK;           // 24 KiB
constexpr int BIG_SMEM = BIG_STAGES * BIG_STAGE;       // 72 KiB  (>= 256 * 132 epilogue tile)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int EPI, int ABL>
__global__ __launch_bounds__(BIG_NT, 2) void gemm_i8_big_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) char smem[BIG_SMEM + BCH * 8];
    unsigned long long t_start = 0, t_loop = 0, t_epi = 0, r_start = 0;
    if constexpr (ABL & 512) {
        t_start = __builtin_amdgcn_s_memtime();
        r_start = __builtin_amdgcn_s_memrealtime();
    }

    const int nblk = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
    const int m0 = tm * BTOK, n0 = tn * BCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wt = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    // ---- LDS-DMA sources: instruction q covers tile rows 16q..16q+15 (1 KiB); lane -> row 16q + lane/4,
    // stored slot lane%4 holds global chunk (lane%4) ^ ((row>>2)&3)
    const int8_t* asrc[4];
    const int8_t* wsrc[2];
    const int lrow = lane >> 2, lslot = lane & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = 16 * (wave + 4 * i) + lrow;
        int c = lslot ^ ((row >> 2) & 3);
        asrc[i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + 16 * c;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = 16 * (wave + 4 * i) + lrow;
        int c = lslot ^ ((row >> 2) & 3);
        wsrc[i] = g.W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + 16 * c;
    }

    const int nk = g.K / BK;
    // DMA piece `idx` (0..3: token tile, 4..5: weight tile) of K step kt
    auto issue_one = [&](int kt, int idx) {
        char* base = smem + (kt % BIG_STAGES) * BIG_STAGE;
        const int koff = kt * BK;
        if (idx < 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[idx] + koff), (lptr_t)(base + 1024 * (wave + 4 * idx)), 16, 0,
                                             0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[idx - 4] + koff),
                                             (lptr_t)(base + BIG_A_BYTES + 1024 * (wave + 4 * (idx - 4))), 16, 0, 0);
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int idx = 0; idx < 6; ++idx) issue_one(kt, idx);
    };

    // Two workgroups share a CU (one wave of each per SIMD).  Launched together they would run in
    // lockstep -- both in the MFMA main loop, then both in the VALU/float64 epilogue -- and the two
    // pipes would never overlap.  Stagger the first generation: the workgroup that landed in the odd
    // wave slot of its SIMD sleeps for about half a main loop, so that from then on one workgroup's
    // epilogue runs under the other's MFMAs.  Later generations inherit the phase shift.  (Speed only.)
    if (g.stagger && blockIdx.x < (unsigned)g.stagger) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;  // HW_ID.wave_id[0]
        if (slot)
            for (int it = 0; it < (nk_of(g) + 1) / 2; ++it) __builtin_amdgcn_s_sleep(64);
    }
    // Start the DMA ring first, then fetch the bias / requant tables under its latency.  The ordinary loads'
    // results are consumed right here, where a full vmcnt(0) drain (which also retires both stages) is wanted
    // anyway; no ordinary load remains in flight once the main loop starts.
    issue(0);
    if (nk > 1) issue(1);
    fill_rq_table(g, smem + BIG_SMEM, n0, BCH, tid);
    v16i acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = n0 + 64 * wc + 32 * i + 8 * q + 4 * h;   // 4 consecutive channels of this register quad
            int4 b4 = make_int4(0, 0, 0, 0);
            if (g.bias != nullptr && c0 < g.N) b4 = *reinterpret_cast<const int4*>(g.bias + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j][4 * q + 0] = b4.x;
                acc[i][j][4 * q + 1] = b4.y;
                acc[i][j][4 * q + 2] = b4.z;
                acc[i][j][4 * q + 3] = b4.w;
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int wrow0 = 64 * wc + l31, arow0 = 128 * wt + l31;
    // fragment byte offsets inside a stage for k-sub-step 0 / 1 (the swizzle depends on the row only)
    int woff[2][2], aoff[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) woff[ks][i] = BIG_A_BYTES + swz(wrow0 + 32 * i, 2 * ks + h);
#pragma unroll
        for (int j = 0; j < 4; ++j) aoff[ks][j] = swz(arow0 + 32 * j, 2 * ks + h);
    }
    v4i wf0[2], af0[4], wf1[2], af1[4];
    bool frags_once = false;
    auto load_frags = [&](const char* st, int ks, v4i (&wf)[2], v4i (&af)[4]) {
        if constexpr (ABL & 128) {
            if (frags_once) return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const v4i*>(st + woff[ks][i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const v4i*>(st + aoff[ks][j]);
    };

    // Pipeline (3 LDS stages, fragments double-buffered in registers):
    //   iteration kt:  read frags(kt, ks=1) | MFMA on frags(kt, ks=0) interleaved with the DMA of stage kt+2
    //                  wait own DMA of stage kt+1 + own LDS reads | barrier B_kt
    //                  read frags(kt+1, ks=0) | MFMA on frags(kt, ks=1)
    // RAW: stage kt+1 is read only after B_kt, which every wave reaches after its counted vmcnt.
    // WAR: the DMA of stage kt+2 overwrites the buffer of stage kt-1; it is issued after B_{kt-1}, and
    //      every wave waited lgkmcnt(0) (all its reads of stage kt-1 returned) before B_{kt-1}.
    auto step = [&](int kt, auto dma_tag, auto last_tag) {
        constexpr bool DMA = decltype(dma_tag)::value && !(ABL & 1);
        constexpr bool LAST = decltype(last_tag)::value;
        const char* st = smem + (kt % BIG_STAGES) * BIG_STAGE;
        load_frags(st, 1, wf1, af1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (!(ABL & 2))
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf0[i], af0[j], acc[i][j], 0, 0, 0);
                else
                    asm volatile("" ::"v"(wf0[i]), "v"(af0[j]));
                if constexpr (DMA)
                    if (4 * i + j < 6) issue_one(kt + 2, 4 * i + j);
            }
        if constexpr (!(ABL & 256)) {
            if constexpr (decltype(dma_tag)::value) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if constexpr (!LAST) load_frags(smem + ((kt + 1) % BIG_STAGES) * BIG_STAGE, 0, wf0, af0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (!(ABL & 2))
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf1[i], af1[j], acc[i][j], 0, 0, 0);
                else
                    asm volatile("" ::"v"(wf1[i]), "v"(af1[j]));
            }
    };
    using T = std::true_type;
    using F = std::false_type;

    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(smem, 0, wf0, af0);
    if constexpr (ABL & 128) {
        load_frags(smem, 1, wf1, af1);
        frags_once = true;
    }
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(kt, T{}, F{});
    if (kt + 1 < nk) { step(kt, F{}, F{}); ++kt; }
    step(kt, F{}, T{});
    __syncthreads();  // every wave is done with the last stage before the tile is reused
    if constexpr (ABL & 512) t_loop = __builtin_amdgcn_s_memtime();
    if constexpr (ABL & 4) {
        int x = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) x ^= acc[i][j][r];
        if (x == 0x7fffffff) reinterpret_cast<int*>(g.out)[tid] = x;
        return;
    }
    epilogue_i8<EPI, 2, 4, BTOK, BIG_NT, ABL>(acc, g, smem, smem + BIG_SMEM, m0, n0, 64 * wc, 128 * wt, tid, h, l31);
    if constexpr (ABL & 512) {   // diagnostic build only: per-workgroup timeline into a buffer nothing else reads
        t_epi = __builtin_amdgcn_s_memtime();
        if (tid == 0 && g.res != nullptr) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<int8_t*>(g.res)) + 8ull * blockIdx.x;
            d[0] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);
            d[1] = t_start; d[2] = t_loop; d[3] = t_epi;
            d[6] = r_start; d[7] = __builtin_amdgcn_s_memrealtime();
        }
    }
}


// ================================================================================================
// XL kernel: block tile 256 tokens x 256 channels x 64 K-bytes, 8 waves (4 channel groups x 2 token
// groups, each 64 channels x 128 tokens = 2 x 4 MFMA tiles), FOUR LDS stages of 32 KiB filled by
// LDS-DMA.  With int8 MFMAs the L2 -> LDS stream is the scarce resource and it is latency bound
// (~1 us per piece under load): the tile moves the fewest bytes per MAC (0.0078 B) and the four-deep
// ring keeps up to three stages (96 KiB per CU) in flight at all times.  One workgroup per CU.
//   iteration kt:  read F(kt, ks=1) | MFMA F(kt, ks=0) interleaved with the 4 DMA pieces of stage kt+3
//                  counted vmcnt: own pieces of stage kt+1 landed | lgkmcnt(0) | barrier B_kt
//                  read F(kt+1, ks=0) | MFMA F(kt, ks=1)
// RAW: stage kt+1 is read only after B_kt.  WAR: stage kt+3 reuses the buffer of stage kt-1, whose
// reads every wave completed (lgkmcnt(0)) before B_{kt-1}; the DMA is issued after B_{kt-1}.
// ================================================================================================
constexpr int XTOK = 256, XCH = 256, XL_NT = 512, XL_STAGES = 4;
constexpr int XL_A_BYTES = XTOK * BK;             // 16 KiB
constexpr int XL_STAGE = (XTOK + XCH) * BK;       // 32 KiB
constexpr int XL_SMEM = XL_STAGES * XL_STAGE;     // 128 KiB (>= 256 * 260 epilogue tile)

template <int EPI, int ABL>
__global__ __launch_bounds__(XL_NT, 2) void gemm_i8_xl_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) char smem[XL_SMEM + XCH * 8];

    const int nblk = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
    const int m0 = tm * XTOK, n0 = tn * XCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wt = wave & 1;  // 4 x 2
    const int h = lane >> 5, l31 = lane & 31;
    fill_rq_table(g, smem + XL_SMEM, n0, XCH, tid);

    // ---- LDS-DMA sources: piece q covers tile rows 16q..16q+15 (1 KiB); lane -> row 16q + lane/4, stored
    // slot lane%4 holds global chunk (lane%4) ^ ((row>>2)&3).  Wave w owns pieces w and w + 8 of each tile.
    const int8_t* asrc[2];
    const int8_t* wsrc[2];
    const int lrow = lane >> 2, lslot = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = 16 * (wave + 8 * i) + lrow;
        int c = lslot ^ ((row >> 2) & 3);
        asrc[i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + 16 * c;
        wsrc[i] = g.W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + 16 * c;
    }

    v16i acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int cn = n0 + 64 * wc + 32 * i + 8 * (r >> 2) + 4 * h + (r & 3);
            int b = (g.bias != nullptr && cn < g.N) ? g.bias[cn] : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j][r] = b;
        }

    const int nk = g.K / BK;
    auto issue_one = [&](int kt, int idx) {
        char* base = smem + (kt % XL_STAGES) * XL_STAGE;
        const int koff = kt * BK;
        if (idx < 2)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[idx] + koff), (lptr_t)(base + 1024 * (wave + 8 * idx)), 16, 0,
                                             0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[idx - 2] + koff),
                                             (lptr_t)(base + XL_A_BYTES + 1024 * (wave + 8 * (idx - 2))), 16, 0, 0);
    };
    auto issue = [&](int kt) {
        if constexpr (!(ABL & 1)) {
#pragma unroll
            for (int idx = 0; idx < 4; ++idx) issue_one(kt, idx);
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // bias / table loads retired before the DMA pipeline starts
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);

    const int wrow0 = 64 * wc + l31, arow0 = 128 * wt + l31;
    int woff[2][2], aoff[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) woff[ks][i] = XL_A_BYTES + swz(wrow0 + 32 * i, 2 * ks + h);
#pragma unroll
        for (int j = 0; j < 4; ++j) aoff[ks][j] = swz(arow0 + 32 * j, 2 * ks + h);
    }
    v4i wf0[2], af0[4], wf1[2], af1[4];
    auto load_frags = [&](const char* st, int ks, v4i (&wf)[2], v4i (&af)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const v4i*>(st + woff[ks][i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const v4i*>(st + aoff[ks][j]);
    };
    // VM = number of this wave's DMA pieces allowed to stay in flight at the barrier (the stages after kt+1)
    auto step = [&](int kt, auto dma_tag, auto vm_tag, auto last_tag) {
        constexpr bool DMA = decltype(dma_tag)::value && !(ABL & 1);
        constexpr int VM = decltype(vm_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        const char* st = smem + (kt % XL_STAGES) * XL_STAGE;
        load_frags(st, 1, wf1, af1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (!(ABL & 2))
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf0[i], af0[j], acc[i][j], 0, 0, 0);
                else
                    asm volatile("" ::"v"(wf0[i]), "v"(af0[j]));
                if constexpr (DMA)
                    if (((4 * i + j) & 1) == 0 && (4 * i + j) < 8) issue_one(kt + 3, (4 * i + j) >> 1);
            }
        if constexpr (VM == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if constexpr (VM == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (!LAST) load_frags(smem + ((kt + 1) % XL_STAGES) * XL_STAGE, 0, wf0, af0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (!(ABL & 2))
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf1[i], af1[j], acc[i][j], 0, 0, 0);
                else
                    asm volatile("" ::"v"(wf1[i]), "v"(af1[j]));
            }
    };
    using T = std::true_type;
    using F = std::false_type;
    using V8 = std::integral_constant<int, 8>;
    using V4 = std::integral_constant<int, 4>;
    using V0 = std::integral_constant<int, 0>;

    // stage 0 landed: everything issued after it may stay in flight
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(smem, 0, wf0, af0);
    int kt = 0;
    for (; kt + 3 < nk; ++kt) step(kt, T{}, V8{}, F{});
    if (kt + 2 < nk) { step(kt, F{}, V4{}, F{}); ++kt; }
    if (kt + 1 < nk) { step(kt, F{}, V0{}, F{}); ++kt; }
    step(kt, F{}, V0{}, T{});

    __syncthreads();
    if constexpr (ABL & 4) {
        int x = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) x ^= acc[i][j][r];
        if (x == 0x7fffffff) reinterpret_cast<int*>(g.out)[tid] = x;
        return;
    }
    epilogue_i8<EPI, 2, 4, XTOK, XL_NT, ABL, XCH>(acc, g, smem, smem + XL_SMEM, m0, n0, 64 * wc, 128 * wt, tid, h, l31);
}


// ================================================================================================
// Persistent form of the 256 x 128 kernel: 2 workgroups per CU loop over tiles (tile = block + k * grid).
// What a relaunch per tile costs -- workgroup dispatch, the cold start of the DMA ring, table loads --
// is hidden: stage 0 of the NEXT tile is prefetched into LDS buffer 0 while this tile's epilogue runs
// (its int8 staging tile lives in buffers 1-2), and the next tile's bias / requant table is fetched
// inside the epilogue, before this tile's stores are issued, into the other half of a double-buffered
// LDS table, so the next main loop starts without a vmcnt(0) drain behind those stores.
// ================================================================================================
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
// One token per physical CU (XCC id, SE, SH, CU of HW_ID): the two co-resident workgroups of the persistent kernel take
// turns in their DMA-bound main loops (see gemm_i8_pers_kernel).  Zero between launches: every holder releases.
__device__ int g_cu_token[2048];

struct PersWork {
    int m0, n0, half;   // m0 < 0: none
};

IVIT_DEV PersWork pers_work(const GemmArgs& g, int i, int b, int G)
{
    const int F = g.tiles_m * g.tiles_n;
    const int t = b + i * G;
    if (t < g.split_from) {
        const PersTile pt = pers_tile(g, t);
        return PersWork{pt.m0, pt.n0, 0};
    }
    // the first index past this workgroup's full tiles: its half tile, if any
    const int first_past = (g.split_from - b + G - 1) / G;   // number of full tiles of workgroup b (b < G)
    const int hb = b;
    if (i == (b < g.split_from ? first_past : 0) && g.split_from + (hb >> 1) < F) {
        const PersTile pt = pers_tile(g, g.split_from + (hb >> 1));
        const int m0 = pt.m0 + 128 * (hb & 1);
        if (m0 < g.M) return PersWork{m0, pt.n0, 1};
    }
    return PersWork{-1, 0, 0};
}

template <int EPI, int EABL = 0>
__global__ __launch_bounds__(BIG_NT, 2) void gemm_i8_pers_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) char smem[PERS_SMEM];
    const int tid = threadIdx.x;
    const unsigned hw_id = __builtin_amdgcn_s_getreg((16 - 1) << 11 | (0 << 6) | 4);    // HW_ID[15:0]: .. cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned xcc_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | (0 << 6) | 20);   // XCC_ID[3:0]
    int* cu_token = &g_cu_token[((xcc_id & 7u) << 8) | ((hw_id >> 8) & 0xffu)];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wt = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int lrow = lane >> 2, lslot = lane & 3;
    const int nk = g.K / BK;
    using T = std::true_type;
    using F = std::false_type;

    const int8_t* asrc[4];
    const int8_t* wsrc[2];
    auto set_sources = [&](const PersWork& w) {   // a half tile uses asrc[0..1] only (token rows 0..127)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = 16 * (wave + 4 * i) + lrow;
            int c = lslot ^ ((row >> 2) & 3);
            asrc[i] = g.A + (int64_t)min(w.m0 + row, g.M - 1) * g.lda + 16 * c;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = 16 * (wave + 4 * i) + lrow;
            int c = lslot ^ ((row >> 2) & 3);
            wsrc[i] = g.W + (int64_t)min(w.n0 + row, g.N - 1) * g.ldw + 16 * c;
        }
    };
    // DMA piece `idx` of K step kt: token pieces first (4, or 2 for a half tile), then the 2 weight pieces
    auto issue_one = [&](int kt, int idx, auto half_tag) {
        constexpr int NA = decltype(half_tag)::value ? 2 : 4;
        char* base = smem + (kt % BIG_STAGES) * BIG_STAGE;
        const int koff = kt * BK;
        if (idx < NA)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[idx] + koff), (lptr_t)(base + 1024 * (wave + 4 * idx)), 16, 0,
                                             0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[idx - NA] + koff),
                                             (lptr_t)(base + BIG_A_BYTES + 1024 * (wave + 4 * (idx - NA))), 16, 0, 0);
    };
    auto issue = [&](int kt, auto half_tag) {
        constexpr int PIECES = decltype(half_tag)::value ? 4 : 6;
#pragma unroll
        for (int idx = 0; idx < PIECES; ++idx) issue_one(kt, idx, half_tag);
    };
    auto issue_rt = [&](int kt, int half) {
        if (half) issue(kt, T{});
        else issue(kt, F{});
    };

    // Fragment reads are issued as inline asm so that their completion is tracked HERE (explicit counted s_waitcnt tied to
    // the registers they guard) and not by the compiler's waitcnt insertion, which drains lgkmcnt to 0 in front of the
    // first MFMA after each group and so exposes a full LDS round trip per half step.  Per lane the token sub-tiles are
    // 2048 B apart and the 2 channel sub-tiles likewise (the swizzle term depends on (row >> 2) & 3 only), so each
    // operand needs one address register per k sub-step and immediate offsets.
    const unsigned smem_base = (unsigned)(__UINTPTR_TYPE__)(lptr_t)smem;
    const int wrow0 = 64 * wc + l31;
    const unsigned wbase[2] = {smem_base + (unsigned)(BIG_A_BYTES + swz(wrow0, h)),
                               smem_base + (unsigned)(BIG_A_BYTES + swz(wrow0, 2 + h))};

    // One work item: a full tile (256 tokens, wave tile 64 ch x 128 tok) or a half tile (128 tokens, 64 ch x 64 tok).
    auto run = [&](auto half_tag, const PersWork& cur, const PersWork& nxt, char* tab, char* tab_next) {
        constexpr bool HALF = decltype(half_tag)::value;
        constexpr int TJ = HALF ? 2 : 4;
        constexpr int WTOK = 32 * TJ;
        constexpr int PIECES = HALF ? 4 : 6;
        const int arow0 = WTOK * wt + l31;
        const unsigned abase[2] = {smem_base + (unsigned)swz(arow0, h), smem_base + (unsigned)swz(arow0, 2 + h)};
        v4i wf0[2], af0[TJ], wf1[2], af1[TJ];
        auto load_frags = [&](unsigned stage_off, int ks, v4i (&wf)[2], v4i (&af)[TJ]) {
            const unsigned wa = wbase[ks] + stage_off, aa = abase[ks] + stage_off;
            asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(wa));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wf[1]) : "v"(wa));
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[1]) : "v"(aa));
            if constexpr (TJ == 4) {
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[2]) : "v"(aa));
                asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[3]) : "v"(aa));
            }
        };
        // wait until at most one group of fragment reads (2 + TJ) / none (together with the DMA wait) is outstanding; the
        // "+v" ties order every later use of the guarded fragments after the wait
        auto wait_frags = [&](v4i (&wf)[2], v4i (&af)[TJ]) {
            if constexpr (TJ == 4)
                asm volatile("s_waitcnt lgkmcnt(6)"
                             : "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])::"memory");
            else
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1])::"memory");
        };
        auto wait_dma_and_frags = [&](auto dma_tag, v4i (&wf)[2], v4i (&af)[TJ]) {
            constexpr bool DMA = decltype(dma_tag)::value;
            if constexpr (TJ == 4) {
                if constexpr (DMA)
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)"
                                 : "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                                 : "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])::"memory");
            } else {
                if constexpr (DMA)
                    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1])::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1])::"memory");
            }
        };
        v16i acc[2][TJ];
        // Pipeline per K step kt (3 LDS stages, fragments double-buffered in registers):
        //   read frags(kt, ks=1)            | wait frags(kt, ks=0) (issued one half step ago; the new reads stay in flight)
        //   MFMA on frags(kt, ks=0) interleaved with the DMA of stage kt+2
        //   wait own DMA of stage kt+1 and all own LDS reads | barrier
        //   read frags(kt+1, ks=0)          | MFMA on frags(kt, ks=1)  (already complete: drained before the barrier)
        // RAW: stage kt+1 is read only after the barrier of step kt, which every wave reaches after its counted vmcnt.
        // WAR: the DMA of stage kt+2 overwrites the buffer of stage kt-1; it is issued after the barrier of step kt-1, and
        //      every wave waited lgkmcnt(0) (all its reads of stage kt-1 returned) before that barrier.
        auto step = [&](int kt, auto dma_tag, auto last_tag) {
            constexpr bool DMA = decltype(dma_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            load_frags((unsigned)((kt % BIG_STAGES) * BIG_STAGE), 1, wf1, af1);
            wait_frags(wf0, af0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf0[i], af0[j], acc[i][j], 0, 0, 0);
                    if constexpr (DMA)
                        if (TJ * i + j < PIECES) issue_one(kt + 2, TJ * i + j, half_tag);
                }
            __builtin_amdgcn_sched_barrier(0);
            wait_dma_and_frags(dma_tag, wf1, af1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (!LAST) load_frags((unsigned)(((kt + 1) % BIG_STAGES) * BIG_STAGE), 0, wf0, af0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf1[i], af1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };

        // stage 0 of this item is in flight (or landed); the table was written during the previous epilogue
        if (nk > 1) {
            issue(1, half_tag);
            if constexpr (HALF) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // own pieces of stage 0 (and everything older)
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                     // everyone's stage 0; table visible
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 b4 = *reinterpret_cast<const int4*>(tab + BCH * 8 + 4 * (64 * wc + 32 * i + 8 * q + 4 * h));
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j][4 * q + 0] = b4.x;
                    acc[i][j][4 * q + 1] = b4.y;
                    acc[i][j][4 * q + 2] = b4.z;
                    acc[i][j][4 * q + 3] = b4.w;
                }
            }
        // The main loop is bound by the CU's global->LDS DMA path (~29 B/clk/CU: a 24 KiB stage per ~830 cycles against
        // 512 cycles of MFMA), the epilogue by VALU.  Two workgroups that run their main loops at the same time just
        // halve each other's DMA rate and then sit in their epilogues together with the DMA path idle.  The per-CU token
        // makes them take turns: one streams its K loop at the full DMA rate while the other requantises and stores.
        if (g.cu_turns) {
            if (tid == 0)
                while (atomicCAS(cu_token, 0, 1) != 0) __builtin_amdgcn_s_sleep(8);
            __syncthreads();
        }
        load_frags(0u, 0, wf0, af0);
        int kt = 0;
        for (; kt + 2 < nk; ++kt) step(kt, T{}, F{});
        if (kt + 1 < nk) { step(kt, F{}, F{}); ++kt; }
        step(kt, F{}, T{});
        __syncthreads();   // all waves are done with every stage: buffers free
        if (g.cu_turns && tid == 0) atomicExch(cu_token, 0);

        // ---- prefetch stage 0 of the next item, then this item's epilogue (staging in buffers 1-2)
        const bool more = nxt.m0 >= 0;   // uniform
        if (more) {
            set_sources(nxt);
            issue_rt(0, nxt.half);
        }
        struct Hook {
            const GemmArgs& g;
            int n0, tid;
            char* dst;
            bool more;
            mutable PersTableLoad ld;
            IVIT_DEV void issue() const { if (more) ld = pers_table_issue(g, n0, tid); }
            IVIT_DEV void consume() const { if (more) pers_table_write(ld, dst, tid); }
        };
        Hook hook{g, nxt.n0, tid, tab_next, more, PersTableLoad{0u, 0, 0, false}};
        epilogue_i8<EPI, 2, TJ, (HALF ? 128 : BTOK), BIG_NT, EABL, BCH, Hook>(acc, g, smem + BIG_STAGE, tab, cur.m0, cur.n0,
                                                                           64 * wc, WTOK * wt, tid, h, l31, hook);
        __syncthreads();   // staging reads done before the next item's stage 1 DMA overwrites buffer 1
    };

    // ---- first item: table + stage 0
    const int G = gridDim.x, b = blockIdx.x;
    PersWork cur = pers_work(g, 0, b, G);
    if (cur.m0 < 0) return;   // uniform
    {
        PersTableLoad tl = pers_table_issue(g, cur.n0, tid);
        pers_table_write(tl, smem + PT_OFF, tid);
    }
    set_sources(cur);
    if (g.stagger && blockIdx.x < (unsigned)g.stagger) {   // see gemm_i8_big_kernel: de-phase the two co-resident groups
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;
        if (slot)
            for (int it = 0; it < g.stagger_units; ++it) __builtin_amdgcn_s_sleep(16);
    }
    issue_rt(0, cur.half);

    for (int it = 0; cur.m0 >= 0; ++it) {
        char* tab = smem + PT_OFF + (it & 1) * PT_BYTES;
        char* tab_next = smem + PT_OFF + ((it + 1) & 1) * PT_BYTES;
        const PersWork nxt = pers_work(g, it + 1, b, G);
        if (cur.half) run(T{}, cur, nxt, tab, tab_next);
        else run(F{}, cur, nxt, tab, tab_next);
        cur = nxt;
    }
}

// ================================================================================================
// Deep-ring form: ONE workgroup per CU (4 waves, wave tile 64 ch x 128 tok as above) with a FIVE-stage LDS ring
// (120 KiB), so that up to four stages (96 KiB) of LDS-DMA are in flight per CU.  Rationale: the
// global->LDS path has a latency of more than two K steps; with three stages per workgroup a stage is awaited one
// step after it was issued and every step waits for the DMA.  Here a stage is issued four steps before it is
// consumed, and the first four stages of the NEXT tile are issued before this tile's epilogue (which has its own
// staging area), so the main loop of a tile starts on data that has already landed.
// ================================================================================================
constexpr int RING_STAGES = 5;
constexpr int RING_BYTES = RING_STAGES * BIG_STAGE;            // 120 KiB
constexpr int RING_EPI_OFF = RING_BYTES;                        // 256 x 132 B int8 staging tile
constexpr int RING_EPI_BYTES = BTOK * (BCH + 4);
constexpr int RING_PT_OFF = RING_EPI_OFF + RING_EPI_BYTES;      // 2 x table
constexpr int RING_SMEM = RING_PT_OFF + 2 * PT_BYTES;           // 159 744 B <= 160 KiB

template <int EPI>
__global__ __launch_bounds__(BIG_NT, 1) void gemm_i8_ring_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) char smem[RING_SMEM];
    static_assert(RING_SMEM <= 160 * 1024, "LDS budget");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wt = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int lrow = lane >> 2, lslot = lane & 3;
    const int nk = g.K / BK;
    const int ntiles = g.tiles_m * g.tiles_n;
    using T = std::true_type;
    using F = std::false_type;

    const int8_t* asrc[4];
    const int8_t* wsrc[2];
    auto set_sources = [&](const PersTile& t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = 16 * (wave + 4 * i) + lrow;
            int c = lslot ^ ((row >> 2) & 3);
            asrc[i] = g.A + (int64_t)min(t.m0 + row, g.M - 1) * g.lda + 16 * c;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = 16 * (wave + 4 * i) + lrow;
            int c = lslot ^ ((row >> 2) & 3);
            wsrc[i] = g.W + (int64_t)min(t.n0 + row, g.N - 1) * g.ldw + 16 * c;
        }
    };
    // DMA piece `idx` (0..3 token tile, 4..5 weight tile) of K step kt into ring buffer kt % RING_STAGES
    auto issue_one = [&](int kt, int idx) {
        char* base = smem + (kt % RING_STAGES) * BIG_STAGE;
        const int koff = kt * BK;
        if (idx < 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[idx] + koff), (lptr_t)(base + 1024 * (wave + 4 * idx)), 16, 0,
                                             0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[idx - 4] + koff),
                                             (lptr_t)(base + BIG_A_BYTES + 1024 * (wave + 4 * (idx - 4))), 16, 0, 0);
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int idx = 0; idx < 6; ++idx) issue_one(kt, idx);
    };

    const unsigned smem_base = (unsigned)(__UINTPTR_TYPE__)(lptr_t)smem;
    const int wrow0 = 64 * wc + l31, arow0 = 128 * wt + l31;
    const unsigned wbase[2] = {smem_base + (unsigned)(BIG_A_BYTES + swz(wrow0, h)),
                               smem_base + (unsigned)(BIG_A_BYTES + swz(wrow0, 2 + h))};
    const unsigned abase[2] = {smem_base + (unsigned)swz(arow0, h), smem_base + (unsigned)swz(arow0, 2 + h)};
    v4i wf0[2], af0[4], wf1[2], af1[4];
    auto load_frags = [&](unsigned stage_off, int ks, v4i (&wf)[2], v4i (&af)[4]) {
        const unsigned wa = wbase[ks] + stage_off, aa = abase[ks] + stage_off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(wa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wf[1]) : "v"(wa));
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[3]) : "v"(aa));
    };
#define RING_TIE(wf, af) "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])
    v16i acc[2][4];
    // One K step.  AHEAD = number of later stages whose DMA may still be in flight when this step ends (each stage is
    // 6 pieces per wave): the counted vmcnt leaves exactly those outstanding, i.e. stage kt+1 has landed.
    // ISSUE: this step also issues the DMA of stage kt + RING_STAGES - 1 into the buffer freed by the previous step.
    auto step = [&](int kt, auto issue_tag, auto ahead_tag, auto last_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        constexpr int AHEAD = decltype(ahead_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        load_frags((unsigned)((kt % RING_STAGES) * BIG_STAGE), 1, wf1, af1);
        asm volatile("s_waitcnt lgkmcnt(6)" : RING_TIE(wf0, af0)::"memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf0[i], af0[j], acc[i][j], 0, 0, 0);
                if constexpr (ISSUE)
                    if (4 * i + j < 6) issue_one(kt + RING_STAGES - 1, 4 * i + j);
            }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (AHEAD == 3) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" : RING_TIE(wf1, af1)::"memory");
        else if constexpr (AHEAD == 2) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" : RING_TIE(wf1, af1)::"memory");
        else if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" : RING_TIE(wf1, af1)::"memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : RING_TIE(wf1, af1)::"memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (!LAST) load_frags((unsigned)(((kt + 1) % RING_STAGES) * BIG_STAGE), 0, wf0, af0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf1[i], af1[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    using A0 = std::integral_constant<int, 0>;
    using A1 = std::integral_constant<int, 1>;
    using A2 = std::integral_constant<int, 2>;
    using A3 = std::integral_constant<int, 3>;
    // issue the first min(nk, RING_STAGES - 1) stages of a tile
    auto prefetch_head = [&]() {
        const int nh = nk < RING_STAGES - 1 ? nk : RING_STAGES - 1;
        for (int kt = 0; kt < nh; ++kt) issue(kt);
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    PersTile cur = pers_tile(g, tile);
    {
        PersTableLoad tl = pers_table_issue(g, cur.n0, tid);
        pers_table_write(tl, smem + RING_PT_OFF, tid);
    }
    set_sources(cur);
    prefetch_head();

    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        char* tab = smem + RING_PT_OFF + (it & 1) * PT_BYTES;
        char* tab_next = smem + RING_PT_OFF + ((it + 1) & 1) * PT_BYTES;
        // The head stages of this tile were issued before the previous epilogue (or just above): wait for stage 0.
        // Everything older (the previous tile's stores included) is allowed to drain with it.
        {
            const int nh = nk < RING_STAGES - 1 ? nk : RING_STAGES - 1;   // stages in flight now
            // one stage stricter than needed: the previous epilogue's stores are younger than these pieces and may retire
            // out of order with respect to loads, so do not let them stand in for DMA pieces in the count
            if (nh >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (nh == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 b4 = *reinterpret_cast<const int4*>(tab + BCH * 8 + 4 * (64 * wc + 32 * i + 8 * q + 4 * h));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j][4 * q + 0] = b4.x;
                    acc[i][j][4 * q + 1] = b4.y;
                    acc[i][j][4 * q + 2] = b4.z;
                    acc[i][j][4 * q + 3] = b4.w;
                }
            }
        load_frags(0u, 0, wf0, af0);
        // steps that still issue a stage (kt + 4 < nk), then the drain: 3, 2, 1, 0 later stages in flight
        int kt = 0;
        for (; kt + RING_STAGES - 1 < nk; ++kt) step(kt, T{}, A3{}, F{});
        if (kt + 3 < nk) { step(kt, F{}, A2{}, F{}); ++kt; }
        if (kt + 2 < nk) { step(kt, F{}, A1{}, F{}); ++kt; }
        if (kt + 1 < nk) { step(kt, F{}, A0{}, F{}); ++kt; }
        step(kt, F{}, A0{}, T{});
        __syncthreads();   // all waves are done with every ring buffer

        // ---- head of the next tile into the (now free) ring, then this tile's epilogue from its own staging area
        const int next = tile + gridDim.x;
        const bool more = next < ntiles;   // uniform
        PersTile nxt = cur;
        if (more) {
            nxt = pers_tile(g, next);
            set_sources(nxt);
            prefetch_head();
        }
        struct Hook {
            const GemmArgs& g;
            int n0, tid;
            char* dst;
            bool more;
            mutable PersTableLoad ld;
            IVIT_DEV void issue() const { if (more) ld = pers_table_issue(g, n0, tid); }
            IVIT_DEV void consume() const { if (more) pers_table_write(ld, dst, tid); }
        };
        Hook hook{g, nxt.n0, tid, tab_next, more, PersTableLoad{0u, 0, 0, false}};
        epilogue_i8<EPI, 2, 4, BTOK, BIG_NT, 0, BCH, Hook>(acc, g, smem + RING_EPI_OFF, tab, cur.m0, cur.n0, 64 * wc,
                                                          128 * wt, tid, h, l31, hook);
        cur = nxt;
        __syncthreads();   // staging tile and table free for the next round
    }
#undef RING_TIE
}

template <int EPI>
int launch_gemm(GemmArgs& g, const char* name, ivit_stream_t stream)
{
    IVIT_REQUIRE(g.A && g.W && g.out, "%s: NULL operand", name);
    IVIT_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "%s: empty problem M=%d N=%d K=%d", name, g.M, g.N, g.K);
    IVIT_REQUIRE(g.K % BK == 0, "%s: K=%d must be a multiple of %d", name, g.K, BK);
    IVIT_REQUIRE(g.lda >= g.K && g.ldw >= g.K && g.lda % 16 == 0 && g.ldw % 16 == 0,
                 "%s: lda=%lld ldw=%lld must be >= K and multiples of 16", name, (long long)g.lda, (long long)g.ldw);
    IVIT_REQUIRE(((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.W % 16 == 0) && ((uintptr_t)g.out % 16 == 0),
                 "%s: operands must be 16-byte aligned", name);
    if (EPI == EPI_I32) {
        IVIT_REQUIRE(g.N % 4 == 0 && g.ldo % 4 == 0 && g.ldo >= g.N, "%s: N=%d ldo=%lld must be multiples of 4", name,
                     g.N, (long long)g.ldo);
    } else {
        IVIT_REQUIRE(g.m && g.e, "%s: NULL requantiser table", name);
        IVIT_REQUIRE(((uintptr_t)g.bias % 16 == 0), "%s: bias must be 16-byte aligned", name);
        IVIT_REQUIRE(g.N % 16 == 0, "%s: N=%d must be a multiple of 16", name, g.N);
        IVIT_REQUIRE(((uintptr_t)g.m % 16 == 0) && ((uintptr_t)g.e % 16 == 0), "%s: m/e tables must be 16-byte aligned",
                     name);
    }
    if (EPI == EPI_RQ || EPI == EPI_RESID)
        IVIT_REQUIRE(g.ldo >= g.N && g.ldo % 16 == 0, "%s: ldo=%lld must be >= N and a multiple of 16", name,
                     (long long)g.ldo);
    if (EPI == EPI_RESID)
        IVIT_REQUIRE(g.res && g.ldr >= g.N && g.ldr % 16 == 0 && ((uintptr_t)g.res % 16 == 0),
                     "%s: residual operand missing or misaligned", name);
    if (EPI == EPI_QKV) {
        IVIT_REQUIRE(g.tokens > 0 && g.heads > 0 && g.head_dim > 0 && g.head_dim % 16 == 0,
                     "%s: bad head geometry tokens=%d heads=%d head_dim=%d", name, g.tokens, g.heads, g.head_dim);
        IVIT_REQUIRE(g.N == 3 * g.heads * g.head_dim && g.M % g.tokens == 0,
                     "%s: N=%d != 3*heads*head_dim or M=%d %% tokens=%d != 0", name, g.N, g.M, g.tokens);
    }
    g.flags = g_debug_flags & (31 | 128 | 256 | 512);
    if constexpr (EPI != EPI_I32) {
        if (g.M >= 2048 && g.N % XCH == 0 && !g_force_small && g_kernel_choice != 1 &&
            (g_debug_flags & 32)) {
            g.tiles_m = (g.M + XTOK - 1) / XTOK;
            g.tiles_n = g.N / XCH;
            dim3 grid(g.tiles_m * g.tiles_n), blk(XL_NT);
            hipStream_t st = ivit_stream(stream);
            if (EPI == EPI_RQ && g.flags != 0) {
                switch (g.flags) {
                    case 1: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 1>), grid, blk, 0, st, g); break;
                    case 2: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 2>), grid, blk, 0, st, g); break;
                    case 4: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 4>), grid, blk, 0, st, g); break;
                    case 5: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 5>), grid, blk, 0, st, g); break;
                    case 6: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 6>), grid, blk, 0, st, g); break;
                    case 7: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 7>), grid, blk, 0, st, g); break;
                    case 8: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 8>), grid, blk, 0, st, g); break;
                    case 16: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 16>), grid, blk, 0, st, g); break;
                    default: hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI_RQ, 3>), grid, blk, 0, st, g); break;
                }
            } else {
                hipLaunchKernelGGL((gemm_i8_xl_kernel<EPI, 0>), grid, blk, 0, st, g);
            }
            IVIT_CHECK_LAUNCH(name);
        }
        if (g.M >= 2048 && g.N >= BCH && !g_force_small && g.flags == 0 && (g_debug_flags & 8192)) {
            g.stagger = 0;
            g.tiles_m = (g.M + BTOK - 1) / BTOK;
            g.tiles_n = (g.N + BCH - 1) / BCH;
            g.split_from = g.tiles_m * g.tiles_n;
            const int ntiles = g.tiles_m * g.tiles_n;
            hipLaunchKernelGGL((gemm_i8_ring_kernel<EPI>), dim3(ntiles < 256 ? ntiles : 256), dim3(BIG_NT), 0,
                               ivit_stream(stream), g);
            IVIT_CHECK_LAUNCH(name);
        }
        if (g.M >= 2048 && g.N >= BCH && !g_force_small && g.flags == 0 && !(g_debug_flags & 1024)) {
            g.stagger = (g_debug_flags & 64) ? 0 : 512;  // 2 workgroups x 256 CUs
            g.tiles_m = (g.M + BTOK - 1) / BTOK;
            g.tiles_n = (g.N + BCH - 1) / BCH;
            const int ntiles = g.tiles_m * g.tiles_n;
            // 2 resident workgroups x 256 CUs; debug bit 12: one workgroup per CU (extra dynamic LDS blocks the second)
            const bool one_per_cu = (g_debug_flags & 4096) != 0;
            const int SLOTS = one_per_cu ? 256 : 512;
            // tail split (see PersWork): R tiles of a last round that is at most half full become 2R half tiles.
            // Measured slower (proj 47.7 -> 50.0 us, fc2 162 -> 173 us): a CU left with one workgroup runs it nearly
            // twice as fast, so the sparse last round is not the cost the tile count suggests.  Opt-in (debug bit 11).
            const int rounds = ntiles / SLOTS, R = ntiles - rounds * SLOTS;
            const bool split = R > 0 && 2 * R <= SLOTS && (g_debug_flags & 2048);
            g.split_from = split ? rounds * SLOTS : ntiles;
            g.cu_turns = (g_debug_flags & 32768) ? 1 : 0;
            // start delay of the second co-resident workgroup: measured (scripts/gemm_ab.py, interleaved) 0..10 units are
            // equivalent and the former half-main-loop delay (26 / 98 units at K = 768 / 3072) cost 4-8 %: off by default
            g.stagger_units = (g_debug_flags >> 16) & 63;
            const int grid = rounds > 0 ? SLOTS : (split ? 2 * R : R);
            if (EPI == EPI_RQ && (g_debug_flags & 16384))   // A/B of epilogue variants (EPI_RQ only)
                hipLaunchKernelGGL((gemm_i8_pers_kernel<EPI_RQ, 32>), dim3(grid), dim3(BIG_NT), one_per_cu ? 20480 : 0,
                                   ivit_stream(stream), g);
            else
                hipLaunchKernelGGL((gemm_i8_pers_kernel<EPI>), dim3(grid), dim3(BIG_NT), one_per_cu ? 20480 : 0,
                                   ivit_stream(stream), g);
            IVIT_CHECK_LAUNCH(name);
        }
        if (g.M >= 2048 && g.N >= BCH && !g_force_small) {
            g.stagger = (g_debug_flags & 64) ? 0 : 512;  // 2 workgroups x 256 CUs
            g.tiles_m = (g.M + BTOK - 1) / BTOK;
            g.tiles_n = (g.N + BCH - 1) / BCH;
            dim3 grid(g.tiles_m * g.tiles_n), blk(BIG_NT);
            hipStream_t st = ivit_stream(stream);
            if (EPI == EPI_RQ && g.flags != 0) {  // perf ablations (scripts/gemm_ablate.py), EPI_RQ only
                switch (g.flags) {
                    case 1: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 1>), grid, blk, 0, st, g); break;
                    case 2: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 2>), grid, blk, 0, st, g); break;
                    case 3: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 3>), grid, blk, 0, st, g); break;
                    case 4: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 4>), grid, blk, 0, st, g); break;
                    case 5: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 5>), grid, blk, 0, st, g); break;
                    case 6: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 6>), grid, blk, 0, st, g); break;
                    case 7: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 7>), grid, blk, 0, st, g); break;
                    case 8: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 8>), grid, blk, 0, st, g); break;
                    case 16: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 16>), grid, blk, 0, st, g); break;
                    case 24: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 24>), grid, blk, 0, st, g); break;
                    case 11: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 11>), grid, blk, 0, st, g); break;
                    case 19: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 19>), grid, blk, 0, st, g); break;
                    case 133: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 133>), grid, blk, 0, st, g); break;
                    case 389: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 389>), grid, blk, 0, st, g); break;
                    case 512: g.res = (const int8_t*)g_stamp_buf; hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 512>), grid, blk, 0, st, g); break;
                    case 515: g.res = (const int8_t*)g_stamp_buf; hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 515>), grid, blk, 0, st, g); break;
                    case 516: g.res = (const int8_t*)g_stamp_buf; hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 516>), grid, blk, 0, st, g); break;
                    case 517: g.res = (const int8_t*)g_stamp_buf; hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 517>), grid, blk, 0, st, g); break;
                    case 518: g.res = (const int8_t*)g_stamp_buf; hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 518>), grid, blk, 0, st, g); break;
                    case 513: g.res = (const int8_t*)g_stamp_buf; hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 513>), grid, blk, 0, st, g); break;
                    default: hipLaunchKernelGGL((gemm_i8_big_kernel<EPI_RQ, 3>), grid, blk, 0, st, g); break;
                }
            } else {
                hipLaunchKernelGGL((gemm_i8_big_kernel<EPI, 0>), grid, blk, 0, st, g);
            }
            IVIT_CHECK_LAUNCH(name);
        }
    }
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(gemm_i8_kernel<EPI>, dim3(g.tiles_m * g.tiles_n), dim3(NT), 0, ivit_stream(stream), g);
    IVIT_CHECK_LAUNCH(name);
}

}  // namespace

IVIT_EXPORT int ivit_gemm_i8_requant(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                     const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, int M, int N,
                                     int K, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K;
    return launch_gemm<EPI_RQ>(g, "ivit_gemm_i8_requant", stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_residual(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                              const int32_t* bias, const uint32_t* m, const int32_t* e,
                                              const int8_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                              uint32_t m_res, int32_t e_res, int8_t* out, int64_t ldo, int M, int N,
                                              int K, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = out; g.ldo = ldo; g.res = res; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
    g.M_main = ivit_dyadic_to_double(m_main, e_main);
    g.M_res = ivit_dyadic_to_double(m_res, e_res);
    IVIT_REQUIRE(g.M_main < 1048576.0 && g.M_res < 1048576.0,
                 "ivit_gemm_i8_requant_residual: residual multiplier >= 2^20 is outside the int8 fast path");
    return launch_gemm<EPI_RESID>(g, "ivit_gemm_i8_requant_residual", stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_qkv(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                         const int32_t* bias, const uint32_t* m, const int32_t* e, int8_t* qkv,
                                         int tokens, int heads, int head_dim, int M, int N, int K,
                                         ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = qkv; g.ldo = 0; g.M = M; g.N = N; g.K = K;
    g.tokens = tokens; g.heads = heads; g.head_dim = head_dim;
    return launch_gemm<EPI_QKV>(g, "ivit_gemm_i8_requant_qkv", stream);
}

IVIT_EXPORT int ivit_gemm_i8_i32(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                 int32_t* out, int64_t ldo, int M, int N, int K, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K;
    return launch_gemm<EPI_I32>(g, "ivit_gemm_i8_i32", stream);
}

// test hook: 1 = always use the 128x128 register-staged kernel (so both kernels stay covered)
IVIT_EXPORT int ivit_debug_force_small_gemm(int on)
{
    g_force_small = (on == 1);   // 1: 128x128 register-staged kernel only
    g_kernel_choice = (on == 2); // 2: at most the 256x128 LDS-DMA kernel
    return IVIT_OK;
}

IVIT_EXPORT int ivit_debug_set_gemm_flags(int flags)
{
    g_debug_flags = flags;
    return IVIT_OK;
}

// diagnostic: device buffer (8 x uint64 per workgroup) receiving {HW_ID | XCC_ID<<32, t_start, t_loop_end, t_end}
// from the stamped build selected by ivit_debug_set_gemm_flags(512)
IVIT_EXPORT int ivit_debug_set_stamp_buffer(void* buf)
{
    g_stamp_buf = buf;
    return IVIT_OK;
}
