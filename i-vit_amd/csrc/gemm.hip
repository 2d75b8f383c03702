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
//
// This file holds the product kernels: the 128 x 128 register-staged kernel for small problems and raw int32 outputs,
// and the persistent 256 x 128 LDS-DMA kernel everything large goes through.  Other forms of the large kernel
// (relaunch per tile with ablation switches, 256 x 256 tiles, deep ring) live in gemm_lab.hip for A/B measurement.
#include "gemm_common.h"

namespace {

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
    if constexpr (EPI != EPI_I32 && EPI != EPI_RQ16 && EPI != EPI_RQ16_RES16) fill_rq_table(g, smem + SMEM_BYTES, n0, BN, tid);

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
    } else if constexpr (EPI == EPI_RQ16_RES16) {
        static_assert(SMEM_BYTES + BN * 8 >= 128 * RQ16S_RS, "staging tile");
        epilogue_rq16_res16_small(acc, g, smem, m0, n0, wm, wn, tid, h, l31);
        return;
    } else if constexpr (EPI == EPI_RQ16) {
        // 16-bit per-channel QuantAct from the registers: a lane holds 4 consecutive channels of one token per register quad,
        // i.e. one 8-byte store.  quant_utils.py:229-230 literally: float64 product (53-bit rounding), then RNE.
        int16_t* out = reinterpret_cast<int16_t*>(g.out);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = n0 + 64 * wn + 32 * i + 8 * q + 4 * h;
                if (c0 >= g.N) continue;
                const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
                const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
                const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z), dyadic_mult(m4.w, e4.w)};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int t = m0 + 64 * wm + 32 * j + l31;
                    if (t >= g.M) continue;
                    int o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = rne_clamp16((double)acc[i][j][4 * q + r] * Mc[r]);
                    }
                    int2 ow;
                    ow.x = (o[0] & 0xffff) | (o[1] << 16);
                    ow.y = (o[2] & 0xffff) | (o[3] << 16);
                    *reinterpret_cast<int2*>(out + (int64_t)t * g.ldo + c0) = ow;
                }
            }
        return;
    } else {
        epilogue_i8<EPI, 2, 2, BM, NT>(acc, g, smem, smem + SMEM_BYTES, m0, n0, 64 * wn, 64 * wm, tid, h, l31);
    }
}

// ================================================================================================
// Persistent form of the 256 x 128 kernel: 2 workgroups per CU loop over tiles (tile = block + k * grid).
// What a relaunch per tile costs -- workgroup dispatch, the cold start of the DMA ring, table loads --
// is hidden: stage 0 of the NEXT tile is prefetched into LDS buffer 0 while this tile's epilogue runs
// (its int8 staging tile lives in buffers 1-2), and the next tile's bias / requant table is fetched
// inside the epilogue, before this tile's stores are issued, into the other half of a double-buffered
// LDS table, so the next main loop starts without a vmcnt(0) drain behind those stores.
// ================================================================================================
// One token per physical CU (XCC id, SE, SH, CU of HW_ID): the two co-resident workgroups of the persistent kernel take
// turns in their DMA-bound main loops (see gemm_i8_pers_kernel).  Zero between launches: every holder releases.
#if IVIT_LAB
__device__ int g_cu_token[2048];
#endif

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
#if IVIT_LAB
    const unsigned hw_id = __builtin_amdgcn_s_getreg((16 - 1) << 11 | (0 << 6) | 4);    // HW_ID[15:0]: .. cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned xcc_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | (0 << 6) | 20);   // XCC_ID[3:0]
    int* cu_token = &g_cu_token[((xcc_id & 7u) << 8) | ((hw_id >> 8) & 0xffu)];
#endif
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
            // block layout: piece (wave + 4i) of the tile IS one 1 KB block row-group, already in LDS order
            if (g.a_blocks) asrc[i] = g.A + (int64_t)min((w.m0 >> 4) + wave + 4 * i, ((g.M + 15) >> 4) - 1) * (g.K >> 6) * 1024 + lane * 16;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = 16 * (wave + 4 * i) + lrow;
            int c = lslot ^ ((row >> 2) & 3);
            wsrc[i] = g.W + (int64_t)min(w.n0 + row, g.N - 1) * g.ldw + 16 * c;
            if (g.w_blocks) wsrc[i] = g.W + (int64_t)min((w.n0 >> 4) + wave + 4 * i, (g.N >> 4) - 1) * (g.K >> 6) * 1024 + lane * 16;
        }
    };
    // bytes from one K step to the next: 64 along a row, or one 1 KB block
    const int kstep_a = g.a_blocks ? 1024 : BK, kstep_w = g.w_blocks ? 1024 : BK;
    // DMA piece `idx` of K step kt: token pieces first (4, or 2 for a half tile), then the 2 weight pieces
    auto issue_one = [&](int kt, int idx, auto half_tag) {
        constexpr int NA = decltype(half_tag)::value ? 2 : 4;
        char* base = smem + (kt % BIG_STAGES) * BIG_STAGE;
        if (idx < NA)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[idx] + kt * kstep_a), (lptr_t)(base + 1024 * (wave + 4 * idx)), 16, 0,
                                             0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[idx - NA] + kt * kstep_w),
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
        v4i bq[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                lds_read16_async(bq[i][q], lds_addr(tab) + (unsigned)(BCH * 8 + 4 * (64 * wc + 32 * i + 8 * q + 4 * h)));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lds_wait(bq[i][q]);
                const v4i b4 = bq[i][q];
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
#if IVIT_LAB
        if (g.cu_turns) {
            if (tid == 0)
                while (atomicCAS(cu_token, 0, 1) != 0) __builtin_amdgcn_s_sleep(8);
            __syncthreads();
        }
#endif
        load_frags(0u, 0, wf0, af0);
        int kt = 0;
        for (; kt + 2 < nk; ++kt) step(kt, T{}, F{});
        if (kt + 1 < nk) { step(kt, F{}, F{}); ++kt; }
        step(kt, F{}, T{});
        __syncthreads();   // all waves are done with every stage: buffers free
#if IVIT_LAB
        if (g.cu_turns && tid == 0) atomicExch(cu_token, 0);
#endif

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
        int tid_o = tid;   // opaque: the epilogue's per-thread addresses are computed here, not carried through the main loop
        asm volatile("" : "+v"(tid_o));
        epilogue_i8<EPI, 2, TJ, (HALF ? 128 : BTOK), BIG_NT, EABL, BCH, Hook>(acc, g, smem + BIG_STAGE, tab, cur.m0, cur.n0,
                                                                           64 * wc, WTOK * wt, tid_o, (tid_o >> 5) & 1, tid_o & 31, hook);
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
// Weights-in-registers form: tile 128 tokens x 256 channels, 4 waves, wave w owns channels [64w, 64w + 64) x all 128 tokens
// (the same 2 x 4 MFMA tiles = 128 accumulators per lane as above).  A wave's weight fragments are read by no other wave of
// the workgroup, so they bypass the LDS: W comes pre-packed in MFMA-fragment order (ivit_pack_weight_frags_i8: one K step of
// a wave = 4 KB contiguous = four 1 KB global_load_dwordx4) and lands in registers two K steps ahead (three rotating
// buffers, nk % 3 == 0).  Only the token tile goes through the LDS (8 KB DMA per K step, read by all four waves):
//   LDS traffic per K step and workgroup   72 KB (24 written by DMA + 48 of fragment reads)  ->  40 KB (8 + 32)
//   bytes through the CU's vector-memory path: 24 KB either way
// The 256 x 128 kernel above keeps the LDS pipe busier (1152 of 128-B cycles per pair of co-resident K steps) than the
// MFMA pipe (1024); this one 640.  The epilogue is the shared one (staging tile in its own LDS region, so the next tile's
// first stage and first weight buffer are already in flight while it runs).
// ================================================================================================
constexpr int WR_TOK = 128, WR_CH = 256, WR_STAGES = 3;
constexpr int WR_STAGE = WR_TOK * BK;                       // 8 KiB
constexpr int WR_RING = WR_STAGES * WR_STAGE;               // 24 KiB
constexpr int WR_CS = WR_TOK * (WR_CH + 4);                 // 32.5 KiB epilogue staging
constexpr int WR_TAB = WR_CH * 12;                          // float2 lohi[256]; int bias[256]
constexpr int WR_LUT = WR_RING + WR_CS + 2 * WR_TAB;        // 256-byte output map (GemmArgs::lut)
constexpr int WR_SMEM = WR_LUT + 256;                       // 62.75 KiB: two workgroups per CU

struct WrWork { int m0, n0, half; };   // m0 < 0: none; half: 0 full tile (128 tok x 256 ch), 1 half tile (64 x 256), 2 narrow tile (128 x 128)

IVIT_DEV WrWork wr_tile(const GemmArgs& g, int t)
{
    const int nblk = g.tiles_m * g.tiles_n;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = t & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (t >> 3);
    const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
    return g.narrow ? WrWork{tm * WR_TOK, tn * (WR_CH / 2), 2} : WrWork{tm * WR_TOK, tn * WR_CH, 0};
}

// Work item i of workgroup b (grid G).  Tiles [0, split_from) are full tiles, tile t with workgroup t % G.  The remaining R
// tiles -- a last round that would leave most workgroups idle -- are processed as 2R half tiles of 64 tokens, one per
// workgroup 0 .. 2R-1, after that workgroup's full tiles: the tail then costs about half a tile time on twice the CUs.
IVIT_DEV WrWork wr_work(const GemmArgs& g, int i, int b, int G)
{
    const int F = g.tiles_m * g.tiles_n;
    const int t = b + i * G;
    if (t < g.split_from) return wr_tile(g, t);
    const int nfull = b < g.split_from ? (g.split_from - b + G - 1) / G : 0;   // full tiles of workgroup b
    if (i == nfull && g.split_from + (b >> 1) < F) {
        WrWork w = wr_tile(g, g.split_from + (b >> 1));
        w.m0 += 64 * (b & 1);
        w.half = 1;
        if (w.m0 < g.M) return w;
    }
    return WrWork{-1, 0, 0};
}

// ABL (lab build only): 1 no epilogue, 2 no weight loads in the loop, 4 no DMA in the loop, 8 no MFMA, 16 time stamps
//
// Narrow tiles (GemmArgs::narrow, round 4; S16 only): 128 tokens x 128 channels per work item, wave w = channel group w & 1
// x token half w >> 1, i.e. every wave does a half tile's arithmetic (64 ch x 64 tok, 16 accumulator tiles) on a full tile's
// token stage (all four waves fill it: two DMA pieces each) and its own channel group's weights (the two waves of a group
// load the same 4 KB per K step; the second hits L2).  For widths that 256-channel tiles fit badly (N = 384, 1152, 576 ...:
// DeiT-S, Swin stages 1-2) and for launches with fewer 256-channel tiles than the chip has workgroup slots (DeiT-S at batch 64:
// 150-600 tiles on 512 slots), where the per-tile serial time, not throughput, sets the kernel's duration.
//
// S16: the same tile on v_mfma_i32_16x16x64_i8 (IVIT_W_FRAGS16) instead of v_mfma_i32_32x32x32_i8.  Same cycles per K step (32
// instructions of 16 cycles against 16 of 32), same registers (32 accumulator tiles of 4), same bytes -- but under the power
// limit the chip holds a higher clock on the 16x16 shape: bare loops on random int8 at this wave tile 4 000 against 3 353 TOPS
// (2 029 against 1 687 MHz; with the token fragments re-read from LDS 3 572 against 3 099; equal on zeros --
// scripts/probes/mfma_shape_probe.hip, profiles/r03a_mfma_shape_*.txt).  A fragment is 16 rows x 64 K bytes with lane
// l = 16 c + r holding chunk c of row r, i.e. simply 1 KB in lane order: the LDS-DMA lays a token piece into its stage in
// exactly that order (per-lane source addresses pick chunk (r, c) out of the row-major or block-layout operand), so the
// fragment read is ds_read_b128 at lane * 16 -- conflict-free by construction, no swizzle on either side.
template <int EPI, int ABL = 0, bool S16 = false>
__global__ __launch_bounds__(BIG_NT, 2) void gemm_i8_wreg_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) char smem[WR_SMEM];
    char* const cs = smem + WR_RING;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int nk = g.K / BK;
    using T = std::true_type;
    using F = std::false_type;

    // ---- sources: 8 DMA pieces of 16 token rows per stage (wave w: pieces w, w + 4; a half tile has pieces 0..3 only);
    //      this wave's 64 weight rows
    const int8_t* asrc[2];
    const int8_t* wsrc = g.W;      // this lane's 16 bytes of the wave's weight piece 0, K step 0
    auto set_sources = [&](const WrWork& w) {
        // per-lane address parts are recomputed from an opaque copy of the lane id for every work item: hoisted out of the tile
        // loop they would occupy registers through the main loop (and were spilled)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int lrow_o = S16 ? (lane_o & 15) : (lane_o >> 2), lslot_o = S16 ? (lane_o >> 4) : (lane_o & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + 4 * i;
            const int row = 16 * piece + lrow_o;
            // 32x32 form: LDS slot 4r + s holds chunk s ^ ((r >> 2) & 3) of row r (= the block layout: identity copy);
            // S16: LDS slot 16 c + r holds chunk c of row r (fragment order)
            const int c = S16 ? lslot_o : (lslot_o ^ ((row >> 2) & 3));
            asrc[i] = g.A + (int64_t)min(w.m0 + row, g.M - 1) * g.lda + 16 * c;
            if (g.a_blocks) {  // uniform block origin + the chunk's position inside the 1 KB block
                const unsigned pos = S16 ? (unsigned)(4 * lrow_o + (lslot_o ^ ((lrow_o >> 2) & 3))) : (unsigned)lane_o;
                asrc[i] = g.A + (int64_t)min((w.m0 >> 4) + piece, ((g.M + 15) >> 4) - 1) * (g.K >> 6) * 1024 + pos * 16u;
            }
        }
        const int cg = min((w.n0 >> 6) + (w.half == 2 ? (wave & 1) : wave), ((g.N + 63) >> 6) - 1);
        wsrc = g.W + (int64_t)cg * nk * 4096 + (unsigned)lane_o * 16u;
    };
    const int kstep_a = g.a_blocks ? 1024 : BK;
    // The sources are RUNNING pointers: every issue advances its pointer in place to the next K step (the pieces of a work item are
    // issued strictly in K order), so no `base + kt * step` temporary pair lives beside the 64-bit bases in the main loop.
    auto issue_dma = [&](int kt, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)asrc[i], (lptr_t)(smem + (kt % WR_STAGES) * WR_STAGE + 1024 * (wave + 4 * i)), 16, 0, 0);
        asrc[i] += kstep_a;
    };
    // weight fragments of K step kt into buffer wr[.]: piece p = 2 i + ks (channel sub-tile i, K half ks) is 1 KB at p * 1024.
    // Per-lane 64-bit addresses (an SGPR base + lane offset form depends on the compiler keeping the base in SGPRs, which it
    // does not under register pressure)
    v4i wr0[4], wr1[4], wr2[4];
    auto issue_w = [&](v4i (&wr)[4], int kt, int p) {
        (void)kt;
        const int8_t* src = wsrc;
        if (p == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wr[0]) : "v"(src));
        if (p == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(wr[1]) : "v"(src));
        if (p == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(wr[2]) : "v"(src));
        if (p == 3) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(wr[3]) : "v"(src));
            wsrc += 4096;     // the next K step's four pieces
        }
    };
    // what goes out ahead of a work item: stage 0, weight buffer 0, stage 1 (the item's own first step issues weight buffer 1)
    auto prefetch = [&](const WrWork& w) {
        set_sources(w);
        issue_dma(0, 0);
        if (w.half != 1) issue_dma(0, 1);
#pragma unroll
        for (int p = 0; p < 4; ++p) issue_w(wr0, 0, p);
        issue_dma(1, 0);
        if (w.half != 1) issue_dma(1, 1);
    };

    const unsigned smem_base = (unsigned)(__UINTPTR_TYPE__)(lptr_t)smem;
    const unsigned abase[2] = {smem_base + (S16 ? (unsigned)lane * 16u : (unsigned)swz(l31, h)),
                               smem_base + (S16 ? (unsigned)lane * 16u : (unsigned)swz(l31, 2 + h))};
    unsigned long long tv[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ABL & 2048: stamps kept in registers until the tile ends
    unsigned long long* stamp = nullptr;   // ABL & 16: [block][tile < 4][32]: tile start, loop start, loop end, epilogue end, step starts

    auto table_issue = [&](int n0) {
        PersTableLoad r{0u, 0, 0, false};
        const int c = n0 + tid;
        if (c < g.N) {
            r.m = g.m[c];
            r.e = g.e[c];
            r.bias = g.bias ? g.bias[c] : 0;
            r.valid = true;
        }
        return r;
    };
    auto table_write = [&](const PersTableLoad& r, char* tab) {
        float2 lh = make_float2(0.f, 0.f);
        if (r.valid) {
            const double M = dyadic_mult(r.m, r.e);
            const float mf = (float)M;
            const int bits = __float_as_int(mf);
            lh.x = ((double)mf > M) ? __int_as_float(bits - 1) : mf;
            lh.y = ((double)mf < M) ? __int_as_float(bits + 1) : mf;
            if constexpr (S16) {   // two more float32 steps on either side: see epilogue_i8_16 (no range test on the accumulator)
                lh.x = __int_as_float(__float_as_int(lh.x) - 2);
                lh.y = __int_as_float(__float_as_int(lh.y) + 2);
            }
        }
        int tid_w = tid;       // opaque: no loop-invariant address part is carried (and spilled) across the main loop
        asm volatile("" : "+v"(tid_w));
        reinterpret_cast<float2*>(tab)[tid_w] = lh;
        reinterpret_cast<int*>(tab + WR_CH * 8)[tid_w] = r.bias;
    };

    // One work item: a full tile (128 tokens: 4 token sub-tiles per wave), a half tile (64 tokens: 2), or a narrow tile (128
    // tokens x 128 channels: 2 token sub-tiles per wave out of the wave's token half, DMA as for a full tile).
    auto run = [&](auto mode_tag, const WrWork& cur, const WrWork& nxt, char* tab, char* tab_next) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool HALF = MODE == 1, NARROW = MODE == 2;
        static_assert(!NARROW || S16, "narrow tiles exist for the 16x16x64 form only");
        constexpr int TJ = MODE ? 2 : 4;
        constexpr int NP = HALF ? 1 : 2;          // DMA pieces per wave and K step
        const int wch = NARROW ? 64 * (wave & 1) : 64 * wave;          // this wave's channels inside the tile
        const int wtok = NARROW ? 64 * (wave >> 1) : 0;                // ... and its first token
        const unsigned abase_s16 = abase[0] + (NARROW ? (unsigned)(wave >> 1) * 4096u : 0u);
        v4i af0[TJ], af1[TJ];
        // 32x32 form: K half ks of the TJ token sub-tiles of 32;  S16: token sub-tiles TJ ks .. TJ ks + TJ - 1 of 16, all 64 K bytes
        auto load_frags = [&](auto stage_tag, auto ks_tag, v4i (&af)[TJ]) {
            constexpr int ST = decltype(stage_tag)::value, ks = decltype(ks_tag)::value;
            constexpr unsigned stage_off = (unsigned)(ST * WR_STAGE);
            if constexpr (S16) {   // one base register (lane * 16), everything else in the offset field
                constexpr int O = ST * WR_STAGE + ks * TJ * 1024;
                lds_read16_async_off<O>(af[0], abase_s16);
                lds_read16_async_off<O + 1024>(af[1], abase_s16);
                if constexpr (TJ == 4) {
                    lds_read16_async_off<O + 2048>(af[2], abase_s16);
                    lds_read16_async_off<O + 3072>(af[3], abase_s16);
                }
            } else {
                const unsigned aa = abase[ks] + stage_off;
                asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(aa));
                asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[1]) : "v"(aa));
                if constexpr (TJ == 4) {
                    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[2]) : "v"(aa));
                    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[3]) : "v"(aa));
                }
            }
        };
        auto wait_frags = [&](v4i (&af)[TJ]) {   // at most the newest group of reads outstanding
            if constexpr (TJ == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[0]), "+v"(af[1])::"memory");
        };
        // own DMA pieces and weight registers of the NEXT K step have landed (the NP + 4 operations of the step after it may be
        // in flight), and every LDS read of this wave has returned
        auto wait_next = [&](auto inflight_tag, v4i (&af)[TJ], v4i (&wn)[4]) {
            constexpr bool INFLIGHT = decltype(inflight_tag)::value && !(ABL & 6);
            if constexpr (TJ == 4) {
                if constexpr (INFLIGHT)
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)"
                                 : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(wn[3])::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                                 : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(wn[3])::"memory");
            } else {
                if constexpr (INFLIGHT && NARROW)
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(wn[3])::"memory");
                else if constexpr (INFLIGHT)
                    asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(wn[3])::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(wn[3])::"memory");
            }
        };
        v16i acc[S16 ? 1 : 2][S16 ? 1 : TJ];      // 32x32 form: channel sub-tile i (32) x token sub-tile j (32)
        // S16: channel sub-tile i (16) x token sub-tile j (16).  The MFMAs are issued through inline asm with the accumulator TIED
        // (destination = source C): as builtin calls the register allocator kept moving the 32 four-register tiles around (82 of
        // 288 MFMAs not in place, ~200 v_mov_b64, and the kernel over its register budget).  What the compiler no longer does for
        // these instructions, and why that is sound here: (a) MFMA -> MFMA on the same accumulator: a tile gets ONE MFMA per K
        // step, 31 others lie between two of its updates; (b) operands come from ds_read / global_load behind explicit s_waitcnt
        // (memory results need no VALU-to-MFMA wait states), the bias preload is VALU moves > 10 instructions before the first
        // MFMA; (c) the first VALU read of an accumulator is in the epilogue, behind the s_nop pair after the loop.
        v4i acc16[S16 ? 4 : 1][S16 ? 2 * TJ : 1];
        auto mfma16 = [&](v4i& c, const v4i& a, const v4i& b) {
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        };
        // K step kt on weight buffer wc (current), wn (next: waited for here), wf (the one after: loaded here)
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        // stage_tag: kt % WR_STAGES as a compile-time constant (nk % 3 == 0: the unrolled steps know their stage)
        auto step = [&](int kt, v4i (&wc)[4], v4i (&wn)[4], v4i (&wf)[4], auto issue_tag, auto last_tag, auto stage_tag) {
            constexpr bool ISSUE = decltype(issue_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr int ST = decltype(stage_tag)::value;
            if constexpr (ABL & 16)
                if (!(ABL & 2048) && stamp && kt < 12) stamp[4 + kt] = __builtin_amdgcn_s_memtime();
            load_frags(stage_tag, I1{}, af1);
            wait_frags(af0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (S16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        if constexpr (!(ABL & 8)) mfma16(acc16[i][j], wc[i], af0[j]);
                        else asm volatile("" : "+v"(acc16[i][j]) : "v"(wc[i]), "v"(af0[j]));
                        if constexpr (ISSUE) {   // NP + 4 loads of K step kt + 2: one behind every other MFMA (full tile: 16 MFMAs,
                            const int n = TJ * i + j;   // six loads), behind each of the first five / six (half / narrow tile: 8 MFMAs)
                            const int u = MODE ? n : (n >> 1);
                            if (MODE || (n & 1) == 0) {
                                if (u < NP) { if constexpr (!(ABL & 4)) issue_dma(kt + 2, u); }
                                else if (u < NP + 4) { if constexpr (!(ABL & 2)) issue_w(wf, kt + 2, u - NP); }
                            }
                        }
                    }
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    if constexpr (!(ABL & 8)) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wc[2 * i], af0[j], acc[i][j], 0, 0, 0);
                    else asm volatile("" : "+v"(acc[i][j]) : "v"(wc[2 * i]), "v"(af0[j]));
                    if constexpr (ISSUE) {   // NP + 4 loads of K step kt + 2 spread over the MFMAs of this half step
                        const int n = TJ * i + j;
                        if constexpr (HALF) {    // four MFMAs, five loads: DMA piece and weight piece 0 behind the first
                            if (n == 0) { if constexpr (!(ABL & 4)) issue_dma(kt + 2, 0); }
                            if constexpr (!(ABL & 2)) issue_w(wf, kt + 2, n);
                        } else {
                            if (n < NP) { if constexpr (!(ABL & 4)) issue_dma(kt + 2, n); }
                            else if (n < NP + 4) { if constexpr (!(ABL & 2)) issue_w(wf, kt + 2, n - NP); }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            wait_next(issue_tag, af1, wn);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (!LAST) load_frags(std::integral_constant<int, (ST + 1) % WR_STAGES>{}, I0{}, af0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (S16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        if constexpr (!(ABL & 8)) mfma16(acc16[i][TJ + j], wc[i], af1[j]);
                        else asm volatile("" : "+v"(acc16[i][TJ + j]) : "v"(wc[i]), "v"(af1[j]));
                    }
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    if constexpr (!(ABL & 8)) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wc[2 * i + 1], af1[j], acc[i][j], 0, 0, 0);
                    else asm volatile("" : "+v"(acc[i][j]) : "v"(wc[2 * i + 1]), "v"(af1[j]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };

        // stage 0, weight buffer 0 and stage 1 of this item are in flight (or landed); the table was written during the last epilogue
#pragma unroll
        for (int p = 0; p < 4; ++p) issue_w(wr1, 1, p);
        // They have landed: a workgroup's first item waits for them in the kernel prologue, every later one in the previous item's
        // epilogue (Hook::consume).  Both waits are UNCONDITIONAL in the instruction stream (round 4): csrc/check_isa.py follows
        // registers with loads in flight through the control flow graph without knowing which branches exclude each other.
        // (A counted wait HERE would also drain the epilogue's stores: vmcnt counts them too.)
        asm volatile("" : "+v"(wr0[0]), "+v"(wr0[1]), "+v"(wr0[2]), "+v"(wr0[3])::"memory");
        __builtin_amdgcn_s_barrier();     // everyone's stage 0; table visible
        asm volatile("" ::: "memory");
        if constexpr (S16) {   // lane (g4, l15): channels 64 wave + 16 i + 4 g4 + r of register r
            int lane_b = lane;     // opaque: the address is derived here, not carried through the tile loop (it was spilled)
            asm volatile("" : "+v"(lane_b));
            const unsigned ba = lds_addr(tab) + (unsigned)(WR_CH * 8 + 4 * (wch + 4 * (lane_b >> 4)));
            v4i bq[4];
            lds_read16_async_off<0>(bq[0], ba);
            lds_read16_async_off<64>(bq[1], ba);
            lds_read16_async_off<128>(bq[2], ba);
            lds_read16_async_off<192>(bq[3], ba);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lds_wait(bq[i]);
#pragma unroll
                for (int j = 0; j < 2 * TJ; ++j) acc16[i][j] = bq[i];
            }
        } else {
        v4i bq[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                lds_read16_async(bq[i][q], lds_addr(tab) + (unsigned)(WR_CH * 8 + 4 * (64 * wave + 32 * i + 8 * q + 4 * h)));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lds_wait(bq[i][q]);
                const v4i b4 = bq[i][q];
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j][4 * q + 0] = b4.x;
                    acc[i][j][4 * q + 1] = b4.y;
                    acc[i][j][4 * q + 2] = b4.z;
                    acc[i][j][4 * q + 3] = b4.w;
                }
            }
        }
        load_frags(I0{}, I0{}, af0);
        if constexpr (ABL & 16)
            { if constexpr (ABL & 2048) tv[1] = __builtin_amdgcn_s_memtime(); else if (stamp) stamp[1] = __builtin_amdgcn_s_memtime(); }
        int kt = 0;
        for (; kt + 3 < nk; kt += 3) {
            step(kt, wr0, wr1, wr2, T{}, F{}, I0{});
            step(kt + 1, wr1, wr2, wr0, T{}, F{}, I1{});
            step(kt + 2, wr2, wr0, wr1, T{}, F{}, I2{});
        }
        step(kt, wr0, wr1, wr2, T{}, F{}, I0{});
        step(kt + 1, wr1, wr2, wr0, F{}, F{}, I1{});
        step(kt + 2, wr2, wr0, wr1, F{}, T{}, I2{});
        if constexpr (ABL & 16)
            { if constexpr (ABL & 2048) tv[2] = __builtin_amdgcn_s_memtime(); else if (stamp) stamp[2] = __builtin_amdgcn_s_memtime(); }
        // every wave's reads of every stage returned before the barrier of the last step: the ring is free
        const bool more = nxt.m0 >= 0;   // uniform
        // Always: a workgroup's last item requests its own first stages again (valid addresses, a free ring, nobody reads them).
        // With the request under `if (more)` and its wait under another `if (more)`, a checker that does not correlate the two
        // branches sees a path with the loads in flight forever; the cost of the extra request is ten loads per workgroup.
        prefetch(more ? nxt : cur);
        struct Hook {
            decltype(table_issue)& ti;
            decltype(table_write)& tw;
            int n0;
            char* dst;
            bool more;
            mutable PersTableLoad ld;
            IVIT_DEV void issue() const { if (more) ld = ti(n0); }
            // also this wave's share of the next item's first stages and weight buffer (issued before the epilogue; every load
            // outstanding here is older than the stores that follow): landed before the barrier at the next item's start
            IVIT_DEV void consume() const
            {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (more) tw(ld, dst);
            }
        };
        Hook hook{table_issue, table_write, nxt.n0, tab_next, more, PersTableLoad{0u, 0, 0, false}};
        if constexpr (ABL & 1) {
            int sum = 0;
            if constexpr (S16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * TJ; ++j) sum ^= acc16[i][j][0] ^ acc16[i][j][1] ^ acc16[i][j][2] ^ acc16[i][j][3];
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum ^= acc[i][j][r];
            }
            if (sum == 0x12345679) reinterpret_cast<int*>(g.out)[tid] = sum;
            hook.issue();
            hook.consume();
        } else {
            int tid_o = tid;   // opaque: the epilogue's per-thread addresses are computed here, not carried through the main loop
            asm volatile("" : "+v"(tid_o));
            if constexpr (S16) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // MFMA results -> VALU reads: see mfma16 (c)
            if constexpr (S16 && (EPI == EPI_RQ || EPI == EPI_RESID || EPI == EPI_QKV) && !(ABL & 8192)) {
                // straight from the registers (lane transpose, no LDS staging); ABL bit 13: the staged form below (A/B, lab)
                epilogue_direct_16<EPI, 2 * TJ, (ABL & (64 | 2048 | 4096 | 32768)), Hook>(acc16, g, tab, cur.m0 + wtok, cur.n0, wch, (tid_o >> 4) & 3, tid_o & 15, hook,
                                                                               g.lut ? reinterpret_cast<const unsigned char*>(smem + WR_LUT) : nullptr,
                                                                               (ABL & 2048) ? tv : nullptr);
                if constexpr ((ABL & 32768) != 0) {      // ShiftGELU of a token panel by the workgroup that completes it (staging region: free here)
                    __builtin_amdgcn_s_setprio(0);
                    gelu_panel_phase<BIG_NT>(g, cs, cur.m0, cur.half, tid_o);
                }
            } else if constexpr (S16) {
                static_assert(!S16 || EPI != EPI_RQ16_RES16, "the 16-bit epilogue exists for the 32x32 form only");
                epilogue_i8_16<EPI, 2 * TJ, BIG_NT, (ABL & (64 | 2048 | 4096)), WR_CH, Hook>(acc16, g, cs, tab, cur.m0, cur.n0, 64 * wave, tid_o, (tid_o >> 4) & 3,
                                                                          tid_o & 15, hook,
                                                                          g.lut ? reinterpret_cast<const unsigned char*>(smem + WR_LUT) : nullptr, (ABL & 2048) ? tv : nullptr);
            } else if constexpr (EPI == EPI_RQ16_RES16)
                epilogue_rq16_res16<TJ, BIG_NT, Hook>(acc, g, cs, cur.m0, cur.n0, 64 * wave, tid_o, (tid_o >> 5) & 1, tid_o & 31, hook);
            else
            epilogue_i8<EPI, 2, TJ, 32 * TJ, BIG_NT, (ABL & 64), WR_CH, Hook>(acc, g, cs, tab, cur.m0, cur.n0, 64 * wave, 0, tid_o, (tid_o >> 5) & 1,
                                                                      tid_o & 31, hook,
                                                                      g.lut ? reinterpret_cast<const unsigned char*>(smem + WR_LUT) : nullptr);
            if constexpr (ABL & 32) __builtin_amdgcn_s_setprio(0);   // lab: main loops back at priority 0 (the epilogue raises it to 2)
        }
    };

    const int G = gridDim.x, b = blockIdx.x;
    WrWork cur = wr_work(g, 0, b, G);
    if (cur.m0 < 0) return;   // uniform
    table_write(table_issue(cur.n0), smem + WR_RING + WR_CS);
    if (g.lut) smem[WR_LUT + tid] = (char)g.lut[tid];     // BIG_NT == 256; visible after the first tile's barriers
    prefetch(cur);
    // stage 0 and weight buffer 0 of the first item: at most ONE younger operation may stay in flight (a half tile has one
    // stage-1 DMA piece behind the weight loads, a full tile two: the stricter count serves both without a second branch on
    // `half`, which a checker that does not correlate branches could pair with the wrong prefetch)
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");

    for (int it = 0; cur.m0 >= 0; ++it) {
        char* tab = smem + WR_RING + WR_CS + (it & 1) * WR_TAB;
        char* tab_next = smem + WR_RING + WR_CS + ((it + 1) & 1) * WR_TAB;
        const WrWork nxt = wr_work(g, it + 1, b, G);
        if constexpr (ABL & 16) {
            unsigned long long* sbase = (ABL & 2048) ? g.stamp : reinterpret_cast<unsigned long long*>(const_cast<int8_t*>(g.res));
            stamp = (tid == 0 && it < 4) ? sbase + (blockIdx.x * 4 + it) * 32 : nullptr;
            if constexpr (ABL & 2048) {    // deferred form: nothing is stored before the tile ends (a store in flight would join the
                tv[0] = __builtin_amdgcn_s_memtime();   // counted vmcnt waits of the main loop and the vmcnt(0) of the epilogue)
                tv[7] = __builtin_amdgcn_s_memrealtime();
            } else if (stamp) {
                stamp[0] = __builtin_amdgcn_s_memtime();
                stamp[16] = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: the shader clock follows from the pair
            }
        }
        if constexpr (S16 && (EPI == EPI_RQ || EPI == EPI_RESID || EPI == EPI_QKV) && !(ABL & (8192 | 32768 | 16))) {
            if (cur.half == 2) run(std::integral_constant<int, 2>{}, cur, nxt, tab, tab_next);
            else if (cur.half) run(std::integral_constant<int, 1>{}, cur, nxt, tab, tab_next);
            else run(std::integral_constant<int, 0>{}, cur, nxt, tab, tab_next);
        } else {
            if (cur.half) run(std::integral_constant<int, 1>{}, cur, nxt, tab, tab_next);
            else run(std::integral_constant<int, 0>{}, cur, nxt, tab, tab_next);
        }
        if constexpr (ABL & 16)
            {
                if constexpr (ABL & 2048) {
                    if (stamp) {
                        stamp[0] = tv[0]; stamp[1] = tv[1]; stamp[2] = tv[2]; stamp[3] = __builtin_amdgcn_s_memtime();
                        stamp[20] = tv[4]; stamp[21] = tv[5]; stamp[22] = tv[6]; stamp[16] = tv[7];
                    }
                } else if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
            }
        cur = nxt;
    }
}

// ================================================================================================
// Skinny-K form (round 4): K <= 128, N <= 320 -- Swin stage 0 (patch embedding K = 64 / N = 96, qkv K = 128 / N = 288; 401 408 rows at batch
// 128).  Such a GEMM is a streaming pass: 154-193 MB of HBM traffic against 22 GOP.  In the tile kernels above it ran at 1.6-2.1 x its
// HBM time, because their epilogues (LDS-staged, two waves per SIMD, N = 288 padded to 384) -- not their two K steps -- fill the time.
// Here the whole weight matrix sits in LDS in v_mfma_i32_16x16x32_i8 fragment order (<= 48 KB), a wave takes strips of 16 tokens:
// K / 32 eight-byte loads per lane, N / 16 x K / 32 MFMAs with the weight fragment read from LDS, the requantisation (float32 bracket
// certificate, float64 on a failed batch: epilogue_direct_16's arithmetic) straight on the accumulators, one 16-byte store per lane and
// batch of four channel sub-tiles.  No inline asm: the compiler schedules it; two workgroups of eight waves per CU.  EPI_RQ (row-major
// output) and EPI_QKV.  Measured (Swin-T b128, profiles/r04x_*): patch embedding 37 -> 24 us, qkv 65.6 -> 62-64 us; at N = 384 (fc1) it lost
// (72 against 63 us) and the tile kernel keeps that shape.
// ================================================================================================
constexpr int SK_MAXN = 320, SK_MAXK = 128, SK_NT = 512, SK_WPB = SK_NT / 64;
constexpr int SK_SMEM = SK_MAXN * SK_MAXK + SK_MAXN * 12 + (SK_MAXN / 4) * 4;      // fragments | float2 lohi[N] | int bias[N] | u32 coff[N / 4]

// NKS = K / 32 at compile time (2: patch embedding, 4: stage-0 qkv; 0: any K, run time).  With a run-time K the `ks < nks` test around
// every MFMA became a scalar branch and each weight fragment a ds_read_b64 / s_waitcnt lgkmcnt(0) / v_mfma triple: 16 exposed LDS
// latencies per batch of four channel sub-tiles.  Straight-line, a batch's fragment reads are issued together (late round 4).
template <int EPI, int NKS = 0>
__global__ __launch_bounds__(SK_NT, 2) void gemm_i8_skinny_kernel(GemmArgs g)
{
    static_assert(EPI == EPI_RQ || EPI == EPI_QKV, "skinny-K form: int8 outputs");
    __shared__ __attribute__((aligned(16))) char smem[SK_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int nks = NKS ? NKS : (g.K >> 5), nsub = g.N >> 4;
    float2* const lohi = reinterpret_cast<float2*>(smem + SK_MAXN * SK_MAXK);
    int* const bias = reinterpret_cast<int*>(smem + SK_MAXN * SK_MAXK + SK_MAXN * 8);
    unsigned* const coff = reinterpret_cast<unsigned*>(smem + SK_MAXN * SK_MAXK + SK_MAXN * 12);      // EPI_QKV: head-major offset of channels 4 j ..
    // ---- weights: fragment (i, ks) = 512 bytes.  The MFMA's K slots are a permutation of the K index (a dot product does not care):
    //      slot (ks, g4, b) = k index 8 nks g4 + 8 ks + b, so that a lane's token bytes are 8 nks CONTIGUOUS bytes of its row (one or
    //      two 16-byte loads, four lanes = the whole row) instead of nks pieces of 8; lane L of a fragment holds W[16 i + (L & 15)][..]
    for (int q = tid; q < nsub * nks * 64; q += SK_NT) {
        const int f = q >> 6, L = q & 63, i = f / nks, ks = f - i * nks;
        *reinterpret_cast<long*>(smem + (size_t)q * 8) =
            *reinterpret_cast<const long*>(g.W + (int64_t)(16 * i + (L & 15)) * g.ldw + 8 * nks * (L >> 4) + 8 * ks);
    }
    if constexpr (EPI == EPI_QKV) {
        const int cdim = g.heads * g.head_dim, nb = g.M / g.tokens;
        for (int j = tid; j < (g.N >> 2); j += SK_NT) {
            const int c = 4 * j, which = c / cdim, rem = c - which * cdim;
            const int hh = rem / g.head_dim, d0 = rem - hh * g.head_dim;
            coff[j] = (unsigned)(((which * nb * g.heads + hh) * g.tokens) * g.head_dim + d0);
        }
    }
    for (int c = tid; c < g.N; c += SK_NT) {
        const int bv = g.bias ? g.bias[c] : 0;      // requested with m / e, not after the float64 arithmetic on them (common.h ln_build_table)
        const double M = dyadic_mult(g.m[c], g.e[c]);
        const float mf = (float)M;
        const int bits = __float_as_int(mf);
        float lo = ((double)mf > M) ? __int_as_float(bits - 1) : mf, hi = ((double)mf < M) ? __int_as_float(bits + 1) : mf;
        lo = __int_as_float(__float_as_int(lo) - 2);      // widened by two float32 steps: no range test on the accumulator
        hi = __int_as_float(__float_as_int(hi) + 2);      // (gemm_common.h epilogue_i8_16)
        lohi[c] = make_float2(lo, hi);
        bias[c] = bv;
    }
    __syncthreads();
    int8_t* const out = reinterpret_cast<int8_t*>(g.out);
    const int nstrips = (g.M + 15) >> 4;
    for (int strip = blockIdx.x * SK_WPB + wave; strip < nstrips; strip += gridDim.x * SK_WPB) {
        const int t = 16 * strip + l15;                   // this lane's token
        const int8_t* arow = g.A + (int64_t)min(t, g.M - 1) * g.lda + 8 * nks * g4;
        long bf[SK_MAXK / 32] = {0l, 0l, 0l, 0l};
        if (nks == 4) {          // uniform: K = 128, 32 bytes per lane
            const int4 u0 = *reinterpret_cast<const int4*>(arow), u1 = *reinterpret_cast<const int4*>(arow + 16);
            bf[0] = ((long)(unsigned)u0.y << 32) | (unsigned)u0.x; bf[1] = ((long)(unsigned)u0.w << 32) | (unsigned)u0.z;
            bf[2] = ((long)(unsigned)u1.y << 32) | (unsigned)u1.x; bf[3] = ((long)(unsigned)u1.w << 32) | (unsigned)u1.z;
        } else if (nks == 2) {   // K = 64, 16 bytes per lane
            const int4 u0 = *reinterpret_cast<const int4*>(arow);
            bf[0] = ((long)(unsigned)u0.y << 32) | (unsigned)u0.x; bf[1] = ((long)(unsigned)u0.w << 32) | (unsigned)u0.z;
        } else {
#pragma unroll
            for (int ks = 0; ks < SK_MAXK / 32; ++ks) bf[ks] = ks < nks ? *reinterpret_cast<const long*>(arow + 8 * ks) : 0l;
        }
        // EPI_QKV: image and token of this lane's row (one division per strip)
        unsigned qrow_off = 0;
        if constexpr (EPI == EPI_QKV) {
            const int tt = min(t, g.M - 1);
            const int qb = tt / g.tokens, qtok = tt - qb * g.tokens;
            qrow_off = (unsigned)((qb * g.heads * g.tokens + qtok) * g.head_dim);
        }
        for (int i0 = 0; i0 < nsub; i0 += 4) {            // four channel sub-tiles per batch: 16 outputs per lane under one certificate
            v4i acc[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = min(i0 + ii, nsub - 1);
                acc[ii] = *reinterpret_cast<const v4i*>(bias + 16 * i + 4 * g4);
#pragma unroll
                for (int ks = 0; ks < SK_MAXK / 32; ++ks)
                    if (ks < nks)
                        acc[ii] = __builtin_amdgcn_mfma_i32_16x16x32_i8(*reinterpret_cast<const long*>(smem + ((size_t)(i * nks + ks) * 64 + lane) * 8),
                                                                         bf[ks], acc[ii], 0, 0, 0);
            }
            int b[4][4];
            unsigned unc = 0;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = min(i0 + ii, nsub - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float2 lh = lohi[16 * i + 4 * g4 + r];
                    const float a = (float)acc[ii][r];
                    const int tl = __float_as_int(__builtin_fmaf(a, lh.x, 12582912.0f));
                    const int th = __float_as_int(__builtin_fmaf(a, lh.y, 12582912.0f));
                    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                    b[ii][r] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);       // low byte = int8 result
                }
            }
            if (__builtin_amdgcn_ballot_w64(unc != 0) != 0) {      // rare: exact float64 evaluation of the batch (quant_utils.py:229-230)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int c0 = 16 * min(i0 + ii, nsub - 1) + 4 * g4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double tq = (double)acc[ii][r] * dyadic_mult(g.m[c0 + r], g.e[c0 + r]) + IVIT_MAGIC;
                        b[ii][r] = clamp_i32((int)(unsigned)__double_as_longlong(tq), -128, 127);
                    }
                }
            }
            // a lane holds one dword (4 channels) per sub-tile; the 4 x 4 dword transpose over the token's four lanes (epilogue_direct_16)
            // leaves lane g4 with the 16 bytes of sub-tile i0 + g4: one 16-byte store per lane, 64 contiguous bytes per row (dword
            // stores from the MFMA layout -- 16-byte pieces of 16 rows per instruction -- made this kernel SLOWER than the tile kernels)
            unsigned D[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                D[ii] = __builtin_amdgcn_perm((unsigned)b[ii][1], (unsigned)b[ii][0], 0x0c0c0400u) |
                        __builtin_amdgcn_perm((unsigned)b[ii][3], (unsigned)b[ii][2], 0x04000c0cu);
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const v2u ab = __builtin_amdgcn_permlane32_swap(D[0], D[2], false, false);
            const v2u cd = __builtin_amdgcn_permlane32_swap(D[1], D[3], false, false);
            const v2u ac = __builtin_amdgcn_permlane16_swap(ab.x, cd.x, false, false);
            const v2u bd = __builtin_amdgcn_permlane16_swap(ab.y, cd.y, false, false);
            const int ci = i0 + g4;            // this lane's sub-tile after the transpose
            if (ci < nsub && t < g.M) {
                const int4 v = make_int4((int)ac.x, (int)ac.y, (int)bd.x, (int)bd.y);
                if constexpr (EPI == EPI_QKV) *reinterpret_cast<int4*>(out + coff[4 * ci] + qrow_off) = v;
                else *reinterpret_cast<int4*>(out + (int64_t)t * g.ldo + 16 * ci) = v;
            }
        }
    }
}

#include "gemm_wp.h"

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
    if (EPI == EPI_I32 || EPI == EPI_RQ16 || EPI == EPI_RQ16_RES16) {
        IVIT_REQUIRE(g.N % 4 == 0 && g.ldo % 4 == 0 && g.ldo >= g.N, "%s: N=%d ldo=%lld must be multiples of 4", name,
                     g.N, (long long)g.ldo);
        if (EPI == EPI_RQ16 || EPI == EPI_RQ16_RES16)
            IVIT_REQUIRE(g.m && g.e && ((uintptr_t)g.m % 16 == 0) && ((uintptr_t)g.e % 16 == 0) && ((uintptr_t)g.out % 8 == 0),
                         "%s: requantiser tables missing or misaligned", name);
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
    if (EPI == EPI_RESID16 || EPI == EPI_RQ16_RES16)
        IVIT_REQUIRE(g.res && g.ldr >= g.N && g.ldr % 8 == 0 && ((uintptr_t)g.res % 16 == 0) && g.ldo >= g.N && g.ldo % 8 == 0,
                     "%s: 16-bit residual / output rows must be 16-byte aligned (ld multiples of 8 elements)", name);
    if (EPI == EPI_QKV) {
        IVIT_REQUIRE(g.tokens > 0 && g.heads > 0 && g.head_dim > 0 && g.head_dim % 16 == 0,
                     "%s: bad head geometry tokens=%d heads=%d head_dim=%d", name, g.tokens, g.heads, g.head_dim);
        IVIT_REQUIRE((int64_t)g.M * g.N < 2147483648ll, "%s: q/k/v output of %lld bytes exceeds the 2 GiB the 32-bit head-major offsets cover",
                     name, (long long)g.M * g.N);
        IVIT_REQUIRE(g.N == 3 * g.heads * g.head_dim && g.M % g.tokens == 0,
                     "%s: N=%d != 3*heads*head_dim or M=%d %% tokens=%d != 0", name, g.N, g.M, g.tokens);
    }
    g.flags = g_debug_flags & (31 | 128 | 256 | 512);
    g.flags2 = g_debug_flags2;
#if IVIT_LAB
    g.stamp = reinterpret_cast<unsigned long long*>(g_stamp_buf);
#endif
    const bool blocks = g.a_blocks || g.w_blocks;
    if constexpr (EPI == EPI_RQ || EPI == EPI_QKV) {
        // skinny-K form: a streaming pass with the whole weight matrix in LDS (Swin stage 0); lab flags2 bit 20: off (A/B, parity of both)
        if (!blocks && !g.w_frags && !g.out_blocks && !g.lut && !g.gelu_ws && g.M >= 8192 && g.K >= 32 && g.K <= SK_MAXK && g.K % 32 == 0 && g.N >= 16 &&
            g.N <= SK_MAXN && g.N % 16 == 0 && g.lda % 16 == 0 && g.ldw % 8 == 0 && ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.W % 8 == 0) &&
            ((uintptr_t)g.out % 16 == 0) && (EPI == EPI_QKV || g.ldo % 16 == 0) && (EPI != EPI_QKV || g.head_dim % 16 == 0) && (int64_t)g.M * (EPI == EPI_QKV ? g.N : g.ldo) < 4294967296ll &&
            !g_force_small && !(IVIT_LAB && ((g_debug_flags2 & (1 << 20)) || (g_debug_flags & (31 | 128 | 256 | 512 | 1024))))) {
            const int nstrips = (g.M + 15) >> 4;
            const int grid = (nstrips + SK_WPB - 1) / SK_WPB < 512 ? (nstrips + SK_WPB - 1) / SK_WPB : 512;      // two workgroups of 8 waves per CU
            const bool any_k = IVIT_LAB && (g_debug_flags2 & (1 << 21));      // lab A/B: the run-time-K instantiation for every K
            if (g.K == 128 && !any_k) hipLaunchKernelGGL((gemm_i8_skinny_kernel<EPI, 4>), dim3(grid), dim3(SK_NT), 0, ivit_stream(stream), g);
            else if (g.K == 64 && !any_k) hipLaunchKernelGGL((gemm_i8_skinny_kernel<EPI, 2>), dim3(grid), dim3(SK_NT), 0, ivit_stream(stream), g);
            else hipLaunchKernelGGL((gemm_i8_skinny_kernel<EPI, 0>), dim3(grid), dim3(SK_NT), 0, ivit_stream(stream), g);
            IVIT_CHECK_LAUNCH(name);
        }
    }
    if (blocks && !g.w_frags) {
        IVIT_REQUIRE(EPI != EPI_I32 && EPI != EPI_RQ16 && g.M >= 2048 && g.N >= BCH && !g_force_small,
                     "%s: block-layout operands need the persistent kernel (M >= 2048, N >= 128, requantising epilogue)", name);
        IVIT_REQUIRE(!g.a_blocks || g.lda == g.K, "%s: a block-layout A operand is dense (lda == K)", name);
        IVIT_REQUIRE(!g.w_blocks || g.ldw == g.K, "%s: a block-layout W operand is dense (ldw == K)", name);
    }
    if (g.w_frags) IVIT_REQUIRE(!g.a_blocks || g.lda == g.K, "%s: a block-layout A operand is dense (lda == K)", name);
    IVIT_REQUIRE(!g.lut || (g.w_frags && EPI == EPI_RQ), "%s: the output map needs the fragment-packed weight form (IVIT_W_FRAGS)", name);
    if (g.w_frags) {
        IVIT_REQUIRE(EPI != EPI_I32 && EPI != EPI_RQ16 && g.M >= 2048 && g.N >= 128 && g.N % 64 == 0 && (g.K / BK) % 3 == 0 && !g.w_blocks && !g_force_small,
                     "%s: the fragment-packed weight needs M >= 2048, N >= 128, N %% 64 == 0, K %% 192 == 0 and a requantising epilogue", name);
        if constexpr (EPI != EPI_I32 && EPI != EPI_RQ16) {
            g.tiles_m = (g.M + WR_TOK - 1) / WR_TOK;
            g.tiles_n = (g.N + WR_CH - 1) / WR_CH;
            // Narrow tiles (128 x 128, wr_tile): built in round 4 for the widths 256-channel tiles fit badly and for launches with few
            // tiles, measured at every GEMM shape of the BASELINE configs (scripts/gemm_narrow_ab.py, profiles/r04f_*) and SLOWER
            // nearly everywhere (x 1.08 - 1.44; DeiT-S attn.proj -5 %, Swin stage-1 qkv equal): a wave's 16 MFMAs per K step carry the
            // same barrier, waits and six loads as 32.  What did pay is running the 256-channel tiles at N = 384 / 1152 / 576 at all
            // (padding of 25 / 10 / 11 %) instead of the LDS-DMA kernel: DeiT-S b64 proj 17 -> 11 us.  Lab flags2 bit 13 selects them.
            g.narrow = 0;
            if constexpr (EPI == EPI_RQ || EPI == EPI_RESID || EPI == EPI_QKV) {
                if (IVIT_LAB && g.w_frags == 2 && !g.lut && !g.gelu_ws && (g_debug_flags2 & 8192) && !(g_debug_flags2 & (256 | 512))) {
                    g.narrow = 1;
                    g.tiles_n = (g.N + WR_CH / 2 - 1) / (WR_CH / 2);
                }
            }
            const int ntiles = g.tiles_m * g.tiles_n;
            // Round 4 experiment, LAB ONLY (flags2 bit 15): the wave-pipelined form (gemm_wp.h: one workgroup of eight waves per CU, a
            // tile's requantisation inside the next tile's main loop).  Exact, and SLOWER than the kernel below (fc1 150 vs 116 us,
            // qkv 121 vs 97 us): with 64 accumulator registers per wave the token fragments are re-read from LDS twice as often per
            // MFMA, and the bare eight-wave loop alone (no epilogue at all) takes 110 us -- profiles/r04p_*, DESIGN.md section 8
            if constexpr (IVIT_LAB != 0 && (EPI == EPI_RQ || EPI == EPI_QKV)) {
                if (g.w_frags == 2 && !g.lut && !g.gelu_ws && !g.narrow && (g.K / BK) >= 12 && (g_debug_flags2 & 32768) && !(g_debug_flags2 & (256 | 512)) &&
                    !(g_debug_flags & 127)) {
                    IVIT_REQUIRE(((int64_t)g.M + 16) * (EPI == EPI_QKV ? g.N : g.ldo) < 4294967296ll,
                                 "%s: IVIT_W_FRAGS16 addresses its output with 32-bit offsets: operand of 4 GiB or more", name);
                    g.split_from = ntiles;
                    if constexpr (EPI == EPI_QKV) {
                        IVIT_REQUIRE((int64_t)g.M * g.tokens < 4294967296ll, "%s: M * tokens must stay below 2^32", name);
                        g.tokens_magic = (unsigned)(4294967296ull / (unsigned)g.tokens) + 1u;
                    }
                    static bool lds_set = false;      // 94 KB of dynamic LDS: beyond the default 64 KB limit of a launch
                    if (!lds_set) {
                        IVIT_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_i8_wp_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         WP_SMEM) == hipSuccess, "%s: the device refuses %d bytes of LDS per workgroup", name, WP_SMEM);
                        lds_set = true;
                    }
#if IVIT_LAB
                    if constexpr (EPI == EPI_RQ) {      // timing ablations (results wrong): flags2 bits 16-19 = ABL of gemm_wp.h
                        const int abl = (g_debug_flags2 >> 16) & 15;
                        if (abl) {
#define IVIT_WP_ABL(v) if (abl == v) { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_i8_wp_kernel<EPI_RQ, v>), hipFuncAttributeMaxDynamicSharedMemorySize, WP_SMEM); \
                                       hipLaunchKernelGGL((gemm_i8_wp_kernel<EPI_RQ, v>), dim3(ntiles < 256 ? ntiles : 256), dim3(WP_NT), WP_SMEM, ivit_stream(stream), g); IVIT_CHECK_LAUNCH(name); }
                            IVIT_WP_ABL(1) IVIT_WP_ABL(3) IVIT_WP_ABL(2)
#undef IVIT_WP_ABL
                        }
                    }
#endif
                    hipLaunchKernelGGL((gemm_i8_wp_kernel<EPI>), dim3(ntiles < 256 ? ntiles : 256), dim3(WP_NT), WP_SMEM, ivit_stream(stream), g);
                    IVIT_CHECK_LAUNCH(name);
                }
            }
            // A sparse last round (R tiles on 512 slots) runs as 2R half tiles of 64 tokens (wr_work) when every half tile still
            // finds a CU of its own (2R <= 256): fc1 at the headline shape, R = 120, 144 -> 136 us.  Beyond that two half tiles
            // share a CU while a lone full tile has one to itself and runs nearly twice as fast: measured slower (N = 768,
            // R = 158: proj 53.8 -> 60.5 us, fc2 124 -> 135 us with the residual epilogue).  Lab bit 27: off, bit 11: up to 2R <= 512.
            const int rounds = ntiles / 512, R = ntiles - rounds * 512;
            const bool split = !g.narrow && rounds > 0 && R > 0 && 2 * R <= ((g_debug_flags & 2048) ? 512 : 256) && !(g_debug_flags & 134217728);
            g.split_from = split ? rounds * 512 : ntiles;
            // Round 4: a launch with at most 256 tiles (DeiT-S attn.proj / fc2 at batch 64: 198; Swin stage 3: 147) runs ALL of them as half
            // tiles of 64 tokens, one per workgroup: such a launch lasts as long as ONE tile (main loop + epilogue, ~15 K cycles), and a
            // half tile's epilogue is half as long, its K steps 16 MFMAs per wave instead of 32 (lab bit 27: off)
            const bool all_halves = g.w_frags == 2 && !g.narrow && !g.gelu_ws && rounds == 0 && 2 * ntiles <= 512 && !(g_debug_flags & 134217728);
            if (all_halves) g.split_from = 0;
            const dim3 grid(all_halves ? 2 * ntiles : (ntiles < 512 ? ntiles : 512));
#if IVIT_LAB
            if constexpr (EPI == EPI_RQ) {   // ablations (scripts/gemm_ab.py --frags): what each stream of the kernel costs
                switch (g_debug_flags & 127) {
#define IVIT_WR_ABL(v) case v: hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI_RQ, v>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g); IVIT_CHECK_LAUNCH(name)
                    case 16: g.res = (const int8_t*)g_stamp_buf; hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI_RQ, 16>), (g_debug_flags & 4096) ? dim3(256) : grid, dim3(BIG_NT), (g_debug_flags & 4096) ? 40960 : 0, ivit_stream(stream), g); IVIT_CHECK_LAUNCH(name);
                    IVIT_WR_ABL(64); IVIT_WR_ABL(32); IVIT_WR_ABL(1); IVIT_WR_ABL(2); IVIT_WR_ABL(4); IVIT_WR_ABL(6); IVIT_WR_ABL(8); IVIT_WR_ABL(14); IVIT_WR_ABL(15); IVIT_WR_ABL(7);
#undef IVIT_WR_ABL
                    default: break;
                }
            }
#endif
            if (g.w_frags == 2) {     // IVIT_W_FRAGS16: the v_mfma_i32_16x16x64_i8 form
                if constexpr (EPI != EPI_RQ16_RES16) {
                    // its epilogue addresses residual and output with 32-bit byte offsets
                    IVIT_REQUIRE(((int64_t)g.M + 16) * (EPI == EPI_QKV ? g.N : g.ldo) < 4294967296ll &&
                                     (EPI != EPI_RESID || ((int64_t)g.M + 16) * g.ldr < 4294967296ll),
                                 "%s: IVIT_W_FRAGS16 addresses its output (and residual) with 32-bit offsets: operand of 4 GiB or more", name);
#if IVIT_LAB
                    if constexpr (EPI == EPI_RQ || EPI == EPI_RESID) {
                        if ((g_debug_flags2 & 256) && g.stamp) {   // stamped timeline (scripts/wreg_timeline.py --s16)
                            if (EPI == EPI_RESID && g.res_f32) hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI, 16 | 2048 | 4096, true>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
                            else hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI, 16 | 2048, true>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
                            IVIT_CHECK_LAUNCH(name);
                        }
                    }
#endif
#if IVIT_LAB
                    if constexpr (EPI == EPI_RQ || EPI == EPI_RESID || EPI == EPI_QKV) {
                        if (g_debug_flags2 & 512) {     // A/B: the LDS-staged epilogue instead of the direct one (scripts/gemm_ab.py 0:512)
                            if (EPI == EPI_RESID && g.res_f32) hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI, 8192 | 4096, true>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
                            else hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI, 8192, true>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
                            IVIT_CHECK_LAUNCH(name);
                        }
                    }
#endif
                    if constexpr (EPI == EPI_RQ && IVIT_LAB != 0) {
                        if (g.gelu_ws) {     // ABL bit 15: ShiftGELU + mlp.qact1 applied per completed token panel (gelu_panel_phase)
                            hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI, 32768, true>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
                            IVIT_CHECK_LAUNCH(name);
                        }
                    }
                    if constexpr (EPI == EPI_RESID) {
                        if (g.res_f32) {     // ABL bit 12: the residual QuantAct on float32 fmas (residual_f32_form)
                            hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI, 4096, true>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
                            IVIT_CHECK_LAUNCH(name);
                        }
                    }
                    hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI, 0, true>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
                    IVIT_CHECK_LAUNCH(name);
                } else {
                    IVIT_REQUIRE(false, "%s: IVIT_W_FRAGS16 has no 16-bit epilogue; pack the weights with ivit_pack_weight_frags_i8", name);
                }
            }
            hipLaunchKernelGGL((gemm_i8_wreg_kernel<EPI>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
            IVIT_CHECK_LAUNCH(name);
        }
    }
    if constexpr (EPI != EPI_I32 && EPI != EPI_RQ16 && EPI != EPI_RQ16_RES16) {
#if IVIT_LAB
        if (EPI <= EPI_QKV && !blocks && (g_debug_flags & (31 | 128 | 256 | 512 | 1024))) {   // a lab form was asked for (tests, scripts)
            int rc = IVIT_OK;
            if (ivit_gemm_lab_launch(EPI, &g, name, stream, &rc)) return rc;
        }
#endif
        if (g.M >= 2048 && g.N >= BCH && !g_force_small) {
            g.stagger = (g_debug_flags & 64) ? 0 : 512;  // 2 workgroups x 256 CUs
            g.tiles_m = (g.M + BTOK - 1) / BTOK;
            g.tiles_n = (g.N + BCH - 1) / BCH;
            const int ntiles = g.tiles_m * g.tiles_n;
            // 2 resident workgroups x 256 CUs; debug bit 12: one workgroup per CU (extra dynamic LDS blocks the second)
            const bool one_per_cu = (g_debug_flags & 4096) != 0;
            // debug bit 26: a 256-workgroup grid WITHOUT the LDS blocker (probe: two streams' GEMMs sharing the CUs)
            const int SLOTS = (one_per_cu || (g_debug_flags & 67108864)) ? 256 : 512;
            // tail split (see PersWork): R tiles of a sparse last round become 2R half tiles (256 x 128 -> two 128 x 128).
            // A CU left with one workgroup runs it nearly twice as fast, so splitting a last round that already occupies more
            // than half of the CUs does not pay (proj 47.7 -> 50.0 us, fc2 162 -> 173 us with 1182 tiles on 512 slots).
            const int rounds = ntiles / SLOTS, R = ntiles - rounds * SLOTS;
            // Default: split only when the 2R half tiles still find a CU each (2R <= 256): then the tail round uses twice the
            // CUs for half the time (fc1, R = 120: 146 -> 140 us); beyond that two half tiles share a CU and it only adds their
            // overhead.  Debug bit 11 forces the split up to 2R <= SLOTS, bit 27 disables it.
            const bool split = R > 0 && !(g_debug_flags & 134217728) &&
                               ((g_debug_flags & 2048) ? 2 * R <= SLOTS : 2 * R <= 256) && !one_per_cu;
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
    }
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(gemm_i8_kernel<EPI>, dim3(g.tiles_m * g.tiles_n), dim3(NT), 0, ivit_stream(stream), g);
    IVIT_CHECK_LAUNCH(name);
}

// W[N][K] row-major -> MFMA-fragment order (include/ivit_hip.h IVIT_W_FRAGS), one 16-byte chunk per thread; rows >= N are zero
__global__ __launch_bounds__(256) void pack_weight_frags_kernel(const int8_t* W, int64_t ldw, int N, int K, int8_t* dst)
{
    const int c16 = K >> 4, n64 = (N + 63) & ~63;
    const int64_t total = (int64_t)n64 * c16;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int n = (int)(q / c16), k = (int)(q - (int64_t)n * c16) << 4;
        int4 v = make_int4(0, 0, 0, 0);
        if (n < N) v = *reinterpret_cast<const int4*>(W + (int64_t)n * ldw + k);
        const int64_t off = ((int64_t)(n >> 6) * (K >> 6) + (k >> 6)) * 4096 + ((((n >> 5) & 1) * 2 + ((k >> 5) & 1)) * 2 + ((k >> 4) & 1)) * 512 + (n & 31) * 16;
        *reinterpret_cast<int4*>(dst + off) = v;
    }
}

// the same for v_mfma_i32_16x16x64_i8 (IVIT_W_FRAGS16): per 64 channels and K step four 1 KB pieces, piece i = channels 16 i ..
// 16 i + 15, lane l = 16 c + r holds K bytes 16 c .. 16 c + 15 of channel 16 i + r
__global__ __launch_bounds__(256) void pack_weight_frags16_kernel(const int8_t* W, int64_t ldw, int N, int K, int8_t* dst)
{
    const int c16 = K >> 4, n64 = (N + 63) & ~63;
    const int64_t total = (int64_t)n64 * c16;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int n = (int)(q / c16), k = (int)(q - (int64_t)n * c16) << 4;
        int4 v = make_int4(0, 0, 0, 0);
        if (n < N) v = *reinterpret_cast<const int4*>(W + (int64_t)n * ldw + k);
        const int64_t off = ((int64_t)(n >> 6) * (K >> 6) + (k >> 6)) * 4096 + ((n >> 4) & 3) * 1024 + ((k >> 4) & 3) * 256 + (n & 15) * 16;
        *reinterpret_cast<int4*>(dst + off) = v;
    }
}

}  // namespace

IVIT_EXPORT int ivit_pack_weight_frags16_i8(const int8_t* W, int64_t ldw, int N, int K, int8_t* dst, ivit_stream_t stream)
{
    IVIT_REQUIRE(W && dst && N > 0 && K > 0 && K % 64 == 0 && ldw >= K && ldw % 16 == 0 && ((uintptr_t)W % 16 == 0) && ((uintptr_t)dst % 16 == 0),
                 "ivit_pack_weight_frags16_i8: bad operand (K must be a multiple of 64, rows 16-byte aligned)");
    const int64_t total = (int64_t)((N + 63) & ~63) * (K >> 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_frags16_kernel, dim3(grid), dim3(256), 0, ivit_stream(stream), W, ldw, N, K, dst);
    IVIT_CHECK_LAUNCH("ivit_pack_weight_frags16_i8");
}

IVIT_EXPORT int ivit_pack_weight_frags_i8(const int8_t* W, int64_t ldw, int N, int K, int8_t* dst, ivit_stream_t stream)
{
    IVIT_REQUIRE(W && dst && N > 0 && K > 0 && K % 64 == 0 && ldw >= K && ldw % 16 == 0 && ((uintptr_t)W % 16 == 0) && ((uintptr_t)dst % 16 == 0),
                 "ivit_pack_weight_frags_i8: bad operand (K must be a multiple of 64, rows 16-byte aligned)");
    const int64_t total = (int64_t)((N + 63) & ~63) * (K >> 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_frags_kernel, dim3(grid), dim3(256), 0, ivit_stream(stream), W, ldw, N, K, dst);
    IVIT_CHECK_LAUNCH("ivit_pack_weight_frags_i8");
}

IVIT_EXPORT int ivit_gemm_i8_requant_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                     const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, int M, int N,
                                     int K, int layouts, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K;
    g.a_blocks = layouts & 1; g.w_blocks = (layouts >> 1) & 1; g.out_blocks = (layouts >> 2) & 1; g.w_frags = (layouts & 16) ? 2 : ((layouts >> 3) & 1);
    IVIT_REQUIRE((layouts & ~31) == 0 && (layouts & 24) != 24, "ivit_gemm_i8_requant_ex: unknown layout bits");
    IVIT_REQUIRE(!g.out_blocks || (N % 64 == 0 && ldo == N && ((int64_t)M + 15) * N < 2147483648ll),
                 "ivit_gemm_i8_requant_ex: block-layout output needs N %% 64 == 0, ldo == N and a buffer below 2 GiB");
    return launch_gemm<EPI_RQ>(g, "ivit_gemm_i8_requant_ex", stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_lut_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                            const uint32_t* m, const int32_t* e, const int8_t* lut, int8_t* out, int64_t ldo,
                                            int M, int N, int K, int layouts, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K; g.lut = lut;
    g.a_blocks = layouts & 1; g.w_blocks = (layouts >> 1) & 1; g.out_blocks = (layouts >> 2) & 1; g.w_frags = (layouts & 16) ? 2 : ((layouts >> 3) & 1);
    IVIT_REQUIRE(lut && (layouts & ~31) == 0 && (layouts & 24) != 24 && g.w_frags, "ivit_gemm_i8_requant_lut_ex: needs a map and IVIT_W_FRAGS");
    IVIT_REQUIRE(!g.out_blocks || (N % 64 == 0 && ldo == N && ((int64_t)M + 15) * N < 2147483648ll),
                 "ivit_gemm_i8_requant_lut_ex: block-layout output needs N %% 64 == 0, ldo == N and a buffer below 2 GiB");
    return launch_gemm<EPI_RQ>(g, "ivit_gemm_i8_requant_lut_ex", stream);
}

#if IVIT_LAB     // the fc1 + ShiftGELU experiment (include/ivit_hip_debug.h): lab library only
IVIT_EXPORT int ivit_gemm_gelu_workspace_bytes(int M, int64_t* bytes)
{
    IVIT_REQUIRE(M > 0 && bytes, "ivit_gemm_gelu_workspace_bytes: M > 0 and a result pointer");
    const int64_t panels = ((int64_t)M + WR_TOK - 1) / WR_TOK;
    *bytes = panels * (int64_t)sizeof(int);
    return IVIT_OK;
}

IVIT_EXPORT int ivit_gemm_i8_requant_gelu_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                             const uint32_t* m, const int32_t* e, const int8_t* gelu_lut, void* workspace,
                                             int8_t* out, int64_t ldo, int M, int N, int K, int layouts, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K; g.gelu_lut = gelu_lut; g.gelu_ws = static_cast<int*>(workspace);
    g.a_blocks = layouts & 1; g.out_blocks = (layouts >> 2) & 1; g.w_frags = (layouts & 16) ? 2 : 0;
    IVIT_REQUIRE(gelu_lut && workspace && ((uintptr_t)workspace % 4 == 0) && (layouts & ~(1 | 4 | 16)) == 0 && g.w_frags == 2,
                 "ivit_gemm_i8_requant_gelu_ex: needs the table, the workspace and IVIT_W_FRAGS16 (| IVIT_A_BLOCKS | IVIT_OUT_BLOCKS)");
    IVIT_REQUIRE(N <= 4096, "ivit_gemm_i8_requant_gelu_ex: N = %d, at most 4096 channels per token", N);
    IVIT_REQUIRE(!g.out_blocks || (N % 64 == 0 && ldo == N && ((int64_t)M + 15) * N < 2147483648ll),
                 "ivit_gemm_i8_requant_gelu_ex: block-layout output needs N %% 64 == 0, ldo == N and a buffer below 2 GiB");
    return launch_gemm<EPI_RQ>(g, "ivit_gemm_i8_requant_gelu_ex", stream);
}
#endif

IVIT_EXPORT int ivit_gemm_i8_requant(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                     const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, int M, int N,
                                     int K, ivit_stream_t stream)
{
    return ivit_gemm_i8_requant_ex(A, lda, W, ldw, bias, m, e, out, ldo, M, N, K, 0, stream);
}

// EPI_RESID: can both two-operand products run as one float32 fma each?  RNE(k * float(M)) (computed as the kernel does, one
// fused multiply-add against 1.5 * 2^23) is compared with RNE(k * M) in float64 for every int8 k: 512 evaluations per launch.
static void residual_f32_form(GemmArgs& g)
{
    g.Mf_main = (float)g.M_main;
    g.Mf_res = (float)g.M_res;
    g.res_f32 = 0;
    if (!(g.M_main > 0.0 && g.M_res > 0.0 && g.M_main < 16384.0 && g.M_res < 16384.0)) return;
    const double Md[2] = {g.M_main, g.M_res};
    const float Mf[2] = {g.Mf_main, g.Mf_res};
    for (int t = 0; t < 2; ++t)
        for (int k = -128; k <= 127; ++k) {
            const float tf = __builtin_fmaf((float)k, Mf[t], 12582912.0f);
            int bits;
            __builtin_memcpy(&bits, &tf, 4);
            const double td = __builtin_fma((double)k, Md[t], IVIT_MAGIC);
            long long lb;
            __builtin_memcpy(&lb, &td, 8);
            if (bits - 0x4B400000 != (int)(unsigned)lb) return;
        }
    g.res_f32 = 1;
}

IVIT_EXPORT int ivit_gemm_i8_requant_residual_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                              const int32_t* bias, const uint32_t* m, const int32_t* e,
                                              const int8_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                              uint32_t m_res, int32_t e_res, int8_t* out, int64_t ldo, int M, int N,
                                              int K, int layouts, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = out; g.ldo = ldo; g.res = res; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
    g.M_main = ivit_dyadic_to_double(m_main, e_main);
    g.M_res = ivit_dyadic_to_double(m_res, e_res);
    residual_f32_form(g);
    IVIT_REQUIRE(g.M_main < 1048576.0 && g.M_res < 1048576.0,
                 "ivit_gemm_i8_requant_residual_ex: residual multiplier >= 2^20 is outside the int8 fast path");
    g.a_blocks = layouts & 1; g.w_blocks = (layouts >> 1) & 1; g.w_frags = (layouts & 16) ? 2 : ((layouts >> 3) & 1);
    IVIT_REQUIRE((layouts & ~27) == 0 && (layouts & 24) != 24, "ivit_gemm_i8_requant_residual_ex: unknown layout bits");
    return launch_gemm<EPI_RESID>(g, "ivit_gemm_i8_requant_residual_ex", stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_residual(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                              const int32_t* bias, const uint32_t* m, const int32_t* e,
                                              const int8_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                              uint32_t m_res, int32_t e_res, int8_t* out, int64_t ldo, int M, int N,
                                              int K, ivit_stream_t stream)
{
    return ivit_gemm_i8_requant_residual_ex(A, lda, W, ldw, bias, m, e, res, ldr, m_main, e_main, m_res, e_res, out, ldo, M, N, K, 0, stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_residual_i16_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                                  const int32_t* bias, const uint32_t* m, const int32_t* e,
                                                  const int16_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                                  uint32_t m_res, int32_t e_res, int16_t* out, int64_t ldo, int M, int N,
                                                  int K, int layouts, ivit_stream_t stream)
{
    GemmArgs g{};
    g.a_blocks = layouts & 1; g.w_blocks = (layouts >> 1) & 1; g.w_frags = (layouts & 16) ? 2 : ((layouts >> 3) & 1);
    IVIT_REQUIRE((layouts & ~27) == 0 && (layouts & 24) != 24, "ivit_gemm_i8_requant_residual_i16_ex: unknown layout bits");
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.res = reinterpret_cast<const int8_t*>(res); g.ldr = ldr;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K;
    g.M_main = ivit_dyadic_to_double(m_main, e_main);
    g.M_res = ivit_dyadic_to_double(m_res, e_res);
    IVIT_REQUIRE(g.M_main < 32768.0 && g.M_res < 32768.0, "ivit_gemm_i8_requant_residual_i16: residual multiplier too large");
    return launch_gemm<EPI_RESID16>(g, "ivit_gemm_i8_requant_residual_i16", stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_residual_i16(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                                  const int32_t* bias, const uint32_t* m, const int32_t* e,
                                                  const int16_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                                  uint32_t m_res, int32_t e_res, int16_t* out, int64_t ldo, int M, int N,
                                                  int K, ivit_stream_t stream)
{
    return ivit_gemm_i8_requant_residual_i16_ex(A, lda, W, ldw, bias, m, e, res, ldr, m_main, e_main, m_res, e_res, out, ldo, M, N, K, 0, stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_qkv_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                         const int32_t* bias, const uint32_t* m, const int32_t* e, int8_t* qkv,
                                         int tokens, int heads, int head_dim, int M, int N, int K,
                                         int layouts, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = qkv; g.ldo = 0; g.M = M; g.N = N; g.K = K;
    g.tokens = tokens; g.heads = heads; g.head_dim = head_dim;
    g.a_blocks = layouts & 1; g.w_blocks = (layouts >> 1) & 1; g.w_frags = (layouts & 16) ? 2 : ((layouts >> 3) & 1);
    IVIT_REQUIRE((layouts & ~27) == 0 && (layouts & 24) != 24, "ivit_gemm_i8_requant_qkv_ex: unknown layout bits");
    return launch_gemm<EPI_QKV>(g, "ivit_gemm_i8_requant_qkv_ex", stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_qkv(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                         const int32_t* bias, const uint32_t* m, const int32_t* e, int8_t* qkv,
                                         int tokens, int heads, int head_dim, int M, int N, int K,
                                         ivit_stream_t stream)
{
    return ivit_gemm_i8_requant_qkv_ex(A, lda, W, ldw, bias, m, e, qkv, tokens, heads, head_dim, M, N, K, 0, stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_i16(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                         const uint32_t* m, const int32_t* e, int16_t* out, int64_t ldo, int M, int N, int K,
                                         ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K;
    return launch_gemm<EPI_RQ16>(g, "ivit_gemm_i8_requant_i16", stream);
}

IVIT_EXPORT int ivit_gemm_i8_requant_i16_residual_i16_ex(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw,
                                                         const int32_t* bias, const uint32_t* m, const int32_t* e,
                                                         const int16_t* res, int64_t ldr, uint32_t m_main, int32_t e_main,
                                                         uint32_t m_res, int32_t e_res, int16_t* out, int64_t ldo, int M, int N,
                                                         int K, int layouts, ivit_stream_t stream)
{
    GemmArgs g{};
    g.a_blocks = layouts & 1; g.w_frags = (layouts & 16) ? 2 : ((layouts >> 3) & 1);
    IVIT_REQUIRE((layouts & ~9) == 0 && (g.w_frags || !g.a_blocks), "ivit_gemm_i8_requant_i16_residual_i16_ex: layouts is 0 or IVIT_W_FRAGS (| IVIT_A_BLOCKS)");
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.m = m; g.e = e;
    g.res = reinterpret_cast<const int8_t*>(res); g.ldr = ldr;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K;
    g.M_main = ivit_dyadic_to_double(m_main, e_main);
    g.M_res = ivit_dyadic_to_double(m_res, e_res);
    IVIT_REQUIRE(g.M_main < 32768.0 && g.M_res < 32768.0, "ivit_gemm_i8_requant_i16_residual_i16_ex: residual multiplier too large");
    IVIT_REQUIRE(ldo % 8 == 0 && ldr % 8 == 0 && ((uintptr_t)res % 16 == 0) && ((uintptr_t)out % 16 == 0) && N % 8 == 0,
                 "ivit_gemm_i8_requant_i16_residual_i16_ex: 16-bit rows must be 16-byte aligned");
    return launch_gemm<EPI_RQ16_RES16>(g, "ivit_gemm_i8_requant_i16_residual_i16_ex", stream);
}

IVIT_EXPORT int ivit_gemm_i8_i32(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const int32_t* bias,
                                 int32_t* out, int64_t ldo, int M, int N, int K, ivit_stream_t stream)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K;
    return launch_gemm<EPI_I32>(g, "ivit_gemm_i8_i32", stream);
}
