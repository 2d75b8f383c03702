// gemm_lab.hip -- forms of the large INT8 GEMM kernel that are NOT on the product path: kept, bit-identical, for A/B
// measurement and ablation through include/ivit_hip_debug.h (scripts/gemm_ablate.py, gemm_ab.py, gemm_timeline.py,
// ring_timeline.py; tests/test_gpu_ops.py::test_gemm_both_kernels_agree).  See DESIGN.md section 5 for what each taught.
#include "gemm_common.h"

int g_kernel_choice = 0;
bool g_force_small = false;
void* g_stamp_buf = nullptr;
int g_debug_flags = 0;   // bits: see include/ivit_hip_debug.h
int g_debug_flags2 = 0;

namespace {

// ================================================================================================
// Large-problem kernel: block tile 256 tokens x 128 channels x 64 K-bytes, 4 waves (2 x 2, each
// 64 channels x 128 tokens = 2 x 4 MFMA tiles, 128 accumulator registers), THREE LDS stages filled by
// LDS-DMA (global_load_lds_dwordx4: no staging registers), one raw s_barrier per K step with a
// counted vmcnt so the next stage's DMA stays in flight across it.  72 KiB LDS and <= 256 registers
// give two workgroups per CU: one block's requant epilogue (VALU/float64 pipe) overlaps the other's
// MFMA main loop.  LDS images are lane-linear per DMA instruction (16 rows x 64 B); the bank swizzle
// is applied on the per-lane SOURCE address and again on the fragment read.
// ================================================================================================
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
// Persistent 256 x 256 kernel: ONE workgroup of 8 waves per CU (4 channel groups x 2 token groups, wave tile 64 ch x
// 128 tok as above, two waves per SIMD), 4-stage ring of 32 KiB.  Why: the CU's global->LDS DMA
// path moves ~29 B/clk, and a 256 x 128 tile needs 24 KiB per K step (>= 830 cycles against 512 cycles of MFMA per
// wave); the 256 x 256 tile needs 32 KiB for twice the MACs (~1100 cycles against 2 x 512 per SIMD), so DMA and MFMA
// are balanced, and its two waves per SIMD overlap each other's stalls in the epilogue.  Stamped per tile: main loop
// 15.1 K cycles + epilogue 8.5 K for 256 x 256, against 2 x (10 K + 7 K + 3 K) for two 256 x 128 tiles.  Persistence
// removes most of what a relaunch of a 512-thread / 128 KiB workgroup costs per tile: stage 0 of the next tile is
// prefetched into ring buffer 3 while the epilogue stages its int8 tile in buffers 0-2, and the next requant table is
// fetched inside the epilogue.  Tile-local stage kt lives in buffer (kt + 3) & 3 for every tile.
// Measured (scripts/gemm_ab.py): on par with the persistent 256 x 128 kernel on every DeiT-B shape (its main loop runs at
// the power-limited MFMA ceiling, 66 % issue at ~1.87 GHz; the epilogue is not overlapped) -- kept for A/B (debug bit 22).
// ================================================================================================
constexpr int XL_RING = XL_STAGES * XL_STAGE;     // 128 KiB (buffers 0-2 >= 256 * 260 epilogue tile)
constexpr int XL_PT_BYTES = XCH * 12;             // { float2 lohi[256]; int bias[256] }
constexpr int XLP_SMEM = XL_RING + 2 * XL_PT_BYTES;

struct XlTable {
    unsigned m;
    int e, bias;
    bool valid;
};

IVIT_DEV PersTile xl_tile(const GemmArgs& g, int t)
{
    const int nblk = g.tiles_m * g.tiles_n;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = t & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (t >> 3);
    const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
    return PersTile{tm * XTOK, tn * XCH};
}

IVIT_DEV XlTable xl_table_issue(const GemmArgs& g, int n0, int tid)
{
    XlTable r{0u, 0, 0, false};
    const int c = n0 + tid;
    if (tid < XCH && c < g.N) {
        r.m = g.m[c];
        r.e = g.e[c];
        r.bias = g.bias ? g.bias[c] : 0;
        r.valid = true;
    }
    return r;
}

IVIT_DEV void xl_table_write(const XlTable& r, char* tab, int tid)
{
    if (tid < XCH) {
        float2 lh = make_float2(0.f, 0.f);
        if (r.valid) {
            const double M = dyadic_mult(r.m, r.e);
            const float mf = (float)M;
            const int bits = __float_as_int(mf);
            lh.x = ((double)mf > M) ? __int_as_float(bits - 1) : mf;
            lh.y = ((double)mf < M) ? __int_as_float(bits + 1) : mf;
        }
        reinterpret_cast<float2*>(tab)[tid] = lh;
        reinterpret_cast<int*>(tab + XCH * 8)[tid] = r.bias;
    }
}

template <int EPI>
__global__ __launch_bounds__(XL_NT, 2) void gemm_i8_xlp_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) char smem[XLP_SMEM];
    const int ntiles = g.tiles_m * g.tiles_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wt = wave & 1;  // 4 x 2
    const int h = lane >> 5, l31 = lane & 31;
    const int lrow = lane >> 2, lslot = lane & 3;
    const int nk = g.K / BK;
    using T = std::true_type;
    using F = std::false_type;

    // LDS-DMA sources: piece q covers tile rows 16q..16q+15 (1 KiB); wave w owns pieces w and w + 8 of each operand tile
    const int8_t* asrc[2];
    const int8_t* wsrc[2];
    auto set_sources = [&](const PersTile& t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = 16 * (wave + 8 * i) + lrow;
            int c = lslot ^ ((row >> 2) & 3);
            asrc[i] = g.A + (int64_t)min(t.m0 + row, g.M - 1) * g.lda + 16 * c;
            wsrc[i] = g.W + (int64_t)min(t.n0 + row, g.N - 1) * g.ldw + 16 * c;
        }
    };
    auto issue_one = [&](int kt, int idx) {
        char* base = smem + ((kt + 3) & 3) * XL_STAGE;
        const int koff = kt * BK;
        if (idx < 2)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[idx] + koff), (lptr_t)(base + 1024 * (wave + 8 * idx)), 16, 0,
                                             0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[idx - 2] + koff),
                                             (lptr_t)(base + XL_A_BYTES + 1024 * (wave + 8 * (idx - 2))), 16, 0, 0);
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int idx = 0; idx < 4; ++idx) issue_one(kt, idx);
    };

    // fragment reads as inline asm with explicit counted waits (see gemm_i8_pers_kernel)
    const unsigned smem_base = (unsigned)(__UINTPTR_TYPE__)(lptr_t)smem;
    const int wrow0 = 64 * wc + l31, arow0 = 128 * wt + l31;
    const unsigned wbase[2] = {smem_base + (unsigned)(XL_A_BYTES + swz(wrow0, h)),
                               smem_base + (unsigned)(XL_A_BYTES + swz(wrow0, 2 + h))};
    const unsigned abase[2] = {smem_base + (unsigned)swz(arow0, h), smem_base + (unsigned)swz(arow0, 2 + h)};
    v4i wf0[2], af0[4], wf1[2], af1[4];
    auto load_frags = [&](int kt, int ks, v4i (&wf)[2], v4i (&af)[4]) {
        const unsigned off = (unsigned)(((kt + 3) & 3) * XL_STAGE);
        const unsigned wa = wbase[ks] + off, aa = abase[ks] + off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(wa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wf[1]) : "v"(wa));
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[3]) : "v"(aa));
    };
#define XLP_TIE(wf, af) "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])
    v16i acc[2][4];
    // One K step; AHEAD = stages after kt+1 whose DMA (4 pieces per wave each) may still be in flight at its end
    auto step = [&](int kt, auto issue_tag, auto ahead_tag, auto last_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        constexpr int AHEAD = decltype(ahead_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        load_frags(kt, 1, wf1, af1);
        asm volatile("s_waitcnt lgkmcnt(6)" : XLP_TIE(wf0, af0)::"memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf0[i], af0[j], acc[i][j], 0, 0, 0);
                if constexpr (ISSUE)
                    if (((4 * i + j) & 1) == 0) issue_one(kt + 3, (4 * i + j) >> 1);
            }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (AHEAD == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : XLP_TIE(wf1, af1)::"memory");
        else if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : XLP_TIE(wf1, af1)::"memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : XLP_TIE(wf1, af1)::"memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (!LAST) load_frags(kt + 1, 0, wf0, af0);
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

    // ---- first tile: table + stage 0
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    PersTile cur = xl_tile(g, tile);
    {
        XlTable tl = xl_table_issue(g, cur.n0, tid);
        xl_table_write(tl, smem + XL_RING, tid);
    }
    set_sources(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // table loads retired before the DMA counting starts
    issue(0);

    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        char* tab = smem + XL_RING + (it & 1) * XL_PT_BYTES;
        char* tab_next = smem + XL_RING + ((it + 1) & 1) * XL_PT_BYTES;
        // stage 0 of this tile is in flight (or landed) in buffer 3; buffers 0-2 are free again (staging tile read out)
        if (nk > 1) issue(1);
        if (nk > 2) issue(2);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // own pieces of stage 0 and everything older
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                        // everyone's stage 0; table visible
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 b4 = *reinterpret_cast<const int4*>(tab + XCH * 8 + 4 * (64 * wc + 32 * i + 8 * q + 4 * h));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j][4 * q + 0] = b4.x;
                    acc[i][j][4 * q + 1] = b4.y;
                    acc[i][j][4 * q + 2] = b4.z;
                    acc[i][j][4 * q + 3] = b4.w;
                }
            }
        load_frags(0, 0, wf0, af0);
        int kt = 0;
        for (; kt + 3 < nk; ++kt) step(kt, T{}, A2{}, F{});
        if (kt + 2 < nk) { step(kt, F{}, A1{}, F{}); ++kt; }
        if (kt + 1 < nk) { step(kt, F{}, A0{}, F{}); ++kt; }
        step(kt, F{}, A0{}, T{});
        __syncthreads();   // all waves are done with every ring buffer

        // ---- stage 0 of the next tile into buffer 3, then this tile's epilogue (staging tile in buffers 0-2)
        const int next = tile + gridDim.x;
        const bool more = next < ntiles;   // uniform
        PersTile nxt = cur;
        if (more) {
            nxt = xl_tile(g, next);
            set_sources(nxt);
            issue(0);
        }
        struct Hook {
            const GemmArgs& g;
            int n0, tid;
            char* dst;
            bool more;
            mutable XlTable ld;
            IVIT_DEV void issue() const { if (more) ld = xl_table_issue(g, n0, tid); }
            IVIT_DEV void consume() const { if (more) xl_table_write(ld, dst, tid); }
        };
        Hook hook{g, nxt.n0, tid, tab_next, more, XlTable{0u, 0, 0, false}};
        epilogue_i8<EPI, 2, 4, XTOK, XL_NT, 0, XCH, Hook>(acc, g, smem, tab, cur.m0, cur.n0, 64 * wc, 128 * wt, tid, h, l31,
                                                         hook);
        cur = nxt;
        __syncthreads();   // staging reads done before the next tile's stages 1-2 overwrite buffers 0-1
    }
#undef XLP_TIE
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

// ================================================================================================
// Ping-pong form: one workgroup per CU (4 waves, wave tile 64 ch x 128 tok), 4-stage ring that runs continuously across
// tiles, and the requantisation (phase 1 of the epilogue) of tile t-1 executed INSIDE the K loop of tile t, one batch of
// 16 outputs per thread in each of the first eight K steps, from a second register set the accumulators are copied
// to at the end of a tile.  Rationale: neither a co-resident workgroup nor a second wave hides the epilogue on this
// chip, but VALU instructions of the SAME wave issue for free in the shadow of its MFMAs (32 cycles each).
// Phase 2 (LDS -> global stores, residual QuantAct) still runs between two K loops.  Needs K >= 512.
// ================================================================================================
constexpr int PP_STAGES = 4;
constexpr int PP_RING = PP_STAGES * BIG_STAGE;             // 96 KiB
constexpr int PP_EPI_OFF = PP_RING;                        // staging tile 256 x 132 B
constexpr int PP_PT_OFF = PP_EPI_OFF + BTOK * (BCH + 4);
constexpr int PP_SMEM = PP_PT_OFF + 2 * PT_BYTES;          // 135 168 B: one workgroup per CU

// phase 1 of epilogue_i8 for ONE unit U = (I, Q, J) of a drained accumulator set: four channels of one token per lane
// (gemm_common.h, TJ = 4, CH = 128).  Branch-free so that it schedules into the MFMA shadow: the float32 certificate is
// only ACCUMULATED here (unc, amax); a tile with any uncertified output is redone exactly by pp_exact_tile.
// lh01 / lh23: (lo, hi) factor pairs of channels cl .. cl+3 read from the table by pp_load_group.
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U>
IVIT_DEV void pp_unit(const v16i (&dr)[2][4], const v4f& lh01, const v4f& lh23, unsigned stg_addr, unsigned& unc,
                      float& amax)
{
    constexpr int I = U >> 4, Q = (U >> 2) & 3, J = U & 3;
    constexpr int CSS = BCH + 4;
    const float lo[4] = {lh01.x, lh01.z, lh23.x, lh23.z};
    const float hi[4] = {lh01.y, lh01.w, lh23.y, lh23.w};
    int b[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const float a = (float)dr[I][J][4 * Q + jj];
        const int tl = __float_as_int(__builtin_fmaf(a, lo[jj], 12582912.0f));
        const int th = __float_as_int(__builtin_fmaf(a, hi[jj], 12582912.0f));
        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
        amax = fmaxf(amax, fabsf(a));
        b[jj] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);
    }
    const unsigned w01 = __builtin_amdgcn_perm((unsigned)b[1], (unsigned)b[0], 0x0c0c0400u);
    const unsigned w23 = __builtin_amdgcn_perm((unsigned)b[3], (unsigned)b[2], 0x04000c0cu);
    const unsigned w = w01 | w23;
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(stg_addr), "v"(w), "n"(J * 32 * CSS + 32 * I + 8 * Q) : "memory");
}

// (lo, hi) pairs of the four channels of group G = (I, Q): two 16-byte LDS reads, waited for by the caller
template <int G>
IVIT_DEV void pp_load_group(v4f& a, v4f& b, unsigned tab_addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(tab_addr), "n"(256 * (G >> 2) + 64 * (G & 3)));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(tab_addr), "n"(256 * (G >> 2) + 64 * (G & 3) + 16));
}

// exact float64 evaluation of a whole drained tile (taken when the certificate of any output of the wave failed)
IVIT_DEV void pp_exact_tile(const v16i (&dr)[2][4], const GemmArgs& g, char* stg, int n0, int wch,
                                                      int wtok, int h, int l31)
{
    constexpr int CSS = BCH + 4;
#pragma unroll 1
    for (int iq = 0; iq < 8; ++iq) {
        const int i = iq >> 2, q = iq & 3;
        const int cl = wch + 32 * i + 8 * q + 4 * h;
        const int c0 = min(n0 + cl, g.N - 4);
        const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
        const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
        const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z),
                              dyadic_mult(m4.w, e4.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int b[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                int av = 0;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        if (ii == i && qq == q) av = dr[ii][j][4 * qq + jj];
                double t = (double)av * Mc[jj] + IVIT_MAGIC;
                b[jj] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
            }
            *reinterpret_cast<int*>(stg + (wtok + 32 * j + l31) * CSS + cl) = pack4_i8(b[0], b[1], b[2], b[3]);
        }
    }
}

// phase 2 of epilogue_i8 (TOK = 256, 256 threads, CH = 128) from the staging tile
template <int EPI>
IVIT_DEV void pp_phase2(const GemmArgs& g, const char* stg, int m0, int n0, int tid)
{
    constexpr int CSS = BCH + 4, CPR = BCH / 16, NIT = BTOK * CPR / BIG_NT;
    int8_t* out = reinterpret_cast<int8_t*>(g.out);
    int v[NIT][4];
    int4 rv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = tid + BIG_NT * it;
        const int tl = q / CPR, cc = q % CPR;
        const int* src = reinterpret_cast<const int*>(stg + tl * CSS + 16 * cc);
        v[it][0] = src[0]; v[it][1] = src[1]; v[it][2] = src[2]; v[it][3] = src[3];
        if constexpr (EPI == EPI_RESID) {
            const int t = min(m0 + tl, g.M - 1), cn = min(n0 + 16 * cc, g.N - 16);
            rv[it] = *reinterpret_cast<const int4*>(g.res + (int64_t)t * g.ldr + cn);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = tid + BIG_NT * it;
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

template <class F, int... S>
IVIT_DEV void pp_for_slots(F&& f, std::integer_sequence<int, S...>)
{
    (f(std::integral_constant<int, S>{}), ...);
}

// UPS = units of phase 1 per K step: the 32 units of a tile take NS = ceil(32 / UPS) steps (needs K / 64 >= NS)
template <int EPI, int UPS>
__global__ __launch_bounds__(BIG_NT, 1) void gemm_i8_pp_kernel(GemmArgs g)
{
    constexpr int NS = (32 + UPS - 1) / UPS;
    constexpr int CSS = BCH + 4;
    __shared__ __attribute__((aligned(16))) char smem[PP_SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wt = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int lrow = lane >> 2, lslot = lane & 3;
    const int nk = g.K / BK;
    const int ntiles = g.tiles_m * g.tiles_n;

    // DMA sources of the current tile and of the next one (the ring runs ahead across the tile boundary)
    const int8_t* src_cur[6];
    const int8_t* src_nxt[6];
    auto set_sources = [&](const int8_t* (&dst)[6], const PersTile& t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = 16 * (wave + 4 * i) + lrow;
            int c = lslot ^ ((row >> 2) & 3);
            dst[i] = g.A + (int64_t)min(t.m0 + row, g.M - 1) * g.lda + 16 * c;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = 16 * (wave + 4 * i) + lrow;
            int c = lslot ^ ((row >> 2) & 3);
            dst[4 + i] = g.W + (int64_t)min(t.n0 + row, g.N - 1) * g.ldw + 16 * c;
        }
    };
    auto issue_piece = [&](int gs, const int8_t* src, int idx) {
        char* base = smem + (gs & 3) * BIG_STAGE;
        char* dst = idx < 4 ? base + 1024 * (wave + 4 * idx) : base + BIG_A_BYTES + 1024 * (wave + 4 * (idx - 4));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };

    const unsigned smem_base = (unsigned)(__UINTPTR_TYPE__)(lptr_t)smem;
    const int wrow0 = 64 * wc + l31, arow0 = 128 * wt + l31;
    const unsigned wbase[2] = {smem_base + (unsigned)(BIG_A_BYTES + swz(wrow0, h)),
                               smem_base + (unsigned)(BIG_A_BYTES + swz(wrow0, 2 + h))};
    const unsigned abase[2] = {smem_base + (unsigned)swz(arow0, h), smem_base + (unsigned)swz(arow0, 2 + h)};
    v4i wf0[2], af0[4], wf1[2], af1[4];
    auto load_frags = [&](int gs, int ks, v4i (&wf)[2], v4i (&af)[4]) {
        const unsigned off = (unsigned)((gs & 3) * BIG_STAGE);
        const unsigned wa = wbase[ks] + off, aa = abase[ks] + off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(wa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wf[1]) : "v"(wa));
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[3]) : "v"(aa));
    };
#define PP_TIE(wf, af) "+v"(wf[0]), "+v"(wf[1]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])
    v16i acc[2][4], dr[2][4];
    char* stg = smem + PP_EPI_OFF;
    // this thread's staging address for (token wtok + l31, channel wch + 4h); units add compile-time offsets
    const unsigned stg_addr = smem_base + (unsigned)(PP_EPI_OFF + (128 * wt + l31) * CSS + 64 * wc + 4 * h);

    // state of the tile being drained (its phase 1 runs inside the next K loop)
    int prev_m0 = 0, prev_n0 = 0;
    unsigned prev_tab_addr = 0;
    unsigned unc = 0;
    float amax = 0.0f;

    // One K step of the stream.  SLOT >= 0: also requantise units [SLOT * UPS, SLOT * UPS + UPS) of the drained tile.
    auto step = [&](int gs, int kt, bool next_ok, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int U0 = SLOT < 0 ? 32 : SLOT * UPS;
        constexpr int U1 = (U0 + UPS < 32) ? U0 + UPS : 32;          // units [U0, U1)
        constexpr int NU = U1 > U0 ? U1 - U0 : 0;
        constexpr int G0 = U0 >> 2, G1 = (U1 - 1) >> 2;              // their table groups (at most two)
        constexpr int UH = U0 + NU / 2;                              // units [U0, UH) in the first half of the step
        v4f ta0, ta1, tb0, tb1;
        if constexpr (NU > 0) {
            pp_load_group<G0>(ta0, ta1, prev_tab_addr);
            if constexpr (G1 != G0) pp_load_group<G1>(tb0, tb1, prev_tab_addr);
        }
        load_frags(gs, 1, wf1, af1);
        if constexpr (NU > 0 && G1 != G0)
            asm volatile("s_waitcnt lgkmcnt(6)" : PP_TIE(wf0, af0), "+v"(ta0), "+v"(ta1), "+v"(tb0), "+v"(tb1)::"memory");
        else if constexpr (NU > 0)
            asm volatile("s_waitcnt lgkmcnt(6)" : PP_TIE(wf0, af0), "+v"(ta0), "+v"(ta1)::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(6)" : PP_TIE(wf0, af0)::"memory");
        auto unit = [&](auto utag) {
            constexpr int U = decltype(utag)::value;
            if constexpr ((U >> 2) == G0) pp_unit<U>(dr, ta0, ta1, stg_addr, unc, amax);
            else pp_unit<U>(dr, tb0, tb1, stg_addr, unc, amax);
        };
        // stage gs + 3 of the stream: this tile's stage kt + 3, else the next tile's stage kt + 3 - nk; the last tile
        // re-fetches its own first stages into the free buffers (harmless) so that the counted waits stay uniform
        const bool own = kt + 3 < nk;
        const int koff = (own ? kt + 3 : kt + 3 - nk) * BK;
        const bool use_cur = own || !next_ok;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf0[i], af0[j], acc[i][j], 0, 0, 0);
                if (4 * i + j < 6) {
                    const int idx = 4 * i + j;
                    issue_piece(gs + 3, (use_cur ? src_cur[idx] : src_nxt[idx]) + koff, idx);
                }
            }
        if constexpr (U0 + 0 < UH) unit(std::integral_constant<int, (U0 + 0 < 32 ? U0 + 0 : 0)>{});
        if constexpr (U0 + 1 < UH) unit(std::integral_constant<int, (U0 + 1 < 32 ? U0 + 1 : 0)>{});
        asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" : PP_TIE(wf1, af1)::"memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_frags(gs + 1, 0, wf0, af0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf1[i], af1[j], acc[i][j], 0, 0, 0);
        if constexpr (UH + 0 < U1) unit(std::integral_constant<int, (UH + 0 < 32 ? UH + 0 : 0)>{});
        if constexpr (UH + 1 < U1) unit(std::integral_constant<int, (UH + 1 < 32 ? UH + 1 : 0)>{});
    };
    using NONE = std::integral_constant<int, -1>;

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    PersTile cur = pers_tile(g, tile);
    {
        PersTableLoad tl = pers_table_issue(g, cur.n0, tid);
        pers_table_write(tl, smem + PP_PT_OFF, tid);
    }
    set_sources(src_cur, cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // head of the stream: stages 0, 1, 2 of the first tile
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int idx = 0; idx < 6; ++idx) issue_piece(st, src_cur[idx] + st * BK, idx);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(0, 0, wf0, af0);

    int gs = 0;
    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        char* tab = smem + PP_PT_OFF + (it & 1) * PT_BYTES;
        char* tab_next = smem + PP_PT_OFF + ((it + 1) & 1) * PT_BYTES;
        const int next = tile + gridDim.x;
        const bool more = next < ntiles;   // uniform
        PersTile nxt = cur;
        if (more) {
            nxt = pers_tile(g, next);
            set_sources(src_nxt, nxt);
        }
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
        long long* stamp = nullptr;
        if constexpr (EPI == EPI_RQ) {
            if (g.res && tid == 0 && it >= 1 && it <= 2)
                stamp = (long long*)g.res + (blockIdx.x * 2 + (it - 1)) * 8;
        }
        if (stamp) stamp[0] = __builtin_amdgcn_s_memtime();
        if (it > 0 && !(g.stagger_units & 2)) {
            // K loop with phase 1 of the previous tile riding in its first NS steps
            int kt = 0;
            pp_for_slots([&](auto s) { step(gs++, kt++, more, s); }, std::make_integer_sequence<int, NS>{});
            for (; kt < nk; ++kt) step(gs++, kt, more, NONE{});
            if (!(g.stagger_units & 1) && __builtin_amdgcn_ballot_w64((unc != 0) | (amax >= 4194304.0f)) != 0)
                pp_exact_tile(dr, g, stg, prev_n0, 64 * wc, 128 * wt, h, l31);
            unc = 0;
            amax = 0.0f;
        } else {
            for (int kt = 0; kt < nk; ++kt) step(gs++, kt, more, NONE{});
        }
        if (stamp) stamp[1] = __builtin_amdgcn_s_memtime();
        __syncthreads();   // staging tile of the previous tile complete
        if (stamp) stamp[2] = __builtin_amdgcn_s_memtime();

        // table of the next tile (its buffer held the previous tile's table, no longer needed)
        PersTableLoad tln{0u, 0, 0, false};
        if (more) tln = pers_table_issue(g, nxt.n0, tid);
        if (it > 0 && !(g.stagger_units & 4)) pp_phase2<EPI>(g, stg, prev_m0, prev_n0, tid);
        if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
        if (more) pers_table_write(tln, tab_next, tid);
        // hand this tile's accumulators to the drain set
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dr[i][j] = acc[i][j];
        prev_m0 = cur.m0; prev_n0 = cur.n0;
        prev_tab_addr = smem_base + (unsigned)(PP_PT_OFF + (it & 1) * PT_BYTES + 8 * (64 * wc + 4 * h));
        if (more) {
#pragma unroll
            for (int idx = 0; idx < 6; ++idx) src_cur[idx] = src_nxt[idx];
        }
        cur = nxt;
        if (stamp) stamp[4] = __builtin_amdgcn_s_memtime();
        __syncthreads();   // staging tile read out, next table visible
        if (stamp) stamp[5] = __builtin_amdgcn_s_memtime();
    }
    // ---- drain the last tile without overlap
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    pp_for_slots(
        [&](auto s) {
            constexpr int G = decltype(s)::value;
            v4f t0, t1;
            pp_load_group<G>(t0, t1, prev_tab_addr);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1)::"memory");
            pp_unit<4 * G + 0>(dr, t0, t1, stg_addr, unc, amax);
            pp_unit<4 * G + 1>(dr, t0, t1, stg_addr, unc, amax);
            pp_unit<4 * G + 2>(dr, t0, t1, stg_addr, unc, amax);
            pp_unit<4 * G + 3>(dr, t0, t1, stg_addr, unc, amax);
        },
        std::make_integer_sequence<int, 8>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (__builtin_amdgcn_ballot_w64((unc != 0) | (amax >= 4194304.0f)) != 0)
        pp_exact_tile(dr, g, stg, prev_n0, 64 * wc, 128 * wt, h, l31);
    __syncthreads();
    pp_phase2<EPI>(g, stg, prev_m0, prev_n0, tid);
#undef PP_TIE
}

template <int EPI>
int launch_lab(GemmArgs& g, const char* name, ivit_stream_t stream, int* handled)
{
    *handled = 1;
    if constexpr (EPI != EPI_I32) {
        if (g.M >= 2048 && g.N % XCH == 0 && !g_force_small && g_kernel_choice != 1 && g.flags == 0 &&
            (g_debug_flags & 4194304)) {   // persistent 256 x 256
            g.tiles_m = (g.M + XTOK - 1) / XTOK;
            g.tiles_n = g.N / XCH;
            const int ntiles = g.tiles_m * g.tiles_n;
            hipLaunchKernelGGL((gemm_i8_xlp_kernel<EPI>), dim3(ntiles < 256 ? ntiles : 256), dim3(XL_NT), 0,
                               ivit_stream(stream), g);
            IVIT_CHECK_LAUNCH(name);
        }
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
        if (g.M >= 2048 && g.N >= BCH && g.K >= 8 * BK && !g_force_small && g.flags == 0 && (g_debug_flags & 8388608)) {
            g.tiles_m = (g.M + BTOK - 1) / BTOK;
            g.tiles_n = (g.N + BCH - 1) / BCH;
            const int ntiles = g.tiles_m * g.tiles_n;
            const dim3 grid(ntiles < 256 ? ntiles : 256);
            g.stagger_units = (g_debug_flags >> 16) & 63;   // ablation bits of the pp kernel (measurement only)
            if (EPI == EPI_RQ && (g_debug_flags & 33554432)) g.res = (const int8_t*)g_stamp_buf;   // time stamps
            if (g.K >= 16 * BK && !(g_debug_flags & 16777216))
                hipLaunchKernelGGL((gemm_i8_pp_kernel<EPI, 2>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
            else if (g.K >= 11 * BK && !(g_debug_flags & 16777216))
                hipLaunchKernelGGL((gemm_i8_pp_kernel<EPI, 3>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
            else
                hipLaunchKernelGGL((gemm_i8_pp_kernel<EPI, 4>), grid, dim3(BIG_NT), 0, ivit_stream(stream), g);
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
        if (g.M >= 2048 && g.N >= BCH && !g_force_small && (g.flags != 0 || (g_debug_flags & 1024))) {   // relaunch-per-tile form
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
    *handled = 0;
    return IVIT_OK;
}

}  // namespace

int ivit_gemm_lab_launch(int epi, void* gemm_args, const char* name, ivit_stream_t stream, int* rc)
{
    GemmArgs& g = *static_cast<GemmArgs*>(gemm_args);
    int handled = 0;
    switch (epi) {
        case EPI_RQ: *rc = launch_lab<EPI_RQ>(g, name, stream, &handled); break;
        case EPI_RESID: *rc = launch_lab<EPI_RESID>(g, name, stream, &handled); break;
        case EPI_QKV: *rc = launch_lab<EPI_QKV>(g, name, stream, &handled); break;
        default: break;
    }
    return handled;
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

IVIT_EXPORT int ivit_debug_set_gemm_flags2(int flags)
{
    g_debug_flags2 = flags;
    return IVIT_OK;
}

// diagnostic: device buffer (8 x uint64 per workgroup) receiving {HW_ID | XCC_ID<<32, t_start, t_loop_end, t_end}
// from the stamped build selected by ivit_debug_set_gemm_flags(512)
IVIT_EXPORT int ivit_debug_set_stamp_buffer(void* buf)
{
    g_stamp_buf = buf;
    return IVIT_OK;
}
