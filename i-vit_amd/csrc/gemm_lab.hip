// gemm_lab.hip -- the one earlier form of the large INT8 GEMM kernel that is kept, bit-identical, for A/B measurement and
// ablation through include/ivit_hip_debug.h: the relaunch-per-tile 256 x 128 LDS-DMA kernel (scripts/gemm_ablate.py, gemm_ab.py,
// gemm_timeline.py; tests/test_gpu_ops.py::test_gemm_both_kernels_agree).  The other forms tried in rounds 1-2 (256 x 256 tile,
// persistent 256 x 256, four-deep ring, producer/consumer split) are gone from the tree; what each taught is in DESIGN.md section 5.
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

template <int EPI>
int launch_lab(GemmArgs& g, const char* name, ivit_stream_t stream, int* handled)
{
    *handled = 1;
    if constexpr (EPI != EPI_I32) {
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
