// gemm_wp.h -- the weights-in-registers GEMM with the EPILOGUE INSIDE THE NEXT TILE'S MAIN LOOP (round 4; included by gemm.hip).
//
// What scripts/probes/coissue_probe.hip measured (profiles/r04j_*): two waves of a SIMD that each issue up to two VALU
// instructions behind every v_mfma_i32_16x16x64_i8 keep the MFMA pipe at its full rate (18.9 against 18.2 cycles per MFMA and
// SIMD; three: 23.7), while the kernel of gemm.hip -- whose two co-resident workgroups run main loop and epilogue one after
// the other -- leaves the pipe idle for the length of every epilogue (tile period 22.4 K cycles for 2 x 384 MFMAs x 18.2 = 14 K).
// So here a tile's requantisation rides in the issue slots of the NEXT tile's MFMAs:
//   * one workgroup of EIGHT waves per CU (two per SIMD), tile 128 tokens x 256 channels; wave w owns channels [32 w, 32 w + 32)
//     x all 128 tokens = 2 x 8 accumulator tiles of 16 x 16 = 64 registers -- and there are TWO such sets: one accumulates tile
//     t while the other, holding tile t - 1, is requantised, packed, transposed and stored between the MFMAs of tile t;
//   * operand paths as in gemm_i8_wreg_kernel<.., S16>: the token tile by LDS-DMA into a 3-stage ring (8 pieces of 1 KB per K
//     step, one per wave), fragment-packed weights (IVIT_W_FRAGS16: a wave's 32 channels are 2 KB of every 4 KB K step)
//     by inline-asm loads into three rotating register buffers, counted s_waitcnt, one barrier per K step;
//   * the old tile's 16 accumulator tiles are 16 UNITS of (4 cvt, 8 fma, 4 v_sad, 4 v_med3, 3 pack) = 23 VALU instructions, one
//     per half K step (8 MFMAs), cut into parts of <= 4 instructions behind successive MFMAs: K steps 0-7; K step 8 transposes
//     and stores (lane (g4, l15) ends with the 16 bytes of channels 16 (g4 & 1) .. + 15 of token 16 (2 jp + (g4 >> 1)) + l15);
//     from K step 9 on the loop is the plain one (K >= 768: nk >= 12);
//   * the certificate (float32 bracket, gemm_common.h epilogue_direct_16) is checked per unit; a failed unit is redone in
//     float64 on the spot (rare, wave-uniform branch);
//   * the per-tile table (bracket + bias of the tile's 256 channels) is requested at the tile's start for the NEXT tile and
//     written at K step 9; the next tile's first stages go out behind the loop.
// EPI_RQ (row-major or block-layout output) and EPI_QKV (head-major q / k / v).  Parity: the same tests as the other forms
// (tests/test_gpu_ops.py, `wreg_tiles` / layouts), bit-identical by construction (same certificate, same fallback).
#pragma once

constexpr int WP_NT = 512;
constexpr int WP_PARK = WR_RING + 2 * WR_TAB + 256;        // 8 waves x 8 KB: the finished tile's second channel sub-tile (see below)
constexpr int WP_SMEM = WP_PARK + 8 * 8192;               // 94.25 KB of the CU's 160 (one workgroup per CU)

template <int OFF>
IVIT_DEV void wp_lds_write16(unsigned a, const v4i& d)
{
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(d), "n"(OFF) : "memory");
}

// ABL (lab build, timing only -- results wrong): 1 no units / transposes / stores inside the loop (the bare eight-wave loop),
// 2 no barrier per K step, 4 no weight loads in the loop, 8 no LDS-DMA in the loop
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(WP_NT, 1) void gemm_i8_wp_kernel(GemmArgs g)
{
    static_assert(EPI == EPI_RQ || EPI == EPI_QKV, "wave-pipelined form: int8 outputs without a second operand");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // WP_SMEM bytes (dynamic: beyond the 64 KB of a static allocation)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // the lane id is RECOMPUTED (v_mbcnt) wherever it is needed: held in a register it lives through every main loop for the
    // sake of a few address computations per tile, and this kernel has no register to spare
    auto lane_id = [] { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
#define lane lane_id()
#define tid (64 * wave + lane_id())
    const int nk = g.K / BK;          // launcher: nk % 3 == 0, nk >= 12

    // ---- sources (running pointers, advanced by every issue)
    const int8_t* asrc = g.A;
    const int8_t* wsrc = g.W;
    auto set_sources = [&](const WrWork& w) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int lrow_o = lane_o & 15, lslot_o = lane_o >> 4;
        const int row = 16 * wave + lrow_o;
        asrc = g.A + (int64_t)min(w.m0 + row, g.M - 1) * g.lda + 16 * lslot_o;
        if (g.a_blocks) {
            const unsigned pos = (unsigned)(4 * lrow_o + (lslot_o ^ ((lrow_o >> 2) & 3)));
            asrc = g.A + (int64_t)min((w.m0 >> 4) + wave, ((g.M + 15) >> 4) - 1) * (g.K >> 6) * 1024 + pos * 16u;
        }
        const int cg = min((w.n0 >> 6) + (wave >> 1), ((g.N + 63) >> 6) - 1);
        wsrc = g.W + (int64_t)cg * nk * 4096 + (wave & 1) * 2048 + (unsigned)lane_o * 16u;
    };
    const int kstep_a = g.a_blocks ? 1024 : BK;
    auto issue_dma = [&](int kt) {
        __builtin_amdgcn_global_load_lds((gptr_t)asrc, (lptr_t)(smem + (kt % WR_STAGES) * WR_STAGE + 1024 * wave), 16, 0, 0);
        asrc += kstep_a;
    };
    v4i wr0[2], wr1[2], wr2[2];
    auto issue_w = [&](v4i (&wr)[2], int p) {
        const int8_t* src = wsrc;
        if (p == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wr[0]) : "v"(src));
        if (p == 1) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(wr[1]) : "v"(src));
            wsrc += 4096;
        }
    };
    auto prefetch = [&](const WrWork& w) {      // stage 0, weight buffer 0, stage 1
        set_sources(w);
        issue_dma(0);
        issue_w(wr0, 0);
        issue_w(wr0, 1);
        issue_dma(1);
    };

    // ---- per-tile table: float2 lohi[256] | int bias[256] (brackets widened by two float32 steps as in the S16 form of gemm.hip)
    auto table_issue = [&](int n0) {
        PersTableLoad r{0u, 0, 0, false};
        const int c = n0 + tid;
        if (tid < WR_CH && c < g.N) {
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
            lh.x = __int_as_float(__float_as_int(lh.x) - 2);
            lh.y = __int_as_float(__float_as_int(lh.y) + 2);
        }
        int tid_w = tid;
        asm volatile("" : "+v"(tid_w));
        if (tid_w < WR_CH) {
            reinterpret_cast<float2*>(tab)[tid_w] = lh;
            reinterpret_cast<int*>(tab + WR_CH * 8)[tid_w] = r.bias;
        }
    };

    const unsigned abase = (unsigned)(__UINTPTR_TYPE__)(lptr_t)smem + (unsigned)lane * 16u;
    using T = std::true_type;
    using F = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // Accumulators [token sub-tile j of 16] of channel sub-tile 0 in two sets (A / B: one accumulates, the other is requantised) and
    // of channel sub-tile 1 in ONE: 256 registers do not hold two full sets beside the operand buffers, so a finished tile's second
    // sub-tile is PARKED in LDS (8 ds_write_b128 per wave at the tile's end, 8 KB per wave) and its units read it back, one
    // ds_read_b128 at the head of their half step (LDS instructions: not the VALU port the scheme is about)
    v4i accA[8], accB[8], acc1[8];
    const unsigned park = (unsigned)(__UINTPTR_TYPE__)(lptr_t)smem + (unsigned)(WP_PARK + 8192 * wave) + (unsigned)lane * 16u;
    auto mfma16 = [&](v4i& c, const v4i& a, const v4i& b) { asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); };

    // ---- everything about the OLD tile (the one being requantised) that its epilogue needs
    struct Old {
        int m0, n0;           // m0 < 0: none (a workgroup's first tile: the units run on zeros and nothing is stored)
    };

    // One work item.  accN accumulates `cur`; accO holds `old` (requantised here).  tab_cur: this tile's table (bias now; its
    // brackets are read when the tile is the old one), tab_old: the old tile's, tab_next: written here for `nxt`.
    auto run = [&](v4i (&accN)[8], v4i (&accO)[8], const WrWork& cur, const Old& old, const WrWork& nxt, char* tab_cur, char* tab_old,
                   char* tab_next) {
        v4i af0[4], af1[4];
        auto load_frags = [&](auto stage_tag, auto half_tag, v4i (&af)[4]) {
            constexpr int O = decltype(stage_tag)::value * WR_STAGE + decltype(half_tag)::value * 4096;
            lds_read16_async_off<O>(af[0], abase);
            lds_read16_async_off<O + 1024>(af[1], abase);
            lds_read16_async_off<O + 2048>(af[2], abase);
            lds_read16_async_off<O + 3072>(af[3], abase);
        };
        auto wait_frags = [&](v4i (&af)[4]) { asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])::"memory"); };
        auto wait_next = [&](auto inflight_tag, v4i (&af)[4], v4i (&wn)[2]) {
            if constexpr (decltype(inflight_tag)::value)
                asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(wn[0]), "+v"(wn[1])::"memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(wn[0]), "+v"(wn[1])::"memory");
        };

        // ---- the old tile's brackets for this lane's 2 x 4 channels (registers for the whole tile: no LDS read inside the loop
        //      beside the fragment reads, whose lgkmcnt waits are counted) and this tile's bias into the accumulators
        v4f lh0, lh1;       // lo0 hi0 lo1 hi1 | lo2 hi2 lo3 hi3 of the channel sub-tile whose units are running (0, from K step 4 on: 1)
        unsigned qa;
        {
            int lane_b = lane;
            asm volatile("" : "+v"(lane_b));
            qa = lds_addr(tab_old) + 8u * (unsigned)(32 * wave + 4 * (lane_b >> 4));
            const unsigned ba = lds_addr(tab_cur) + (unsigned)(WR_CH * 8 + 4 * (32 * wave + 4 * (lane_b >> 4)));
            v4i bq[2];
            lds_read16_async_off<0>(lh0, qa);
            lds_read16_async_off<16>(lh1, qa);
            lds_read16_async_off<0>(bq[0], ba);
            lds_read16_async_off<64>(bq[1], ba);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lh0), "+v"(lh1), "+v"(bq[0]), "+v"(bq[1])::"memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                accN[j] = bq[0];
                acc1[j] = bq[1];
                // materialised HERE: left to the scheduler the copies (v_mov_b64) sink to just in front of the tile's first MFMAs,
                // and an MFMA issued right behind them read stale halves of its C operand (half of the dwords by lane parity: first
                // parity run of this kernel); the fragment reads, the table request and the first weight loads lie in between now
                asm volatile("" : "+v"(accN[j]), "+v"(acc1[j]));
            }
        }
        load_frags(I0{}, I0{}, af0);
        // the next tile's table operands: in flight through the first K steps (older than every load the loop waits for)
        const bool more = nxt.m0 >= 0;
        PersTableLoad tl = table_issue(more ? nxt.n0 : cur.n0);

        // ---- a unit: accumulator tile (i, j) of the old tile -> one packed dword (4 channels of one token), in parts
        float ua[4];
        int utl[4];
        unsigned uunc = 0;
        unsigned D[2][8];
        const int ncol_old = old.n0 + 32 * wave;
        v4i pk;       // a parked accumulator tile on its way back (units 8-15)
        auto unit_fetch = [&](auto u_tag) {      // at the head of the unit's half step, in front of the fragment reads
            constexpr int U = decltype(u_tag)::value;
            if constexpr (U >= 8) lds_read16_async_off<(U & 7) * 1024>(pk, park);
        };
        auto unit_part = [&](auto u_tag, int part) {
            constexpr int U = decltype(u_tag)::value, i = U >> 3, j = U & 7;
            v4i& acc = (i == 0) ? accO[j] : pk;
            if (part == 0) {
                uunc = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) ua[r] = (float)acc[r];
            } else if (part == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)      // (inline asm: left to the compiler the four become two v_pk_fma_f32 and four moves)
                    asm("v_fmaak_f32 %0, %1, %2, 0x4b400000" : "=v"(utl[r]) : "v"(ua[r]), "v"(r < 2 ? lh0[2 * r] : lh1[2 * r - 4]));
            } else if (part == 2 || part == 3) {      // the upper bracket and the certificate, two outputs per part
#pragma unroll
                for (int r = 2 * (part - 2); r < 2 * (part - 2) + 2; ++r) {
                    int th;
                    asm("v_fmaak_f32 %0, %1, %2, 0x4b400000" : "=v"(th) : "v"(ua[r]), "v"(r < 2 ? lh0[2 * r + 1] : lh1[2 * r - 3]));
                    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(uunc) : "v"(utl[r]), "v"(th), "v"(uunc));
                }
                if (part == 3 && __builtin_amdgcn_ballot_w64(uunc != 0) != 0) {      // rare: exact float64 evaluation of the unit (quant_utils.py:229-230)
                    const int c0 = min(ncol_old + 16 * i + 4 * (lane >> 4), g.N - 4);
                    const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
                    const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
                    const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z), dyadic_mult(m4.w, e4.w)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double t = (double)acc[r] * Mc[r] + IVIT_MAGIC;
                        utl[r] = 0x4B400000 + clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                    }
                }
            } else if (part == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) utl[r] = clamp_i32(utl[r], 0x4B400000 - 128, 0x4B400000 + 127);      // low byte = int8 result
            } else if (part == 5) {
                D[i][j] = __builtin_amdgcn_perm((unsigned)utl[1], (unsigned)utl[0], 0x0c0c0400u) |
                          __builtin_amdgcn_perm((unsigned)utl[3], (unsigned)utl[2], 0x04000c0cu);
            }
        };
        // ---- store addressing of the old tile (derived at K step 8, not carried through the loop).  EPI_QKV: the channel part and
        //      the (image, token) of the lane's first row by division once per tile, 32 rows further per pair by increment
        int8_t* const out = reinterpret_cast<int8_t*>(g.out);
        unsigned st_off0 = 0;          // EPI_QKV: offset of the channel part; else of (row old.m0 + 16 (g4 >> 1) + l15, column c)
        int st_b = 0, st_tok = 0, st_t0 = 0;
        bool st_colok = false;
        auto store_setup = [&]() {
            int lane_s = lane;
            asm volatile("" : "+v"(lane_s));
            const int g4 = lane_s >> 4, l15 = lane_s & 15;
            const int c = ncol_old + 16 * (g4 & 1);
            st_colok = old.m0 >= 0 && ncol_old < g.N;
            st_t0 = max(old.m0, 0) + 16 * (g4 >> 1) + l15;
            if constexpr (EPI == EPI_QKV) {
                // channel part: scalar arithmetic on the wave's two 16-channel chunks (a division by a loop-invariant per lane would
                // park its reciprocal in a register through every main loop); row part: multiply-high by the launcher's constant
                const int cdim = g.heads * g.head_dim, nb = g.M / g.tokens;
                unsigned offc[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int cs = __builtin_amdgcn_readfirstlane(min(ncol_old + 16 * k, g.N - 16));
                    const int which = cs / cdim, rem = cs - which * cdim;
                    const int hh = rem / g.head_dim, d0 = rem - hh * g.head_dim;
                    offc[k] = (unsigned)(((which * nb * g.heads + hh) * g.tokens) * g.head_dim + d0);
                }
                st_off0 = (g4 & 1) ? offc[1] : offc[0];
                const int tt = min(st_t0, g.M - 1);
                st_b = (int)__umulhi((unsigned)tt, g.tokens_magic);
                st_tok = tt - st_b * g.tokens;
            } else if (g.out_blocks) {
                st_off0 = (unsigned)min(c, g.N - 16);
            } else {
                st_off0 = (unsigned)st_t0 * (unsigned)g.ldo + (unsigned)c;
            }
        };
        auto store_pair = [&](auto jp_tag) {      // token sub-tiles 2 jp, 2 jp + 1: a 4 x 4 dword transpose over the token's four lanes
            constexpr int jp = decltype(jp_tag)::value;
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            // lane g4 <- the dwords of lanes 0..3 in column k = g4, columns k = i + 2 jj: (i = 0, jj = 0), (1, 0), (0, 1), (1, 1)
            const v2u ab = __builtin_amdgcn_permlane32_swap(D[0][2 * jp], D[0][2 * jp + 1], false, false);
            const v2u cd = __builtin_amdgcn_permlane32_swap(D[1][2 * jp], D[1][2 * jp + 1], false, false);
            const v2u ac = __builtin_amdgcn_permlane16_swap(ab.x, cd.x, false, false);
            const v2u bd = __builtin_amdgcn_permlane16_swap(ab.y, cd.y, false, false);
            const int t = st_t0 + 32 * jp;
            unsigned off;
            if constexpr (EPI == EPI_QKV) {
                if (jp > 0) {       // 32 rows further
                    if (g.tokens >= 32) {
                        st_tok += 32;
                        if (st_tok >= g.tokens) { st_tok -= g.tokens; ++st_b; }
                    } else {
                        const int tt = min(t, g.M - 1);
                        st_b = (int)__umulhi((unsigned)tt, g.tokens_magic);
                        st_tok = tt - st_b * g.tokens;
                    }
                }
                off = st_off0 + (unsigned)((st_b * g.heads * g.tokens + st_tok) * g.head_dim);
            } else if (g.out_blocks) {
                off = block_off(block_row(t, g.N), block_col((int)st_off0));
            } else {
                off = st_off0 + (unsigned)(32 * jp) * (unsigned)g.ldo;
            }
            if (st_colok && t < g.M) store16_pol<IVIT_STORE_POLICY>(out + off, make_int4((int)ac.x, (int)ac.y, (int)bd.x, (int)bd.y));
        };

        // ---- K step kt; EPS: what of the old tile rides along: 0..7 = units 2 EPS, 2 EPS + 1; 8 = transposes + stores; 9 = the next
        //      tile's table; -1 = nothing
        auto step = [&](v4i (&wc)[2], v4i (&wn)[2], v4i (&wf)[2], auto issue_tag, auto last_tag, auto stage_tag, auto eps_tag) {
            constexpr bool ISSUE = decltype(issue_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr int ST = decltype(stage_tag)::value;
            constexpr int EPS = decltype(eps_tag)::value;
            if constexpr (EPS == 4) {      // the second channel sub-tile's brackets replace the first's
                lds_read16_async_off<128>(lh0, qa);
                lds_read16_async_off<144>(lh1, qa);
            }
            if constexpr (EPS >= 0 && EPS < 8) unit_fetch(std::integral_constant<int, 2 * (EPS < 8 ? EPS : 0)>{});
            load_frags(stage_tag, I1{}, af1);
            wait_frags(af0);
            if constexpr (EPS == 4) asm volatile("" : "+v"(lh0), "+v"(lh1));
            if constexpr (EPS >= 4 && EPS < 8) asm volatile("" : "+v"(pk));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (n < 4) mfma16(accN[n & 3], wc[0], af0[n & 3]);
                else mfma16(acc1[n & 3], wc[1], af0[n & 3]);
                if constexpr (ISSUE) {      // the three loads of K step kt + 2 behind the first MFMAs
                    if (n == 0) { if constexpr (!(ABL & 8)) issue_dma(ST + 2); }
                    if (n == 1) { if constexpr (!(ABL & 4)) issue_w(wf, 0); }
                    if (n == 2) { if constexpr (!(ABL & 4)) issue_w(wf, 1); }
                }
                if constexpr (EPS >= 0 && EPS < 8 && !(ABL & 1)) unit_part(std::integral_constant<int, 2 * (EPS < 8 ? EPS : 0)>{}, n);
                if constexpr (EPS == 8 && !(ABL & 1)) {
                    if (n == 0) store_setup();
                    if (n == 3) store_pair(I0{});
                    if (n == 6) store_pair(I1{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (EPS >= 0 && EPS < 8) unit_fetch(std::integral_constant<int, 2 * (EPS < 8 ? EPS : 0) + 1>{});
            wait_next(issue_tag, af1, wn);
            if constexpr (EPS >= 4 && EPS < 8) asm volatile("" : "+v"(pk));
            if constexpr (!(ABL & 2)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (!LAST) load_frags(std::integral_constant<int, (ST + 1) % WR_STAGES>{}, I0{}, af0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (n < 4) mfma16(accN[4 + (n & 3)], wc[0], af1[n & 3]);
                else mfma16(acc1[4 + (n & 3)], wc[1], af1[n & 3]);
                if constexpr (EPS >= 0 && EPS < 8 && !(ABL & 1)) unit_part(std::integral_constant<int, 2 * (EPS < 8 ? EPS : 0) + 1>{}, n);
                if constexpr (EPS == 8 && !(ABL & 1)) {
                    if (n == 1) store_pair(I2{});
                    if (n == 5) store_pair(std::integral_constant<int, 3>{});
                }
                if constexpr (EPS == 9) {
                    if (n == 0 && more) table_write(tl, tab_next);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using E = std::integral_constant<int, -1>;
#define WP_EPS(k) std::integral_constant<int, k>{}
        // issue_dma's stage argument is only used modulo 3: the unrolled steps pass their stage + 2
#pragma unroll
        for (int p = 0; p < 2; ++p) issue_w(wr1, p);
        asm volatile("" : "+v"(wr0[0]), "+v"(wr0[1])::"memory");
        step(wr0, wr1, wr2, T{}, F{}, I0{}, WP_EPS(0));
        step(wr1, wr2, wr0, T{}, F{}, I1{}, WP_EPS(1));
        step(wr2, wr0, wr1, T{}, F{}, I2{}, WP_EPS(2));
        step(wr0, wr1, wr2, T{}, F{}, I0{}, WP_EPS(3));
        step(wr1, wr2, wr0, T{}, F{}, I1{}, WP_EPS(4));
        step(wr2, wr0, wr1, T{}, F{}, I2{}, WP_EPS(5));
        step(wr0, wr1, wr2, T{}, F{}, I0{}, WP_EPS(6));
        step(wr1, wr2, wr0, T{}, F{}, I1{}, WP_EPS(7));
        step(wr2, wr0, wr1, T{}, F{}, I2{}, WP_EPS(8));
        step(wr0, wr1, wr2, T{}, F{}, I0{}, WP_EPS(9));
        int kt = 10;
        for (; kt + 5 <= nk; kt += 3) {       // steps 10 .. nk - 3 in triples (nk = 12: none)
            step(wr1, wr2, wr0, T{}, F{}, I1{}, E{});
            step(wr2, wr0, wr1, T{}, F{}, I2{}, E{});
            step(wr0, wr1, wr2, T{}, F{}, I0{}, E{});
        }
        step(wr1, wr2, wr0, F{}, F{}, I1{}, E{});
        step(wr2, wr0, wr1, F{}, T{}, I2{}, E{});
#undef WP_EPS
        // the ring is free (every wave's reads returned before the last barrier): the next item's first stages go out; a
        // workgroup's last item requests its own again (see gemm_i8_wreg_kernel: unconditional for the ISA lint)
        prefetch(more ? nxt : cur);
        // park the second channel sub-tile (its registers accumulate the next tile from here on)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA results -> LDS data reads
        wp_lds_write16<0>(park, acc1[0]);
        wp_lds_write16<1024>(park, acc1[1]);
        wp_lds_write16<2048>(park, acc1[2]);
        wp_lds_write16<3072>(park, acc1[3]);
        wp_lds_write16<4096>(park, acc1[4]);
        wp_lds_write16<5120>(park, acc1[5]);
        wp_lds_write16<6144>(park, acc1[6]);
        wp_lds_write16<7168>(park, acc1[7]);
    };

    // ---- the last tile of a workgroup: its units, transposes and stores without a loop to hide in
    auto flush = [&](v4i (&accO)[8], const Old& old, char* tab_old) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // MFMA results -> VALU reads; the parking writes done
        int lane_b = lane;
        asm volatile("" : "+v"(lane_b));
        const int g4 = lane_b >> 4, l15 = lane_b & 15;
        const float2* lh = reinterpret_cast<const float2*>(tab_old);
        const int ncol_old = old.n0 + 32 * wave;
        const v4i* parked = reinterpret_cast<const v4i*>(smem + WP_PARK + 8192 * wave) + lane_b;
        unsigned D[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float lo[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 v = lh[32 * wave + 16 * i + 4 * g4 + r];
                lo[r] = v.x;
                hi[r] = v.y;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const v4i acc = i == 0 ? accO[j] : parked[64 * j];
                int b[4];
                unsigned unc = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = (float)acc[r];
                    const int tl = __float_as_int(__builtin_fmaf(a, lo[r], 12582912.0f));
                    const int th = __float_as_int(__builtin_fmaf(a, hi[r], 12582912.0f));
                    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                    b[r] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);
                }
                if (__builtin_amdgcn_ballot_w64(unc != 0) != 0) {
                    const int c0 = min(ncol_old + 16 * i + 4 * g4, g.N - 4);
                    const uint4 m4 = *reinterpret_cast<const uint4*>(g.m + c0);
                    const int4 e4 = *reinterpret_cast<const int4*>(g.e + c0);
                    const double Mc[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z), dyadic_mult(m4.w, e4.w)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double t = (double)acc[r] * Mc[r] + IVIT_MAGIC;
                        b[r] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                    }
                }
                D[i][j] = __builtin_amdgcn_perm((unsigned)b[1], (unsigned)b[0], 0x0c0c0400u) | __builtin_amdgcn_perm((unsigned)b[3], (unsigned)b[2], 0x04000c0cu);
            }
        }
        int8_t* const out = reinterpret_cast<int8_t*>(g.out);
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const v2u ab = __builtin_amdgcn_permlane32_swap(D[0][2 * jp], D[0][2 * jp + 1], false, false);
            const v2u cd = __builtin_amdgcn_permlane32_swap(D[1][2 * jp], D[1][2 * jp + 1], false, false);
            const v2u ac = __builtin_amdgcn_permlane16_swap(ab.x, cd.x, false, false);
            const v2u bd = __builtin_amdgcn_permlane16_swap(ab.y, cd.y, false, false);
            const int t = old.m0 + 16 * (2 * jp + (g4 >> 1)) + l15;
            const int c = ncol_old + 16 * (g4 & 1);
            const bool ok = ncol_old < g.N && t < g.M;
            unsigned off;
            if constexpr (EPI == EPI_QKV) {
                const int cc = min(c, g.N - 16);
                const int cdim = g.heads * g.head_dim;
                const int which = cc / cdim, rem = cc - which * cdim;
                const int hh = rem / g.head_dim, d0 = rem - hh * g.head_dim;
                const int nb = g.M / g.tokens;
                const int tt = min(t, g.M - 1);
                const int qb = (int)__umulhi((unsigned)tt, g.tokens_magic), qtok = tt - qb * g.tokens;
                off = (unsigned)((((which * nb + qb) * g.heads + hh) * g.tokens + qtok) * g.head_dim + d0);
            } else if (g.out_blocks) {
                off = block_off(block_row(t, g.N), block_col(min(c, g.N - 16)));
            } else {
                off = (unsigned)t * (unsigned)g.ldo + (unsigned)c;
            }
            if (ok) store16_pol<IVIT_STORE_POLICY>(out + off, make_int4((int)ac.x, (int)ac.y, (int)bd.x, (int)bd.y));
        }
    };

    const int G = gridDim.x, b = blockIdx.x;
    WrWork cur = wr_work(g, 0, b, G);
    if (cur.m0 < 0) return;   // uniform
    char* const tab0 = smem + WR_RING;
    char* const tab1 = smem + WR_RING + WR_TAB;
    table_write(table_issue(cur.n0), tab0);
    // the "old tile" of the first item: zeros, never stored; its brackets are whatever tab1 holds -- make them finite
    if (tid < WR_CH) reinterpret_cast<float2*>(tab1)[tid] = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        accB[j] = v4i{0, 0, 0, 0};
        reinterpret_cast<v4i*>(smem + WP_PARK + 8192 * wave)[64 * j + lane] = v4i{0, 0, 0, 0};
    }
    prefetch(cur);
    Old old{-1, 0};
    for (int it = 0;; it += 2) {
        // stage 0 and weight buffer 0 of `cur` have landed when at most the stage-1 piece behind them is in flight (the old
        // tile's stores are older: they only make this wait longer); then everyone's stage 0 and the table are visible
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        WrWork nxt = wr_work(g, it + 1, b, G);
        run(accA, accB, cur, old, nxt, tab0, tab1, tab1);
        old = Old{cur.m0, cur.n0};
        if (nxt.m0 < 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummy prefetch: nothing may be in flight at the end
            flush(accA, old, tab0);
            return;
        }
        cur = nxt;
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        nxt = wr_work(g, it + 2, b, G);
        run(accB, accA, cur, old, nxt, tab1, tab0, tab0);
        old = Old{cur.m0, cur.n0};
        if (nxt.m0 < 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            flush(accB, old, tab1);
            return;
        }
        cur = nxt;
    }
}
#undef lane
#undef tid
