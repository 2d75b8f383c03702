// attention.hip -- fused integer attention core, one workgroup per (image, head).
// Replaces, for one head, the chain of /root/reference/models/vit_quant.py:72-85
//   matmul_1 (QuantMatMul, quant_modules.py:404-409) -> * scale -> qact_attn1 (fixedpoint_mul)
//   -> IVITIntSoftmax (Shiftmax, ivit_modules.py:150-179) -> matmul_2 -> qact2
// without materialising the [B,H,T,T] score tensor.
//
// MFMA formulation (v_mfma_i32_16x16x64_i8: the whole head dimension is ONE instruction deep):
//   S^T tile (16 keys x 16 queries) = K_tile . Q_tile^T: keys are the "A" rows, queries the "B" columns, so a
//   lane owns ONE query (col = lane&15) and 4 keys of each 16-key tile (key = 16kt + 4(lane>>4) + r); the lanes
//   l, l^16, l^32, l^48 share a query.  A whole Shiftmax row (<= 208 keys) is 13 x 4 registers in four lanes:
//   row max / row sum are register reductions plus two cross-lane exchanges.  ~110 VGPRs, so several
//   workgroups share a CU and one wave's Shiftmax arithmetic (VALU) runs under another's MFMAs.
//   O^T tile (16 d x 16 queries) = Vt . P^T over key steps of 64: the packed int8 probabilities of four
//   consecutive key tiles ARE the "B" fragment (byte 4t + r of step s = key 64s + 16t + 4g + r); V is transposed
//   once per workgroup into LDS with its keys stored in exactly that byte order, so the "A" fragment is a single
//   ds_read_b128.  Both LDS images are swizzled for conflict-free 16-lane-group reads.
// Shiftmax's integer exponential depends only on (row max - k) in [0,255]: tabulated once per workgroup in LDS
// (256 x u32) with the reference's arithmetic (shiftexp_int); the row sum is an exact u32 sum rounded once to
// float32 (= the reference's float32 sum whenever that is exact).
#include "common.h"

#if IVIT_LAB
extern int g_ln_ablate;     // rowops.hip (ivit_debug_ln_ablate); bits 20-24 ablate phases of attention_kernel<0> (scripts/attn_ablate.py)
#else
constexpr int g_ln_ablate = 0;
#endif

namespace {

constexpr int NT = 256;
constexpr int HD = 64;                 // head dim = one 16x16x64 MFMA deep
constexpr int NKT = 13;                // key tiles of 16 (tokens <= 208)
constexpr int KP = NKT * 16;           // 208 K rows in LDS
constexpr int NKS = 4;                 // key steps of 64 for P.V (256 key slots; P = 0 beyond T)
constexpr int VT_ROW = NKS * 64;       // 256 B per d row: one bank row, chunk j stored at j ^ (d & 15)
constexpr int K_BYTES = KP * HD;       // 13312
constexpr int VT_BYTES = HD * VT_ROW;  // 16384
constexpr int LUT_OFF = K_BYTES + VT_BYTES;
constexpr int SMEM_BYTES = LUT_OFF + 2 * 256 * 4;   // exponent table as u32 (exact row sum) and as float32 (the product of :175)

struct AttnArgs {
    const int8_t* qkv;
    int8_t* out;
    int batch, heads, tokens;
    double Ms, Mo;
    int x0;    // floor(-1/s_attn)
    int ksat;  // first table index whose argument is clamped at n*x0: every later entry is identical
    int out_blocks;   // output in the GEMM block layout (common.h: ivit_block_offset), row length heads * 64
    // natural-scale ("compat") Shiftmax: exp_int as a function of (row max q, q), [256][256] u32 indexed
    // (qmax + 128) * 256 + (q + 128).  The reference's Shiftmax runs its whole float32 sequence on phi(q) = fl(fl(q*s)/s)
    // (ivit_modules.py:165-170; the .to(int32) of :166 is discarded), so the exponent is no longer a function of qmax - q
    // alone; the host tabulates it with the reference's float32 steps (prepare.shiftexp2d).  NULL: power-of-two scale.
    const unsigned* exp2d;
    // The same table in BAND form for the LDS path: band[(qmax + 128) * band_w + j] = exp_int of (qmax, q = qmax - j),
    // j < band_w; band_w is a multiple of 16 and entry band_w - 1 is already the saturated value -|x0| (the argument is
    // clamped at n * x0 from there on, ivit_modules.py:155), so index min(qmax - q, band_w - 1) covers every q.  Each wave
    // stages the band rows of its 16 queries in LDS per query tile: 64 LDS lanes per cycle instead of one table address per
    // cycle through the texture addresser (the full-table gather made the kernel 2.6x slower).  0: use exp2d.
    const unsigned* band;
    int band_w;
    // I-BERT softmax (MODE 3): exp_int after the internal QuantAct(16), as the float32 the reference sums and multiplies, for
    // every (row max, q): [256][256], built by ivit_ibert_softmax_build_table
    const float* ib_table;
    float nMs32;  // RQ32 kernels: -Ms as float32 (a power of two: the float32 product with a score accumulator is exact)
    int parts;  // workgroups per (image, head): the query tiles are dealt out among them (small batches: batch * heads << CUs)
    int abl;   // lab build: 1 no score requant, 2 no table lookups, 4 no probability products, 8 no P.V + output, 16 one query tile per wave
};

constexpr int BAND_PAD = 4;   // dwords: keeps slice rows 16-byte aligned and rotates their banks

// K image: 64-byte rows; 16-byte chunk c of row r at slot (c + 2*((r>>2)&1)) & 3.  A 16x16x64 fragment read has
// lanes 0-15 on rows 0-15 chunk 0, lanes 16-31 chunk 1, ...; with this rotation every ds_read_b128 lane group
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) touches 16 distinct 16-byte slots of the 256-byte bank row.
IVIT_DEV int kswz(int r, int c) { return r * HD + (((c + 2 * ((r >> 2) & 1)) & 3) << 4); }

// MODE 0: power-of-two input scale (256-entry table); 1: natural scale, band rows in LDS; 2: natural scale, full-table gather;
// 3 / 4: the I-BERT softmax (ibert_modules.py:237-319) from its (row max, q) table (3: gathered from global memory, 4: band rows
// staged in LDS like mode 1), row sum in torch's float32 reduction order
// PB: width of the softmax output (softmax_bw, vit_quant.py:184): 8, or 16 -- probabilities up to 2^15 as three 7-bit planes
// (p = c + 128 b + 16384 a), one P.V MFMA set per plane (the a plane only when a wave has such a score)
// GENT ("general T", Shiftmax modes only): any token count up to 207 -- every key tile takes the masked path of the last one
// (keys >= T carry the sentinel), all 13 key tiles are still walked: for geometries other than 14 x 14 patches, not tuned.
// RQ32 (MODE 0, PB 8): the requantisation of the scores in float32 -- one v_cvt_f32_i32 + one v_fma_f32 against the magic constant
// instead of v_cvt_f64_i32 + v_fma_f64 (4 + 4 cycles per wave instruction against 3.2 + 2, profiles/r04_valu_price_list.txt) --
// when the multiplier Ms is a power of two (every scale of the power-of-two regime is, and head_dim^-0.5 = 1/8): |S| <= 2^20 is
// exact in float32, S * Ms is exact, the fma rounds once, to nearest even at integer granularity, exactly as the float64 path.
// The scores are then carried with the magic constant's exponent bits in place (RQ_OFF + nk): differences and minima commute
// with the offset, so the table index nk + rmax needs no correction.
constexpr int RQ_OFF = 0x4B400000;
template <int MODE, int PB = 8, int OCC = (PB == 8 ? 4 : 3), bool GENT = false, bool RQ32 = false>
__global__ __launch_bounds__(NT, OCC) void attention_kernel(AttnArgs a)
{
    static_assert(!RQ32 || (MODE == 0 && PB == 8), "RQ32: Shiftmax with a power-of-two input scale, 8-bit probabilities");
    constexpr int SOFF = RQ32 ? RQ_OFF : 0;       // offset the (negated) scores are carried with
    constexpr int PADK = SOFF + 1000;             // sentinel of the padding keys
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    extern __shared__ __attribute__((aligned(16))) unsigned band_lds[];   // [4 waves][16 queries][band_w + BAND_PAD], compat only
    const int T = a.tokens;
    const int bh = blockIdx.x / a.parts, part = blockIdx.x - bh * a.parts;
    const int b = bh / a.heads, hh = bh - b * a.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int64_t plane = (int64_t)a.batch * a.heads * T * HD;
    const int8_t* qg = a.qkv + (int64_t)bh * T * HD;
    const int8_t* kg = qg + plane;
    const int8_t* vg = qg + 2 * plane;

    // ---- Shiftmax exponent table: lut[i] = int_exp_shift(-i), i = kmax - k in [0,255]
    if constexpr (MODE < 3) {
        const unsigned e0 = shiftexp_int(-tid, a.x0, 15);
        reinterpret_cast<unsigned*>(smem + LUT_OFF)[tid] = e0;
        reinterpret_cast<float*>(smem + LUT_OFF)[256 + tid] = (float)e0;   // same address + 1 KB: one more ds_read, one cvt fewer per score
    }

    // ---- K tile [key][64]; rows >= T are never consumed unmasked.  ALL of a thread's K and V chunks are requested before the
    //      first is written to LDS (T <= 208: at most 4 K chunks and one V work item of 4 chunks per thread): one memory latency
    //      per workgroup instead of one per loop iteration (round 4: a staging-only launch took 24 us = 8 us per round of
    //      workgroups, profiles/r04g_*)
    static_assert(KP * 4 <= 4 * NT && (KP / 4) * 4 <= NT, "staging: four K chunks and one V work item per thread");
    v4i kst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = min(tid + NT * i, T * 4 - 1);
        kst[i] = *reinterpret_cast<const v4i*>(kg + (int64_t)(q >> 2) * HD + 16 * (q & 3));
    }
    // ---- V transposed: Vt[d][chunk j = 4s + g'][byte 4t + r] = V[key = 64s + 16t + 4g' + r][d], chunk j of row d
    //      stored at position j ^ (d & 15).  One work item = 4 consecutive keys x 16 d: the 4x4 byte blocks are
    //      transposed in registers (v_perm_b32), so every LDS write is a whole dword (4 keys of one d).
    const bool v_item = tid < ((T + 3) >> 2) * 4;
    v4i vst[4];
    {
        const int kg4 = min(tid, ((T + 3) >> 2) * 4 - 1) >> 2, c = tid & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            vst[r] = *reinterpret_cast<const v4i*>(vg + (int64_t)min(4 * kg4 + r, T - 1) * HD + 16 * c);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + NT * i;
        if (q < T * 4) *reinterpret_cast<v4i*>(smem + kswz(q >> 2, q & 3)) = kst[i];
    }
    if (v_item) {
        const int q = tid;
        const int kg4 = q >> 2, c = q & 3;      // keys 4*kg4 .. 4*kg4+3, d = 16c .. 16c+15
        v4i (&v)[4] = vst;
        const int key0 = 4 * kg4;
        const int j = 4 * (key0 >> 6) + ((key0 >> 2) & 3);
        const int boff = 4 * ((key0 >> 4) & 3);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            // rows = keys r (v[r][w] holds d = 16c+4w .. +3 in its bytes) -> columns: dword bb = 4 keys of d = 16c+4w+bb
            const unsigned a0 = (unsigned)v[0][w], a1 = (unsigned)v[1][w], a2 = (unsigned)v[2][w], a3 = (unsigned)v[3][w];
            const unsigned lo01 = __builtin_amdgcn_perm(a1, a0, 0x05010400u);  // a0.b0 a1.b0 a0.b1 a1.b1
            const unsigned hi01 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);  // a0.b2 a1.b2 a0.b3 a1.b3
            const unsigned lo23 = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
            const unsigned hi23 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
            const unsigned t[4] = {__builtin_amdgcn_perm(lo23, lo01, 0x05040100u),   // d+0: a0.b0 a1.b0 a2.b0 a3.b0
                                   __builtin_amdgcn_perm(lo23, lo01, 0x07060302u),   // d+1
                                   __builtin_amdgcn_perm(hi23, hi01, 0x05040100u),   // d+2
                                   __builtin_amdgcn_perm(hi23, hi01, 0x07060302u)};  // d+3
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int d = 16 * c + 4 * w + bb;
                *reinterpret_cast<unsigned*>(smem + K_BYTES + d * VT_ROW + ((j ^ (d & 15)) << 4) + boff) = t[bb];
            }
        }
    }
    __syncthreads();

    const unsigned* lut = reinterpret_cast<const unsigned*>(smem + LUT_OFF);
    const int nqt = (T + 15) >> 4;

    // 13 query tiles over 4 waves: one wave gets four tiles, the others three.  Which wave that is rotates with the
    // workgroup index, so that the co-resident workgroups of a CU do not all put their extra tile on the same SIMD
    // the Q fragment of a wave's NEXT query tile is requested while the current tile's probabilities are multiplied with V (its
    // registers are free there): the load's latency no longer opens every tile
    const int qt0 = ((wave + bh) & 3) + 4 * part, qt_end = (IVIT_LAB && (a.abl & 16)) ? 4 : nqt;
    v4i qf_next = *reinterpret_cast<const v4i*>(qg + (int64_t)min(qt0 * 16 + l15, T - 1) * HD + 16 * g);
    for (int qt = qt0; qt < qt_end; qt += 4 * a.parts) {
        const int qrow = qt * 16 + l15;  // this lane's query
        const v4i qf = qf_next;

        // ---- S^T = K . Q^T, requantised to the 8-bit Shiftmax input (qact_attn1)
        // scores are kept NEGATED (nk = -k = RNE(S * -Ms): RNE is symmetric): the table index max - k = nk + max is then one
        // v_add_lshl_u32 per score instead of a subtract and a shift
        int s[NKT][4];
        int nmin = PADK;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const v4i kf = *reinterpret_cast<const v4i*>(smem + kswz(16 * kt + l15, g));
            v4i acc = {0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf, qf, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // |S| <= 64*128*128 = 2^20, m < 2^32: the product is exact in float64
                int nk;
                if constexpr (RQ32) {
                    const int tb = __float_as_int(__builtin_fmaf((float)acc[r], a.nMs32, 12582912.0f));     // RQ_OFF + RNE(-S * Ms)
                    nk = (IVIT_LAB && (a.abl & 1)) ? SOFF + (acc[r] & 127) : clamp_i32(tb, SOFF - 127, SOFF + 128);
                } else {
                    nk = (IVIT_LAB && (a.abl & 1)) ? (acc[r] & 127) : clamp_i32(requant_exact(acc[r], -a.Ms), -127, 128);
                }
                if (GENT || kt == NKT - 1) nk = (16 * kt + 4 * g + r < T) ? nk : PADK;
                s[kt][r] = nk;
                nmin = min(nmin, nk);
            }
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most 4 K fragments in flight (registers)
        }
        nmin = rows_allmin_i32(nmin);      // over the four lanes of a query (common.h: permlane swaps, no LDS round trip)
        int rmax = -nmin;                // RQ32: -(RQ_OFF + nmin), so that nk + rmax is the plain difference
        asm volatile("" : "+v"(rmax));   // opaque: keeps (nk + rmax) << 2 one v_add_lshl_u32 instead of a subtract and a shift

        // ---- Shiftmax (ivit_modules.py:164-175): e = exp_int(k - max), sum, factor, e*factor >> 24
        // k - max is in [-255, 0] for every real key: the 256-entry table covers it without a clamp (the entries from
        // ksat on are identical anyway); only the padding keys of the last tile carry the -1000 sentinel
        unsigned esum = 0;
        if constexpr (MODE == 1) {     // natural input scale, band rows of this tile's 16 queries staged in LDS
            const int W = a.band_w, stride = W + BAND_PAD;
            unsigned* slice = band_lds + (wave * 16 + l15) * stride;
            {
                const uint4* src = reinterpret_cast<const uint4*>(a.band + (size_t)(rmax + 128) * W) + g * (W >> 4);
                uint4* dst = reinterpret_cast<uint4*>(slice) + g * (W >> 4);
                for (int i = 0; i < (W >> 4); ++i) dst[i] = src[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nk = s[kt][r];
                    unsigned e = slice[min(nk + rmax, W - 1)];     // the 1000 sentinel of the padding keys clamps, too
                    if (GENT || kt == NKT - 1) e = (nk == 1000) ? 0u : e;
                    s[kt][r] = (int)e;
                    esum += e;
                }
            __builtin_amdgcn_wave_barrier();    // every lane has read its slice before the next tile overwrites it
        } else if constexpr (MODE >= 3) {
            // IBERTIntSoftmax: e = table[row max][q] (float32), S = e.sum() in float32 IN TORCH'S ORDER (ATen SumKernel, rowsum.h:
            // for 192 <= T < 208 the 32 partials p = key % 32 take their six elements key = p, p + 32, .. in turn, vectors of
            // keys 192.. join partials 0..7, the scalar tail goes first into the final accumulator, then the eight lane sums
            // ((P[l] + P[l+8]) + P[l+16]) + P[l+24] are added left to right).  With key = 16 kt + 4 g + r a lane holds the whole
            // sequence of its eight partials (hi = kt & 1, r): they are summed in registers, exchanged among the four lanes of
            // a query and combined by every lane alike.  The values are integers for a power-of-two range of the internal
            // QuantAct and fl(fl(k * s) / s) otherwise: the order matters then.
            const float* row2d = a.ib_table + ((rmax + 128) << 8) + 128;       // mode 3: entry of q = -nk
            const int W = a.band_w, stride = W + BAND_PAD;
            const unsigned* slice = band_lds + (wave * 16 + l15) * stride;     // mode 4: this query's band row
            if constexpr (MODE == 4) {
                const uint4* src = reinterpret_cast<const uint4*>(a.band + (size_t)(rmax + 128) * W) + g * (W >> 4);
                uint4* dst = reinterpret_cast<uint4*>(band_lds + (wave * 16 + l15) * stride) + g * (W >> 4);
                for (int i = 0; i < (W >> 4); ++i) dst[i] = src[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            float e12[4];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nk = s[kt][r];
                    float e;
                    if constexpr (MODE == 4) e = __int_as_float((int)slice[min(nk + rmax, W - 1)]);   // the 1000 sentinel clamps, too
                    else e = row2d[-min(nk, 128)];
                    if (kt == NKT - 1) {
                        e = (nk == 1000) ? 0.0f : e;
                        e12[r] = e;
                    }
                    s[kt][r] = __float_as_int(e);
                }
            if constexpr (MODE == 4) __builtin_amdgcn_wave_barrier();    // every lane has read its slice before the next tile overwrites it
            const int nv = T >> 3;                 // 8-float vectors: 24 or 25 (the launcher keeps T < 208)
            float Pp[2][4];
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float acc = 0.0f;
#pragma unroll
                    for (int m = 0; m < 6; ++m) acc += __int_as_float(s[2 * m + hi][r]);
                    Pp[hi][r] = acc;
                }
            if (nv > 24 && g < 2) {                // vector 24 = keys 192 .. 199 -> partials 0 .. 7 (lanes g = 0, 1; hi = 0)
#pragma unroll
                for (int r = 0; r < 4; ++r) Pp[0][r] += e12[r];
            }
            // all-gather over the four lanes of a query (lane = 16 g + l15) in VALU instructions: v_permlane32_swap of a value
            // with itself leaves every lane with the values of rows {0,1} and {2,3} of 16 lanes, v_permlane16_swap of each of
            // those with itself then separates row 0 / 1 and row 2 / 3: out[g'] = the value lane 16 g' + l15 held
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            auto gather4 = [&](float x, float (&out)[4]) {
                const unsigned xb = (unsigned)__float_as_int(x);
                const v2u h = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);       // h.x: rows 0,1,0,1; h.y: rows 2,3,2,3
                const v2u a01 = __builtin_amdgcn_permlane16_swap(h.x, h.x, false, false);   // .x: row 0 everywhere, .y: row 1
                const v2u a23 = __builtin_amdgcn_permlane16_swap(h.y, h.y, false, false);   // .x: row 2, .y: row 3
                out[0] = __int_as_float((int)a01.x);
                out[1] = __int_as_float((int)a01.y);
                out[2] = __int_as_float((int)a23.x);
                out[3] = __int_as_float((int)a23.y);
            };
            float fin = 0.0f;
            {
                float t12[4][4];                   // [r][g']
#pragma unroll
                for (int r = 0; r < 4; ++r) gather4(e12[r], t12[r]);
#pragma unroll
                for (int gp = 0; gp < 4; ++gp)     // scalar tail: keys 8 nv .. T - 1 in order
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 192 + 4 * gp + r;
                        if (key >= 8 * nv && key < T) fin += t12[r][gp];
                    }
            }
            float A[4][2][4];
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t[4];
                    gather4(Pp[hi][r], t);
#pragma unroll
                    for (int gp = 0; gp < 4; ++gp) A[gp][hi][r] = t[gp];
                }
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const int gm = l >> 2, r = l & 3;
                const float v = ((A[gm][0][r] + A[2 + gm][0][r]) + A[gm][1][r]) + A[2 + gm][1][r];
                fin += v;
            }
            esum = (unsigned)__float_as_int(fin);   // carried as bits to the common code below
        } else if constexpr (MODE == 2) {      // one L2-resident gather per score instead of the LDS lookup
            const unsigned* row2d = a.exp2d + ((rmax + 128) << 8) + 128;       // entry of q = -nk
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nk = s[kt][r];
                    unsigned e = row2d[-min(nk, 128)];
                    if (GENT || kt == NKT - 1) e = (nk == 1000) ? 0u : e;
                    s[kt][r] = (int)e;
                    esum += e;
                }
        } else
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned e, ef;
                if (GENT || kt == NKT - 1) {
                    const int idx = min(rmax + s[kt][r], 255);
                    e = lut[idx];
                    ef = lut[256 + idx];
                    const bool pad = s[kt][r] == PADK;
                    e = pad ? 0u : e;
                    ef = pad ? 0u : ef;
                } else if (IVIT_LAB && (a.abl & 2)) {
                    e = (unsigned)s[kt][r];
                    ef = (unsigned)__float_as_int(1.0f);
                } else {
                    e = lut[rmax + s[kt][r]];
                    ef = lut[256 + rmax + s[kt][r]];
                }
                s[kt][r] = (int)ef;       // float32 bit pattern of the exponent
                esum += e;
            }
        constexpr bool ef_is_float = MODE == 0 || MODE >= 3;
        float factor;
        if constexpr (MODE >= 3) {
            factor = floorf(4294967296.0f / __int_as_float((int)esum));        // ibert_modules.py:313
        } else {
            esum = rows_allsum_u32(esum);
            float S = (float)esum;                                     // exp_int.sum (:171)
            S = fminf(S, 2147483648.0f);                               // clamp_max_(2**31-1) in float32 (:173)
            factor = floorf((1.0f / S) * 2147483648.0f);               // (:174)
        }

        // packed probabilities: dword t of key step ks = bytes r = 0..3 of key tile 4ks + t
        v4i pk[NKS];
        v4i pkb[PB == 16 ? NKS : 1];       // 16-bit probabilities: the second 7-bit plane
        v4i pkh[(MODE >= 3 || PB == 16) ? NKS : 1];      // I-BERT: probabilities reach 128 (a one-hot row): 128 = 127 + 1, the 1 in a second operand
        // I-BERT: p = floor(fl32(e * factor) / 2^25) in [0, 128] (ibert_modules.py:314, output_bit = 8).  factor / 2 is an exact
        // scaling, so u = trunc(e * (factor / 2)) has p in its top byte like the Shiftmax product below -- except p = 128, which
        // shows as the sign bit of u: the OR of all u of the tile is tested once and the tile repacked in that rare case.
        const float factor_h = factor * 0.5f;
        unsigned any_u = 0;
        constexpr bool SDWA_PACK = MODE < 3 && PB == 8;
        if constexpr (SDWA_PACK) {
            // p = floor(fl32(e * factor) / 2^24) (:175) = trunc(fl32(e * (factor * 2^-24))): the scaling by a power of two commutes
            // with the float32 rounding of the product (no operand or result is subnormal: factor >= 1, e >= 1 or e == 0), and
            // p < 128.  v_cvt_u32_f32 with SDWA destination select writes the truncated value straight into byte r of the packed
            // dword (UNUSED_PRESERVE keeps the other bytes; semantics checked by scripts/probes/cvt_pack_probe.hip): one
            // instruction per score instead of a conversion plus 3/4 of a byte permute / OR.  The four dwords of a key step are
            // written round-robin, so that no instruction reads the register the previous one wrote with a destination select
            // (gfx940+ dst_sel forwarding hazard: one wait state, which the compiler cannot insert inside inline asm).
            const float factor24 = factor * 5.9604644775390625e-08f;     // 2^-24, exact
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                float pf[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kt2 = 4 * ks + t < NKT ? 4 * ks + t : 0;
                        const float ev = ef_is_float ? __int_as_float(s[kt2][r]) : (float)(unsigned)s[kt2][r];
                        pf[t][r] = (IVIT_LAB && (a.abl & 4)) ? 1.0f : ev * factor24;     // lab bit 2: no products
                    }
                unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
                if (4 * ks + 3 < NKT) {
                    // (the first write of each dword zero-fills its other bytes: UNUSED_PAD, early-clobber outputs, no initialisation)
                    asm("v_cvt_u32_f32_sdwa %0, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %1, %5 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %2, %6 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %3, %7 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %0, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %1, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %2, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %3, %11 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %0, %12 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %1, %13 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %2, %14 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %3, %15 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %0, %16 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %1, %17 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %2, %18 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "v_cvt_u32_f32_sdwa %3, %19 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                        "s_nop 0"
                        : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                        : "v"(pf[0][0]), "v"(pf[1][0]), "v"(pf[2][0]), "v"(pf[3][0]), "v"(pf[0][1]), "v"(pf[1][1]), "v"(pf[2][1]), "v"(pf[3][1]),
                          "v"(pf[0][2]), "v"(pf[1][2]), "v"(pf[2][2]), "v"(pf[3][2]), "v"(pf[0][3]), "v"(pf[1][3]), "v"(pf[2][3]), "v"(pf[3][3]));
                } else if (4 * ks < NKT) {      // the last key step: one key tile (13 = 3 x 4 + 1), a chain on one register
                    static_assert(NKT % 4 == 1, "the tail key step packs exactly one key tile");
                    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\ts_nop 0\n\t"
                        "v_cvt_u32_f32_sdwa %0, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 0\n\t"
                        "v_cvt_u32_f32_sdwa %0, %3 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 0\n\t"
                        "v_cvt_u32_f32_sdwa %0, %4 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 0"
                        : "=&v"(w0)
                        : "v"(pf[0][0]), "v"(pf[0][1]), "v"(pf[0][2]), "v"(pf[0][3]));
                }
                pk[ks] = v4i{(int)w0, (int)w1, (int)w2, (int)w3};
            }
        } else
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                unsigned w = 0;
                if constexpr (MODE >= 3 || PB == 16) pkh[ks][t] = 0;
                if constexpr (PB == 16) pkb[ks][t] = 0;
                if (4 * ks + t < NKT) {
                    unsigned p[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float ev = ef_is_float ? __int_as_float(s[4 * ks + t][r]) : (float)(unsigned)s[4 * ks + t][r];
                        if constexpr (MODE >= 3) {
                            p[r] = (unsigned)(ev * factor_h);
                            any_u |= p[r];
                        } else if (IVIT_LAB && (a.abl & 4)) {
                            p[r] = (unsigned)s[4 * ks + t][r];
                        } else {
                            p[r] = (unsigned)(ev * factor);  // float32 product (:175), < 2^31
                        }
                    }
                    if constexpr (PB == 16) {
                        // p16 = u >> 16 = floor(fl32(e * factor) / 2^16) (Shiftmax :175) resp. / 2^17 (I-BERT :314) with u as above:
                        // plane c = bits 16..22 (byte 2 & 0x7f), plane b = bits 23..29 (byte 3 of u << 1, & 0x7f), plane a = bits 30, 31
                        if constexpr (MODE < 3) any_u |= p[0] | p[1] | p[2] | p[3];
                        const unsigned c4 = __builtin_amdgcn_perm(p[1], p[0], 0x0c0c0602u) | __builtin_amdgcn_perm(p[3], p[2], 0x06020c0cu);
                        const unsigned b4 = __builtin_amdgcn_perm(p[1] << 1, p[0] << 1, 0x0c0c0703u) |
                                            __builtin_amdgcn_perm(p[3] << 1, p[2] << 1, 0x07030c0cu);
                        w = c4 & 0x7f7f7f7fu;
                        pkb[ks][t] = (int)(b4 & 0x7f7f7f7fu);
                    } else {
                    // floor(. / 2^24) = the top byte of each product: gather the four top bytes with two byte permutes
                    const unsigned lo = __builtin_amdgcn_perm(p[1], p[0], 0x0c0c0703u);  // [p0.b3, p1.b3, 0, 0]
                    const unsigned hi = __builtin_amdgcn_perm(p[3], p[2], 0x07030c0cu);  // [0, 0, p2.b3, p3.b3]
                    w = lo | hi;
                    }
                }
                pk[ks][t] = (int)w;
            }
        const bool any_hi = (MODE >= 3 || PB == 16) &&
                            __builtin_amdgcn_ballot_w64((any_u >> (PB == 16 ? 30 : 31)) != 0) != 0;   // wave-uniform, almost never
        if constexpr (PB == 16) {
            if (any_hi) {      // plane a (bits 30, 31 of u: p16 >= 16384) from the products again
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        unsigned w = 0;
                        if (4 * ks + t < NKT) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float ev = ef_is_float ? __int_as_float(s[4 * ks + t][r]) : (float)(unsigned)s[4 * ks + t][r];
                                const unsigned u = (unsigned)(ev * (MODE >= 3 ? factor_h : factor));
                                w |= (u >> 30) << (8 * r);
                            }
                        }
                        pkh[ks][t] = (int)w;
                    }
            }
        } else if constexpr (MODE >= 3) {
            if (any_hi) {      // a byte 0x80 (p = 128) becomes 127 in pk and 1 in pkh
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const unsigned w = (unsigned)pk[ks][t];
                        const unsigned h128 = (w >> 7) & 0x01010101u;          // 1 in every byte that is 0x80
                        pk[ks][t] = (int)(w - h128);                          // 0x80 -> 0x7f (no borrow: the byte is >= 1)
                        pkh[ks][t] = (int)h128;
                    }
            }
        }

        const bool hi_pass = any_hi;
        qf_next = *reinterpret_cast<const v4i*>(qg + (int64_t)min((qt + 4 * a.parts) * 16 + l15, T - 1) * HD + 16 * g);
        // ---- O^T = Vt . P^T, requantised (attn.qact2), 4 consecutive d per dword
        const int64_t orow_idx = (int64_t)b * T + qrow;
        const BlockRow obrow = block_row((int)orow_idx, a.heads * HD);
        int8_t* orow = a.out + orow_idx * ((int64_t)a.heads * HD) + hh * HD;
        if (IVIT_LAB && (a.abl & 8)) {
            if (pk[0][0] == 0x12345678) a.out[0] = 1;
            continue;
        }
        unsigned wq[4];    // wq[dt]: bytes d = 16 dt + 4 g + 0..3 of this lane's query
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            v4i acc = {0, 0, 0, 0};
            v4i accb = {0, 0, 0, 0}, acca = {0, 0, 0, 0};
            const int d = 16 * dt + l15;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const v4i vf = *reinterpret_cast<const v4i*>(smem + K_BYTES + d * VT_ROW + (((4 * ks + g) ^ (d & 15)) << 4));
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf, pk[ks], acc, 0, 0, 0);
                if constexpr (PB == 16) {
                    accb = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf, pkb[ks], accb, 0, 0, 0);
                    if (hi_pass) acca = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf, pkh[ks], acca, 0, 0, 0);
                } else if constexpr (MODE >= 3) {
                    if (hi_pass) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf, pkh[ks], acc, 0, 0, 0);
                }
            }
            int ob[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int o;
                if constexpr (PB == 16) {
                    // O = sum p16 * v up to 208 * 32768 * 128 < 2^30: the reference's float64 product rounds at 53 bits first
                    // (quant_utils.py:229-230), so product and rounding are two steps here
                    const int O = acc[r] + (accb[r] << 7) + (acca[r] << 14);
                    const double t = (double)O * a.Mo + IVIT_MAGIC;
                    o = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                } else {
                    // |O| <= 208*127*128 < 2^22: exact float64 product
                    o = clamp_i32(requant_exact(acc[r], a.Mo), -128, 127);
                }
                ob[r] = o;
            }
            // the four low bytes by two byte permutes and an OR (was: and / shift / or per byte)
            wq[dt] = __builtin_amdgcn_perm((unsigned)ob[1], (unsigned)ob[0], 0x0c0c0400u) | __builtin_amdgcn_perm((unsigned)ob[3], (unsigned)ob[2], 0x04000c0cu);
        }
        // A query's 64 output bytes sit as 4 x 4 dwords in its four lanes (g = lane >> 4).  A 4 x 4 word transpose across those
        // lanes -- two v_permlane32_swap, two v_permlane16_swap -- leaves lane g with the 16 CONTIGUOUS bytes d = 16 g .. 16 g + 15:
        // one 16-byte store per lane instead of four 4-byte ones (a quarter of the store instructions and of the segments the
        // memory pipeline has to merge).
        {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const v2u ab = __builtin_amdgcn_permlane32_swap(wq[0], wq[2], false, false);    // g < 2: (w0[g], w0[g+2]); g >= 2: (w2[g-2], w2[g])
            const v2u cd = __builtin_amdgcn_permlane32_swap(wq[1], wq[3], false, false);    // g < 2: (w1[g], w1[g+2]); g >= 2: (w3[g-2], w3[g])
            const v2u ac = __builtin_amdgcn_permlane16_swap(ab.x, cd.x, false, false);      // (w_g[0], w_g[1]) of the lanes 0, 1
            const v2u bd = __builtin_amdgcn_permlane16_swap(ab.y, cd.y, false, false);      // (w_g[2], w_g[3])
            if (qrow < T) {
                const v4i chunk = {(int)ac.x, (int)ac.y, (int)bd.x, (int)bd.y};
                if (a.out_blocks)   // column = 64 hh + 16 g: chunk index g, column block hh
                    *reinterpret_cast<v4i*>(a.out + obrow.base + (unsigned)hh * 1024u + (((unsigned)g ^ obrow.rs) << 4)) = chunk;
                else
                    *reinterpret_cast<v4i*>(orow + 16 * g) = chunk;
            }
        }
    }
}

// Workgroups per (image, head).  Every workgroup stages K and V^T of its head in LDS (~3 us) and walks query tiles (13 at
// T = 197: 4 per wave, ~3.5 us each); `slots` workgroups are resident at once.  With batch * heads a multiple of the slots one
// workgroup per head is right (DeiT-B b256: 3072 = 3 rounds of 1024; 4 of 768 before round 3); a launch that fills only part of a round is bound by that
// walk -- DeiT-S b64 (384 heads) took a whole round's 17 us -- so the tiles are dealt out among 2 or 4 workgroups per head when
// the model below says the launch gets shorter (each stages K / V^T itself: L2 hits).
int attention_parts(int batch_heads, int tokens, int slots)
{
    const int nqt = (tokens + 15) >> 4;
#if IVIT_LAB
    if ((g_ln_ablate >> 28) & 7) return (g_ln_ablate >> 28) & 7;      // lab: forced (scripts/attn_parts.py)
#endif
    int best = 1;
    double best_t = 0.0;
    for (int p = 1; p <= 4; p *= 2) {
        if (p > 1 && 4 * (p / 2) >= nqt) break;     // no wave would lose a tile
        const int rounds = (batch_heads * p + slots - 1) / slots, tiles = (nqt + 4 * p - 1) / (4 * p);
        const double t = rounds * (3.0 + 3.45 * tiles);
        if (p == 1 || t < 0.95 * best_t) { best = p; best_t = t; }
    }
    return best;
}

}  // namespace

IVIT_EXPORT int ivit_attention_fused_i8_ex(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens,
                                        int head_dim, uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o,
                                        int32_t e_o, int out_blocks, ivit_stream_t stream)
{
    return ivit_attention_fused_i8_compat(qkv, out, batch, heads, tokens, head_dim, m_s, e_s, s_attn, m_o, e_o, nullptr,
                                          out_blocks, stream);
}

IVIT_EXPORT int ivit_attention_fused_i8_compat(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens,
                                               int head_dim, uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o,
                                               int32_t e_o, const uint32_t* exp2d, int out_blocks, ivit_stream_t stream)
{
    return ivit_attention_fused_i8_compat_band(qkv, out, batch, heads, tokens, head_dim, m_s, e_s, s_attn, m_o, e_o, exp2d,
                                               nullptr, 0, out_blocks, stream);
}

IVIT_EXPORT int ivit_attention_fused_i8_compat_band(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens,
                                                    int head_dim, uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o,
                                                    int32_t e_o, const uint32_t* exp2d, const uint32_t* band, int band_w,
                                                    int out_blocks, ivit_stream_t stream)
{
    return ivit_attention_fused_i8_wide(qkv, out, batch, heads, tokens, head_dim, m_s, e_s, s_attn, m_o, e_o, exp2d, band, band_w, 8,
                                        out_blocks, stream);
}

IVIT_EXPORT int ivit_attention_fused_i8_wide(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim,
                                             uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o, int32_t e_o,
                                             const uint32_t* exp2d, const uint32_t* band, int band_w, int softmax_bits,
                                             int out_blocks, ivit_stream_t stream)
{
    IVIT_REQUIRE(softmax_bits == 8 || softmax_bits == 16, "ivit_attention_fused_i8_wide: softmax_bits must be 8 or 16");
    IVIT_REQUIRE(qkv && out, "ivit_attention_fused_i8: NULL operand");
    IVIT_REQUIRE(batch > 0 && heads > 0, "ivit_attention_fused_i8: empty batch");
    if (head_dim != HD || tokens < 1 || tokens > KP) {
        ivit_set_error("ivit_attention_fused_i8: unsupported geometry head_dim=%d tokens=%d (need 64, 1..208)",
                       head_dim, tokens);
        return IVIT_ERR_UNSUPPORTED;
    }
    IVIT_REQUIRE(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((heads * head_dim) % 16 == 0),
                 "ivit_attention_fused_i8: misaligned operand (16-byte rows)");
    IVIT_REQUIRE(s_attn > 0.0f, "ivit_attention_fused_i8: scale must be positive");
    IVIT_REQUIRE(out_blocks == 0 || (out_blocks == 1 && ((uintptr_t)out % 16 == 0) &&
                                     ((int64_t)batch * tokens + 15) * heads * head_dim < 2147483648ll),
                 "ivit_attention_fused_i8_ex: bad output layout (block-layout buffers stay below 2 GiB)");
    IVIT_REQUIRE((uintptr_t)exp2d % 4 == 0, "ivit_attention_fused_i8_compat: misaligned exponent table");
    IVIT_REQUIRE(band_w == 0 || (band && band_w >= 16 && band_w <= 256 && band_w % 16 == 0 && (uintptr_t)band % 16 == 0),
                 "ivit_attention_fused_i8_compat_band: band table must be 16-byte aligned, width a multiple of 16 in [16, 256]");
    AttnArgs a{};
    a.abl = (g_ln_ablate >> 20) & 31;
    a.exp2d = exp2d;
    a.band = band;
    a.band_w = band_w;
    a.out_blocks = out_blocks;
    a.qkv = qkv; a.out = out; a.batch = batch; a.heads = heads; a.tokens = tokens;
    a.Ms = ivit_dyadic_to_double(m_s, e_s);
    a.Mo = ivit_dyadic_to_double(m_o, e_o);
    // a power-of-two score multiplier (m = 2^k): the float32 requantisation of the RQ32 kernels is exact (lab bit 26: off, A/B)
    const bool ms_pow2 = m_s != 0 && (m_s & (m_s - 1)) == 0 && a.Ms >= 1e-30 && !(IVIT_LAB && (g_ln_ablate & (1 << 26)));
    a.nMs32 = (float)-a.Ms;
    const float x0f = __builtin_floorf((1.0f / s_attn) * -1.0f);  // ivit_modules.py:154
    IVIT_REQUIRE(x0f <= -1.0f && x0f >= -4096.0f, "ivit_attention_fused_i8: x0=%g outside [-4096,-1]", (double)x0f);
    a.x0 = (int)x0f;
    // exact u32 row sum: tokens * |x0| * 2^15 must stay below 2^32
    IVIT_REQUIRE((double)tokens * (double)(-a.x0) * 32768.0 < 4294967296.0,
                 "ivit_attention_fused_i8: Shiftmax row sum could overflow 32 bits (x0=%d)", a.x0);
    IVIT_REQUIRE(a.Ms < 2048.0 && a.Mo < 512.0, "ivit_attention_fused_i8: requant multiplier too large");
    a.ksat = 255;
    for (int i = 0; i < 256; ++i) {
        const int d = -i;
        const int x = d + (d >> 1) - (d >> 4);  // ivit_modules.py:151 (arithmetic shifts = floor)
        if (x <= 15 * a.x0) { a.ksat = i; break; }
    }
    const size_t band_lds_bytes = band_w ? (size_t)4 * 16 * (band_w + BAND_PAD) * sizeof(unsigned) : 0;
    {   // resident workgroups: 4 per CU for 8-bit probabilities (128 VGPRs), 3 for the 16-bit planes; the band rows add dynamic LDS
        const int by_regs = softmax_bits == 16 ? 3 : 4, by_lds = (int)(163840 / (SMEM_BYTES + 512 + band_lds_bytes));
        a.parts = attention_parts(batch * heads, tokens, 256 * (by_regs < by_lds ? by_regs : by_lds));
    }
    const dim3 grid(batch * heads * a.parts), blk(NT);
    hipStream_t st = ivit_stream(stream);
    if (tokens <= 16 * (NKT - 1)) {      // fewer than 193 tokens: the general-T form
        if (softmax_bits == 16) {
            if (band_w) hipLaunchKernelGGL((attention_kernel<1, 16, 3, true>), grid, blk, band_lds_bytes, st, a);
            else if (exp2d) hipLaunchKernelGGL((attention_kernel<2, 16, 3, true>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((attention_kernel<0, 16, 3, true>), grid, blk, 0, st, a);
        } else {
            if (band_w) hipLaunchKernelGGL((attention_kernel<1, 8, 4, true>), grid, blk, band_lds_bytes, st, a);
            else if (exp2d) hipLaunchKernelGGL((attention_kernel<2, 8, 4, true>), grid, blk, 0, st, a);
            else if (ms_pow2) hipLaunchKernelGGL((attention_kernel<0, 8, 4, true, true>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((attention_kernel<0, 8, 4, true>), grid, blk, 0, st, a);
        }
        IVIT_CHECK_LAUNCH("ivit_attention_fused_i8");
    }
    if (softmax_bits == 16) {
        if (band_w) hipLaunchKernelGGL((attention_kernel<1, 16>), grid, blk, band_lds_bytes, st, a);
        else if (exp2d) hipLaunchKernelGGL((attention_kernel<2, 16>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((attention_kernel<0, 16>), grid, blk, 0, st, a);
    } else {
        if (band_w) hipLaunchKernelGGL(attention_kernel<1>, grid, blk, band_lds_bytes, st, a);
        else if (exp2d) hipLaunchKernelGGL(attention_kernel<2>, grid, blk, 0, st, a);
        else if (IVIT_LAB && (g_ln_ablate & (1 << 25))) hipLaunchKernelGGL((attention_kernel<0, 8, 3>), grid, blk, 0, st, a);   // lab A/B: three workgroups per CU as before round 3
        else if (ms_pow2) hipLaunchKernelGGL((attention_kernel<0, 8, 4, false, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL(attention_kernel<0>, grid, blk, 0, st, a);
    }
    IVIT_CHECK_LAUNCH("ivit_attention_fused_i8");
}

IVIT_EXPORT int ivit_attention_fused_i8_ibert(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim,
                                              uint32_t m_s, int32_t e_s, uint32_t m_o, int32_t e_o, const float* table,
                                              const float* band, int band_w, int out_blocks, ivit_stream_t stream)
{
    return ivit_attention_fused_i8_ibert_wide(qkv, out, batch, heads, tokens, head_dim, m_s, e_s, m_o, e_o, table, band, band_w, 8,
                                              out_blocks, stream);
}

IVIT_EXPORT int ivit_attention_fused_i8_ibert_wide(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens, int head_dim,
                                                   uint32_t m_s, int32_t e_s, uint32_t m_o, int32_t e_o, const float* table,
                                                   const float* band, int band_w, int softmax_bits, int out_blocks,
                                                   ivit_stream_t stream)
{
    IVIT_REQUIRE(softmax_bits == 8 || softmax_bits == 16, "ivit_attention_fused_i8_ibert_wide: softmax_bits must be 8 or 16");
    IVIT_REQUIRE(qkv && out && table && batch > 0 && heads > 0, "ivit_attention_fused_i8_ibert: bad operand");
    IVIT_REQUIRE(band_w == 0 || (band && band_w >= 16 && band_w <= 256 && band_w % 16 == 0 && (uintptr_t)band % 16 == 0),
                 "ivit_attention_fused_i8_ibert: band table must be 16-byte aligned, width a multiple of 16 in [16, 256]");
    if (head_dim != HD || tokens <= 16 * (NKT - 1) || tokens >= KP) {
        ivit_set_error("ivit_attention_fused_i8_ibert: unsupported geometry head_dim=%d tokens=%d (need 64, 193..207)", head_dim, tokens);
        return IVIT_ERR_UNSUPPORTED;
    }
    IVIT_REQUIRE(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)table % 4 == 0),
                 "ivit_attention_fused_i8_ibert: misaligned operand");
    IVIT_REQUIRE(out_blocks == 0 || (out_blocks == 1 && ((uintptr_t)out % 16 == 0) &&
                                     ((int64_t)batch * tokens + 15) * heads * head_dim < 2147483648ll),
                 "ivit_attention_fused_i8_ibert: bad output layout (block-layout buffers stay below 2 GiB)");
    AttnArgs a{};
    a.ib_table = table;
    a.out_blocks = out_blocks;
    a.qkv = qkv; a.out = out; a.batch = batch; a.heads = heads; a.tokens = tokens;
    a.Ms = ivit_dyadic_to_double(m_s, e_s);
    a.Mo = ivit_dyadic_to_double(m_o, e_o);
    IVIT_REQUIRE(a.Ms < 2048.0 && a.Mo < 512.0, "ivit_attention_fused_i8_ibert: requant multiplier too large");
    a.band = reinterpret_cast<const unsigned*>(band);
    a.band_w = band_w;
    const size_t band_lds_bytes = band_w ? (size_t)4 * 16 * (band_w + BAND_PAD) * sizeof(unsigned) : 0;
    {
        const int by_regs = softmax_bits == 16 ? 3 : 4, by_lds = (int)(163840 / (SMEM_BYTES + 512 + band_lds_bytes));
        a.parts = attention_parts(batch * heads, tokens, 256 * (by_regs < by_lds ? by_regs : by_lds));
    }
    if (softmax_bits == 16) {
        if (band_w) hipLaunchKernelGGL((attention_kernel<4, 16>), dim3(batch * heads * a.parts), dim3(NT), band_lds_bytes, ivit_stream(stream), a);
        else hipLaunchKernelGGL((attention_kernel<3, 16>), dim3(batch * heads * a.parts), dim3(NT), 0, ivit_stream(stream), a);
    } else {
        if (band_w) hipLaunchKernelGGL(attention_kernel<4>, dim3(batch * heads * a.parts), dim3(NT), band_lds_bytes, ivit_stream(stream), a);
        else hipLaunchKernelGGL(attention_kernel<3>, dim3(batch * heads * a.parts), dim3(NT), 0, ivit_stream(stream), a);
    }
    IVIT_CHECK_LAUNCH("ivit_attention_fused_i8_ibert");
}

IVIT_EXPORT int ivit_attention_fused_i8(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens,
                                        int head_dim, uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o,
                                        int32_t e_o, ivit_stream_t stream)
{
    return ivit_attention_fused_i8_ex(qkv, out, batch, heads, tokens, head_dim, m_s, e_s, s_attn, m_o, e_o, 0, stream);
}
