// rowops.hip -- row-wise integer kernels: I-LayerNorm, ShiftGELU, stand-alone Shiftmax, and the
// small element-wise pieces around them.  All are one-wavefront-per-row (64 lanes), reductions by
// cross-lane exchange, HBM-bound by design (1 byte in + 1 byte out per element).
// Reference: /root/reference/models/quantization_utils/ivit_modules.py (IVITIntLayerNorm :30-65,
// IVITIntGELU :89-126, IVITIntSoftmax :150-179), quant_utils.py (fixedpoint_mul :193-253,
// SymmetricQuantFunction :79-97).
//
// Float32 / float64 instructions appear below ONLY where the reference's own float emulation
// rounds (24-bit products, correctly rounded quotients, the float64 requant product); each such
// site is a single IEEE operation on exactly representable operands, compiled with
// -ffp-contract=off, and is listed in DESIGN.md.
#include <limits.h>

#include "common.h"
#include "rowsum.h"

// tests / A-B timing: 1 = always the wave-per-row LayerNorm kernel, 2 = the half-wave-per-row one wherever it exists
// (ivit_debug_ln_wave_per_row)
#if IVIT_LAB
int g_ln_wave_per_row = 0;
int g_ln_ablate = 0;     // lab build only (ivit_debug_ln_ablate): 1 no element chain, 2 no statistics, 4 no stores, 8 no table build
int g_ln_stream_cfg = 0; // lab build only (ivit_debug_ln_stream_cfg): ring depth / workgroups per CU of the streaming kernel
unsigned long long* g_ln_stamps = nullptr;   // lab build only (ivit_debug_ln_stamp_buffer): 8 x s_memrealtime per wave of the streaming kernel
#else
constexpr int g_ln_wave_per_row = 0;
constexpr int g_ln_ablate = 0;
constexpr int g_ln_stream_cfg = 0;
#endif

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int WPB = NT / 64;  // waves (rows in flight) per block

IVIT_DEV int sx8(int v, int byte) { return (int)(int8_t)(v >> (8 * byte)); }

IVIT_DEV int pack4(int a, int b, int c, int d)
{
    return (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | ((d & 0xff) << 24);
}

static inline int grid_for_rows(int64_t rows, int rows_per_wave = 1)
{
    int64_t blocks = (rows + WPB * rows_per_wave - 1) / (WPB * rows_per_wave);
    return (int)(blocks < 4096 ? blocks : 4096);
}

// ------------------------------------------------------------------------------------------------
// I-LayerNorm
// ------------------------------------------------------------------------------------------------
struct LnArgs {
    const void* x;
    int64_t ldx;
    int rows, C;
    const float* bias_int;
    const float* s_ln;
    const uint32_t* m;
    const int32_t* e;
    void* out;
    int64_t ldo;
    int out_blocks;   // int8 output in the GEMM block layout (common.h: ivit_block_offset), row length C
    // natural-scale ("compat") inputs, ivit_layernorm_i8_compat: the reference's LayerNorm does not see the integer q its
    // input QuantAct produced but phi(q) = fl(fl(q*s)/s) (quant_modules.py:387, ivit_modules.py:36)
    const int8_t* remap;   // [256] k'(q) = trunc(phi(q)), indexed q + 128   (ivit_modules.py:38 .to(int32))
    const float* phi;      // [256] phi(q), indexed q + 128                  (the mean, :37, is taken over these)
    int abl;               // lab build: timing ablations (results wrong), always 0 in the product
    int outer;             // compat, > 0: the reference reduces over a transposed view of contiguous extent `outer` (row = .. * outer +
                           // column) -- torch's outer-reduction order
#if IVIT_LAB
    unsigned long long* stamps;   // lab build: wave timeline of the streaming kernel (nullptr = none)
#endif
};

// Per-row statistics exactly as ivit_modules.py:36-51 computes them.
//   sum  : exact integer sum of the row
//   returns mean_int and factor = floor(2^31 / std_int) (float32)
IVIT_DEV void ln_mean(int sum, int C, int& mean_int)
{
    // :37  x_int.mean() in float32 (= sum / C, the sum of integers is exact below 2^24), torch.round
    float mean = (float)sum / (float)C;
    mean_int = (int)rintf(mean);
}

IVIT_DEV float ln_factor(long long var)
{
    // :45-49  ten Newton steps; var_int / k promotes to float32, all divisions correctly rounded
    float varf = (float)var;
    float t = 65536.0f;
#pragma unroll 1
    for (int it = 0; it < 10; ++it) t = floorf((t + floorf(varf / t)) * 0.5f);
    // :51  (2**31-1)/std  ==  reciprocal(std) * 2^31 in float32
    return floorf((1.0f / t) * 2147483648.0f);
}

// ivit_modules.py:45-52 for 0 <= var < 2^24 (8-bit inputs, C <= 1024: SURVEY Appendix A.5): ten steps
// t <- floor((t + floor(var / t)) / 2) from t = 2^16, then floor(2^31 / t) / 2, all in float32 as the reference.
// floor(fl(var / t)) == floor(var / t) exactly there (a non-integer quotient is at least 1/t below the next integer and
// var < 2^24), so the IEEE division may be replaced by any exact integer quotient.  t >= 2^(16-k) >= 64 in step k (each step
// at most halves t), so var / t < 2^18 and var * rcp(t) (rcp: 1 ulp) is within 2^-4 of it: its floor is off by at most one,
// and r = fma(-q, t, var) is exact (|r| <= 2t, an integer), which tells which way.
IVIT_DEV float ln_newton10(float varf)
{
    float t = 65536.0f;
#pragma unroll
    for (int it = 0; it < 10; ++it) {
        float q = floorf(varf * __builtin_amdgcn_rcpf(t));
        const float r = __builtin_fmaf(-q, t, varf);
        q = (r >= t) ? q + 1.0f : q;
        q = (r < 0.0f) ? q - 1.0f : q;
        t = floorf((t + q) * 0.5f);
    }
    return t;
}

// The ten steps WITHOUT iterating, where that is provably the same (checked for every var in [0, 2^24) against the float32
// recurrence: scripts/probes/ln_newton_exhaustive.py): for var >= LN_NEWTON_CONVERGED the recurrence has converged to
// s = floor(sqrt(var)) by step ten, except when var + 1 is a perfect square, where it alternates between s and s + 1.
// Those rows (about one in 2 s) and rows with a small variance take the literal loop -- wave-uniformly, any lane.
// s from v_sqrt_f32 (1 ulp) with an exact remainder fix-up (s * s and var are integers below 2^24: the fma is exact).
constexpr float LN_NEWTON_CONVERGED = 142883.0f;
IVIT_DEV float ln_std10(int var)
{
    const float varf = (float)var;
    float s = floorf(__builtin_amdgcn_sqrtf(varf));
    float r = __builtin_fmaf(-s, s, varf);                  // var - s^2
    // s one too large (r < 0) / one too small (r > 2 s): step d = -1 / +1 / 0, then r' = var - (s + d)^2 = r - d (2 s + d)
    const float d = (r < 0.0f ? -1.0f : 0.0f) + (r > 2.0f * s ? 1.0f : 0.0f);
    r = __builtin_fmaf(-d, 2.0f * s + d, r);
    s += d;
    const bool slow = varf < LN_NEWTON_CONVERGED || r == 2.0f * s;     // var + 1 == (s + 1)^2
    if (__builtin_amdgcn_ballot_w64(slow) != 0) return ln_newton10(varf);
    return s;
}

IVIT_DEV float ln_hfactor_small(int var)
{
    const float t = ln_std10(var);
    return floorf((1.0f / t) * 2147483648.0f) * 0.5f;    // :51-52 (the /2 of :52 is an exact scaling)
}



// float32 sum of phi(q_i) over one row in the order torch's CPU sum kernel uses (ATen native/cpu/SumKernel.cpp:
// vectorized_inner_sum -> row_sum -> multi_row_sum with 8-float vectors and 4 accumulator rows: 32 partial sums, element
// i goes to partial i % 32 in increasing i with a 4-level cascade, then rows, then the 8 lanes left to right; restated
// and checked against torch in oracle/ivit_oracle.c ivo_torch_rowsum_f32).  Called by the whole wave for the rare rows
// whose mean is an exact .5 tie, where this order decides the reference's result.  phi_lds: [256] floats.
IVIT_DEV float torch_rowsum_phi(const int8_t* qrow, int C, const float* phi_lds, int lane, int outer = 0, int row = 0)
{
    if (outer) {   // the reduced dimension is not the contiguous one (Swin patch embedding): rowsum.h torch_outer_rowsum
        const bool tail_column = row % outer >= (outer & ~31);
        if (!tail_column && C % 16 == 0 && C < 256 && ((uintptr_t)qrow % 16 == 0)) {
            // round 4: the cascade's groups of 16 elements (added in sequence, then the group sums in sequence: no second-level fold
            // below 256 elements) on one lane each -- one 16-byte load per lane instead of C dependent byte loads by one lane, which made
            // every candidate tie row (2.6 % of the rows at C = 96) cost ~50 K cycles: Swin-T's patch norm 162 us against 65 us
            float gs = 0.f;
            if (lane < (C >> 4)) {
                const int4 w = *reinterpret_cast<const int4*>(qrow + 16 * lane);
                const int ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const float v = phi_lds[(int)(int8_t)(ww[d] >> (8 * bb)) + 128];
                        gs = (d == 0 && bb == 0) ? v : gs + v;
                    }
            }
            float S = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gs), 0));
            for (int k = 1; k < (C >> 4); ++k) S += __shfl(gs, k);
            return S;
        }
        return torch_outer_rowsum([&](int i) { return phi_lds[(int)qrow[i] + 128]; }, C, tail_column);
    }
    const int vec_size = C >> 3, size_ilp = vec_size >> 2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (lane < 32) {
        int lg = 0;
        while ((1 << lg) < size_ilp) ++lg;
        const int lp = max(4, lg / 4), step = 1 << lp, mask = step - 1;
        int i = 0;
        while (i + step <= size_ilp) {
            for (int j = 0; j < step; ++j, ++i) acc0 += phi_lds[(int)qrow[i * 32 + lane] + 128];
            acc1 += acc0; acc0 = 0.f;
            if ((i & (mask << lp)) == 0) {
                acc2 += acc1; acc1 = 0.f;
                if ((i & (mask << (2 * lp))) == 0) { acc3 += acc2; acc2 = 0.f; }
            }
        }
        for (; i < size_ilp; ++i) acc0 += phi_lds[(int)qrow[i * 32 + lane] + 128];
        acc0 += acc1; acc0 += acc2; acc0 += acc3;
    }
    if (lane < 8)   // vectors beyond the last group of four join accumulator row 0
        for (int i = size_ilp * 4; i < vec_size; ++i) acc0 += phi_lds[(int)qrow[i * 8 + lane] + 128];
    const float p1 = __shfl(acc0, (lane + 8) & 63), p2 = __shfl(acc0, (lane + 16) & 63), p3 = __shfl(acc0, (lane + 24) & 63);
    const float v = ((acc0 + p1) + p2) + p3;    // lanes 0-7: the 8 vector lanes
    float fin = 0.f;
    for (int i = vec_size * 8; i < C; ++i) fin += phi_lds[(int)qrow[i] + 128];
#pragma unroll
    for (int l = 0; l < 8; ++l) fin += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    return fin;
}

// int8 in -> int8 out, NJ dwords (4 channels each) per lane, per-channel constants kept in registers
// across the rows a wave processes.
//
// The tail of the reference's chain for one element is  x = y * s_ln (:63, float32),  z = round(x / s_ln)
// (quant_utils.py:220, float32 quotient),  out = clamp8(RNE(float64(z) * M)) (:229-230), M = m * 2^-e.  It costs six
// float64 instructions per element when evaluated literally.  Fast path: z = y * (1 + eps) with |eps| <= 2^-22 (two
// float32 roundings of relative size 2^-24 each, plus the round() step, which is a no-op for |y| >= 2^23, moves
// 2^22 <= |y| < 2^23 by at most 1/2 <= |y| * 2^-23 and gives back z = y exactly for |y| < 2^22), so the real number
// the reference rounds lies between y * lo and y * hi for float32 lo <= M * (1 - 1.25 * 2^-22), hi >= M * (1 + 1.25 * 2^-22).
// t_lo = fma(y, lo, 1.5 * 2^23) and t_hi = fma(y, hi, 1.5 * 2^23) are RNE(y * lo) and RNE(y * hi) exactly (one rounding,
// ulp 1) while |y * hi| < 2^22; RNE is monotone, so t_lo == t_hi certifies the reference's result.  Products beyond
// 2^22 saturate the int8 clamp on either side whatever their rounding (float bit patterns are monotone), so they need
// no separate range test.  A row with any uncertified element (about 1 % of the rows) is redone literally.
//
// COMPAT (natural activation scales): the row is seen through phi.  Bytes are remapped to k' = trunc(phi(q)) through a
// 256-byte LDS table before anything else; the mean is round(fl(sum phi / C)) -- equal to RNE(sum q / C) except on rows
// whose integer sum is an exact .5 tie (1 in C), where torch's float32 reduction order over the phi values decides and
// torch_rowsum_phi restates it; everything downstream of (k', mean) is the same arithmetic.
template <int NJ, bool COMPAT = false>
__global__ __launch_bounds__(NT, (NJ <= 3 && !COMPAT ? 3 : NJ <= 8 ? 2 : 1)) void layernorm_i8_kernel(LnArgs a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row addresses in SALU
    const int C = a.C, nd = C >> 2;
    __shared__ unsigned char s_remap[COMPAT ? 256 : 4];
    __shared__ float s_phi[COMPAT ? 256 : 1];
    if constexpr (COMPAT) {
        s_remap[threadIdx.x] = (unsigned char)a.remap[threadIdx.x];   // NT == 256
        s_phi[threadIdx.x] = a.phi[threadIdx.x];
        __syncthreads();
    }
    float bias[NJ][4], lo[NJ][4], hi[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int d = lane + 64 * j;
        if (d < nd) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias_int + 4 * d);
            const float4 s4 = *reinterpret_cast<const float4*>(a.s_ln + 4 * d);
            const uint4 m4 = *reinterpret_cast<const uint4*>(a.m + 4 * d);
            const int4 e4 = *reinterpret_cast<const int4*>(a.e + 4 * d);
            bias[j][0] = b4.x; bias[j][1] = b4.y; bias[j][2] = b4.z; bias[j][3] = b4.w;
            const float sl[4] = {s4.x, s4.y, s4.z, s4.w};
            const double Mq[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z),
                                  dyadic_mult(m4.w, e4.w)};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double lod = Mq[c] * (1.0 - 1.25 / 4194304.0), hid = Mq[c] * (1.0 + 1.25 / 4194304.0);
                float lf = (float)lod, hf = (float)hid;
                if ((double)lf > lod) lf = __int_as_float(__float_as_int(lf) - 1);   // largest float32 <= lod (lod > 0)
                if ((double)hf < hid) hf = __int_as_float(__float_as_int(hf) + 1);   // smallest float32 >= hid
                // the error bound on z needs normal float32 products y * s_ln: a degenerate scale, or a multiplier outside
                // the normal float32 range, never certifies
                const bool ok = fabsf(sl[c]) >= 1e-30f && fabsf(sl[c]) <= 1e30f && lod > 1e-35 && hid < 1e30;
                lo[j][c] = ok ? lf : 0.0f;
                hi[j][c] = ok ? hf : __builtin_inff();
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) { bias[j][c] = 0.f; lo[j][c] = 0.f; hi[j][c] = 0.f; }
        }
    }
    const int8_t* xin = reinterpret_cast<const int8_t*>(a.x);
    int8_t* out = reinterpret_cast<int8_t*>(a.out);
    BlockCol bcol[NJ];   // block-layout output: the column part of this lane's store addresses
#pragma unroll
    for (int j = 0; j < NJ; ++j) bcol[j] = block_col(4 * (lane + 64 * j));
    // G rows per wave and iteration.  The per-row statistics (mean division, ten Newton steps, reciprocal) are scalar
    // work that a one-row-per-wave kernel repeats in all 64 lanes; here lane r evaluates them for row r of the group
    // (G rows in parallel across lanes) and the results come back as wave-uniform values through v_readlane.
    constexpr int G = (NJ <= 3) ? 8 : (NJ <= 4) ? 4 : 1;
    // Streaming schedule: every wave owns one CONTIGUOUS, balanced run of row groups (the former grid-stride loop left 46 %
    // of the waves with one iteration and the rest with two at the headline shape), and the loads of group g + 1 are issued
    // before group g is computed, so that after the first group the HBM latency sits under the element chain of the previous
    // one instead of in front of every iteration (the launch is one resident wave set: all waves would otherwise load,
    // compute and store in lockstep).
    const int n_groups = (a.rows + G - 1) / G;
    const int n_waves = gridDim.x * WPB, wave_id = blockIdx.x * WPB + wave;
    const int gq = n_groups / n_waves, gr = n_groups - gq * n_waves;
    const int g_begin = wave_id * gq + min(wave_id, gr), g_end = g_begin + gq + (wave_id < gr ? 1 : 0);
    int wn[G][NJ];
    auto load_group = [&](int row0) {
#pragma unroll
        for (int rr = 0; rr < G; ++rr) {
            const int* xr = reinterpret_cast<const int*>(xin + (int64_t)min(row0 + rr, a.rows - 1) * a.ldx);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int d = lane + 64 * j;
                wn[rr][j] = (d < nd) ? xr[d] : 0;
            }
        }
    };
    if (g_begin < g_end) load_group(g_begin * G);
    for (int gi = g_begin; gi < g_end; ++gi) {
        const int row0 = gi * G;
        int w[G][NJ], sum[G], var[G];   // var[]: sum of squares
        int sumq[COMPAT ? G : 1];       // COMPAT: integer sum of the un-remapped row (the mean is over phi(q) ~ q)
#pragma unroll
        for (int rr = 0; rr < G; ++rr)
#pragma unroll
            for (int j = 0; j < NJ; ++j) w[rr][j] = wn[rr][j];
        if (gi + 1 < g_end) load_group(row0 + G);     // in flight during this group's arithmetic
#pragma unroll
        for (int rr = 0; rr < G; ++rr) {
            sum[rr] = 0;
            var[rr] = 0;
            if constexpr (COMPAT) sumq[rr] = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (COMPAT) {
                    sumq[rr] = __builtin_amdgcn_sdot4(w[rr][j], 0x01010101, sumq[rr], false);
                    const unsigned u = (unsigned)w[rr][j] ^ 0x80808080u;      // q + 128 per byte
                    w[rr][j] = (int)((unsigned)s_remap[u & 255] | ((unsigned)s_remap[(u >> 8) & 255] << 8) |
                                     ((unsigned)s_remap[(u >> 16) & 255] << 16) | ((unsigned)s_remap[u >> 24] << 24));
                }
                // v_dot4_i32_i8: sum and sum of squares of the 4 int8 of a dword, one instruction each
                sum[rr] = __builtin_amdgcn_sdot4(w[rr][j], 0x01010101, sum[rr], false);
                var[rr] = __builtin_amdgcn_sdot4(w[rr][j], w[rr][j], var[rr], false);   // <= 4096 * 128^2 = 2^26
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int rr = 0; rr < G; ++rr) {
                sum[rr] += __shfl_xor(sum[rr], o);
                var[rr] += __shfl_xor(var[rr], o);
                if constexpr (COMPAT) sumq[rr] += __shfl_xor(sumq[rr], o);
            }
        // lane rr: statistics of row rr, computed once.  mean_int as the reference (:37); the variance sum (:40-42)
        // sum_c (x_c - mean)^2 = sum x^2 - 2*mean*sum x + C*mean^2 exactly, in integers (all terms < 2^31)
        int my_sum = sum[0], my_sq = var[0];
#pragma unroll
        for (int rr = 1; rr < G; ++rr) {
            my_sum = (lane == rr) ? sum[rr] : my_sum;
            my_sq = (lane == rr) ? var[rr] : my_sq;
        }
        int my_mean;
        if constexpr (COMPAT) {
            int my_sumq = sumq[0];
#pragma unroll
            for (int rr = 1; rr < G; ++rr) my_sumq = (lane == rr) ? sumq[rr] : my_sumq;
            ln_mean(my_sumq, C, my_mean);      // = round(fl(sum phi / C)) unless the row is a tie (|sum phi - sum q| < 0.3)
            // 2*sum - C == 0 (mod 2C)  <=>  the mean is an exact .5 tie; one unit of slack on either side
            int r2 = (2 * my_sumq - C) % (2 * C);
            r2 = r2 < 0 ? r2 + 2 * C : r2;
            const bool tie = (r2 <= 2 || r2 >= 2 * C - 2) && lane < G && row0 + lane < a.rows;
            unsigned long long tm = __builtin_amdgcn_ballot_w64(tie);
            while (tm) {                       // wave-uniform: rows decided by the float32 reduction order
                const int rr = __builtin_ctzll(tm);
                tm &= tm - 1;
                const float S = torch_rowsum_phi(xin + (int64_t)(row0 + rr) * a.ldx, C, s_phi, lane, a.outer, row0 + rr);
                const int fixed = (int)rintf(S / (float)C);     // ivit_modules.py:37
                my_mean = (lane == rr) ? fixed : my_mean;
            }
        } else {
            ln_mean(my_sum, C, my_mean);
        }
        const int my_var = my_sq - 2 * my_mean * my_sum + C * my_mean * my_mean;
        int mean_int[G];
#pragma unroll
        for (int rr = 0; rr < G; ++rr) mean_int[rr] = __builtin_amdgcn_readlane(my_mean, rr);
        const float my_factor = ln_factor((long long)my_var);   // :45-51, lane rr <-> row rr
#pragma unroll
        for (int rr = 0; rr < G; ++rr) {
            if (row0 + rr >= a.rows) continue;
            // :52 floor((y * factor) / 2): halving commutes with the float32 product (exact scaling), so fold it into the factor
            const float hfactor = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_factor), rr)) * 0.5f;
            const float mean128 = (float)(mean_int[rr] + 128);
            int res[NJ];
            unsigned unc = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned wu = (unsigned)w[rr][j] ^ 0x80808080u;      // bytes x + 128: v_cvt_f32_ubyteN below
                int o[4];
                // two channels per packed float32 instruction (v_pk_add / v_pk_mul / v_pk_fma: the same IEEE operations,
                // two lanes of data per issue slot); floor, the certificate and the clamp stay scalar
#pragma unroll
                for (int c = 0; c < 4; c += 2) {
                    const v2f xf = {(float)((wu >> (8 * c)) & 0xffu), (float)((wu >> (8 * c + 8)) & 0xffu)};
                    const v2f dl = xf - (v2f){mean128, mean128};           // x - mean, exact
                    const v2f pr = dl * (v2f){hfactor, hfactor};           // :52  float32 product (the /2 is in the factor)
                    const v2f vv = {floorf(pr.x), floorf(pr.y)};
                    const v2f y = vv + (v2f){bias[j][c], bias[j][c + 1]};  // :61  float32 add
                    const v2f tlv = __builtin_elementwise_fma(y, (v2f){lo[j][c], lo[j][c + 1]}, (v2f){12582912.0f, 12582912.0f});
                    const v2f thv = __builtin_elementwise_fma(y, (v2f){hi[j][c], hi[j][c + 1]}, (v2f){12582912.0f, 12582912.0f});
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int tl = __float_as_int(tlv[k]), th = __float_as_int(thv[k]);
                        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                        o[c + k] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);   // low byte = int8 result
                    }
                }
                const unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                const unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x04000c0cu);
                res[j] = (int)(w01 | w23);
            }
            // lanes beyond the row (d >= nd) hold lo = hi = 0: always certified
            if (__builtin_amdgcn_ballot_w64(unc != 0) != 0) {
                // literal evaluation of the row (wave-uniform branch)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    int d = lane + 64 * j;
                    if (d >= nd) continue;
                    const float4 s4 = *reinterpret_cast<const float4*>(a.s_ln + 4 * d);
                    const uint4 m4 = *reinterpret_cast<const uint4*>(a.m + 4 * d);
                    const int4 e4 = *reinterpret_cast<const int4*>(a.e + 4 * d);
                    const float sl[4] = {s4.x, s4.y, s4.z, s4.w};
                    const double Mq[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z),
                                          dyadic_mult(m4.w, e4.w)};
                    int o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float dl = (float)(sx8(w[rr][j], c) - mean_int[rr]);
                        float v = floorf(dl * hfactor);                    // :52
                        float y = v + bias[j][c];                          // :61
                        float x = y * sl[c];                               // :63  float32 product
                        // quant_utils.py:220  z = round(x / s): the correctly rounded float32 quotient,
                        // obtained as RN24(RN53(x * RN53(1/s))) (no midpoint can lie within 2^-52 of x/s)
                        float qf = (float)((double)x * (1.0 / (double)sl[c]));
                        float z = rintf(qf);
                        double p = (double)z * Mq[c];                      // :229 float64 product
                        double t = p + IVIT_MAGIC;                         // :230 round half to even
                        o[c] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                    }
                    res[j] = pack4(o[0], o[1], o[2], o[3]);
                }
            }
            if (a.out_blocks) {
                const BlockRow br = block_row(row0 + rr, C);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (lane + 64 * j < nd) *reinterpret_cast<int*>(out + block_off(br, bcol[j])) = res[j];
            } else {
                int* orow = reinterpret_cast<int*>(out + (int64_t)(row0 + rr) * a.ldo);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    int d = lane + 64 * j;
                    if (d < nd) orow[d] = res[j];
                }
            }
        }
    }
}

// Half a wave per row (the form the fused engines use, C <= 1536): lanes 0-31 own row r, lanes 32-63 row r + 1, NJ dwords
// per lane (d = l32 + 32 j); G = 8 rows (4 row pairs) per wave and iteration.  Every load and store instruction touches
// two full 128-byte lines per row pair, in the row-major layout AND in the GEMM block layout (where rows 2q, 2q + 1 of a
// 64-byte column block share a line) -- with a whole wave per row the block-layout stores were 64-byte half lines
// (36.7 vs 29.6 us per call).  The per-channel constants (bias, requant bracket) are computed once per WORKGROUP into LDS
// and read 16 bytes at a time per dword of channels; row sums reduce over 32 lanes; the statistics of the 8 rows are
// evaluated in lanes 0-3 of each half.  Arithmetic exactly as layernorm_i8_kernel (certificate, literal fallback).
// Round 4, small launches (DeiT-S b64: 12 608 rows of 384, 25 launches per forward, 9.8 us each = a chain of latencies, not
// bandwidth): G2 = 2 or 1 row pairs per wave instead of 4 puts 2-4 x as many waves on the chip for the same rows, the first
// group's rows are requested BEFORE the constants table is derived (its loads and float64 arithmetic run under their flight),
// and the ten Newton steps are the sqrt shortcut of ln_std10 where it is proven (var < 2^24: C <= 1024).
// Sum over the 32 lanes of a half wave, every lane ends with the total: four DPP steps inside the rows of 16 lanes and one
// v_permlane16_swap of the value with itself (odd rows of the one copy against even rows of the other: the two copies then hold
// row 0 | row 0 and row 1 | row 1 of each half).  No LDS instruction: the __shfl_xor form was five ds_bpermute_b32 round trips per
// value, 12 per row pair with the broadcasts -- at 401 408 rows of 96 channels the LDS pipe, not the VALU, set the pace.
IVIT_DEV int half_wave_allreduce(int v)
{
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    const v2u_ r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(r.x + r.y);
}

template <int NJ, int G2 = 4>
__global__ __launch_bounds__(NT, NJ <= 3 ? 4 : 3) void layernorm_i8_pair_kernel(LnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds_tab[];   // [C] bias | [C] lo | [C] hi
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int C = a.C, nd = C >> 2;
    float* t_bias = lds_tab;
    float* t_lo = lds_tab + C;
    float* t_hi = lds_tab + 2 * C;
    const int8_t* xin = reinterpret_cast<const int8_t*>(a.x);
    int8_t* out = reinterpret_cast<int8_t*>(a.out);
    const int row_first = (blockIdx.x * WPB + wave) * (2 * G2);
    int w[G2][NJ];
    auto load_rows = [&](int row0) {
#pragma unroll
        for (int q = 0; q < G2; ++q) {
            const int row = min(row0 + 2 * q + half, a.rows - 1);
            const int* xr = reinterpret_cast<const int*>(xin + (int64_t)row * a.ldx);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int d = l32 + 32 * j;
                w[q][j] = (d < nd) ? xr[d] : 0;
            }
        }
    };
    load_rows(row_first);
    ln_build_table<NT>(a.m, a.e, a.s_ln, a.bias_int, C, t_bias, t_lo, t_hi);
    __syncthreads();
    for (int row0 = row_first; row0 < a.rows; row0 += gridDim.x * WPB * (2 * G2)) {
        if (row0 != row_first) load_rows(row0);      // uniform per wave
        int sum[G2], sq[G2];
#pragma unroll
        for (int q = 0; q < G2; ++q) {
            sum[q] = 0;
            sq[q] = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                sum[q] = __builtin_amdgcn_sdot4(w[q][j], 0x01010101, sum[q], false);
                sq[q] = __builtin_amdgcn_sdot4(w[q][j], w[q][j], sq[q], false);   // <= 1536 * 128^2 < 2^25
            }
        }
        if (IVIT_LAB && (a.abl & (1 << 27))) {      // lab A/B: the ds_bpermute butterfly of rounds 2-4
#pragma unroll
            for (int o = 16; o > 0; o >>= 1)
#pragma unroll
                for (int q = 0; q < G2; ++q) {
                    sum[q] += __shfl_xor(sum[q], o);
                    sq[q] += __shfl_xor(sq[q], o);
                }
        } else {
#pragma unroll
            for (int q = 0; q < G2; ++q) {
                sum[q] = half_wave_allreduce(sum[q]);
                sq[q] = half_wave_allreduce(sq[q]);
            }
        }
        // lane (half, l32 = q) : statistics of row 2q + half, computed once; ivit_modules.py:37, 40-51
        int my_sum = sum[0], my_sq = sq[0];
#pragma unroll
        for (int q = 1; q < G2; ++q) {
            my_sum = (l32 == q) ? sum[q] : my_sum;
            my_sq = (l32 == q) ? sq[q] : my_sq;
        }
        int my_mean;
        ln_mean(my_sum, C, my_mean);
        const int my_var = my_sq - 2 * my_mean * my_sum + C * my_mean * my_mean;   // exact, see layernorm_i8_kernel
        // :52: the /2 folds into the factor.  C <= 1024: var < 2^24, the range ln_std10's shortcut is proven on
        const float my_hfactor = (NJ <= 8 && C <= 1024) ? ln_hfactor_small(my_var) : ln_factor((long long)my_var) * 0.5f;
        int mean_int[G2];
        float hfactor[G2], mean128[G2];
#pragma unroll
        for (int q = 0; q < G2; ++q) {
            mean_int[q] = __shfl(my_mean, (lane & 32) | q);
            hfactor[q] = __shfl(my_hfactor, (lane & 32) | q);
            mean128[q] = (float)(mean_int[q] + 128);
        }
        // results are stored as they are produced; a row pair whose certificate fails is recomputed literally below and
        // stored again (same lane, same addresses: program order)
        unsigned unc[G2];
#pragma unroll
        for (int q = 0; q < G2; ++q) unc[q] = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = min(l32 + 32 * j, nd - 1);
            const float4 b4 = *reinterpret_cast<const float4*>(t_bias + 4 * d);
            const float4 l4 = *reinterpret_cast<const float4*>(t_lo + 4 * d);
            const float4 h4 = *reinterpret_cast<const float4*>(t_hi + 4 * d);
            const float bias[4] = {b4.x, b4.y, b4.z, b4.w}, lo[4] = {l4.x, l4.y, l4.z, l4.w}, hi[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int q = 0; q < G2; ++q) {
                const unsigned wu = (unsigned)w[q][j] ^ 0x80808080u;   // bytes x + 128: v_cvt_f32_ubyteN below
                int o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float dl = (float)((wu >> (8 * c)) & 0xffu) - mean128[q];   // x - mean, exact
                    const float v = floorf(dl * hfactor[q]);               // :52
                    const float y = v + bias[c];                           // :61
                    const int tl = __float_as_int(__builtin_fmaf(y, lo[c], 12582912.0f));
                    const int th = __float_as_int(__builtin_fmaf(y, hi[c], 12582912.0f));
                    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc[q]) : "v"(tl), "v"(th), "v"(unc[q]));
                    o[c] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);
                }
                const unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                const unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x04000c0cu);
                const int row = row0 + 2 * q + half, dd = l32 + 32 * j;
                if (row < a.rows && dd < nd) {
                    const int64_t off = a.out_blocks ? (int64_t)block_off(block_row(row, C), block_col(4 * dd)) : (int64_t)row * a.ldo + 4 * dd;
                    *reinterpret_cast<int*>(out + off) = (int)(w01 | w23);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the constants of one channel group live at a time
        }
#pragma unroll
        for (int q = 0; q < G2; ++q) {
            const int row = row0 + 2 * q + half;
            if (__builtin_amdgcn_ballot_w64(unc[q] != 0) != 0) {
                // literal evaluation of the row pair (wave-uniform branch), as layernorm_i8_kernel
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int d = min(l32 + 32 * j, nd - 1);
                    const float4 b4 = *reinterpret_cast<const float4*>(t_bias + 4 * d);
                    const float4 s4 = *reinterpret_cast<const float4*>(a.s_ln + 4 * d);
                    const uint4 m4 = *reinterpret_cast<const uint4*>(a.m + 4 * d);
                    const int4 e4 = *reinterpret_cast<const int4*>(a.e + 4 * d);
                    const float bias[4] = {b4.x, b4.y, b4.z, b4.w}, sl[4] = {s4.x, s4.y, s4.z, s4.w};
                    const double Mq[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z),
                                          dyadic_mult(m4.w, e4.w)};
                    int o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float dl = (float)(sx8(w[q][j], c) - mean_int[q]);
                        float v = floorf(dl * hfactor[q]);
                        float y = v + bias[c];
                        float x = y * sl[c];                               // :63
                        float qf = (float)((double)x * (1.0 / (double)sl[c]));   // quant_utils.py:220, see layernorm_i8_kernel
                        float z = rintf(qf);
                        double p = (double)z * Mq[c];                      // :229
                        double t = p + IVIT_MAGIC;                         // :230
                        o[c] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                    }
                    const int dd = l32 + 32 * j;
                    if (row < a.rows && dd < nd) {
                        const int64_t off = a.out_blocks ? ivit_block_offset(row, 4 * dd, C) : (int64_t)row * a.ldo + 4 * dd;
                        *reinterpret_cast<int*>(out + off) = pack4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
    }
}

// module-level form: int32 in (8- or 16-bit values), float32 out = y * s_ln (ivit_modules.py:63)
// ------------------------------------------------------------------------------------------------
// layernorm_i8_v2_kernel: the default int8 I-LayerNorm for C <= 1024 (NJ <= 4 dwords per lane).
//
// What the counters said about the kernels above at the headline shape (profiles/r02d_row_kernel_counters.json: 175 VALU
// wave-instructions per row, of which the element chain is 90; VALU busy 48 % of the kernel): half of the instructions
// were per-row overhead -- the 6-step all-reduce of every row's two sums (24 / row), the statistics (ten Newton steps with
// IEEE divisions) evaluated once per 8 rows, the per-wave prologue that derives 12 * NJ bracket constants in float64.  Here:
//   * a wave takes G = 16 rows at a time, one contiguous balanced run of rows per wave (no grid-stride remainder);
//   * the 16 x 2 row sums are reduced by a TRANSPOSE reduction: v_permlane32_swap / v_permlane16_swap exchange the halves
//     of two registers in one instruction, so each step halves the number of live values; 15 exchanges per sum type instead
//     of 96 shuffles, and lane l ends up with the totals of row l >> 2;
//   * the statistics run once per 16 rows, all lanes in parallel (lane l: row l >> 2), results broadcast by v_readlane;
//   * per-channel constants (bias, the float32 bracket of the requantiser) are derived once per WORKGROUP into LDS and read
//     16 bytes at a time per dword of channels and group of rows, not kept in 36 registers per wave;
//   * channels outer, rows inner in the element chain: one set of constants serves 16 rows; results overwrite the row
//     registers, a row with an uncertified element (~1 %) is re-read and redone literally.
// Arithmetic identical to layernorm_i8_kernel (same certificate, same literal fallback, same COMPAT handling).
// (COMPAT at groups of 8: three workgroups per CU -- at four the remap tables' extra registers spilled six dwords; natural-scale
// DeiT-B b256 6.95 -> 6.90 ms)
template <int NJ, bool COMPAT, int G>
__global__ __launch_bounds__(NT, (G == 4 && NJ <= 3 ? 5 : G == 8 && NJ <= 3 ? (COMPAT ? 3 : 4) : NJ <= 1 ? 4 : NJ <= 3 ? 3 : 2)) void layernorm_i8_v2_kernel(LnArgs a)
{
    static_assert(G == 4 || G == 8 || G == 16, "G");
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float lds_tab[];   // [C] bias | [C] lo | [C] hi
    __shared__ unsigned char s_remap[COMPAT ? 256 : 4];
    __shared__ float s_phi[COMPAT ? 256 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, nd = C >> 2;
    float* t_bias = lds_tab;
    float* t_lo = lds_tab + C;
    float* t_hi = lds_tab + 2 * C;
    const int abl = IVIT_LAB ? a.abl : 0;
    if (!(abl & 8)) ln_build_table<NT>(a.m, a.e, a.s_ln, a.bias_int, C, t_bias, t_lo, t_hi);
    if constexpr (COMPAT) {
        s_remap[tid] = (unsigned char)a.remap[tid];   // NT == 256
        s_phi[tid] = a.phi[tid];
    }
    __syncthreads();
    const int8_t* xin = reinterpret_cast<const int8_t*>(a.x);
    int8_t* out = reinterpret_cast<int8_t*>(a.out);
#if IVIT_LAB
    if ((abl & 128) && (blockIdx.x & 1)) {      // lab: start every other workgroup late (phase-overlap experiment)
        const int units = (abl >> 8) & 15;
        for (int it = 0; it < units; ++it) __builtin_amdgcn_s_sleep(127);
    }
#endif
    BlockCol bcol[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bcol[j] = block_col(4 * (lane + 64 * j));
    // this wave's contiguous run of rows
    const int n_waves = gridDim.x * WPB, wave_id = blockIdx.x * WPB + wave;
    const int rq = a.rows / n_waves, rrm = a.rows - rq * n_waves;
    const int r_begin = wave_id * rq + min(wave_id, rrm), r_end = r_begin + rq + (wave_id < rrm ? 1 : 0);
    // (Tried and measured, scripts/ln_ablate.py: issuing the loads of group g + 1 before group g is computed, with groups of 8
    // rows, does not overlap anything -- 29.5 us against 28.1 us for plain groups of 16 at the headline shape.  The phases of
    // this kernel add up almost exactly: read + reduce 10 us, statistics 1.5 us, element chain 11.5 us, stores 6 us.)
    auto load_group = [&](int row0, auto& dst) {
        const int nr = min(G, r_end - row0);
#pragma unroll
        for (int rr = 0; rr < G; ++rr) {
            const int* xr = reinterpret_cast<const int*>(xin + (int64_t)(row0 + min(rr, nr - 1)) * a.ldx);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // NJ = ceil(nd / 64): only the last dword column can be partial.  Unconditional loads from a clamped address
                // and a select, not a predicated load: branches around the loads make the compiler's s_waitcnt insertion
                // fall back to vmcnt(0) at the first use, which also waits for the STORES of the previous group
                const int d = lane + 64 * j;
                if (j < NJ - 1) {
                    dst[rr][j] = xr[d];
                } else {
                    const int v = xr[min(d, nd - 1)];
                    dst[rr][j] = (d < nd) ? v : 0;
                }
            }
        }
    };
    int step = ((wave_id & 1) != 0 && (abl & 64) != 0) ? G / 2 : G;     // lab bit 6: the de-phasing experiment (no gain)
    for (int row0 = r_begin; row0 < r_end; row0 += step, step = G) {
        const int nrow = min(step, r_end - row0);
        int w[G][NJ];
        int s1[G], s2[G];
        int sq[COMPAT ? G : 1];
        load_group(row0, w);
#pragma unroll
        for (int rr = 0; rr < G; ++rr) {
            s1[rr] = 0;
            s2[rr] = 0;
            if constexpr (COMPAT) sq[rr] = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (COMPAT) {
                    sq[rr] = __builtin_amdgcn_sdot4(w[rr][j], 0x01010101, sq[rr], false);
                    const unsigned u = (unsigned)w[rr][j] ^ 0x80808080u;      // q + 128 per byte
                    w[rr][j] = (int)((unsigned)s_remap[u & 255] | ((unsigned)s_remap[(u >> 8) & 255] << 8) |
                                     ((unsigned)s_remap[(u >> 16) & 255] << 16) | ((unsigned)s_remap[u >> 24] << 24));
                }
                s1[rr] = __builtin_amdgcn_sdot4(w[rr][j], 0x01010101, s1[rr], false);
                s2[rr] = __builtin_amdgcn_sdot4(w[rr][j], w[rr][j], s2[rr], false);
            }
        }
        // transpose reduction: 16 values per lane -> 1; lane l ends with the totals of row (l >> 2) & 15.
        // step "32": rows 0-7 stay in lanes 0-31, rows 8-15 go to lanes 32-63 (v_permlane32_swap exchanges the upper half
        // of its first operand with the lower half of its second); step "16" likewise inside each half; then two steps with
        // plain exchanges (2 values, 1 value), then the last two butterfly levels.
        // (G = 8: one level less -- lane l ends with row (l >> 3) & 7.)
        constexpr int LPR = 64 / G;          // lanes that end up holding one row's totals
        auto treduce = [&](int (&v)[G]) -> int {
            int t8[8], t4[4], t2[2];
            if constexpr (G == 4) {
                for (int i = 0; i < 2; ++i) {
                    const v2u r = __builtin_amdgcn_permlane32_swap((unsigned)v[i], (unsigned)v[i + 2], false, false);
                    t2[i] = (int)(r.x + r.y);
                }
                const v2u r = __builtin_amdgcn_permlane16_swap((unsigned)t2[0], (unsigned)t2[1], false, false);
                int t = (int)(r.x + r.y);
                t += __shfl_xor(t, 8);
                t += __shfl_xor(t, 4);
                t += __shfl_xor(t, 2);
                t += __shfl_xor(t, 1);
                return t;
            } else if constexpr (G == 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const v2u r = __builtin_amdgcn_permlane32_swap((unsigned)v[i], (unsigned)v[i + 8], false, false);
                    t8[i] = (int)(r.x + r.y);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v2u r = __builtin_amdgcn_permlane16_swap((unsigned)t8[i], (unsigned)t8[i + 4], false, false);
                    t4[i] = (int)(r.x + r.y);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v2u r = __builtin_amdgcn_permlane32_swap((unsigned)v[i], (unsigned)v[i + 4], false, false);
                    t4[i] = (int)(r.x + r.y);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const v2u r = __builtin_amdgcn_permlane16_swap((unsigned)t4[i], (unsigned)t4[i + 2], false, false);
                    t2[i] = (int)(r.x + r.y);
                }
            }
            const bool up8 = (lane & 8) != 0, up4 = (lane & 4) != 0;
            int t = 0;
            if constexpr (G == 4) {
            } else if constexpr (G == 16) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int keep = up8 ? t4[i + 2] : t4[i], send = up8 ? t4[i] : t4[i + 2];
                    t2[i] = keep + __shfl_xor(send, 8);
                }
                const int keep = up4 ? t2[1] : t2[0], send = up4 ? t2[0] : t2[1];
                t = keep + __shfl_xor(send, 4);
            } else {
                const int keep = up8 ? t2[1] : t2[0], send = up8 ? t2[0] : t2[1];
                t = keep + __shfl_xor(send, 8);
                t += __shfl_xor(t, 4);
            }
            t += __shfl_xor(t, 2);
            t += __shfl_xor(t, 1);
            return t;
        };
        const int my_sum = treduce(s1), my_sq = treduce(s2);     // lane l: row l / LPR
        const int my_row = (lane / LPR) & (G - 1);
        int my_mean;
        if constexpr (COMPAT) {
            const int my_sumq = treduce(sq);
            ln_mean(my_sumq, C, my_mean);      // = round(fl(sum phi / C)) unless the row is a tie (see layernorm_i8_kernel)
            int r2 = (2 * my_sumq - C) % (2 * C);
            r2 = r2 < 0 ? r2 + 2 * C : r2;
            const bool tie = (r2 <= 2 || r2 >= 2 * C - 2) && (lane & (LPR - 1)) == 0 && my_row < nrow;
            unsigned long long tm = __builtin_amdgcn_ballot_w64(tie);
            while (tm) {                       // wave-uniform: rows decided by the float32 reduction order
                const int tl_ = __builtin_ctzll(tm);
                tm &= tm - 1;
                const int rr = tl_ / LPR;
                const float S = torch_rowsum_phi(xin + (int64_t)(row0 + rr) * a.ldx, C, s_phi, lane, a.outer, row0 + rr);
                const int fixed = (int)rintf(S / (float)C);     // ivit_modules.py:37
                my_mean = (my_row == rr) ? fixed : my_mean;
            }
        } else {
            ln_mean(my_sum, C, my_mean);
        }
        const int my_var = my_sq - 2 * my_mean * my_sum + C * my_mean * my_mean;
        const float my_hfactor = (abl & 2) ? 1.0f : ln_factor((long long)my_var) * 0.5f;   // :45-52 (the /2 of :52 is an exact scaling)
        float mean128[G], hfac[G];      // wave-uniform (SGPRs)
#pragma unroll
        for (int rr = 0; rr < G; ++rr) {
            mean128[rr] = (float)(__builtin_amdgcn_readlane(my_mean, LPR * rr) + 128);
            hfac[rr] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_hfactor), LPR * rr));
        }
        unsigned unc[G];
#pragma unroll
        for (int rr = 0; rr < G; ++rr) unc[rr] = 0;
        if (!(abl & 1))
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = min(lane + 64 * j, nd - 1);
            const float4 b4 = *reinterpret_cast<const float4*>(t_bias + 4 * d);
            const float4 l4 = *reinterpret_cast<const float4*>(t_lo + 4 * d);
            const float4 h4 = *reinterpret_cast<const float4*>(t_hi + 4 * d);
            const bool live = lane + 64 * j < nd;
            const float bias[4] = {b4.x, b4.y, b4.z, b4.w};
            const float lo[4] = {live ? l4.x : 0.f, live ? l4.y : 0.f, live ? l4.z : 0.f, live ? l4.w : 0.f};
            const float hi[4] = {live ? h4.x : 0.f, live ? h4.y : 0.f, live ? h4.z : 0.f, live ? h4.w : 0.f};
#pragma unroll
            for (int rr = 0; rr < G; ++rr) {
                if (rr >= nrow) continue;   // wave-uniform
                const unsigned wu = (unsigned)w[rr][j] ^ 0x80808080u;
                int o[4];
                unsigned u = unc[rr];
#pragma unroll
                for (int c = 0; c < 4; c += 2) {
                    const v2f xf = {(float)((wu >> (8 * c)) & 0xffu), (float)((wu >> (8 * c + 8)) & 0xffu)};
                    const v2f dl = xf - (v2f){mean128[rr], mean128[rr]};   // x - mean, exact
                    const v2f pr = dl * (v2f){hfac[rr], hfac[rr]};         // :52
                    const v2f vv = {floorf(pr.x), floorf(pr.y)};
                    const v2f y = vv + (v2f){bias[c], bias[c + 1]};        // :61
                    const v2f tlv = __builtin_elementwise_fma(y, (v2f){lo[c], lo[c + 1]}, (v2f){12582912.0f, 12582912.0f});
                    const v2f thv = __builtin_elementwise_fma(y, (v2f){hi[c], hi[c + 1]}, (v2f){12582912.0f, 12582912.0f});
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int tl = __float_as_int(tlv[k]), th = __float_as_int(thv[k]);
                        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(u) : "v"(tl), "v"(th), "v"(u));
                        o[c + k] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);
                    }
                }
                unc[rr] = u;
                const unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                const unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x04000c0cu);
                w[rr][j] = (int)(w01 | w23);      // the input dword is dead from here on
            }
        }
#pragma unroll
        for (int rr = 0; rr < G; ++rr) {
            if (rr >= nrow) continue;       // wave-uniform
            const int row = row0 + rr;
            if (__builtin_amdgcn_ballot_w64(unc[rr] != 0) != 0) {
                // literal evaluation of the row (wave-uniform, ~1 % of the rows): re-read it, as layernorm_i8_kernel
                const int* xr = reinterpret_cast<const int*>(xin + (int64_t)row * a.ldx);
                const int mean_i = (int)mean128[rr] - 128;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int d = lane + 64 * j;
                    if (d >= nd) continue;
                    int wv = xr[d];
                    if constexpr (COMPAT) {
                        const unsigned u = (unsigned)wv ^ 0x80808080u;
                        wv = (int)((unsigned)s_remap[u & 255] | ((unsigned)s_remap[(u >> 8) & 255] << 8) |
                                   ((unsigned)s_remap[(u >> 16) & 255] << 16) | ((unsigned)s_remap[u >> 24] << 24));
                    }
                    const float4 s4 = *reinterpret_cast<const float4*>(a.s_ln + 4 * d);
                    const uint4 m4 = *reinterpret_cast<const uint4*>(a.m + 4 * d);
                    const int4 e4 = *reinterpret_cast<const int4*>(a.e + 4 * d);
                    const float4 b4 = *reinterpret_cast<const float4*>(t_bias + 4 * d);
                    const float sl[4] = {s4.x, s4.y, s4.z, s4.w}, bias[4] = {b4.x, b4.y, b4.z, b4.w};
                    const double Mq[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z),
                                          dyadic_mult(m4.w, e4.w)};
                    int o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float dl = (float)(sx8(wv, c) - mean_i);
                        float v = floorf(dl * hfac[rr]);                   // :52
                        float y = v + bias[c];                             // :61
                        float x = y * sl[c];                               // :63
                        float qf = (float)((double)x * (1.0 / (double)sl[c]));   // quant_utils.py:220, see layernorm_i8_kernel
                        float z = rintf(qf);
                        double p = (double)z * Mq[c];                      // :229
                        double t = p + IVIT_MAGIC;                         // :230
                        o[c] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                    }
                    w[rr][j] = pack4(o[0], o[1], o[2], o[3]);
                }
            }
            if (abl & 4) continue;
            if (a.out_blocks) {
                const BlockRow br = block_row(row, C);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (j < NJ - 1 || lane + 64 * j < nd) *reinterpret_cast<int*>(out + block_off(br, bcol[j])) = w[rr][j];
            } else {
                int* orow = reinterpret_cast<int*>(out + (int64_t)row * a.ldo);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int d = lane + 64 * j;
                    if (j < NJ - 1 || d < nd) orow[d] = w[rr][j];
                }
            }
        }
    }
}

#include "ln_stream.h"

__global__ __launch_bounds__(NT) void layernorm_i32_f32_kernel(LnArgs a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row addresses in SALU
    const int C = a.C;
    const int32_t* xin = reinterpret_cast<const int32_t*>(a.x);
    float* out = reinterpret_cast<float*>(a.out);
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const int32_t* xr = xin + (int64_t)row * a.ldx;
        // float32 sum of integers (exact below 2^24; 16-bit rows may exceed it: then this is
        // RN24 of the exact sum)
        int sum = 0;  // |sum| < 2^31 for C <= 4096 and 16-bit values
        for (int c = lane; c < C; c += 64) sum += xr[c];
        sum = wave_reduce_sum_i32(sum);
        float mean = (float)sum / (float)C;
        int mean_int = (int)rintf(mean);
        long long var = 0;
        for (int c = lane; c < C; c += 64) {
            long long d = (long long)xr[c] - mean_int;
            var += d * d;
        }
        // 64-bit wave reduction
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int vlo = __shfl_xor((int)(var & 0xffffffffll), o);
            int vhi = __shfl_xor((int)(var >> 32), o);
            var += ((long long)vhi << 32) | (unsigned)vlo;
        }
        const float factor = ln_factor(var);
        for (int c = lane; c < C; c += 64) {
            float dl = (float)(xr[c] - mean_int);
            float v = floorf((dl * factor) * 0.5f);
            float y = v + a.bias_int[c];
            out[(int64_t)row * a.ldo + c] = y * a.s_ln[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ShiftGELU
// ------------------------------------------------------------------------------------------------
// One element of IVITIntGELU.forward (ivit_modules.py:105-123): returns sigmoid_int in [0,127].
//   e  = exp_int(k - kmax), em = exp_int(-kmax)  (both exact integers)
IVIT_DEV int gelu_sigmoid(float e, float em)
{
    float S = e + em;                                            // :116 float32 add
    S = fminf(S, 2147483648.0f);                                 // :118
    float factor = floorf((1.0f / S) * 2147483648.0f);           // :119
    float pr = e * factor;                                       // :120 float32 product
    return (int)(((unsigned)pr) >> 24);                          //      floor(. / 2^24)
}

struct GeluArgs {
    const int8_t* x;
    int64_t ldx;
    int rows, L;
    int x0;      // floor(-1/(s*1.702))
    double Mq;   // per-tensor requantiser (mlp.qact1)
    const int8_t* lut;
    void* out;
    int64_t ldo;
    int out_blocks;   // output in the GEMM block layout (common.h: ivit_block_offset), row length L
    int in_blocks;    // input in the block layout (table form only)
    const int8_t* remap;   // table build only, may be NULL: k'(q) = trunc(phi(q)) for a natural input scale (ivit_modules.py:106-107)
};

// direct arithmetic; OUT_I32: module-level int32 output k*sig, else fused requant -> int8
template <bool OUT_I32>
__global__ __launch_bounds__(NT) void shiftgelu_kernel(GeluArgs a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row addresses in SALU
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const int8_t* xr = a.x + (int64_t)row * a.ldx;
        int kmax = -128;
        for (int i = lane; i < a.L; i += 64) kmax = max(kmax, (int)xr[i]);
        kmax = wave_reduce_max_i32(kmax);                         // :110
        const float em = shiftexp_f32(-kmax, a.x0, 23);           // :115
        for (int i = lane; i < a.L; i += 64) {
            int k = xr[i];
            float e = shiftexp_f32(k - kmax, a.x0, 23);           // :111-113
            int sig = gelu_sigmoid(e, em);
            int v = k * sig;                                      // :123
            if (OUT_I32) {
                reinterpret_cast<int32_t*>(a.out)[(int64_t)row * a.ldo + i] = v;
            } else {
                reinterpret_cast<int8_t*>(a.out)[(int64_t)row * a.ldo + i] =
                    (int8_t)clamp_i32(requant_exact(v, a.Mq), -128, 127);
            }
        }
    }
}

// lut[(kmax+128)*256 + (k+128)] for every (kmax, k <= kmax); entries with k > kmax are unused (0).
__global__ __launch_bounds__(NT) void shiftgelu_lut_kernel(GeluArgs a)
{
    const int idx = blockIdx.x * NT + threadIdx.x;  // 65536 entries
    int kmax = (idx >> 8) - 128, k = (idx & 255) - 128;
    int8_t r = 0;
    const bool used = k <= kmax;
    if (a.remap) {      // the row maximum of k' is k'(max q): trunc(phi) is monotone
        kmax = a.remap[kmax + 128];
        k = a.remap[k + 128];
    }
    if (used) {
        // exp_int(-kmax): int_exp_shift clamps at n*x0, positive arguments included (:95 torch.max)
        float em = shiftexp_f32(-kmax, a.x0, 23);
        float e = shiftexp_f32(k - kmax, a.x0, 23);
        int v = k * gelu_sigmoid(e, em);
        r = (int8_t)clamp_i32(requant_exact(v, a.Mq), -128, 127);
    }
    reinterpret_cast<int8_t*>(a.out)[idx] = r;
}

// table form: row max by wave reduction, the row's 256-byte table slice staged in LDS, byte gather.
// NJ dwords per lane stay in registers between the max pass and the gather (one HBM read per byte); two rows per
// wave and iteration keep twice the loads in flight (the kernel is HBM-latency bound).  NJ = 0: generic re-read form.
// INB: the input is in the GEMM block layout (compile time: with a run-time `a.in_blocks ? .. : ..` around every load the compiler
// branched per load and put an s_waitcnt vmcnt(0) in front of each row-major one -- NJ x RW serial HBM latencies per iteration; Swin's
// row-major MLP ran at 2.5-4.2 TB/s where the block-layout path of the ViT engines reached 6).  Loads are unconditional: a dword
// beyond the row clamps to the row's last one and is replaced by -128 bytes afterwards.
template <int NJ, bool INB = false>
__global__ __launch_bounds__(NT) void shiftgelu_lut_apply_kernel(GeluArgs a)
{
    constexpr int RW = 2;
    constexpr int NJR = NJ > 0 ? NJ : 1;
    __shared__ __attribute__((aligned(16))) unsigned char tab[WPB][RW][256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row addresses in SALU
    const int nd = a.L >> 2;
    for (int row0 = (blockIdx.x * WPB + wave) * RW; row0 < a.rows; row0 += gridDim.x * WPB * RW) {
        const int* xr[RW];
        int kmax[RW];
        int w[RW][NJR];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            xr[r] = reinterpret_cast<const int*>(a.x + (int64_t)min(row0 + r, a.rows - 1) * a.ldx);
            kmax[r] = -128;
        }
        if constexpr (NJ > 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const int d = lane + 64 * j, dd = min(d, nd - 1);
                    int v;
                    if constexpr (INB) v = *reinterpret_cast<const int*>(a.x + block_off(block_row(min(row0 + r, a.rows - 1), a.L), block_col(4 * dd)));
                    else v = xr[r][dd];
                    w[r][j] = (d < nd) ? v : (int)0x80808080;
                }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    kmax[r] = max(max(kmax[r], sx8(w[r][j], 0)), max(sx8(w[r][j], 1), max(sx8(w[r][j], 2), sx8(w[r][j], 3))));
        } else {
            for (int d = lane; d < nd; d += 64) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    int v = a.in_blocks ? *reinterpret_cast<const int*>(a.x + block_off(block_row(min(row0 + r, a.rows - 1), a.L), block_col(4 * d)))
                                        : xr[r][d];
                    kmax[r] = max(max(kmax[r], sx8(v, 0)), max(sx8(v, 1), max(sx8(v, 2), sx8(v, 3))));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) kmax[r] = wave_allmax_i32(kmax[r]);      // common.h: DPP + permlane swaps, no LDS round trips
#pragma unroll
        for (int r = 0; r < RW; ++r)
            reinterpret_cast<int*>(tab[wave][r])[lane] =
                reinterpret_cast<const int*>(a.lut + (int64_t)(kmax[r] + 128) * 256)[lane];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the table slices are in LDS
        const BlockRow brow[RW] = {block_row(row0, a.L), block_row(row0 + 1, a.L)};
        auto map4 = [&](int r, unsigned v) {
            v ^= 0x80808080u;  // k + 128 per byte
            const unsigned char* tb = tab[wave][r];
            return (unsigned)tb[v & 255] | ((unsigned)tb[(v >> 8) & 255] << 8) | ((unsigned)tb[(v >> 16) & 255] << 16) |
                   ((unsigned)tb[v >> 24] << 24);
        };
        if constexpr (NJ > 0) {
            if constexpr (!INB) {      // all gathers first (in place over the inputs; every byte indexes inside the slice), then the stores
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < RW; ++r) w[r][j] = (int)map4(r, (unsigned)w[r][j]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const int d = lane + 64 * j;
                    if (d < nd && row0 + r < a.rows) {
                        const int64_t off = a.out_blocks ? (int64_t)block_off(brow[r], block_col(4 * d)) : (int64_t)(row0 + r) * a.ldo + 4 * d;
                        *reinterpret_cast<int*>(reinterpret_cast<int8_t*>(a.out) + off) = INB ? (int)map4(r, (unsigned)w[r][j]) : w[r][j];
                    }
                }
        } else {
            for (int d = lane; d < nd; d += 64) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    if (row0 + r >= a.rows) continue;
                    const int64_t off = a.out_blocks ? ivit_block_offset(row0 + r, 4 * d, a.L) : (int64_t)(row0 + r) * a.ldo + 4 * d;
                    const unsigned vin = a.in_blocks ? *reinterpret_cast<const unsigned*>(a.x + block_off(brow[r], block_col(4 * d))) : (unsigned)xr[r][d];
                    *reinterpret_cast<int*>(reinterpret_cast<int8_t*>(a.out) + off) = (int)map4(r, vin);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// stand-alone Shiftmax (module-level)
// ------------------------------------------------------------------------------------------------
template <typename TX>
struct SmArgs {
    const TX* x;
    int64_t ldx;
    int rows, L, x0;
    int8_t* out;
    int64_t ldo;
};

template <typename TX>
__global__ __launch_bounds__(NT) void shiftmax_kernel(SmArgs<TX> a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row addresses in SALU
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const TX* xr = a.x + (int64_t)row * a.ldx;
        int kmax = INT_MIN;
        for (int i = lane; i < a.L; i += 64) kmax = max(kmax, (int)xr[i]);
        kmax = wave_reduce_max_i32(kmax);
        unsigned long long sum = 0;
        for (int i = lane; i < a.L; i += 64) sum += shiftexp_int((int)xr[i] - kmax, a.x0, 15);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            unsigned lo = (unsigned)__shfl_xor((int)(sum & 0xffffffffull), o);
            unsigned hi = (unsigned)__shfl_xor((int)(sum >> 32), o);
            sum += ((unsigned long long)hi << 32) | lo;
        }
        float S = (float)sum;                                     // :171 (RN24 of the exact sum)
        S = fminf(S, 2147483648.0f);                              // :173
        const float factor = floorf((1.0f / S) * 2147483648.0f);  // :174
        for (int i = lane; i < a.L; i += 64) {
            unsigned e = shiftexp_int((int)xr[i] - kmax, a.x0, 15);
            float pr = (float)e * factor;                         // :175
            a.out[(int64_t)row * a.ldo + i] = (int8_t)(((unsigned)pr) >> 24);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// element-wise pieces
// ------------------------------------------------------------------------------------------------
IVIT_DEV int quant_sym_i8(float x, float inv_scale)
{
    float v = rintf(inv_scale * x);  // quant_utils.py:49 round(1./scale * input)
    v = fminf(fmaxf(v, -128.0f), 127.0f);
    return (int)v;
}

__global__ __launch_bounds__(NT) void quantize_kernel(const float* x, int8_t* out, int64_t n, float inv_scale)
{
    int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * NT * 4;
    for (; i + 3 < n; i += stride) {
        float4 v = *reinterpret_cast<const float4*>(x + i);
        *reinterpret_cast<int*>(out + i) = pack4(quant_sym_i8(v.x, inv_scale), quant_sym_i8(v.y, inv_scale),
                                                 quant_sym_i8(v.z, inv_scale), quant_sym_i8(v.w, inv_scale));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t t = n & ~(int64_t)3; t < n; ++t) out[t] = (int8_t)quant_sym_i8(x[t], inv_scale);
}

// SymmetricQuantFunction at any width up to 32 bits (quant_utils.py:79-97): clamp(round(1/s * x), -2^(b-1), 2^(b-1) - 1)
__global__ __launch_bounds__(NT) void quantize_i32_kernel(const float* x, int32_t* out, int64_t n, float inv_scale, float lo,
                                                          float hi)
{
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        float v = rintf(inv_scale * x[i]);
        v = fminf(fmaxf(v, lo), hi);
        out[i] = (int32_t)fminf(v, 2147483520.0f);
    }
}

// img [B, chans, hw, hw] -> A [B*gh*gw, chans*patch*patch]; one thread quantises 4 consecutive kw
__global__ __launch_bounds__(NT) void patchify_kernel(const float* img, int8_t* A, int64_t lda, int batch, int chans,
                                                      int hw, int patch, float inv_scale)
{
    const int g = hw / patch;
    const int pw4 = patch >> 2;
    const int64_t total = (int64_t)batch * chans * hw * (hw >> 2);  // groups of 4 pixels
    if (total < 2147483648ll) {      // uniform: 32-bit index arithmetic (the 64-bit divisions below cost more than the 20 bytes they place)
        const unsigned w4 = (unsigned)(hw >> 2), tot = (unsigned)total;
        for (unsigned q = blockIdx.x * NT + threadIdx.x; q < tot; q += gridDim.x * NT) {
            const unsigned r = q / w4, x4 = q - r * w4;
            const unsigned r2 = r / (unsigned)hw, y = r - r2 * (unsigned)hw;
            const unsigned b = r2 / (unsigned)chans, c = r2 - b * (unsigned)chans;
            const float4 v = *reinterpret_cast<const float4*>(img + (size_t)r * hw + 4 * x4);      // r = (b * chans + c) * hw + y
            const unsigned px = x4 / (unsigned)pw4, kw = (x4 - px * pw4) * 4;
            const unsigned py = y / (unsigned)patch, kh = y - py * patch;
            const int64_t row = ((int64_t)b * g + py) * g + px;
            const unsigned col = (c * patch + kh) * patch + kw;
            *reinterpret_cast<int*>(A + row * lda + col) = pack4(quant_sym_i8(v.x, inv_scale), quant_sym_i8(v.y, inv_scale),
                                                                 quant_sym_i8(v.z, inv_scale), quant_sym_i8(v.w, inv_scale));
        }
        return;
    }
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        // source order (coalesced float4 reads): b, c, y, x4
        int x4 = (int)(q % (hw >> 2));
        int64_t r = q / (hw >> 2);
        int y = (int)(r % hw);
        r /= hw;
        int c = (int)(r % chans);
        int b = (int)(r / chans);
        float4 v = *reinterpret_cast<const float4*>(img + (((int64_t)b * chans + c) * hw + y) * hw + 4 * x4);
        int px = x4 / pw4, kw = (x4 - px * pw4) * 4;
        int py = y / patch, kh = y - py * patch;
        int64_t row = ((int64_t)b * g + py) * g + px;
        int col = (c * patch + kh) * patch + kw;
        *reinterpret_cast<int*>(A + row * lda + col) = pack4(quant_sym_i8(v.x, inv_scale), quant_sym_i8(v.y, inv_scale),
                                                             quant_sym_i8(v.z, inv_scale), quant_sym_i8(v.w, inv_scale));
    }
}

// The same from uint8 pixels: the float pipeline in front of the model (ToTensor: v / 255; Normalize: (x - mean[c]) / std[c]) and
// the input QuantAct depend on (channel, pixel value) alone -- 3 x 256 results, tabulated by the caller with the float32 sequence
// itself; a quarter of the input bytes.  One thread: 4 consecutive pixels of a row.
__global__ __launch_bounds__(NT) void patchify_u8_kernel(const uint8_t* img, int8_t* A, int64_t lda, int batch, int chans, int hw,
                                                         int patch, const int8_t* lut)
{
    __shared__ int8_t s_lut[4 * 256];
    for (int i = threadIdx.x; i < chans * 256; i += NT) s_lut[i] = lut[i];
    __syncthreads();
    const int g = hw / patch;
    const int pw4 = patch >> 2;
    const int64_t total = (int64_t)batch * chans * hw * (hw >> 2);
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        int x4 = (int)(q % (hw >> 2));
        int64_t r = q / (hw >> 2);
        int y = (int)(r % hw);
        r /= hw;
        int c = (int)(r % chans);
        int b = (int)(r / chans);
        const unsigned v = *reinterpret_cast<const unsigned*>(img + (((int64_t)b * chans + c) * hw + y) * hw + 4 * x4);
        const int8_t* t = s_lut + 256 * c;
        int px = x4 / pw4, kw = (x4 - px * pw4) * 4;
        int py = y / patch, kh = y - py * patch;
        int64_t row = ((int64_t)b * g + py) * g + px;
        int col = (c * patch + kh) * patch + kw;
        *reinterpret_cast<int*>(A + row * lda + col) = pack4(t[v & 255], t[(v >> 8) & 255], t[(v >> 16) & 255], t[v >> 24]);
    }
}

// cls row + position embedding, vit_quant.py:290-296 (see ivit_hip.h)
__global__ __launch_bounds__(NT) void embed_kernel(const int8_t* patch, const int16_t* pos_add, const int8_t* cls_row,
                                                   double Mq, int8_t* out, int batch, int tokens, int C)
{
    const int cd = C >> 2;
    const int64_t total = (int64_t)batch * tokens * cd;
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        int d = (int)(q % cd);
        int64_t r = q / cd;
        int tok = (int)(r % tokens);
        int b = (int)(r / tokens);
        int res;
        if (tok == 0) {
            res = *reinterpret_cast<const int*>(cls_row + 4 * d);
        } else {
            int w = *reinterpret_cast<const int*>(patch + ((int64_t)b * (tokens - 1) + tok - 1) * C + 4 * d);
            const int16_t* pa = pos_add + (int64_t)tok * C + 4 * d;
            int o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = clamp_i32(requant_exact(sx8(w, c), Mq) + (int)pa[c], -128, 127);
            res = pack4(o[0], o[1], o[2], o[3]);
        }
        *reinterpret_cast<int*>(out + ((int64_t)b * tokens + tok) * C + 4 * d) = res;
    }
}

// the same with a 16-bit patch embedding (patch_embed_bw = 16) and a 16-bit block input (block_input_bw = 16, vit_quant.py:180-187):
// out16 = clamp16(RNE(patch16 * Mq) + pos_add[tok]), cls row precomputed
__global__ __launch_bounds__(NT) void embed16_kernel(const int16_t* patch, const int32_t* pos_add, const int16_t* cls_row,
                                                     double Mq, int16_t* out, int batch, int tokens, int C)
{
    const int cd = C >> 2;
    const int64_t total = (int64_t)batch * tokens * cd;
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        const int d = (int)(q % cd);
        const int64_t r = q / cd;
        const int tok = (int)(r % tokens);
        const int b = (int)(r / tokens);
        int2 res;
        if (tok == 0) {
            res = *reinterpret_cast<const int2*>(cls_row + 4 * d);
        } else {
            const int2 w = *reinterpret_cast<const int2*>(patch + ((int64_t)b * (tokens - 1) + tok - 1) * C + 4 * d);
            const int4 pa = *reinterpret_cast<const int4*>(pos_add + (int64_t)tok * C + 4 * d);
            const int k[4] = {(int)(int16_t)w.x, w.x >> 16, (int)(int16_t)w.y, w.y >> 16};
            const int p4[4] = {pa.x, pa.y, pa.z, pa.w};
            int o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = clamp_i32(requant_exact(k[c], Mq) + p4[c], -32768, 32767);
            res.x = (o[0] & 0xffff) | (o[1] << 16);
            res.y = (o[2] & 0xffff) | (o[3] << 16);
        }
        *reinterpret_cast<int2*>(out + ((int64_t)b * tokens + tok) * C + 4 * d) = res;
    }
}

// classifier: logits_f32 = float(acc) * s_acc (quant_modules.py:225-226); arg-max, first index on ties
__global__ __launch_bounds__(NT) void head_argmax_kernel(const int32_t* acc, const float* s_acc, int batch, int N,
                                                         float* logits, int32_t* top1)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row addresses in SALU
    for (int row = blockIdx.x * WPB + wave; row < batch; row += gridDim.x * WPB) {
        float best = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int n = lane; n < N; n += 64) {
            float v = (float)acc[(int64_t)row * N + n] * s_acc[n];
            if (logits) logits[(int64_t)row * N + n] = v;
            if (v > best) { best = v; bi = n; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o);
            int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) top1[row] = bi;
    }
}

// generic QuantAct on int32 (fixedpoint_mul), bit-faithful general form
__global__ __launch_bounds__(NT) void requant_i32_kernel(const int32_t* z, int64_t rows, int C, const uint32_t* m,
                                                         const int32_t* e, int n_me, const int32_t* z2,
                                                         const uint32_t* m2, const int32_t* e2, int n_me2, int bits,
                                                         int32_t* out)
{
    const int64_t total = rows * C;
    const double lo = -__builtin_ldexp(1.0, bits - 1), hi = __builtin_ldexp(1.0, bits - 1) - 1.0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int c = (int)(i % C);
        int im = (n_me == 1) ? 0 : c;
        double o = requant_double((double)z[i], dyadic_mult(m[im], e[im]));
        if (z2) {
            int i2 = (n_me2 == 1) ? 0 : c;
            o = requant_double((double)z2[i], dyadic_mult(m2[i2], e2[i2])) + o;
        }
        o = (double)(float)o;  // quant_utils.py:249 output.type(torch.float) precedes the clamp
        o = fmin(fmax(o, lo), hi);
        out[i] = (int32_t)o;
    }
}

__global__ __launch_bounds__(NT) void residual_requant_kernel(const int8_t* a, double Ma, const int8_t* b, double Mb,
                                                              int8_t* out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        out[i] = (int8_t)clamp_i32(requant_exact(a[i], Ma) + requant_exact(b[i], Mb), -128, 127);
}

__global__ __launch_bounds__(NT) void f32_to_i32_kernel(const float* x, int64_t rows, int C, const float* s, int n_s,
                                                        int mode, int32_t* z)
{
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        float q = x[i] / s[n_s == 1 ? 0 : (int)(i % C)];  // correctly rounded float32 quotient
        q = (mode == 0) ? rintf(q) : truncf(q);
        q = fminf(fmaxf(q, -2147483648.0f), 2147483520.0f);
        z[i] = (int32_t)q;
    }
}

__global__ __launch_bounds__(NT) void i32_to_f32_kernel(const int32_t* z, int64_t rows, int C, const float* s, int n_s,
                                                        float* y)
{
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT)
        y[i] = (float)z[i] * s[n_s == 1 ? 0 : (int)(i % C)];
}

__global__ __launch_bounds__(NT) void narrow_i32_i8_kernel(const int32_t* z, int8_t* out, int64_t n, int* overflow)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        int v = z[i];
        bad |= (v < -128 || v > 127);
        out[i] = (int8_t)clamp_i32(v, -128, 127);
    }
    if (bad && overflow) atomicOr(overflow, 1);
}

// module-level QuantMatMul: small batched products, one output element per thread (not the hot path:
// the engine uses the fused MFMA attention kernel)
template <bool PV, typename TX = int8_t>
__global__ __launch_bounds__(NT) void bgemm_kernel(const TX* X, const int8_t* Y, int32_t* O, int batch, int Tq,
                                                   int Tk, int D)
{
    const int cols = PV ? D : Tk, red = PV ? Tk : D;
    const int64_t total = (int64_t)batch * Tq * cols;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * NT) {
        int j = (int)(idx % cols);
        int64_t r = idx / cols;
        int i = (int)(r % Tq);
        int b = (int)(r / Tq);
        const TX* xr = X + ((int64_t)b * Tq + i) * red;
        int acc = 0;
        if (PV) {
            const int8_t* yb = Y + (int64_t)b * Tk * D + j;
            for (int k = 0; k < red; ++k) acc += (int)xr[k] * (int)yb[(int64_t)k * D];
        } else {
            const int8_t* yr = Y + ((int64_t)b * Tk + j) * D;
            for (int k = 0; k < red; ++k) acc += (int)xr[k] * (int)yr[k];
        }
        O[idx] = acc;
    }
}

static inline int ew_grid(int64_t n)
{
    int64_t b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// row-major <-> block layout (common.h: ivit_block_offset), 16 bytes per thread
template <bool TO_BLOCKS>
__global__ __launch_bounds__(NT) void relayout_kernel(const int8_t* src, int8_t* dst, int64_t ld, int64_t rows, int K)
{
    const int c16 = K >> 4;
    const int64_t total = rows * c16;
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        const int64_t r = q / c16;
        const int c = (int)(q - r * c16) << 4;
        const int64_t rm = r * ld + c, bl = ivit_block_offset(r, c, K);
        if (TO_BLOCKS) *reinterpret_cast<int4*>(dst + bl) = *reinterpret_cast<const int4*>(src + rm);
        else *reinterpret_cast<int4*>(dst + rm) = *reinterpret_cast<const int4*>(src + bl);
    }
}

static int gelu_x0(float s, const char* who, int* x0_out)
{
    IVIT_REQUIRE(s > 0.0f, "%s: scale must be positive", who);
    const float s_sig = s * 1.702f;                                   // ivit_modules.py:108
    const float x0f = __builtin_floorf((1.0f / s_sig) * -1.0f);       // :94
    IVIT_REQUIRE(x0f <= -1.0f && x0f >= -255.0f, "%s: x0=%g outside [-255,-1] (scale %g)", who, (double)x0f, (double)s);
    *x0_out = (int)x0f;
    return IVIT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// calibration statistics: min / max of a float tensor (QuantAct running_stat mode, quant_modules.py:310-349)
// float -> order-preserving uint32 key so that integer atomics reduce across workgroups
// ------------------------------------------------------------------------------------------------
IVIT_DEV unsigned f32_key(float v)
{
    const unsigned b = (unsigned)__float_as_int(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

IVIT_DEV float key_f32(unsigned k)
{
    return __int_as_float((int)((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k));
}

__global__ void minmax_init_kernel(unsigned* keys)
{
    keys[0] = 0xffffffffu;   // running min
    keys[1] = 0u;            // running max
}

__global__ __launch_bounds__(NT) void minmax_kernel(const float* x, int64_t n, unsigned* keys)
{
    unsigned kmin = 0xffffffffu, kmax = 0u;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float v = x[i];
        if (v == v) {   // NaNs do not take part (torch.min/max would propagate them; a calibrated range must not)
            const unsigned k = f32_key(v);
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o));
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(keys, kmin);
        atomicMax(keys + 1, kmax);
    }
}

__global__ void minmax_finish_kernel(unsigned* keys)
{
    const unsigned a = keys[0], b = keys[1];
    reinterpret_cast<float*>(keys)[0] = key_f32(a);
    reinterpret_cast<float*>(keys)[1] = key_f32(b);
}

// ================================================================================================

// layernorm_i8_v2_kernel: one resident set of workgroups, every wave a contiguous run of rows
template <bool COMPAT>
static int launch_ln_v2(const LnArgs& a, hipStream_t st, const char* who)
{
    const int nj = (a.C / 4 + 63) / 64;
    const int resident = 256 * (nj <= 1 ? 4 : nj <= 3 ? 3 : 2);     // = the kernel's __launch_bounds__ occupancy
    // One group of 8 rows per wave and more workgroups than fit at once (measured at the headline shape, scripts/ln_ablate.py:
    // 25.5 us; groups of 16 rows on one resident set of workgroups 28-30 us, groups of 8 on a resident set 27.8 us): a wave's
    // timeline is serial -- first loads, arithmetic, store drain -- so shorter waves that start as others finish overlap those
    // phases across waves, which neither prefetching nor de-phasing a resident set achieved.
    // At four times the rows the order flips (90.9 vs 87.6 us: the statistics are paid once per 16 rows and a resident set then
    // has several groups per wave anyway); groups of 4 rows: 27.0 us.  Lab bits 4-5: 1 / 2 / 3 force groups of 8 / 16 / 4.
    const int force = (a.abl >> 4) & 3;
    const bool g8 = force == 1 || force == 3 || (force == 0 && a.rows <= 131072);
    const bool g4 = force == 3;
    const int gsz = g4 ? 4 : g8 ? 8 : 16;
    int grid = (int)(((int64_t)a.rows + gsz * WPB - 1) / (gsz * WPB));    // one full group per wave
    if (grid > resident && !g8) grid = resident;
    if (grid < 1) grid = 1;
    const size_t lds = (size_t)3 * a.C * sizeof(float);
#define IVIT_LN_V2(NJv)                                                                                              \
    do {                                                                                                             \
        if (g4) hipLaunchKernelGGL((layernorm_i8_v2_kernel<NJv, COMPAT, 4>), dim3(grid), dim3(NT), lds, st, a);      \
        else if (g8) hipLaunchKernelGGL((layernorm_i8_v2_kernel<NJv, COMPAT, 8>), dim3(grid), dim3(NT), lds, st, a); \
        else hipLaunchKernelGGL((layernorm_i8_v2_kernel<NJv, COMPAT, 16>), dim3(grid), dim3(NT), lds, st, a);        \
    } while (0)
    if (nj <= 1) IVIT_LN_V2(1);
    else if (nj <= 2) IVIT_LN_V2(2);
    else if (nj <= 3) IVIT_LN_V2(3);
    else IVIT_LN_V2(4);
#undef IVIT_LN_V2
    IVIT_CHECK_LAUNCH(who);
}

IVIT_EXPORT int ivit_tile_operand_i8(const int8_t* src, int64_t ld, int64_t rows, int K, int8_t* dst, ivit_stream_t stream)
{
    IVIT_REQUIRE(src && dst && rows > 0 && K > 0 && K % 64 == 0 && ld >= K && ld % 16 == 0 && ((uintptr_t)src % 16 == 0) &&
                     ((uintptr_t)dst % 16 == 0),
                 "ivit_tile_operand_i8: bad operand (K must be a multiple of 64, rows 16-byte aligned)");
    hipLaunchKernelGGL(relayout_kernel<true>, dim3(ew_grid(rows * (K >> 4))), dim3(NT), 0, ivit_stream(stream), src, dst, ld, rows,
                       K);
    IVIT_CHECK_LAUNCH("ivit_tile_operand_i8");
}

IVIT_EXPORT int ivit_untile_operand_i8(const int8_t* src, int64_t rows, int K, int8_t* dst, int64_t ld, ivit_stream_t stream)
{
    IVIT_REQUIRE(src && dst && rows > 0 && K > 0 && K % 64 == 0 && ld >= K && ld % 16 == 0 && ((uintptr_t)src % 16 == 0) &&
                     ((uintptr_t)dst % 16 == 0),
                 "ivit_untile_operand_i8: bad operand (K must be a multiple of 64, rows 16-byte aligned)");
    hipLaunchKernelGGL(relayout_kernel<false>, dim3(ew_grid(rows * (K >> 4))), dim3(NT), 0, ivit_stream(stream), src, dst, ld,
                       rows, K);
    IVIT_CHECK_LAUNCH("ivit_untile_operand_i8");
}

#if IVIT_LAB
IVIT_EXPORT int ivit_debug_ln_wave_per_row(int on)
{
    g_ln_wave_per_row = on;
    return IVIT_OK;
}

IVIT_EXPORT int ivit_debug_ln_ablate(int bits)
{
    g_ln_ablate = bits;
    return IVIT_OK;
}

IVIT_EXPORT int ivit_debug_ln_stream_cfg(int cfg)
{
    g_ln_stream_cfg = cfg;
    return IVIT_OK;
}

IVIT_EXPORT int ivit_debug_ln_stamp_buffer(void* buf)
{
    g_ln_stamps = reinterpret_cast<unsigned long long*>(buf);
    return IVIT_OK;
}
#endif

IVIT_EXPORT int ivit_layernorm_i8_ex(const int8_t* x, int64_t ldx, int rows, int C, const float* bias_int,
                                  const float* s_ln, const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo,
                                     int out_blocks, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && bias_int && s_ln && m && e, "ivit_layernorm_i8: NULL operand");
    IVIT_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 4096, "ivit_layernorm_i8: rows=%d C=%d unsupported", rows, C);
    IVIT_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= C && ldo >= C && ((uintptr_t)x % 4 == 0) &&
                     ((uintptr_t)out % 4 == 0),
                 "ivit_layernorm_i8: rows must be 4-byte aligned");
    IVIT_REQUIRE(((uintptr_t)bias_int % 16 == 0) && ((uintptr_t)s_ln % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                     ((uintptr_t)e % 16 == 0),
                 "ivit_layernorm_i8: per-channel tables must be 16-byte aligned");
    IVIT_REQUIRE(out_blocks == 0 || (out_blocks == 1 && C % 64 == 0 && ldo == C && ((uintptr_t)out % 16 == 0) &&
                                     ((int64_t)rows + 15) * C < 2147483648ll),
                 "ivit_layernorm_i8_ex: block-layout output needs C %% 64 == 0, ldo == C and a buffer below 2 GiB");
    LnArgs a{x, ldx, rows, C, bias_int, s_ln, m, e, out, ldo, out_blocks, nullptr, nullptr, g_ln_ablate, 0};
#if IVIT_LAB
    a.stamps = g_ln_stamps;
#endif
    hipStream_t st = ivit_stream(stream);
    // v2 where a row fills the wave (measured at 50 432 rows: C = 768 28.1 us against 30.3; C = 384 19.0 against 17.5 for the
    // half-wave form; at C = 96 -- Swin's patch norm -- only 24 of 64 lanes would hold data)
    // the streaming kernel (ln_stream.h) for the ViT widths from ~12 MB of rows (below, its fixed costs -- four workgroups per
    // CU, the constants table before the first row -- lose against the kernels below: 12 608 x 384 10.5 vs 8.7 us, r04e);
    // lab form 3: never, lab form 4: wherever it applies
    if (((g_ln_wave_per_row == 0 && ln_stream_pays(a)) || g_ln_wave_per_row == 4) && ln_stream_takes(a))
        return launch_ln_stream<false>(a, st, "ivit_layernorm_i8", g_ln_stream_cfg);
    if (C >= 512 && C <= 1024 && (g_ln_wave_per_row == 0 || g_ln_wave_per_row >= 3)) return launch_ln_v2<false>(a, st, "ivit_layernorm_i8");
    // half a wave per row (constants in LDS, 3 * C floats) where it is the faster form: measured 17.5 vs 20.7 us at
    // C = 384 and 14.0 vs 14.9 us at C = 192, but 36 vs 30 us at C = 768 (rows = 50 432)
    if ((C <= 384 || g_ln_wave_per_row == 2) && C <= 1536 && g_ln_wave_per_row != 1) {
        const int nj2 = (C / 4 + 31) / 32;
        // row pairs per wave: 4 from 16 K rows (4 waves per SIMD busy either way), fewer below so that a small launch still fills the
        // chip (lab: ln_ablate bits 21-22 = 1 / 2 / 3 force 4 / 2 / 1)
        int g2 = rows > 16384 ? 4 : rows > 8192 ? 2 : 1;
        if (nj2 > 3) g2 = 4;
        if ((g_ln_ablate >> 21) & 3) g2 = nj2 > 3 ? 4 : 8 >> ((g_ln_ablate >> 21) & 3);
        int grid = grid_for_rows(rows, 2 * g2);
        if (grid > 1024) grid = 1024;          // 4 waves per SIMD resident
        const size_t lds = (size_t)3 * C * sizeof(float);
        // C <= 128 (Swin's patch norm, 401 408 rows of 96): one dword per lane -- the NJ = 2 form computes a second, fully masked one
        // -- and 8 row pairs per wave from 64 K rows (the row statistics of 16 rows in one pass of lanes 0-7)
        const bool one = nj2 == 1 && g2 == 4 && !(g_ln_ablate & (1u << 25));
        if (one && rows > 65536 && !(g_ln_ablate & (1u << 26))) {
            grid = grid_for_rows(rows, 16);
            if (grid > 1024) grid = 1024;
            hipLaunchKernelGGL((layernorm_i8_pair_kernel<1, 8>), dim3(grid), dim3(NT), lds, st, a);
        } else if (one) hipLaunchKernelGGL((layernorm_i8_pair_kernel<1, 4>), dim3(grid), dim3(NT), lds, st, a);
        else if (nj2 <= 2 && g2 == 1) hipLaunchKernelGGL((layernorm_i8_pair_kernel<2, 1>), dim3(grid), dim3(NT), lds, st, a);
        else if (nj2 <= 2 && g2 == 2) hipLaunchKernelGGL((layernorm_i8_pair_kernel<2, 2>), dim3(grid), dim3(NT), lds, st, a);
        else if (nj2 <= 3 && g2 == 1) hipLaunchKernelGGL((layernorm_i8_pair_kernel<3, 1>), dim3(grid), dim3(NT), lds, st, a);
        else if (nj2 <= 3 && g2 == 2) hipLaunchKernelGGL((layernorm_i8_pair_kernel<3, 2>), dim3(grid), dim3(NT), lds, st, a);
        else if (nj2 <= 2) hipLaunchKernelGGL(layernorm_i8_pair_kernel<2>, dim3(grid), dim3(NT), lds, st, a);
        else if (nj2 <= 3) hipLaunchKernelGGL(layernorm_i8_pair_kernel<3>, dim3(grid), dim3(NT), lds, st, a);
        else if (nj2 <= 6) hipLaunchKernelGGL(layernorm_i8_pair_kernel<6>, dim3(grid), dim3(NT), lds, st, a);
        else if (nj2 <= 8) hipLaunchKernelGGL(layernorm_i8_pair_kernel<8>, dim3(grid), dim3(NT), lds, st, a);
        else hipLaunchKernelGGL(layernorm_i8_pair_kernel<12>, dim3(grid), dim3(NT), lds, st, a);
        IVIT_CHECK_LAUNCH("ivit_layernorm_i8");
    }
    const int nj = (C / 4 + 63) / 64;
    // each wave sets up its per-channel constants (bias and the requant bracket, 12*NJ registers per lane) once: launch no more workgroups than
    // stay resident (256 CUs x waves/SIMD at the kernel's register count) and let them stride over the rows
    const int resident = 256 * (nj <= 3 ? 3 : nj <= 8 ? 2 : 1);     // = the kernel's __launch_bounds__ occupancy
    int grid = grid_for_rows(rows, nj <= 3 ? 8 : nj <= 4 ? 4 : 1);   // rows per wave and iteration: G of the kernel
    if (grid > resident) grid = resident;
    if (nj <= 1) hipLaunchKernelGGL(layernorm_i8_kernel<1>, dim3(grid), dim3(NT), 0, st, a);
    else if (nj <= 2) hipLaunchKernelGGL(layernorm_i8_kernel<2>, dim3(grid), dim3(NT), 0, st, a);
    else if (nj <= 3) hipLaunchKernelGGL(layernorm_i8_kernel<3>, dim3(grid), dim3(NT), 0, st, a);
    else if (nj <= 4) hipLaunchKernelGGL(layernorm_i8_kernel<4>, dim3(grid), dim3(NT), 0, st, a);
    else if (nj <= 8) hipLaunchKernelGGL(layernorm_i8_kernel<8>, dim3(grid), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(layernorm_i8_kernel<16>, dim3(grid), dim3(NT), 0, st, a);
    IVIT_CHECK_LAUNCH("ivit_layernorm_i8");
}

IVIT_EXPORT int ivit_layernorm_i8_compat(const int8_t* x, int64_t ldx, int rows, int C, const float* bias_int,
                                         const float* s_ln, const uint32_t* m, const int32_t* e, const int8_t* remap,
                                         const float* phi, int8_t* out, int64_t ldo, int flags, ivit_stream_t stream)
{
    const int out_blocks = flags & 1, outer = flags >> 8;     // IVIT_LN_OUT_BLOCKS, IVIT_LN_OUTER_MEAN(L)
    IVIT_REQUIRE((flags & 0xfe) == 0 && outer >= 0 && (outer == 0 || rows % outer == 0), "ivit_layernorm_i8_compat: bad flags");
    IVIT_REQUIRE(x && out && bias_int && s_ln && m && e && remap && phi, "ivit_layernorm_i8_compat: NULL operand");
    IVIT_REQUIRE(rows > 0 && C >= 32 && C % 8 == 0 && C <= 4096, "ivit_layernorm_i8_compat: rows=%d C=%d unsupported", rows, C);
    IVIT_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= C && ldo >= C && ((uintptr_t)x % 4 == 0) &&
                     ((uintptr_t)out % 4 == 0),
                 "ivit_layernorm_i8_compat: rows must be 4-byte aligned");
    IVIT_REQUIRE(((uintptr_t)bias_int % 16 == 0) && ((uintptr_t)s_ln % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                     ((uintptr_t)e % 16 == 0) && ((uintptr_t)phi % 4 == 0),
                 "ivit_layernorm_i8_compat: per-channel tables must be 16-byte aligned");
    IVIT_REQUIRE(out_blocks == 0 || (out_blocks == 1 && C % 64 == 0 && ldo == C && ((uintptr_t)out % 16 == 0) &&
                                     ((int64_t)rows + 15) * C < 2147483648ll),
                 "ivit_layernorm_i8_compat: block-layout output needs C %% 64 == 0, ldo == C and a buffer below 2 GiB");
    LnArgs a{x, ldx, rows, C, bias_int, s_ln, m, e, out, ldo, out_blocks, remap, phi, g_ln_ablate, outer};
#if IVIT_LAB
    a.stamps = g_ln_stamps;
#endif
    hipStream_t st = ivit_stream(stream);
    // C < 512 as well (Swin's patch norm, C = 96, 401 408 rows at batch 128: 255 us in the one-row-per-wave kernel below, whose waves
    // are three-quarters idle at that width): the grouped kernel keeps 8 / 16 rows per wave in flight whatever the width
    if (((g_ln_wave_per_row == 0 && ln_stream_pays(a)) || g_ln_wave_per_row == 4) && ln_stream_takes(a))
        return launch_ln_stream<true>(a, st, "ivit_layernorm_i8_compat", g_ln_stream_cfg);
    if (C <= 1024 && (g_ln_wave_per_row == 0 || g_ln_wave_per_row >= 3)) return launch_ln_v2<true>(a, st, "ivit_layernorm_i8_compat");
    const int nj = (C / 4 + 63) / 64;
    const int resident = 256 * (nj <= 8 ? 2 : 1);                   // = the kernel's __launch_bounds__ occupancy
    int grid = grid_for_rows(rows, nj <= 3 ? 8 : nj <= 4 ? 4 : 1);
    if (grid > resident) grid = resident;
    if (nj <= 1) hipLaunchKernelGGL((layernorm_i8_kernel<1, true>), dim3(grid), dim3(NT), 0, st, a);
    else if (nj <= 2) hipLaunchKernelGGL((layernorm_i8_kernel<2, true>), dim3(grid), dim3(NT), 0, st, a);
    else if (nj <= 3) hipLaunchKernelGGL((layernorm_i8_kernel<3, true>), dim3(grid), dim3(NT), 0, st, a);
    else if (nj <= 4) hipLaunchKernelGGL((layernorm_i8_kernel<4, true>), dim3(grid), dim3(NT), 0, st, a);
    else if (nj <= 8) hipLaunchKernelGGL((layernorm_i8_kernel<8, true>), dim3(grid), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((layernorm_i8_kernel<16, true>), dim3(grid), dim3(NT), 0, st, a);
    IVIT_CHECK_LAUNCH("ivit_layernorm_i8_compat");
}

IVIT_EXPORT int ivit_layernorm_i8(const int8_t* x, int64_t ldx, int rows, int C, const float* bias_int,
                                  const float* s_ln, const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo,
                                  ivit_stream_t stream)
{
    return ivit_layernorm_i8_ex(x, ldx, rows, C, bias_int, s_ln, m, e, out, ldo, 0, stream);
}

IVIT_EXPORT int ivit_layernorm_i32_f32(const int32_t* x, int64_t ldx, int rows, int C, const float* bias_int,
                                       const float* s_ln, float* out, int64_t ldo, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && bias_int && s_ln, "ivit_layernorm_i32_f32: NULL operand");
    IVIT_REQUIRE(rows > 0 && C > 0 && C <= 4096 && ldx >= C && ldo >= C, "ivit_layernorm_i32_f32: bad shape");
    LnArgs a{x, ldx, rows, C, bias_int, s_ln, nullptr, nullptr, out, ldo, 0, nullptr, nullptr, 0, 0};
#if IVIT_LAB
    a.stamps = nullptr;
#endif
    hipLaunchKernelGGL(layernorm_i32_f32_kernel, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_layernorm_i32_f32");
}

IVIT_EXPORT int ivit_shiftgelu_i8(const int8_t* x, int64_t ldx, int rows, int L, float s, uint32_t m, int32_t e,
                                  int8_t* out, int64_t ldo, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && rows > 0 && L > 0 && ldx >= L && ldo >= L, "ivit_shiftgelu_i8: bad operand");
    GeluArgs a{};
    a.x = x; a.ldx = ldx; a.rows = rows; a.L = L; a.out = out; a.ldo = ldo;
    a.Mq = ivit_dyadic_to_double(m, e);
    IVIT_REQUIRE(a.Mq < 65536.0, "ivit_shiftgelu_i8: requant multiplier too large");
    int rc = gelu_x0(s, "ivit_shiftgelu_i8", &a.x0);
    if (rc) return rc;
    hipLaunchKernelGGL(shiftgelu_kernel<false>, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_shiftgelu_i8");
}

IVIT_EXPORT int ivit_shiftgelu_i8_i32(const int8_t* x, int64_t ldx, int rows, int L, float s, int32_t* out,
                                      int64_t ldo, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && rows > 0 && L > 0 && ldx >= L && ldo >= L, "ivit_shiftgelu_i8_i32: bad operand");
    GeluArgs a{};
    a.x = x; a.ldx = ldx; a.rows = rows; a.L = L; a.out = out; a.ldo = ldo; a.Mq = 0.0;
    int rc = gelu_x0(s, "ivit_shiftgelu_i8_i32", &a.x0);
    if (rc) return rc;
    hipLaunchKernelGGL(shiftgelu_kernel<true>, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_shiftgelu_i8_i32");
}

IVIT_EXPORT int ivit_shiftgelu_build_lut(float s, uint32_t m, int32_t e, int8_t* lut, ivit_stream_t stream)
{
    return ivit_shiftgelu_build_lut_ex(s, m, e, nullptr, lut, stream);
}

IVIT_EXPORT int ivit_shiftgelu_build_lut_ex(float s, uint32_t m, int32_t e, const int8_t* remap, int8_t* lut,
                                            ivit_stream_t stream)
{
    IVIT_REQUIRE(lut, "ivit_shiftgelu_build_lut: NULL table");
    GeluArgs a{};
    a.out = lut;
    a.remap = remap;
    a.Mq = ivit_dyadic_to_double(m, e);
    IVIT_REQUIRE(a.Mq < 65536.0, "ivit_shiftgelu_build_lut: requant multiplier too large");
    int rc = gelu_x0(s, "ivit_shiftgelu_build_lut", &a.x0);
    if (rc) return rc;
    hipLaunchKernelGGL(shiftgelu_lut_kernel, dim3(65536 / NT), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_shiftgelu_build_lut");
}

// Short rows (L <= 384: Swin stage 0, 401 408 rows of 384 bytes per launch at batch 128): HALF a wave per row, four rows per wave and
// iteration.  With a whole wave per row only 96 of 2 x 192 lane-dwords carry data at L = 384, every row pays its own 256-byte table
// slice and reduction chain, and the launch ran at 3.4 TB/s (90 us) where the wide rows of the ViT MLP reach 5.9 (round 4).
// Maximum over the 32 lanes of a half wave in every lane (DPP inside the rows of 16, v_permlane16_swap across them: half_wave_allreduce)
IVIT_DEV int half_wave_allmax(int v)
{
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));   // row_mirror
    const v2u_ r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)r.x, (int)r.y);
}

// PIPE (lab bit 28; measured 57.2 us against 55.3 without at 401 408 rows of 384, so not the product form): the rows of the NEXT
// iteration are requested right after this iteration's table slices -- the slices are the older loads, so the wait for them leaves the
// rows in flight (loads return in order).  What the launch was missing were unconditional loads (shiftgelu_lut_apply_kernel): 72.8 ->
// 55-57 us (5.4-5.6 TB/s) with them, eight waves per SIMD hide the chain of an iteration without a prefetch.
template <int NJ, bool INB, bool PIPE = false>
__global__ __launch_bounds__(NT) void shiftgelu_lut_apply_half_kernel(GeluArgs a)
{
    constexpr int RP = 2;                      // row pairs per wave and iteration
    __shared__ __attribute__((aligned(16))) unsigned char tab[WPB][2 * RP][256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int nd = a.L >> 2;
    const int stride = gridDim.x * WPB * (2 * RP);
    int w[RP][NJ];
    auto load_rows = [&](int (&dst)[RP][NJ], int row0) {
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            const int row = min(row0 + 2 * p + half, a.rows - 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int d = l32 + 32 * j, dd = min(d, nd - 1);      // unconditional loads: see shiftgelu_lut_apply_kernel
                int v;
                if constexpr (INB) v = *reinterpret_cast<const int*>(a.x + block_off(block_row(row, a.L), block_col(4 * dd)));
                else v = reinterpret_cast<const int*>(a.x + (int64_t)row * a.ldx)[dd];
                dst[p][j] = (d < nd) ? v : (int)0x80808080;
            }
        }
    };
    const int row_first = (blockIdx.x * WPB + wave) * (2 * RP);
    if (PIPE && row_first < a.rows) load_rows(w, row_first);
    for (int row0 = row_first; row0 < a.rows; row0 += stride) {
        if (!PIPE) load_rows(w, row0);
        int kmax[RP];
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            kmax[p] = -128;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                kmax[p] = max(max(kmax[p], sx8(w[p][j], 0)), max(sx8(w[p][j], 1), max(sx8(w[p][j], 2), sx8(w[p][j], 3))));
        }
#pragma unroll
        for (int p = 0; p < RP; ++p) kmax[p] = half_wave_allmax(kmax[p]);
        int2 slice[RP];
#pragma unroll
        for (int p = 0; p < RP; ++p)           // each half wave fetches its row's 256-byte slice: 8 bytes per lane
            slice[p] = reinterpret_cast<const int2*>(a.lut + (int64_t)(kmax[p] + 128) * 256)[l32];
        int wn[RP][NJ];
        if (PIPE) load_rows(wn, row0 + stride);      // unconditional (rows clamp to the last one: the final iteration's prefetch is not used)
#pragma unroll
        for (int p = 0; p < RP; ++p) reinterpret_cast<int2*>(tab[wave][2 * p + half])[l32] = slice[p];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): the table slices are in LDS
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            const int row = row0 + 2 * p + half;
            const unsigned char* tb = tab[wave][2 * p + half];
            unsigned o[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {        // the gathers of the whole row first (every byte indexes inside the slice), then the stores
                const unsigned v = (unsigned)w[p][j] ^ 0x80808080u;      // k + 128 per byte
                o[j] = (unsigned)tb[v & 255] | ((unsigned)tb[(v >> 8) & 255] << 8) | ((unsigned)tb[(v >> 16) & 255] << 16) | ((unsigned)tb[v >> 24] << 24);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int d = l32 + 32 * j;
                if (d < nd && row < a.rows) {
                    const int64_t off = a.out_blocks ? (int64_t)block_off(block_row(row, a.L), block_col(4 * d)) : (int64_t)row * a.ldo + 4 * d;
                    *reinterpret_cast<int*>(reinterpret_cast<int8_t*>(a.out) + off) = (int)o[j];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (PIPE) {
#pragma unroll
            for (int p = 0; p < RP; ++p)
#pragma unroll
                for (int j = 0; j < NJ; ++j) w[p][j] = wn[p][j];
        }
    }
}

IVIT_EXPORT int ivit_shiftgelu_lut_i8_ex(const int8_t* x, int64_t ldx, int rows, int L, const int8_t* lut, int8_t* out,
                                         int64_t ldo, int layouts, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && lut, "ivit_shiftgelu_lut_i8: NULL operand");
    IVIT_REQUIRE(rows > 0 && L > 0 && L % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= L && ldo >= L &&
                     ((uintptr_t)x % 4 == 0) && ((uintptr_t)out % 4 == 0) && ((uintptr_t)lut % 4 == 0),
                 "ivit_shiftgelu_lut_i8: rows=%d L=%d must be 4-byte aligned rows", rows, L);
    GeluArgs a{};
    const int out_blocks = layouts & 1, in_blocks = (layouts >> 1) & 1;   // bit 0: output, bit 1: input in the block layout
    IVIT_REQUIRE((layouts & ~3) == 0 && (!in_blocks || (L % 64 == 0 && ldx == L)) && (x != out || in_blocks == out_blocks),
                 "ivit_shiftgelu_lut_i8_ex: bad layouts (in place needs the same layout on both sides)");
    IVIT_REQUIRE(out_blocks == 0 || (out_blocks == 1 && L % 64 == 0 && ldo == L && ((uintptr_t)out % 16 == 0) &&
                                     ((int64_t)rows + 15) * L < 2147483648ll),
                 "ivit_shiftgelu_lut_i8_ex: block-layout output needs L %% 64 == 0, ldo == L and a buffer below 2 GiB");
    a.x = x; a.ldx = ldx; a.rows = rows; a.L = L; a.lut = lut; a.out = out; a.ldo = ldo; a.out_blocks = out_blocks; a.in_blocks = in_blocks;
    const dim3 grid(grid_for_rows(rows, 2)), blk(NT);
    hipStream_t st = ivit_stream(stream);
    const int nj = (L / 4 + 63) / 64;
    if (L <= 384 && rows >= 4096 && !(g_ln_ablate & (1 << 24))) {      // lab bit 24: the whole-wave-per-row form (A/B, parity of both)
        const dim3 gridh(grid_for_rows(rows, 4));
#define IVIT_GELU_HALF(NJ_, ...) do { if (in_blocks) hipLaunchKernelGGL((shiftgelu_lut_apply_half_kernel<NJ_, true, ##__VA_ARGS__>), gridh, blk, 0, st, a); \
                                      else hipLaunchKernelGGL((shiftgelu_lut_apply_half_kernel<NJ_, false, ##__VA_ARGS__>), gridh, blk, 0, st, a); } while (0)
        if (L <= 128) IVIT_GELU_HALF(1);
        else if (L <= 256) IVIT_GELU_HALF(2);
        else if (IVIT_LAB && (g_ln_ablate & (1 << 28))) IVIT_GELU_HALF(3, true);   // lab A/B: with the prefetch of the next iteration's rows
        else IVIT_GELU_HALF(3);
#undef IVIT_GELU_HALF
        IVIT_CHECK_LAUNCH("ivit_shiftgelu_lut_i8");
    }
#define IVIT_GELU_WAVE(NJ_) do { if (in_blocks) hipLaunchKernelGGL((shiftgelu_lut_apply_kernel<NJ_, true>), grid, blk, 0, st, a); \
                                 else hipLaunchKernelGGL((shiftgelu_lut_apply_kernel<NJ_, false>), grid, blk, 0, st, a); } while (0)
    if (nj <= 3) IVIT_GELU_WAVE(3);
    else if (nj <= 6) IVIT_GELU_WAVE(6);
    else if (nj <= 12) IVIT_GELU_WAVE(12);
    else hipLaunchKernelGGL(shiftgelu_lut_apply_kernel<0>, grid, blk, 0, st, a);
#undef IVIT_GELU_WAVE
    IVIT_CHECK_LAUNCH("ivit_shiftgelu_lut_i8");
}

IVIT_EXPORT int ivit_shiftgelu_lut_i8(const int8_t* x, int64_t ldx, int rows, int L, const int8_t* lut, int8_t* out,
                                      int64_t ldo, ivit_stream_t stream)
{
    return ivit_shiftgelu_lut_i8_ex(x, ldx, rows, L, lut, out, ldo, 0, stream);
}

template <typename TX>
static int launch_shiftmax(const char* who, const TX* x, int64_t ldx, int rows, int L, float s, int8_t* out, int64_t ldo,
                           ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && rows > 0 && L > 1 && ldx >= L && ldo >= L, "%s: bad operand (L must be > 1)", who);
    IVIT_REQUIRE(s > 0.0f, "%s: scale must be positive", who);
    const float x0f = __builtin_floorf((1.0f / s) * -1.0f);
    IVIT_REQUIRE(x0f <= -1.0f && x0f >= -65535.0f, "%s: x0=%g outside [-65535,-1]", who, (double)x0f);
    SmArgs<TX> a{x, ldx, rows, L, (int)x0f, out, ldo};
    hipLaunchKernelGGL(shiftmax_kernel<TX>, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH(who);
}

IVIT_EXPORT int ivit_shiftmax_i8(const int8_t* x, int64_t ldx, int rows, int L, float s, int8_t* out, int64_t ldo,
                                 ivit_stream_t stream)
{
    return launch_shiftmax<int8_t>("ivit_shiftmax_i8", x, ldx, rows, L, s, out, ldo, stream);
}

IVIT_EXPORT int ivit_shiftmax_i32_i8(const int32_t* x, int64_t ldx, int rows, int L, float s, int8_t* out, int64_t ldo,
                                     ivit_stream_t stream)
{
    return launch_shiftmax<int32_t>("ivit_shiftmax_i32_i8", x, ldx, rows, L, s, out, ldo, stream);
}

IVIT_EXPORT int ivit_minmax_f32(const float* x, int64_t n, float* out_min_max, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out_min_max && n > 0, "ivit_minmax_f32: bad operand");
    IVIT_REQUIRE(((uintptr_t)out_min_max % 4 == 0) && ((uintptr_t)x % 4 == 0), "ivit_minmax_f32: misaligned");
    hipStream_t st = ivit_stream(stream);
    unsigned* keys = reinterpret_cast<unsigned*>(out_min_max);
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, st, keys);
    hipLaunchKernelGGL(minmax_kernel, dim3(ew_grid(n) < 2048 ? ew_grid(n) : 2048), dim3(NT), 0, st, x, n, keys);
    hipLaunchKernelGGL(minmax_finish_kernel, dim3(1), dim3(1), 0, st, keys);
    IVIT_CHECK_LAUNCH("ivit_minmax_f32");
}

IVIT_EXPORT int ivit_quantize_input_f32_i8(const float* x, int8_t* out, int64_t n, float inv_scale,
                                           ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && n > 0, "ivit_quantize_input_f32_i8: bad operand");
    IVIT_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 4 == 0), "ivit_quantize_input_f32_i8: misaligned");
    hipLaunchKernelGGL(quantize_kernel, dim3(ew_grid(n / 4 + 1)), dim3(NT), 0, ivit_stream(stream), x, out, n,
                       inv_scale);
    IVIT_CHECK_LAUNCH("ivit_quantize_input_f32_i8");
}

IVIT_EXPORT int ivit_quantize_input_f32_i32(const float* x, int32_t* out, int64_t n, float inv_scale, int bits,
                                            ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && n > 0 && bits >= 2 && bits <= 32, "ivit_quantize_input_f32_i32: bad operand");
    const float lo = -__builtin_ldexpf(1.0f, bits - 1), hi = __builtin_ldexpf(1.0f, bits - 1) - 1.0f;   // float32, as torch.clamp
    hipLaunchKernelGGL(quantize_i32_kernel, dim3(ew_grid(n)), dim3(NT), 0, ivit_stream(stream), x, out, n, inv_scale, lo, hi);
    IVIT_CHECK_LAUNCH("ivit_quantize_input_f32_i32");
}

static int launch_patchify(const char* who, const float* img, int8_t* A, int64_t lda, int batch, int chans, int hw, int patch,
                           float inv_scale, ivit_stream_t stream)
{
    IVIT_REQUIRE(img && A && batch > 0 && chans > 0, "%s: bad operand", who);
    IVIT_REQUIRE(patch > 0 && patch % 4 == 0 && hw % patch == 0, "%s: hw=%d patch=%d", who, hw, patch);
    IVIT_REQUIRE(lda >= (int64_t)chans * patch * patch && lda % 4 == 0, "%s: lda=%lld too small or not a multiple of 4", who,
                 (long long)lda);
    IVIT_REQUIRE(((uintptr_t)img % 16 == 0) && ((uintptr_t)A % 4 == 0), "%s: misaligned", who);
    const int64_t total = (int64_t)batch * chans * hw * (hw / 4);
    hipLaunchKernelGGL(patchify_kernel, dim3(ew_grid(total)), dim3(NT), 0, ivit_stream(stream), img, A, lda, batch, chans,
                       hw, patch, inv_scale);
    IVIT_CHECK_LAUNCH(who);
}

IVIT_EXPORT int ivit_quantize_patchify_f32_i8(const float* img, int8_t* A, int batch, int chans, int hw, int patch,
                                              float inv_scale, ivit_stream_t stream)
{
    return launch_patchify("ivit_quantize_patchify_f32_i8", img, A, (int64_t)chans * patch * patch, batch, chans, hw, patch,
                           inv_scale, stream);
}

IVIT_EXPORT int ivit_quantize_patchify_u8_i8(const uint8_t* img, int8_t* A, int64_t lda, int batch, int chans, int hw, int patch,
                                             const int8_t* lut, ivit_stream_t stream)
{
    IVIT_REQUIRE(img && A && lut && batch > 0 && chans > 0 && chans <= 4 && hw > 0 && patch > 0 && hw % patch == 0 && patch % 4 == 0,
                 "ivit_quantize_patchify_u8_i8: bad shape (up to 4 channels, hw %% patch == 0, patch %% 4 == 0)");
    IVIT_REQUIRE(lda >= (int64_t)chans * patch * patch && lda % 4 == 0 && ((uintptr_t)img % 4 == 0) && ((uintptr_t)A % 4 == 0),
                 "ivit_quantize_patchify_u8_i8: lda too small or misaligned operand");
    const int64_t total = (int64_t)batch * chans * hw * (hw / 4);
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(ew_grid(total)), dim3(NT), 0, ivit_stream(stream), img, A, lda, batch, chans, hw,
                       patch, lut);
    IVIT_CHECK_LAUNCH("ivit_quantize_patchify_u8_i8");
}

IVIT_EXPORT int ivit_quantize_patchify_ld_f32_i8(const float* img, int8_t* A, int64_t lda, int batch, int chans, int hw,
                                                 int patch, float inv_scale, ivit_stream_t stream)
{
    return launch_patchify("ivit_quantize_patchify_ld_f32_i8", img, A, lda, batch, chans, hw, patch, inv_scale, stream);
}

IVIT_EXPORT int ivit_embed_assemble_i8(const int8_t* patch, const int16_t* pos_add, const int8_t* cls_row, uint32_t m,
                                       int32_t e, int8_t* out, int batch, int tokens, int C, ivit_stream_t stream)
{
    IVIT_REQUIRE(patch && pos_add && cls_row && out, "ivit_embed_assemble_i8: NULL operand");
    IVIT_REQUIRE(batch > 0 && tokens > 1 && C > 0 && C % 4 == 0, "ivit_embed_assemble_i8: bad shape");
    IVIT_REQUIRE(((uintptr_t)patch % 4 == 0) && ((uintptr_t)out % 4 == 0) && ((uintptr_t)cls_row % 4 == 0) &&
                     ((uintptr_t)pos_add % 8 == 0),
                 "ivit_embed_assemble_i8: misaligned");
    const double Mq = ivit_dyadic_to_double(m, e);
    IVIT_REQUIRE(Mq < 1048576.0, "ivit_embed_assemble_i8: requant multiplier too large");
    const int64_t total = (int64_t)batch * tokens * (C / 4);
    hipLaunchKernelGGL(embed_kernel, dim3(ew_grid(total)), dim3(NT), 0, ivit_stream(stream), patch, pos_add, cls_row,
                       Mq, out, batch, tokens, C);
    IVIT_CHECK_LAUNCH("ivit_embed_assemble_i8");
}

IVIT_EXPORT int ivit_embed_assemble_i16(const int16_t* patch, const int32_t* pos_add, const int16_t* cls_row, uint32_t m,
                                        int32_t e, int16_t* out, int batch, int tokens, int C, ivit_stream_t stream)
{
    IVIT_REQUIRE(patch && pos_add && cls_row && out, "ivit_embed_assemble_i16: NULL operand");
    IVIT_REQUIRE(batch > 0 && tokens > 1 && C > 0 && C % 4 == 0, "ivit_embed_assemble_i16: bad shape");
    IVIT_REQUIRE(((uintptr_t)patch % 8 == 0) && ((uintptr_t)out % 8 == 0) && ((uintptr_t)cls_row % 8 == 0) &&
                     ((uintptr_t)pos_add % 16 == 0),
                 "ivit_embed_assemble_i16: misaligned");
    const double Mq = ivit_dyadic_to_double(m, e);
    IVIT_REQUIRE(Mq < 32768.0, "ivit_embed_assemble_i16: requant multiplier too large");
    const int64_t total = (int64_t)batch * tokens * (C / 4);
    hipLaunchKernelGGL(embed16_kernel, dim3(ew_grid(total)), dim3(NT), 0, ivit_stream(stream), patch, pos_add, cls_row,
                       Mq, out, batch, tokens, C);
    IVIT_CHECK_LAUNCH("ivit_embed_assemble_i16");
}

IVIT_EXPORT int ivit_head_argmax(const int32_t* acc, const float* s_acc, int batch, int N, float* logits_f32,
                                 int32_t* top1, ivit_stream_t stream)
{
    IVIT_REQUIRE(acc && s_acc && top1 && batch > 0 && N > 0, "ivit_head_argmax: bad operand");
    hipLaunchKernelGGL(head_argmax_kernel, dim3(grid_for_rows(batch)), dim3(NT), 0, ivit_stream(stream), acc, s_acc,
                       batch, N, logits_f32, top1);
    IVIT_CHECK_LAUNCH("ivit_head_argmax");
}

IVIT_EXPORT int ivit_requant_i32(const int32_t* z, int64_t rows, int C, const uint32_t* m, const int32_t* e, int n_me,
                                 const int32_t* z2, const uint32_t* m2, const int32_t* e2, int n_me2, int bits,
                                 int32_t* out, ivit_stream_t stream)
{
    IVIT_REQUIRE(z && m && e && out && rows > 0 && C > 0, "ivit_requant_i32: bad operand");
    IVIT_REQUIRE(n_me == 1 || n_me == C, "ivit_requant_i32: n_me=%d must be 1 or C=%d", n_me, C);
    IVIT_REQUIRE(z2 == nullptr || (m2 && e2 && (n_me2 == 1 || n_me2 == C)), "ivit_requant_i32: bad identity branch");
    IVIT_REQUIRE(bits == 8 || bits == 16 || bits == 32 || bits == 4, "ivit_requant_i32: bits=%d", bits);
    hipLaunchKernelGGL(requant_i32_kernel, dim3(ew_grid(rows * C)), dim3(NT), 0, ivit_stream(stream), z, rows, C, m, e,
                       n_me, z2, m2, e2, n_me2, bits, out);
    IVIT_CHECK_LAUNCH("ivit_requant_i32");
}

IVIT_EXPORT int ivit_residual_requant_i8(const int8_t* a, uint32_t m_a, int32_t e_a, const int8_t* b, uint32_t m_b,
                                         int32_t e_b, int8_t* out, int64_t n, ivit_stream_t stream)
{
    IVIT_REQUIRE(a && b && out && n > 0, "ivit_residual_requant_i8: bad operand");
    const double Ma = ivit_dyadic_to_double(m_a, e_a), Mb = ivit_dyadic_to_double(m_b, e_b);
    IVIT_REQUIRE(Ma < 1048576.0 && Mb < 1048576.0, "ivit_residual_requant_i8: multiplier too large");
    hipLaunchKernelGGL(residual_requant_kernel, dim3(ew_grid(n)), dim3(NT), 0, ivit_stream(stream), a, Ma, b, Mb, out,
                       n);
    IVIT_CHECK_LAUNCH("ivit_residual_requant_i8");
}

IVIT_EXPORT int ivit_bgemm_qk_i8(const int8_t* Q, const int8_t* K, int32_t* S, int batch, int Tq, int Tk, int D,
                                 ivit_stream_t stream)
{
    IVIT_REQUIRE(Q && K && S && batch > 0 && Tq > 0 && Tk > 0 && D > 0, "ivit_bgemm_qk_i8: bad operand");
    hipLaunchKernelGGL(bgemm_kernel<false>, dim3(ew_grid((int64_t)batch * Tq * Tk)), dim3(NT), 0, ivit_stream(stream),
                       Q, K, S, batch, Tq, Tk, D);
    IVIT_CHECK_LAUNCH("ivit_bgemm_qk_i8");
}

IVIT_EXPORT int ivit_bgemm_pv_i8(const int8_t* P, const int8_t* V, int32_t* O, int batch, int Tq, int Tk, int D,
                                 ivit_stream_t stream)
{
    IVIT_REQUIRE(P && V && O && batch > 0 && Tq > 0 && Tk > 0 && D > 0, "ivit_bgemm_pv_i8: bad operand");
    hipLaunchKernelGGL(bgemm_kernel<true>, dim3(ew_grid((int64_t)batch * Tq * D)), dim3(NT), 0, ivit_stream(stream), P,
                       V, O, batch, Tq, Tk, D);
    IVIT_CHECK_LAUNCH("ivit_bgemm_pv_i8");
}

IVIT_EXPORT int ivit_bgemm_pv_i16_i8(const int16_t* P, const int8_t* V, int32_t* O, int batch, int Tq, int Tk, int D,
                                     ivit_stream_t stream)
{
    IVIT_REQUIRE(P && V && O && batch > 0 && Tq > 0 && Tk > 0 && D > 0, "ivit_bgemm_pv_i16_i8: bad operand");
    IVIT_REQUIRE((int64_t)Tk * 32768 * 128 < 2147483648ll * 64, "ivit_bgemm_pv_i16_i8: Tk too large for int32 accumulation");
    hipLaunchKernelGGL((bgemm_kernel<true, int16_t>), dim3(ew_grid((int64_t)batch * Tq * D)), dim3(NT), 0, ivit_stream(stream),
                       P, V, O, batch, Tq, Tk, D);
    IVIT_CHECK_LAUNCH("ivit_bgemm_pv_i16_i8");
}

IVIT_EXPORT int ivit_bgemm_pv_i32_i8(const int32_t* P, const int8_t* V, int32_t* O, int batch, int Tq, int Tk, int D, int64_t p_absmax,
                                     ivit_stream_t stream)
{
    IVIT_REQUIRE(P && V && O && batch > 0 && Tq > 0 && Tk > 0 && D > 0, "ivit_bgemm_pv_i32_i8: bad operand");
    IVIT_REQUIRE(p_absmax >= 0 && p_absmax * 128 * Tk < 2147483648ll, "ivit_bgemm_pv_i32_i8: |P| <= %lld over Tk=%d keys overflows int32",
                 (long long)p_absmax, Tk);
    hipLaunchKernelGGL((bgemm_kernel<true, int32_t>), dim3(ew_grid((int64_t)batch * Tq * D)), dim3(NT), 0, ivit_stream(stream),
                       P, V, O, batch, Tq, Tk, D);
    IVIT_CHECK_LAUNCH("ivit_bgemm_pv_i32_i8");
}

IVIT_EXPORT int ivit_f32_to_i32(const float* x, int64_t rows, int C, const float* s, int n_s, int mode, int32_t* z,
                                ivit_stream_t stream)
{
    IVIT_REQUIRE(x && s && z && rows > 0 && C > 0 && (n_s == 1 || n_s == C) && (mode == 0 || mode == 1),
                 "ivit_f32_to_i32: bad operand");
    hipLaunchKernelGGL(f32_to_i32_kernel, dim3(ew_grid(rows * C)), dim3(NT), 0, ivit_stream(stream), x, rows, C, s, n_s,
                       mode, z);
    IVIT_CHECK_LAUNCH("ivit_f32_to_i32");
}

IVIT_EXPORT int ivit_i32_to_f32(const int32_t* z, int64_t rows, int C, const float* s, int n_s, float* y,
                                ivit_stream_t stream)
{
    IVIT_REQUIRE(z && s && y && rows > 0 && C > 0 && (n_s == 1 || n_s == C), "ivit_i32_to_f32: bad operand");
    hipLaunchKernelGGL(i32_to_f32_kernel, dim3(ew_grid(rows * C)), dim3(NT), 0, ivit_stream(stream), z, rows, C, s, n_s,
                       y);
    IVIT_CHECK_LAUNCH("ivit_i32_to_f32");
}

IVIT_EXPORT int ivit_narrow_i32_i8(const int32_t* z, int8_t* out, int64_t n, int32_t* overflow_flag,
                                   ivit_stream_t stream)
{
    IVIT_REQUIRE(z && out && n > 0, "ivit_narrow_i32_i8: bad operand");
    hipLaunchKernelGGL(narrow_i32_i8_kernel, dim3(ew_grid(n)), dim3(NT), 0, ivit_stream(stream), z, out, n,
                       overflow_flag);
    IVIT_CHECK_LAUNCH("ivit_narrow_i32_i8");
}
