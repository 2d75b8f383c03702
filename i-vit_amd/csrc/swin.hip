// swin.hip -- the additional integer kernels of the Swin path (config 5): 16-bit residual stream, windowed
// attention with relative-position bias and shift mask, patch merging, token average pooling.
// Reference: /root/reference/models/swin_quant.py (WindowAttention :121-169, SwinTransformerBlock :251-301,
// PatchMerging :328-349, SwinTransformer.forward_features :539-558); operators as in rowops.hip / attention.hip.
// The fork's swin_quant.py does not run as shipped (SURVEY.md finding 6); semantics are SURVEY Appendix A.8, pinned by
// tests/golden/swin_tiny.npz (generated from the reference with harness-side shims).
#include "common.h"
#include "rowsum.h"

#if IVIT_LAB
extern int g_ln_ablate;     // rowops.hip; bits 16-19: workgroup cap of the tiled 16-bit LayerNorm in units of 256 (0 = default);
                            // bit 20: natural-scale 16-bit LayerNorm sums its rows through LDS (the round-3 form);
                            // bit 23: window attention requantises its scores in float64 also where float32 is exact
#else
constexpr int g_ln_ablate = 0;
#endif

namespace {

constexpr int NT = 256;
constexpr int WPB = NT / 64;

static inline int ew_grid(int64_t n)
{
    int64_t b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

static inline int grid_for_rows(int64_t rows, int rpw = 1)
{
    int64_t blocks = (rows + WPB * rpw - 1) / (WPB * rpw);
    return (int)(blocks < 4096 ? blocks : 4096);
}

IVIT_DEV int pack4(int a, int b, int c, int d)
{
    return (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | ((d & 0xff) << 24);
}

// ------------------------------------------------------------------------------------------------
// Row maps: the window partition / cyclic shift of SwinTransformerBlock.forward (swin_quant.py:258-271, 278-289) are
// pure row permutations of a [B, H*W, C] tensor.  win_row(r) = index, in window-major order
// (image, window row, window col, row in window, col in window), of token r = (b, y, x) after rolling by -shift.
// ------------------------------------------------------------------------------------------------
struct WinMap {
    int H, W, ws, shift;  // ws == 0: identity
};

IVIT_DEV int64_t win_row(const WinMap& m, int64_t r)
{
    if (m.ws == 0) return r;
    const int L = m.H * m.W;
    const int b = (int)(r / L);
    const int t = (int)(r - (int64_t)b * L);
    int y = t / m.W, x = t - y * m.W;
    // shifted_x[y'] = x[(y' + shift) mod H]  =>  token (y, x) lands at y' = (y - shift) mod H
    y = y - m.shift; if (y < 0) y += m.H;
    x = x - m.shift; if (x < 0) x += m.W;
    const int wy = y / m.ws, iy = y - wy * m.ws, wx = x / m.ws, ix = x - wx * m.ws;
    const int nwx = m.W / m.ws;
    return ((int64_t)b * (m.H / m.ws) * nwx + (int64_t)wy * nwx + wx) * (m.ws * m.ws) + iy * m.ws + ix;
}

// the inverse: row R of the window-ordered tensor (window index, position in the window) -> row of the image-ordered one
// (window_reverse + the roll back, swin_quant.py:278-287)
IVIT_DEV int64_t win_row_inv(const WinMap& m, int64_t R)
{
    if (m.ws == 0) return R;
    const int T = m.ws * m.ws, nwx = m.W / m.ws, nwy = m.H / m.ws;
    const int64_t widx = R / T;
    const int pos = (int)(R - widx * T);
    const int iy = pos / m.ws, ix = pos - iy * m.ws;
    const int b = (int)(widx / (nwy * nwx)), wrem = (int)(widx - (int64_t)b * (nwy * nwx));
    const int wy = wrem / nwx, wx = wrem - wy * nwx;
    int y = wy * m.ws + iy + m.shift; if (y >= m.H) y -= m.H;
    int x = wx * m.ws + ix + m.shift; if (x >= m.W) x -= m.W;
    return (int64_t)b * m.H * m.W + (int64_t)y * m.W + x;
}

// ------------------------------------------------------------------------------------------------
// 8 -> 16 bit QuantAct (SwinTransformer.qact1, swin_quant.py:546; also used as an exact int8 -> int16 widening)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void requant_i8_i16_kernel(const int8_t* x, double Mq, int16_t* out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        out[i] = (int16_t)clamp_i32(requant_exact((int)x[i], Mq), -32768, 32767);
}

// ------------------------------------------------------------------------------------------------
// Two-operand 16-bit QuantAct of the residual connections (swin_quant.py:293, 299; quant_utils.py:232-245):
//   out[r] = clamp16(RNE(a[map(r)] * Ma) + RNE(res[r] * Mr));  a is int16 (attn.qact4) or int8 (mlp.qact2)
// ------------------------------------------------------------------------------------------------
template <typename TA>
__global__ __launch_bounds__(NT) void residual_i16_kernel(const TA* a, const uint32_t* m_pre, const int32_t* e_pre, double Ma,
                                                          const int16_t* res, double Mr, int16_t* out, int64_t rows, int C,
                                                          WinMap map)
{
    const int c4 = C >> 2;
    const int64_t total = rows * c4;
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        const int64_t r = q / c4;
        const int c = (int)(q - r * c4) * 4;
        const int64_t ra = win_row(map, r);
        int av[4];
        if (sizeof(TA) == 1) {
            const int w = *reinterpret_cast<const int*>(reinterpret_cast<const int8_t*>(a) + ra * C + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = (int)(int8_t)(w >> (8 * i));
        } else if (sizeof(TA) == 2) {
            const int2 w = *reinterpret_cast<const int2*>(reinterpret_cast<const int16_t*>(a) + ra * C + c);
            av[0] = (int)(int16_t)w.x; av[1] = w.x >> 16; av[2] = (int)(int16_t)w.y; av[3] = w.y >> 16;
        } else {  // raw GEMM accumulators: the 16-bit QuantAct behind the projection first (attn.qact4, swin_quant.py:166)
            const v4i w = *reinterpret_cast<const v4i*>(reinterpret_cast<const int32_t*>(a) + ra * C + c);
            const v4i mm = *reinterpret_cast<const v4i*>(m_pre + c);
            const v4i ee = *reinterpret_cast<const v4i*>(e_pre + c);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                av[i] = clamp_i32(requant_exact(w[i], dyadic_mult((uint32_t)mm[i], ee[i])), -32768, 32767);
        }
        const int2 rw = *reinterpret_cast<const int2*>(res + r * C + c);
        const int rv[4] = {(int)(int16_t)rw.x, rw.x >> 16, (int)(int16_t)rw.y, rw.y >> 16};
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = clamp_i32(requant_exact(av[i], Ma) + requant_exact(rv[i], Mr), -32768, 32767);
        int2 ow;
        ow.x = (o[0] & 0xffff) | (o[1] << 16);
        ow.y = (o[2] & 0xffff) | (o[3] << 16);
        *reinterpret_cast<int2*>(out + r * C + c) = ow;
    }
}

// ------------------------------------------------------------------------------------------------
// I-LayerNorm on the 16-bit stream + QuantAct(8) (+ optional row map on the OUTPUT: window partition)
// ivit_modules.py:30-65 with 16-bit inputs: var up to ~3.5e10 -> the Newton iteration runs on RN24(var) in float32
// with correctly rounded divisions, exactly as the reference does.
// ------------------------------------------------------------------------------------------------
struct Ln16Args {
    const int16_t* x;
    int rows, C;
    const float* bias_int;
    const float* s_ln;
    const uint32_t* m;
    const int32_t* e;
    int8_t* out;
    int64_t ldo;
    WinMap map;
    int outer;     // compat kernels, > 0: the reference takes the mean over a transposed view of contiguous extent `outer` (the first
                   // LayerNorm behind the Swin patch embedding, whose layout travels through the elementwise QuantActs): torch's
                   // outer-reduction order (rowsum.h torch_outer_rowsum), row = image * outer + column
    int lab_lds_sum;   // lab A/B: the row sums through LDS (the round-3 form) where the register form applies
};

__global__ __launch_bounds__(NT) void layernorm_i16_i8_kernel(Ln16Args a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const int16_t* xr = a.x + (int64_t)row * C;
        int sum = 0;  // |sum| < 2^31 for C <= 4096
        for (int c = lane; c < C; c += 64) sum += xr[c];
        sum = wave_reduce_sum_i32(sum);
        const float mean = (float)sum / (float)C;                    // :37 (RN24 of the exact sum, / C)
        const int mean_int = (int)rintf(mean);
        long long var = 0;
        for (int c = lane; c < C; c += 64) {
            long long d = (long long)xr[c] - mean_int;
            var += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int vlo = __shfl_xor((int)(var & 0xffffffffll), o);
            int vhi = __shfl_xor((int)(var >> 32), o);
            var += ((long long)vhi << 32) | (unsigned)vlo;
        }
        float varf = (float)var, t = 65536.0f;                       // :45-49
#pragma unroll 1
        for (int it = 0; it < 10; ++it) t = floorf((t + floorf(varf / t)) * 0.5f);
        const float factor = floorf((1.0f / t) * 2147483648.0f);     // :51
        int8_t* orow = a.out + win_row(a.map, row) * a.ldo;
        for (int c = lane; c < C; c += 64) {
            float dl = (float)(xr[c] - mean_int);
            float v = floorf((dl * factor) * 0.5f);                   // :52
            float y = v + a.bias_int[c];                              // :61
            float s = a.s_ln[c];
            float x = y * s;                                          // :63
            float z = rintf(x / s);                                   // quant_utils.py:220 (correctly rounded quotient)
            double p = (double)z * dyadic_mult(a.m[c], a.e[c]);       // :229
            double tt = p + IVIT_MAGIC;                               // :230
            orow[c] = (int8_t)clamp_i32((int)(unsigned)__double_as_longlong(tt), -128, 127);
        }
    }
}

// Natural (non power-of-two) scale of the 16-bit input: IVITIntLayerNorm sees phi(q) = fl(fl(q*s)/s) (QuantAct hands on q*s,
// quant_modules.py:387; ivit_modules.py:36 divides by s again).  Literal: the float32 mean over the phi values in torch's
// CPU reduction order (rowsum.h), `.to(int32)` truncation, then the usual chain.  One wave per row.
struct Ln16LitArgs {
    Ln16Args b;
    float s_in;
};

__global__ __launch_bounds__(NT) void layernorm_i16_i8_literal_kernel(Ln16LitArgs al)
{
    const Ln16Args& a = al.b;
    const float s_in = al.s_in;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const int16_t* xr = a.x + (int64_t)row * C;
        auto xint = [&](int c) { return ((float)xr[c] * s_in) / s_in; };       // :36 on the float view q * s
        const float S = a.outer ? torch_outer_rowsum(xint, C, row % a.outer >= (a.outer & ~31)) : torch_rowsum(xint, C, lane);
        const int mean_int = (int)rintf(S / (float)C);                         // :37
        long long var = 0;
        for (int c = lane; c < C; c += 64) {
            const long long d = (long long)(int)truncf(xint(c)) - mean_int;    // :38-40
            var += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int vlo = __shfl_xor((int)(var & 0xffffffffll), o);
            int vhi = __shfl_xor((int)(var >> 32), o);
            var += ((long long)vhi << 32) | (unsigned)vlo;
        }
        float varf = (float)var, t = 65536.0f;                                 // :45-49
#pragma unroll 1
        for (int it = 0; it < 10; ++it) t = floorf((t + floorf(varf / t)) * 0.5f);
        const float factor = floorf((1.0f / t) * 2147483648.0f);               // :51
        int8_t* orow = a.out + win_row(a.map, row) * a.ldo;
        for (int c = lane; c < C; c += 64) {
            float dl = (float)((int)truncf(xint(c)) - mean_int);
            float v = floorf((dl * factor) * 0.5f);                             // :52
            float y = v + a.bias_int[c];                                        // :61
            float s = a.s_ln[c];
            float x = y * s;                                                    // :63
            float z = rintf(x / s);                                             // quant_utils.py:220
            double p = (double)z * dyadic_mult(a.m[c], a.e[c]);                 // :229
            double tt = p + IVIT_MAGIC;                                         // :230
            orow[c] = (int8_t)clamp_i32((int)(unsigned)__double_as_longlong(tt), -128, 127);
        }
    }
}

// Round 4: the per-channel constants (bias, float32 bracket (lo, hi) of the QuantAct multiplier: the certificate of
// layernorm_i8_kernel, rowops.hip) of the tiled 16-bit kernels are derived ONCE PER WORKGROUP into LDS -- one or two channels per
// thread -- and read per chunk of 8 channels where the element chain needs them.  Rounds 1-3 kept them in 72 registers per lane
// (NJ = 3), derived by every lane for its 24 channels in float64: 195-244 VGPRs = two waves per SIMD, where this VALU-bound chain
// issues its half-rate instructions at 4.4 cycles instead of 3.2, and a prologue that a launch of few rows
// (Swin stages 2-3: 14-29 MB) spent most of its time in.
// The ten float32 Newton steps of ivit_modules.py:45-49 on varf = RN24(var) WITHOUT iterating, where that is provably the same
// (round 4; scripts/probes/ln16_newton_exhaustive.py walks every float32 value in [2^24, 2^37), ln_newton_exhaustive.py every
// integer below 2^24): t10 == floor(sqrt(varf)) unless varf lies within 2^(ex - 22) below the next square (then t10 is that or one
// more).  Those rows (<= 3 %), and rows below 2^24 that rowops.hip's rule sends there, run the literal loop -- wave-uniformly.
// 40 % of the instructions of a C = 384 / 768 row group were this loop (ten IEEE divisions per group of 4 / 2 rows).
IVIT_DEV float ln16_std10(float varf)
{
    const double v = (double)varf;                          // an integer below 2^37, exact
    double s = __builtin_floor(__builtin_sqrt(v));
    s = (s * s > v) ? s - 1.0 : s;                          // (exact products: s < 2^19)
    s = ((s + 1.0) * (s + 1.0) <= v) ? s + 1.0 : s;
    const double gap = (s + 1.0) * (s + 1.0) - v;
    const bool slow = varf < 16777216.0f ? (varf < 142883.0f || gap == 1.0)      // rowops.hip LN_NEWTON_CONVERGED, ln_std10
                                          : gap <= (double)(varf * 2.384185791015625e-07f);   // 2^-22
    if (__builtin_amdgcn_ballot_w64(slow) != 0) {
        float t = 65536.0f;
#pragma unroll 1
        for (int it = 0; it < 10; ++it) t = floorf((t + floorf(varf / t)) * 0.5f);
        return t;
    }
    return (float)s;
}

IVIT_DEV void ln16_build_table(const Ln16Args& a, float* t_bias, float* t_lo, float* t_hi)
{
    ln_build_table<NT>(a.m, a.e, a.s_ln, a.bias_int, a.C, t_bias, t_lo, t_hi);      // common.h: every load before the first use
}

// Sub-wave tiling of the same computation for C % 8 == 0, C <= 1536: LPR lanes share a row (64 / LPR rows per wave and
// iteration), each lane owns NJ chunks of 8 channels (one 16-byte load each) and keeps their per-channel constants in
// registers for the whole kernel.  The per-row scalar work (mean division, Newton steps) is evaluated once per
// wave-iteration for all rows of the wave in parallel; with LPR = C / 24 every lane is busy (C = 96 * 2^k: NJ = 3).
template <int LPR, int NJ>
__global__ __launch_bounds__(NT, 4) void layernorm_i16_i8_tiled_kernel(Ln16Args a)
{
    extern __shared__ __attribute__((aligned(16))) float ln16_tab[];      // [C] bias | [C] lo | [C] hi
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const int C = a.C, nd = C >> 3;
    float* t_bias = ln16_tab;
    float* t_lo = ln16_tab + C;
    float* t_hi = ln16_tab + 2 * C;
    ln16_build_table(a, t_bias, t_lo, t_hi);
    __syncthreads();
    const float fC = (float)C;
    for (int row0 = (blockIdx.x * WPB + wave) * RPW; row0 < a.rows; row0 += gridDim.x * WPB * RPW) {
        const int row = row0 + grp;
        const bool live = row < a.rows;
        const int16_t* xr = a.x + (int64_t)min(row, a.rows - 1) * C;
        v4i w[NJ];
        int sum = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = sub + LPR * j;
            w[j] = (d < nd) ? *reinterpret_cast<const v4i*>(xr + 8 * d) : v4i{0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q) sum += (int)(int16_t)w[j][q] + (w[j][q] >> 16);
        }
        sum = lanes_allsum_i32<LPR>(sum);      // common.h: DPP / permlane swaps instead of ds_bpermute butterflies
        const int mean_int = (int)rintf((float)sum / fC);                       // :37
        unsigned long long var = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (sub + LPR * j < nd) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned d0 = (unsigned)abs((int)(int16_t)w[j][q] - mean_int);
                    const unsigned d1 = (unsigned)abs((w[j][q] >> 16) - mean_int);
                    var += (unsigned long long)d0 * d0;
                    var += (unsigned long long)d1 * d1;
                }
            }
        }
        var = lanes_allsum_u64<LPR>(var);
        const float t = ln16_std10((float)var);                                 // :45-49
        const float hfactor = floorf((1.0f / t) * 2147483648.0f) * 0.5f;        // :51; the /2 of :52 commutes (exact scaling)
        int8_t* orow = a.out + win_row(a.map, min(row, a.rows - 1)) * a.ldo;
        int2 res[NJ];
        unsigned unc = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int o[8];
            const int dt = min(sub + LPR * j, nd - 1);      // this chunk's constants: 6 x 16 bytes from the workgroup's table
            float bias8[8], lo8[8], hi8[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 b4 = *reinterpret_cast<const float4*>(t_bias + 8 * dt + 4 * h);
                const float4 l4 = *reinterpret_cast<const float4*>(t_lo + 8 * dt + 4 * h);
                const float4 h4 = *reinterpret_cast<const float4*>(t_hi + 8 * dt + 4 * h);
                bias8[4 * h] = b4.x; bias8[4 * h + 1] = b4.y; bias8[4 * h + 2] = b4.z; bias8[4 * h + 3] = b4.w;
                lo8[4 * h] = l4.x; lo8[4 * h + 1] = l4.y; lo8[4 * h + 2] = l4.z; lo8[4 * h + 3] = l4.w;
                hi8[4 * h] = h4.x; hi8[4 * h + 1] = h4.y; hi8[4 * h + 2] = h4.z; hi8[4 * h + 3] = h4.w;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int xv = (c & 1) ? (w[j][c >> 1] >> 16) : (int)(int16_t)w[j][c >> 1];
                const float dl = (float)(xv - mean_int);
                const float v = floorf(dl * hfactor);                           // :52
                const float y = v + bias8[c];                                   // :61
                const int tl = __float_as_int(__builtin_fmaf(y, lo8[c], 12582912.0f));
                const int th = __float_as_int(__builtin_fmaf(y, hi8[c], 12582912.0f));
                asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                o[c] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);       // low byte = int8 result
            }
            res[j].x = pack4(o[0], o[1], o[2], o[3]);
            res[j].y = pack4(o[4], o[5], o[6], o[7]);
            __builtin_amdgcn_sched_barrier(0);      // one chunk's constants live at a time
        }
        if (__builtin_amdgcn_ballot_w64(unc != 0) != 0) {
            // literal evaluation (wave-uniform branch): x = y * s_ln (:63), z = round(x / s_ln) (quant_utils.py:220),
            // RNE(float64(z) * M) (:229-230)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int d = min(sub + LPR * j, nd - 1);
                int o[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 s4 = *reinterpret_cast<const float4*>(a.s_ln + 8 * d + 4 * h);
                    const uint4 m4 = *reinterpret_cast<const uint4*>(a.m + 8 * d + 4 * h);
                    const int4 e4 = *reinterpret_cast<const int4*>(a.e + 8 * d + 4 * h);
                    const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
                    const double MM[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z),
                                          dyadic_mult(m4.w, e4.w)};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int c = 4 * h + cc;
                        const int xv = (c & 1) ? (w[j][c >> 1] >> 16) : (int)(int16_t)w[j][c >> 1];
                        const float dl = (float)(xv - mean_int);
                        const float v = floorf(dl * hfactor);
                        const float y = v + t_bias[8 * d + c];
                        const float x = y * ss[cc];
                        const float z = rintf((float)((double)x * (1.0 / (double)ss[cc])));   // see layernorm_i8_kernel
                        const double tt = (double)z * MM[cc] + IVIT_MAGIC;
                        o[c] = clamp_i32((int)(unsigned)__double_as_longlong(tt), -128, 127);
                    }
                }
                res[j].x = pack4(o[0], o[1], o[2], o[3]);
                res[j].y = pack4(o[4], o[5], o[6], o[7]);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = sub + LPR * j;
            if (live && d < nd) *reinterpret_cast<int2*>(orow + 8 * d) = res[j];
        }
    }
}

// The tiled kernel for a 16-bit input carried at a NATURAL scale s_in (ivit_layernorm_i16_i8_compat): per element
//   phi = fl(fl(q * s_in) / s_in)   -- the float the reference's LayerNorm sees (quant_modules.py:387, ivit_modules.py:36); the
//       quotient by the invariant s_in is the 3-instruction form q0 = x * r, e = fma(-s, q0, x), phi = fma(e, r, q0) with
//       r = RN(1 / s_in), correctly rounded for every 16-bit q at this s_in (checked exhaustively on the host,
//       prepare.markstein_division_ok; the literal kernel above is the fallback);
//   k' = trunc(phi) replaces q in everything downstream (ivit_modules.py:38);
//   the mean is round(fl(S / C)) with S the float32 sum of the phi values in torch's CPU reduction order: the phi values of
//   the wave's rows go through LDS once and two rows at a time are summed by the two halves of the wave (rowsum32 below).
template <int CTRL>
IVIT_DEV float dpp_f32(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

template <class F>
IVIT_DEV float rowsum32(F elem, int n, int l32, int base)
{
    // rowsum.h torch_rowsum on 32 lanes (lanes base .. base + 31 of the wave): same partial sums, same order
    const int vec_size = n >> 3, size_ilp = vec_size >> 2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    {
        int lg = 0;
        while ((1 << lg) < size_ilp) ++lg;
        const int lp = max(4, lg / 4), step = 1 << lp, mask = step - 1;
        int i = 0;
        while (i + step <= size_ilp) {
            for (int j = 0; j < step; ++j, ++i) acc0 += elem(i * 32 + l32);
            acc1 += acc0; acc0 = 0.f;
            if ((i & (mask << lp)) == 0) {
                acc2 += acc1; acc1 = 0.f;
                if ((i & (mask << (2 * lp))) == 0) { acc3 += acc2; acc2 = 0.f; }
            }
        }
        for (; i < size_ilp; ++i) acc0 += elem(i * 32 + l32);
        acc0 += acc1; acc0 += acc2; acc0 += acc3;
    }
    if (l32 < 8)
        for (int i = size_ilp * 4; i < vec_size; ++i) acc0 += elem(i * 8 + l32);
    const float p1 = __shfl(acc0, base + ((l32 + 8) & 31)), p2 = __shfl(acc0, base + ((l32 + 16) & 31)),
                p3 = __shfl(acc0, base + ((l32 + 24) & 31));
    const float v = ((acc0 + p1) + p2) + p3;    // lanes base .. base + 7: the 8 vector lanes
    float fin = 0.f;
    for (int i = vec_size * 8; i < n; ++i) fin += elem(i);
#pragma unroll
    for (int l = 0; l < 8; ++l) fin += __shfl(v, base + l);
    return fin;
}

template <int LPR, int NJ>
__global__ __launch_bounds__(NT, NJ >= 2 ? 3 : 4) void layernorm_i16_i8_tiled_compat_kernel(Ln16Args a, float s_in, float r_in)
{
    __shared__ float s_phi[WPB][64 * 8 * NJ];      // [wave][row of the wave][channel]
    __shared__ float s_sum[WPB][64];
    constexpr int RPW = 64 / LPR;
    constexpr bool REGSUM = LPR <= 16 && LPR * NJ < 64;      // fewer than 16 steps per vector lane, a row inside one DPP row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const int C = a.C, nd = C >> 3;
    const bool regsum = a.outer == 0 && nd == LPR * NJ && !(IVIT_LAB && a.lab_lds_sum);     // uniform
    // transposed view: groups of 16 need C < 256 (no second-level fold) and no tail columns (outer % 32 == 0: rowsum.h)
    const bool regouter = a.outer != 0 && (a.outer & 31) == 0 && nd == LPR * NJ && C < 256 && !(IVIT_LAB && a.lab_lds_sum);
    extern __shared__ __attribute__((aligned(16))) float ln16_tab[];      // [C] bias | [C] lo | [C] hi (ln16_build_table)
    float* t_bias = ln16_tab;
    float* t_lo = ln16_tab + C;
    float* t_hi = ln16_tab + 2 * C;
    ln16_build_table(a, t_bias, t_lo, t_hi);
    __syncthreads();
    const float fC = (float)C;
    for (int row0 = (blockIdx.x * WPB + wave) * RPW; row0 < a.rows; row0 += gridDim.x * WPB * RPW) {
        const int row = row0 + grp;
        const bool live = row < a.rows;
        const int16_t* xr = a.x + (int64_t)min(row, a.rows - 1) * C;
        v4i w[NJ];
        float ph[NJ][8];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = sub + LPR * j;
            w[j] = (d < nd) ? *reinterpret_cast<const v4i*>(xr + 8 * d) : v4i{0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int qv = (c & 1) ? (w[j][c >> 1] >> 16) : (int)(int16_t)w[j][c >> 1];
                const float x = (float)qv * s_in;                                   // quant_modules.py:387
                const float q0 = x * r_in;                                          // :36  x / s_in (Markstein, see above)
                const float e = __builtin_fmaf(-s_in, q0, x);
                ph[j][c] = __builtin_fmaf(e, r_in, q0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                           // :38 .to(int32) truncates
                const int k0 = (int)ph[j][2 * q], k1 = (int)ph[j][2 * q + 1];
                w[j][q] = (k0 & 0xffff) | (k1 << 16);
            }
        }
        float S;
        if (REGSUM && regouter) {
            // The reference's mean over the TRANSPOSED view (a.outer, every LayerNorm of Swin stage 0): torch's outer reduction adds a
            // row's elements in sequence, 16 at a time, and the group sums in sequence (rowsum.h torch_cascade_sum: no second-level
            // fold below 256 elements).  A group is two chunks of 8 = the chunks j of an even lane and its odd neighbour: the even
            // lane adds its 8, the odd lane CONTINUES that sum with its own 8; the group sums sit on the odd lanes and are added in
            // the order (j, lane) by the group's first lane.
            float gsum[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float p = ph[j][0];
#pragma unroll
                for (int c = 1; c < 8; ++c) p += ph[j][c];
                float q = dpp_f32<0xa0>(p);                         // quad_perm [0,0,2,2]: the even neighbour's partial sum
#pragma unroll
                for (int c = 0; c < 8; ++c) q += ph[j][c];
                gsum[j] = q;                                        // meaningful on odd lanes
            }
            float fin = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int o = 1; o < LPR; o += 2) {
                    float t = gsum[j];
                    if (o == 1) t = dpp_f32<0x101>(t);              // row_shl:o -- lane l reads lane l + o
                    if (o == 3) t = dpp_f32<0x103>(t);
                    if (o == 5) t = dpp_f32<0x105>(t);
                    if (o == 7) t = dpp_f32<0x107>(t);
                    if (o == 9) t = dpp_f32<0x109>(t);
                    if (o == 11) t = dpp_f32<0x10b>(t);
                    if (o == 13) t = dpp_f32<0x10d>(t);
                    if (o == 15) t = dpp_f32<0x10f>(t);
                    fin = (j == 0 && o == 1) ? t : fin + t;
                }
            S = __shfl(fin, lane & ~(LPR - 1));
        } else if (REGSUM && regsum) {
            // torch's order without leaving the registers.  C = 8 LPR NJ < 512: 32 vector lanes p = e mod 32 each add their C / 32
            // elements in sequence (no cascade fold below 16 steps), the 8 lanes of a vector are ((a[l] + a[l+8]) + a[l+16]) + a[l+24],
            // the 8 results are added in sequence (rowsum.h).  Element e = 8 d + c of chunk d = sub + LPR j is vector lane
            // 8 (sub & 3) + c, step (LPR / 4) j + (sub >> 2): lane (sub & 3) of the row's first quad accumulates its own three
            // chunks and those of lanes sub + 4, + 8, + 12 (DPP row shifts: a row group never straddles a DPP row of 16), the quad
            // combines, everybody reads the result from the group's first lane.
            constexpr int H = LPR / 4;
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = ph[0][c];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int hh = 0; hh < H; ++hh) {
                    if (j == 0 && hh == 0) continue;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float t = ph[j][c];
                        if (hh == 1) t = dpp_f32<0x104>(t);         // row_shl:4: lane l reads lane l + 4
                        if (hh == 2) t = dpp_f32<0x108>(t);
                        if (hh == 3) t = dpp_f32<0x10c>(t);
                        acc[c] += t;
                    }
                }
            float fin = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float v = dpp_f32<0x00>(acc[c]);                    // quad_perm [0,0,0,0]
                v += dpp_f32<0x55>(acc[c]);
                v += dpp_f32<0xaa>(acc[c]);
                v += dpp_f32<0xff>(acc[c]);
                fin = c ? fin + v : v;
            }
            S = __shfl(fin, lane & ~(LPR - 1));
        } else {
        float* prow = s_phi[wave] + grp * C;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = sub + LPR * j;
            if (d < nd) {
                *reinterpret_cast<float4*>(prow + 8 * d) = make_float4(ph[j][0], ph[j][1], ph[j][2], ph[j][3]);
                *reinterpret_cast<float4*>(prow + 8 * d + 4) = make_float4(ph[j][4], ph[j][5], ph[j][6], ph[j][7]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (a.outer) {   // transposed view in the reference: a lane sums one row in the outer-reduction order (serial cascade)
            if (lane < RPW) {
                const float* pr = s_phi[wave] + lane * C;
                const int rw = min(row0 + lane, a.rows - 1);
                s_sum[wave][lane] = torch_outer_rowsum([&](int i) { return pr[i]; }, C, rw % a.outer >= (a.outer & ~31));
            }
        } else {   // float32 row sums in torch's order: the two halves of the wave take two rows per round
            const int half = lane >> 5, l32 = lane & 31;
#pragma unroll 1
            for (int r0 = 0; r0 < RPW; r0 += 2) {
                const int rr = min(r0 + half, RPW - 1);
                const float* pr = s_phi[wave] + rr * C;
                const float Sr = rowsum32([&](int i) { return pr[i]; }, C, l32, 32 * half);
                if (l32 == 0 && r0 + half < RPW) s_sum[wave][rr] = Sr;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        S = s_sum[wave][grp];
        __builtin_amdgcn_wave_barrier();                                        // before the next iteration overwrites s_phi
        }
        const int mean_int = (int)rintf(S / fC);                                // :37
        unsigned long long var = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (sub + LPR * j < nd) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned d0 = (unsigned)abs((int)(int16_t)w[j][q] - mean_int);
                    const unsigned d1 = (unsigned)abs((w[j][q] >> 16) - mean_int);
                    var += (unsigned long long)d0 * d0;
                    var += (unsigned long long)d1 * d1;
                }
            }
        }
        var = lanes_allsum_u64<LPR>(var);
        const float t = ln16_std10((float)var);                                 // :45-49
        const float hfactor = floorf((1.0f / t) * 2147483648.0f) * 0.5f;        // :51; the /2 of :52 commutes (exact scaling)
        int8_t* orow = a.out + win_row(a.map, min(row, a.rows - 1)) * a.ldo;
        int2 res[NJ];
        unsigned unc = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int o[8];
            const int dt = min(sub + LPR * j, nd - 1);      // this chunk's constants: 6 x 16 bytes from the workgroup's table
            float bias8[8], lo8[8], hi8[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 b4 = *reinterpret_cast<const float4*>(t_bias + 8 * dt + 4 * h);
                const float4 l4 = *reinterpret_cast<const float4*>(t_lo + 8 * dt + 4 * h);
                const float4 h4 = *reinterpret_cast<const float4*>(t_hi + 8 * dt + 4 * h);
                bias8[4 * h] = b4.x; bias8[4 * h + 1] = b4.y; bias8[4 * h + 2] = b4.z; bias8[4 * h + 3] = b4.w;
                lo8[4 * h] = l4.x; lo8[4 * h + 1] = l4.y; lo8[4 * h + 2] = l4.z; lo8[4 * h + 3] = l4.w;
                hi8[4 * h] = h4.x; hi8[4 * h + 1] = h4.y; hi8[4 * h + 2] = h4.z; hi8[4 * h + 3] = h4.w;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int xv = (c & 1) ? (w[j][c >> 1] >> 16) : (int)(int16_t)w[j][c >> 1];
                const float dl = (float)(xv - mean_int);
                const float v = floorf(dl * hfactor);                           // :52
                const float y = v + bias8[c];                                   // :61
                const int tl = __float_as_int(__builtin_fmaf(y, lo8[c], 12582912.0f));
                const int th = __float_as_int(__builtin_fmaf(y, hi8[c], 12582912.0f));
                asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                o[c] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);       // low byte = int8 result
            }
            res[j].x = pack4(o[0], o[1], o[2], o[3]);
            res[j].y = pack4(o[4], o[5], o[6], o[7]);
            __builtin_amdgcn_sched_barrier(0);      // one chunk's constants live at a time
        }
        if (__builtin_amdgcn_ballot_w64(unc != 0) != 0) {
            // literal evaluation (wave-uniform branch): x = y * s_ln (:63), z = round(x / s_ln) (quant_utils.py:220),
            // RNE(float64(z) * M) (:229-230)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int d = min(sub + LPR * j, nd - 1);
                int o[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 s4 = *reinterpret_cast<const float4*>(a.s_ln + 8 * d + 4 * h);
                    const uint4 m4 = *reinterpret_cast<const uint4*>(a.m + 8 * d + 4 * h);
                    const int4 e4 = *reinterpret_cast<const int4*>(a.e + 8 * d + 4 * h);
                    const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
                    const double MM[4] = {dyadic_mult(m4.x, e4.x), dyadic_mult(m4.y, e4.y), dyadic_mult(m4.z, e4.z),
                                          dyadic_mult(m4.w, e4.w)};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int c = 4 * h + cc;
                        const int xv = (c & 1) ? (w[j][c >> 1] >> 16) : (int)(int16_t)w[j][c >> 1];
                        const float dl = (float)(xv - mean_int);
                        const float v = floorf(dl * hfactor);
                        const float y = v + t_bias[8 * d + c];
                        const float x = y * ss[cc];
                        const float z = rintf((float)((double)x * (1.0 / (double)ss[cc])));   // see layernorm_i8_kernel
                        const double tt = (double)z * MM[cc] + IVIT_MAGIC;
                        o[c] = clamp_i32((int)(unsigned)__double_as_longlong(tt), -128, 127);
                    }
                }
                res[j].x = pack4(o[0], o[1], o[2], o[3]);
                res[j].y = pack4(o[4], o[5], o[6], o[7]);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = sub + LPR * j;
            if (live && d < nd) *reinterpret_cast<int2*>(orow + 8 * d) = res[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PatchMerging gather (swin_quant.py:337-344): [B, H*W, C] int16 -> [B, H/2*W/2, 4C], channel blocks
// (0::2,0::2), (1::2,0::2), (0::2,1::2), (1::2,1::2)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void patch_merge_kernel(const int16_t* x, int16_t* out, int B, int H, int W, int C)
{
    const int c4 = C >> 2;
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t total = (int64_t)B * H2 * W2 * 4 * c4;
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        int cc = (int)(q % c4);
        int64_t r = q / c4;
        int blk = (int)(r & 3);
        r >>= 2;
        int x2 = (int)(r % W2);
        r /= W2;
        int y2 = (int)(r % H2);
        int b = (int)(r / H2);
        const int dy = blk & 1, dx = blk >> 1;
        const int2 v = *reinterpret_cast<const int2*>(x + (((int64_t)b * H + 2 * y2 + dy) * W + 2 * x2 + dx) * C + 4 * cc);
        *reinterpret_cast<int2*>(out + (((int64_t)b * H2 + y2) * W2 + x2) * (4 * C) + blk * C + 4 * cc) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Token average pooling + QuantAct (swin_quant.py:554-555): z = round(fl32(sum_t k[t][c] / T)), requant -> int8
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void avgpool_kernel(const int8_t* x, int8_t* out, int B, int T, int C, double Mq)
{
    const int64_t total = (int64_t)B * C;
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        const int b = (int)(q / C), c = (int)(q - (int64_t)b * C);
        int sum = 0;
        for (int t = 0; t < T; ++t) sum += x[((int64_t)b * T + t) * C + c];
        const float z = rintf((float)sum / (float)T);
        out[q] = (int8_t)clamp_i32(requant_exact((int)z, Mq), -128, 127);
    }
}

// ------------------------------------------------------------------------------------------------
// Windowed attention: one wave per (window, head); T = ws*ws <= 64 tokens, head_dim 32.
//   S^T = K . Q^T on v_mfma_i32_16x16x32_i8 (the head dimension is one instruction deep), qact_attn1 (8 bit),
//   + relative position bias through the two-operand qact2 (the bias operand RNE(k_tab * m / 2^e) is a load-time
//   constant table [nH, T, 64] int16, keys padded to 64), + shift mask (the integer -100/s wherever the region ids of
//   query and key differ, added after the clamp, swin_quant.py:149-155), Shiftmax, O^T = Vt . P^T on v_mfma_i32_16x16x64_i8 (all 64 key slots in one step), qact3.
// ------------------------------------------------------------------------------------------------
struct WinAttnArgs {
    const int8_t* qkv;      // [3][B_][nH][T][32]
    int8_t* out;            // [B_*T, nH*32] with row stride ldo
    int64_t ldo;
    const int16_t* bias;    // [nH][T][64]
    const uint8_t* region;  // [nW][64] or NULL
    int mask_value;
    int nwin, heads, T, nW;
    double Ms, Mb, Mo;      // qact_attn1; qact2 main operand; qact3
    float Ms32, Mb32;       // rq32: the same as float32 -- both powers of two (every scale of the power-of-two regime): |S| <= 2^19 and
    int rq32;               // |kS| <= 128 make both products exact in float32, each fma rounds once, to nearest even, as the float64 form
    int x0, ksat;
    // natural Shiftmax input scale: phi[q + 128] = fl(fl(q*s)/s) for unmasked scores and phim[q + 128] =
    // fl(fl(fl(q*s) - 100)/s) for scores under the shift mask (swin_quant.py:151-156 adds float -100 to q*s before the softmax
    // divides by s), both float32 [256] on the device; NULL: power-of-two scale
    const float* phi;
    const float* phim;
    // natural scale, table form (round 4): band[(qmax + 128) * band_w + min(qmax - q, band_w - 1)] = Shiftmax's exp_int of the score
    // q under the row maximum qmax (prepare.shiftexp_band: the float32 sequence of ivit_modules.py:150-170 evaluated on the host for
    // every pair; entry band_w - 1 is the saturated value).  The host hands this over only when every score under the shift mask
    // saturates whatever the maximum and no masked score can be the maximum (prepare.window_shiftexp_band): masked scores then take
    // the saturated entry.  NULL: the literal form above (phi / phim) or the power-of-two form.
    const unsigned* band;
    int band_w;
    const unsigned* band1;  // ... and when the rows of that table do not depend on the maximum (often: prepare.window_shiftexp_band):
                            // its one row [band_w], used like the power-of-two form's table of distances
    // ws != 0: the output rows go to their IMAGE positions (window reverse + roll back applied here): the projection is
    // row-wise, so attn.proj and the residual QuantAct behind it then need no row map (and fuse into one GEMM)
    WinMap omap;
    int omap_inv;           // 65536 / ws + 1: (q * omap_inv) >> 16 == q / ws for q < 64
};

constexpr int WHD = 32;
constexpr int WVT_ROW = 64;                      // Vt row: 64 key slots
constexpr int WVT_BYTES = WHD * WVT_ROW;         // 2 KiB per wave
constexpr int WLUT_OFF = WPB * WVT_BYTES;

constexpr int WBAND_PAD = 4;     // dwords: keeps the slices 16-byte aligned and rotates their banks

// SM: Shiftmax form -- 0 the table of distances to the row maximum (power-of-two scales; natural scales whose band table has one
// row), 1 the literal float32 sequence on the phi tables, 2 band rows per row maximum staged through LDS.  Template parameters, not
// run-time branches: with all three forms in one body the kernel took 135 VGPRs = three workgroups per CU instead of four, and the
// power-of-two form lost 12 % (44 -> 49.5 us per call in Swin-T, profiles/r04q vs r04h).
template <int SM, bool RQ32>
__global__ __launch_bounds__(NT, 4) void window_attention_kernel(WinAttnArgs a)
{
    __shared__ __attribute__((aligned(16))) char smem[WLUT_OFF + 256 * 4];
    __shared__ float s_phi[2][256];
    extern __shared__ __attribute__((aligned(16))) unsigned band_lds[];     // [4 waves][16 queries][band_w + WBAND_PAD], band form only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int T = a.T;
    reinterpret_cast<unsigned*>(smem + WLUT_OFF)[tid] = a.band1 ? a.band1[min(tid, a.band_w - 1)] : shiftexp_int(-tid, a.x0, 15);
    constexpr bool band = SM == 2, compat = SM == 1;
    if constexpr (compat) {
        s_phi[0][tid] = a.phi[tid];
        s_phi[1][tid] = a.phim[tid];
    }
    __syncthreads();
    const unsigned* lut = reinterpret_cast<const unsigned*>(smem + WLUT_OFF);
    char* vt = smem + wave * WVT_BYTES;
    const int64_t plane = (int64_t)a.nwin * a.heads * T * WHD;
    const int npairs = a.nwin * a.heads;

    for (int pair = blockIdx.x * WPB + wave; pair < npairs; pair += gridDim.x * WPB) {
        const int win = pair / a.heads, hh = pair - win * a.heads;
        // image position of the window (wave-uniform): its image, its window row / column
        int64_t img_base = 0;
        int wy0 = 0, wx0 = 0;
        if (a.omap.ws) {
            const int nwx = a.omap.W / a.omap.ws;
            const int bimg = win / a.nW, wrem = win - bimg * a.nW;
            const int wy = wrem / nwx;
            wy0 = wy * a.omap.ws + a.omap.shift;
            wx0 = (wrem - wy * nwx) * a.omap.ws + a.omap.shift;
            img_base = (int64_t)bimg * a.omap.H * a.omap.W;
        }
        const int8_t* qg = a.qkv + (int64_t)pair * T * WHD;
        const int8_t* kg = qg + plane;
        const int8_t* vg = qg + 2 * plane;
        // ---- V transposed into this wave's LDS tile: Vt[d][chunk g'][byte 4t + r] = V[key 16t + 4g' + r][d];
        //      chunk j of row d at slot (j + 2*((d>>2)&1)) & 3.  Work item = 4 keys x 16 d (13 x 2 items).
        __builtin_amdgcn_wave_barrier();
        if (lane < ((T + 3) >> 2) * 2) {
            const int kg4 = lane >> 1, c = lane & 1;
            v4i v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const v4i*>(vg + (int64_t)min(4 * kg4 + r, T - 1) * WHD + 16 * c);
            const int key0 = 4 * kg4;
            const int j = (key0 >> 2) & 3, boff = 4 * ((key0 >> 4) & 3);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned a0 = (unsigned)v[0][w], a1 = (unsigned)v[1][w], a2 = (unsigned)v[2][w], a3 = (unsigned)v[3][w];
                const unsigned lo01 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), hi01 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
                const unsigned lo23 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), hi23 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                const unsigned t4[4] = {__builtin_amdgcn_perm(lo23, lo01, 0x05040100u), __builtin_amdgcn_perm(lo23, lo01, 0x07060302u),
                                        __builtin_amdgcn_perm(hi23, hi01, 0x05040100u), __builtin_amdgcn_perm(hi23, hi01, 0x07060302u)};
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int d = 16 * c + 4 * w + bb;
                    *reinterpret_cast<unsigned*>(vt + d * WVT_ROW + (((j + 2 * ((d >> 2) & 1)) & 3) << 4) + boff) = t4[bb];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)

        // K fragments of the 4 key tiles: lane (key 16kt + l15, 8 bytes at 8g)
        long kf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            kf[kt] = *reinterpret_cast<const long*>(kg + (int64_t)min(16 * kt + l15, T - 1) * WHD + 8 * g);

        // shift mask (swin_quant.py:223-249): tokens of different regions of the rolled image do not attend to each other
        unsigned kreg[4] = {0u, 0u, 0u, 0u};
        const uint8_t* regrow = a.region ? a.region + (win % a.nW) * 64 : nullptr;
        if (regrow) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) kreg[kt] = *reinterpret_cast<const unsigned*>(regrow + 16 * kt + 4 * g);
        }
        for (int qt = 0; qt < 4; ++qt) {
            const int qrow = 16 * qt + l15;
            if (16 * qt >= T) break;  // uniform
            const int qld = min(qrow, T - 1);
            const long qf = *reinterpret_cast<const long*>(qg + (int64_t)qld * WHD + 8 * g);
            const int16_t* brow = a.bias + ((int64_t)hh * T + qld) * 64 + 4 * g;
            const unsigned qreg = regrow ? regrow[qld] : 0u;
            int s[4][4];
            int rmax = -100000;
            float xv[4][4], xmax = -__builtin_inff();     // compat: the float view x / s of every score
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                v4i acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(kf[kt], qf, acc, 0, 0, 0);
                const int2 bw = *reinterpret_cast<const int2*>(brow + 16 * kt);
                const int bv[4] = {(int)(int16_t)bw.x, bw.x >> 16, (int)(int16_t)bw.y, bw.y >> 16};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * g + r;
                    int ka = -100000;
                    xv[kt][r] = -__builtin_inff();
                    if (key < T) {
                        if constexpr (RQ32) {      // round 4: 7 float32 / integer instructions instead of 6 float64 ones + 3 (attention.hip RQ32)
                            const int tb = clamp_i32(__float_as_int(__builtin_fmaf((float)acc[r], a.Ms32, 12582912.0f)), 0x4B400000 - 128, 0x4B400000 + 127);
                            const float kSf = __int_as_float(tb) - 12582912.0f;                  // qact_attn1, exact small integer
                            ka = clamp_i32(__float_as_int(__builtin_fmaf(kSf, a.Mb32, 12582912.0f)) - 0x4B400000 + bv[r], -128, 127);
                        } else {
                        const int kS = clamp_i32(requant_exact(acc[r], a.Ms), -128, 127);        // qact_attn1
                        ka = clamp_i32(requant_exact(kS, a.Mb) + bv[r], -128, 127);              // qact2 (two operands)
                        }
                        const bool masked = ((kreg[kt] >> (8 * r)) & 0xffu) != qreg;
                        if constexpr (compat) xv[kt][r] = s_phi[masked ? 1 : 0][ka + 128];
                        if (masked) ka = band ? -50000 : ka + a.mask_value;                      // shift mask, after the clamp
                    }
                    s[kt][r] = ka;
                    rmax = max(rmax, ka);
                    xmax = fmaxf(xmax, xv[kt][r]);
                }
            }
            rmax = rows_allmax_i32(rmax);      // over the four lanes of a query (common.h: permlane swaps, no LDS round trip)
            unsigned esum = 0;
            if constexpr (compat) {
                // Shiftmax's float32 sequence on the phi values themselves (ivit_modules.py:150-170), per score
                xmax = fmaxf(xmax, __shfl_xor(xmax, 16));
                xmax = fmaxf(xmax, __shfl_xor(xmax, 32));
                const float x0f = (float)a.x0;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        unsigned e = 0u;
                        if (s[kt][r] != -100000) {
                            const float d = xv[kt][r] - xmax;                                    // :168
                            float x = (d + floorf(d / 2.0f)) - floorf(d / 16.0f);                // :151
                            x = fmaxf(x, 15.0f * x0f);                                           // :155
                            const float qq = floorf(x / x0f);                                    // :157
                            const float rr = x - x0f * qq;                                       // :158
                            const float ex = floorf((rr / 2.0f - x0f) * ldexpf(1.0f, 15 - (int)qq));   // :159-160
                            e = (unsigned)fmaxf(ex, 0.0f);
                        }
                        s[kt][r] = (int)e;
                        esum += e;
                    }
            } else if constexpr (band) {
                // the band rows of this tile's 16 queries go through LDS (as attention.hip MODE 1): the four lanes of a query copy
                // its row (band_w dwords, 16 bytes per lane and step), then every score is one LDS gather.  (Gathers straight from
                // the L2-resident table were as slow as the literal float sequence: profiles/r04m_*.)
                const int W = a.band_w, W1 = W - 1, stride = W + WBAND_PAD;
                const int rm = max(rmax, -128);
                unsigned* slice = band_lds + (wave * 16 + l15) * stride;
                __builtin_amdgcn_wave_barrier();      // the previous tile's gathers are done
                {
                    const uint4* src = reinterpret_cast<const uint4*>(a.band + (size_t)(rm + 128) * W);
                    uint4* dst = reinterpret_cast<uint4*>(slice);
                    for (int i = g; i < (W >> 2); i += 4) dst[i] = src[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                unsigned ev[4][4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ev[kt][r] = slice[min(rm - max(s[kt][r], -50000), W1)];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned e = (s[kt][r] == -100000) ? 0u : ev[kt][r];
                        s[kt][r] = (int)e;
                        esum += e;
                    }
            } else
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    unsigned e = (s[kt][r] == -100000) ? 0u : lut[min(rmax - s[kt][r], a.ksat) & 255];
                    s[kt][r] = (int)e;
                    esum += e;
                }
            esum = rows_allsum_u32(esum);
            float S = fminf((float)esum, 2147483648.0f);                 // ivit_modules.py:171-173
            const float factor = floorf((1.0f / S) * 2147483648.0f);     // :174
            v4i pk;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                unsigned w = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = (float)(unsigned)s[t][r] * factor;       // :175
                    w |= ((((unsigned)pr) >> 24) & 0xffu) << (8 * r);
                }
                pk[t] = (int)w;
            }
            int64_t orow_idx = (int64_t)win * T + qrow;
            if (a.omap.ws) {      // window reverse + roll back: (iy, ix) of the query in its window -> (y, x) of the image
                const int qr = min(qrow, T - 1);
                const int iy = (qr * a.omap_inv) >> 16, ix = qr - iy * a.omap.ws;      // qr / ws for qr < 64 (checked by the launcher)
                int y = wy0 + iy, x = wx0 + ix;
                y = y >= a.omap.H ? y - a.omap.H : y;
                x = x >= a.omap.W ? x - a.omap.W : x;
                orow_idx = img_base + y * a.omap.W + x;
            }
            int8_t* orow = a.out + orow_idx * a.ldo + hh * WHD;
            unsigned wq[2];      // wq[dt]: bytes d = 16 dt + 4 g + 0..3 of this lane's query
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int d = 16 * dt + l15;
                const v4i vf = *reinterpret_cast<const v4i*>(vt + d * WVT_ROW + (((g + 2 * ((d >> 2) & 1)) & 3) << 4));
                v4i acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf, pk, acc, 0, 0, 0);
                unsigned w = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    w |= ((unsigned)clamp_i32(requant_exact(acc[r], a.Mo), -128, 127) & 0xffu) << (8 * r);
                wq[dt] = w;
            }
            // the query's 32 bytes sit as 2 x 4 dwords in its four lanes: a word exchange (v_permlane32_swap, v_permlane16_swap)
            // leaves lane g with the 8 contiguous bytes d = 8 g .. 8 g + 7 -> one 8-byte store instead of two 4-byte ones
            {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u ab = __builtin_amdgcn_permlane32_swap(wq[0], wq[1], false, false);   // g < 2: (w0[g], w0[g+2]); g >= 2: (w1[g-2], w1[g])
                const v2u pr = __builtin_amdgcn_permlane16_swap(ab.x, ab.y, false, false);     // lane g: words 2 (g & 1), 2 (g & 1) + 1 of w_{g >> 1}
                if (qrow < T) *reinterpret_cast<int2*>(orow + 8 * g) = make_int2((int)pr.x, (int)pr.y);
            }
        }
    }
}

}  // namespace

// ================================================================================================
IVIT_EXPORT int ivit_requant_i8_i16(const int8_t* x, uint32_t m, int32_t e, int16_t* out, int64_t n, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && n > 0, "ivit_requant_i8_i16: bad operand");
    const double Mq = ivit_dyadic_to_double(m, e);
    IVIT_REQUIRE(Mq < 8388608.0, "ivit_requant_i8_i16: multiplier too large");
    hipLaunchKernelGGL(requant_i8_i16_kernel, dim3(ew_grid(n)), dim3(NT), 0, ivit_stream(stream), x, Mq, out, n);
    IVIT_CHECK_LAUNCH("ivit_requant_i8_i16");
}

static int check_map(const char* who, int64_t rows, int H, int W, int ws, int shift)
{
    if (ws == 0) return IVIT_OK;
    IVIT_REQUIRE(H > 0 && W > 0 && ws > 0 && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws && rows % ((int64_t)H * W) == 0,
                 "%s: bad window map H=%d W=%d ws=%d shift=%d rows=%lld", who, H, W, ws, shift, (long long)rows);
    return IVIT_OK;
}

IVIT_EXPORT int ivit_residual_requant_i16(const void* a, int a_bits, const uint32_t* m_pre, const int32_t* e_pre,
                                          uint32_t m_a, int32_t e_a, const int16_t* res, uint32_t m_r, int32_t e_r,
                                          int16_t* out, int64_t rows, int C, int H, int W, int ws, int shift,
                                          ivit_stream_t stream)
{
    IVIT_REQUIRE(a && res && out && rows > 0 && C > 0 && C % 4 == 0, "ivit_residual_requant_i16: bad operand");
    IVIT_REQUIRE(a_bits == 8 || a_bits == 16 || a_bits == 32, "ivit_residual_requant_i16: a_bits must be 8, 16 or 32");
    IVIT_REQUIRE((a_bits == 32) == (m_pre != nullptr && e_pre != nullptr),
                 "ivit_residual_requant_i16: (m_pre, e_pre) are required for, and only for, int32 accumulators");
    IVIT_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)res % 8 == 0) && ((uintptr_t)out % 8 == 0) &&
                     ((uintptr_t)m_pre % 16 == 0) && ((uintptr_t)e_pre % 16 == 0),
                 "ivit_residual_requant_i16: misaligned operand");
    int rc = check_map("ivit_residual_requant_i16", rows, H, W, ws, shift);
    if (rc) return rc;
    const double Ma = ivit_dyadic_to_double(m_a, e_a), Mr = ivit_dyadic_to_double(m_r, e_r);
    IVIT_REQUIRE(Ma < 32768.0 && Mr < 32768.0, "ivit_residual_requant_i16: multiplier too large");
    WinMap map{H, W, ws, shift};
    const int grid = ew_grid(rows * (C / 4));
    hipStream_t st = ivit_stream(stream);
    if (a_bits == 8)
        hipLaunchKernelGGL(residual_i16_kernel<int8_t>, dim3(grid), dim3(NT), 0, st, reinterpret_cast<const int8_t*>(a),
                           m_pre, e_pre, Ma, res, Mr, out, rows, C, map);
    else if (a_bits == 16)
        hipLaunchKernelGGL(residual_i16_kernel<int16_t>, dim3(grid), dim3(NT), 0, st, reinterpret_cast<const int16_t*>(a),
                           m_pre, e_pre, Ma, res, Mr, out, rows, C, map);
    else
        hipLaunchKernelGGL(residual_i16_kernel<int32_t>, dim3(grid), dim3(NT), 0, st, reinterpret_cast<const int32_t*>(a),
                           m_pre, e_pre, Ma, res, Mr, out, rows, C, map);
    IVIT_CHECK_LAUNCH("ivit_residual_requant_i16");
}

IVIT_EXPORT int ivit_layernorm_i16_i8(const int16_t* x, int rows, int C, const float* bias_int, const float* s_ln,
                                      const uint32_t* m, const int32_t* e, int8_t* out, int64_t ldo, int H, int W, int ws,
                                      int shift, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && bias_int && s_ln && m && e && rows > 0 && C > 0 && C <= 4096 && ldo >= C,
                 "ivit_layernorm_i16_i8: bad operand");
    int rc = check_map("ivit_layernorm_i16_i8", rows, H, W, ws, shift);
    if (rc) return rc;
    Ln16Args a{x, rows, C, bias_int, s_ln, m, e, out, ldo, WinMap{H, W, ws, shift}, 0, 0};
    hipStream_t st = ivit_stream(stream);
    const bool tiled = C % 8 == 0 && C <= 1536 && ldo % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 8 == 0) &&
                       ((uintptr_t)bias_int % 16 == 0) && ((uintptr_t)s_ln % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                       ((uintptr_t)e % 16 == 0);
    if (!tiled) {
        hipLaunchKernelGGL(layernorm_i16_i8_kernel, dim3(grid_for_rows(rows)), dim3(NT), 0, st, a);
        IVIT_CHECK_LAUNCH("ivit_layernorm_i16_i8");
    }
    const int nd = C / 8;
    int lpr = 4;
    while ((nd + lpr - 1) / lpr > 3) lpr *= 2;   // smallest group of lanes that covers a row with <= 3 chunks per lane
    const int nj = (nd + lpr - 1) / lpr;
    // persistent-style launch: the per-channel constants (24 f64 reciprocals per lane) are set up once per wave, so keep
    // the grid at the number of resident workgroups (2 per CU at ~195 VGPRs) and let each wave stride over many rows
    int nblk = grid_for_rows(rows, 64 / lpr);
    // four workgroups per CU (128 VGPRs since the constants moved to LDS, round 4); each builds the table once and strides over rows
    const int cap = ((g_ln_ablate >> 16) & 15) ? 256 * ((g_ln_ablate >> 16) & 15) : 1024;
    if (nblk > cap) nblk = cap;
    const dim3 grid(nblk), blk(NT);
    const size_t tab_bytes = (size_t)3 * C * sizeof(float);
#define LN16_CASE(L, J) \
    if (lpr == L && nj == J) hipLaunchKernelGGL((layernorm_i16_i8_tiled_kernel<L, J>), grid, blk, tab_bytes, st, a)
    LN16_CASE(4, 1); LN16_CASE(4, 2); LN16_CASE(4, 3);
    LN16_CASE(8, 2); LN16_CASE(8, 3);
    LN16_CASE(16, 2); LN16_CASE(16, 3);
    LN16_CASE(32, 2); LN16_CASE(32, 3);
    LN16_CASE(64, 2); LN16_CASE(64, 3);
#undef LN16_CASE
    IVIT_CHECK_LAUNCH("ivit_layernorm_i16_i8");
}

IVIT_EXPORT int ivit_layernorm_i16_i8_compat(const int16_t* x, int rows, int C, float s_in, int fast_division,
                                             const float* bias_int, const float* s_ln, const uint32_t* m, const int32_t* e,
                                             int8_t* out, int64_t ldo, int H, int W, int ws, int shift, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && bias_int && s_ln && m && e && rows > 0 && C > 0 && C <= 4096 && ldo >= C && s_in > 0.0f,
                 "ivit_layernorm_i16_i8_compat: bad operand");
    int rc = check_map("ivit_layernorm_i16_i8_compat", rows, H, W, ws, shift);
    if (rc) return rc;
    const int outer = fast_division >> 8;     // IVIT_LN_OUTER_MEAN(L)
    fast_division &= 255;
    IVIT_REQUIRE(outer >= 0 && (outer == 0 || rows % outer == 0) && fast_division <= 1, "ivit_layernorm_i16_i8_compat: bad flags");
    Ln16Args b{x, rows, C, bias_int, s_ln, m, e, out, ldo, WinMap{H, W, ws, shift}, outer, (g_ln_ablate >> 20) & 1};   // lab bit 20: sums through LDS
    hipStream_t st = ivit_stream(stream);
    const bool tiled = fast_division && C % 8 == 0 && C <= 1536 && ldo % 8 == 0 && ((uintptr_t)x % 16 == 0) &&
                       ((uintptr_t)out % 8 == 0) && ((uintptr_t)bias_int % 16 == 0) && ((uintptr_t)s_ln % 16 == 0) &&
                       ((uintptr_t)m % 16 == 0) && ((uintptr_t)e % 16 == 0);
    if (tiled) {
        const int nd = C / 8;
        int lpr = 4;
        while ((nd + lpr - 1) / lpr > 3) lpr *= 2;
        const int nj = (nd + lpr - 1) / lpr;
        int nblk = grid_for_rows(rows, 64 / lpr);
        if (nblk > (nj >= 2 ? 768 : 1024)) nblk = nj >= 2 ? 768 : 1024;      // = the kernel's launch bounds: 3 / 4 workgroups per CU
        const size_t tab_bytes = (size_t)3 * C * sizeof(float);
        const float r_in = 1.0f / s_in;
        bool launched = false;
#define LN16C_CASE(L, J)                                                                                                  \
    if (lpr == L && nj == J) {                                                                                            \
        hipLaunchKernelGGL((layernorm_i16_i8_tiled_compat_kernel<L, J>), dim3(nblk), dim3(NT), tab_bytes, st, b, s_in, r_in);     \
        launched = true;                                                                                                  \
    }
        LN16C_CASE(4, 1); LN16C_CASE(4, 2); LN16C_CASE(4, 3);
        LN16C_CASE(8, 2); LN16C_CASE(8, 3);
        LN16C_CASE(16, 2); LN16C_CASE(16, 3);
        LN16C_CASE(32, 2); LN16C_CASE(32, 3);
        LN16C_CASE(64, 2); LN16C_CASE(64, 3);
#undef LN16C_CASE
        if (launched) IVIT_CHECK_LAUNCH("ivit_layernorm_i16_i8_compat");
    }
    Ln16LitArgs a{b, s_in};
    hipLaunchKernelGGL(layernorm_i16_i8_literal_kernel, dim3(grid_for_rows(rows)), dim3(NT), 0, st, a);
    IVIT_CHECK_LAUNCH("ivit_layernorm_i16_i8_compat");
}

IVIT_EXPORT int ivit_patch_merge_i16(const int16_t* x, int16_t* out, int batch, int H, int W, int C, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && batch > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0,
                 "ivit_patch_merge_i16: bad operand");
    IVIT_REQUIRE(((uintptr_t)x % 8 == 0) && ((uintptr_t)out % 8 == 0), "ivit_patch_merge_i16: misaligned operand");
    const int64_t total = (int64_t)batch * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(patch_merge_kernel, dim3(ew_grid(total)), dim3(NT), 0, ivit_stream(stream), x, out, batch, H, W, C);
    IVIT_CHECK_LAUNCH("ivit_patch_merge_i16");
}

IVIT_EXPORT int ivit_avgpool_requant_i8(const int8_t* x, int8_t* out, int batch, int tokens, int C, uint32_t m, int32_t e,
                                        ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && batch > 0 && tokens > 0 && C > 0, "ivit_avgpool_requant_i8: bad operand");
    const double Mq = ivit_dyadic_to_double(m, e);
    IVIT_REQUIRE(Mq < 1048576.0, "ivit_avgpool_requant_i8: multiplier too large");
    hipLaunchKernelGGL(avgpool_kernel, dim3(ew_grid((int64_t)batch * C)), dim3(NT), 0, ivit_stream(stream), x, out, batch,
                       tokens, C, Mq);
    IVIT_CHECK_LAUNCH("ivit_avgpool_requant_i8");
}

IVIT_EXPORT int ivit_window_attention_i8(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add,
                                         const uint8_t* mask_region, int mask_value, int windows, int windows_per_image,
                                         int heads, int tokens, int head_dim, uint32_t m_s, int32_t e_s, uint32_t m_b,
                                         int32_t e_b, float s_attn, uint32_t m_o, int32_t e_o, ivit_stream_t stream)
{
    return ivit_window_attention_i8_compat(qkv, out, ldo, bias_add, mask_region, mask_value, windows, windows_per_image, heads,
                                           tokens, head_dim, m_s, e_s, m_b, e_b, s_attn, m_o, e_o, nullptr, nullptr, stream);
}

static int window_attention_launch(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add,
                                   const uint8_t* mask_region, int mask_value, int windows, int windows_per_image,
                                   int heads, int tokens, int head_dim, uint32_t m_s, int32_t e_s, uint32_t m_b,
                                   int32_t e_b, float s_attn, uint32_t m_o, int32_t e_o, const float* phi,
                                   const float* phi_masked, WinMap omap, ivit_stream_t stream, const uint32_t* band = nullptr,
                                   int band_w = 0, int band_rows = 256)
{
    IVIT_REQUIRE(qkv && out && bias_add, "ivit_window_attention_i8: NULL operand");
    // (16 band rows per wave in LDS: 4 x 16 x (192 + 4) dwords = 49 KB beside the kernel's 11 KB stay below the 64 KB of a default launch)
    IVIT_REQUIRE(band_w == 0 ? band == nullptr : (band && band_w >= 16 && band_w <= 192 && band_w % 16 == 0 && (uintptr_t)band % 16 == 0),
                 "ivit_window_attention_i8_band: the band table must be 16-byte aligned, its width a multiple of 16 in [16, 192]");
    IVIT_REQUIRE(windows > 0 && heads > 0 && windows_per_image > 0 && windows % windows_per_image == 0,
                 "ivit_window_attention_i8: bad window counts");
    if (head_dim != WHD || tokens < 2 || tokens > 64) {
        ivit_set_error("ivit_window_attention_i8: unsupported geometry head_dim=%d tokens=%d (need 32, 2..64)", head_dim, tokens);
        return IVIT_ERR_UNSUPPORTED;
    }
    IVIT_REQUIRE(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 8 == 0) && ldo % 8 == 0 && ldo >= (int64_t)heads * head_dim,
                 "ivit_window_attention_i8: misaligned operand or ldo too small");
    IVIT_REQUIRE(((uintptr_t)bias_add % 8 == 0) && ((uintptr_t)mask_region % 4 == 0), "ivit_window_attention_i8: misaligned table");
    IVIT_REQUIRE(mask_value <= 0 && mask_value >= -32768, "ivit_window_attention_i8: mask_value=%d outside [-32768, 0]", mask_value);
    IVIT_REQUIRE(s_attn > 0.0f, "ivit_window_attention_i8: scale must be positive");
    IVIT_REQUIRE((phi == nullptr) == (phi_masked == nullptr), "ivit_window_attention_i8_compat: phi and phi_masked go together");
    WinAttnArgs a;
    a.omap = omap;
    a.omap_inv = omap.ws ? 65536 / omap.ws + 1 : 0;
    for (int qv = 0; qv < 64 && omap.ws; ++qv)
        IVIT_REQUIRE(((qv * a.omap_inv) >> 16) == qv / omap.ws, "ivit_window_attention_i8_unwindow: window size %d unsupported", omap.ws);
    a.phi = phi; a.phim = phi_masked;
    IVIT_REQUIRE(band_rows == 256 || band_rows == 1, "ivit_window_attention_i8_band: band_rows must be 256 or 1");
    a.band = band_rows == 256 ? band : nullptr;
    a.band1 = band_rows == 1 ? band : nullptr;
    a.band_w = band_w;
    a.qkv = qkv; a.out = out; a.ldo = ldo; a.bias = bias_add; a.region = mask_region; a.mask_value = mask_value;
    a.nwin = windows; a.heads = heads; a.T = tokens; a.nW = windows_per_image;
    a.Ms = ivit_dyadic_to_double(m_s, e_s);
    a.Mb = ivit_dyadic_to_double(m_b, e_b);
    a.Mo = ivit_dyadic_to_double(m_o, e_o);
    IVIT_REQUIRE(a.Ms < 2048.0 && a.Mb < 1048576.0 && a.Mo < 512.0, "ivit_window_attention_i8: requant multiplier too large");
    // both score multipliers powers of two (m = 2^k) and in float32's range: the float32 form of the two requantisations is exact
    a.rq32 = m_s != 0 && (m_s & (m_s - 1)) == 0 && m_b != 0 && (m_b & (m_b - 1)) == 0 && a.Ms >= 1e-30 && a.Mb >= 1e-30 && a.Mb <= 4096.0 &&
             !(g_ln_ablate & (1 << 23));      // lab bit 23: the float64 form (A/B, parity of both forms)
    a.Ms32 = (float)a.Ms;
    a.Mb32 = (float)a.Mb;
    const float x0f = __builtin_floorf((1.0f / s_attn) * -1.0f);
    IVIT_REQUIRE(x0f <= -1.0f && x0f >= -4096.0f, "ivit_window_attention_i8: x0=%g outside [-4096,-1]", (double)x0f);
    a.x0 = (int)x0f;
    // exact u32 row sum: tokens * |x0| * 2^15 must stay below 2^32
    IVIT_REQUIRE((double)tokens * (double)(-a.x0) * 32768.0 < 4294967296.0,
                 "ivit_window_attention_i8: Shiftmax row sum could overflow 32 bits (x0=%d)", a.x0);
    a.ksat = 255;
    for (int i = 0; i < 256; ++i) {
        const int d = -i;
        if (d + (d >> 1) - (d >> 4) <= 15 * a.x0) { a.ksat = i; break; }
    }
    if (a.band1) {      // one row for every maximum: the distance table of the power-of-two form, with the host's values
        a.ksat = band_w - 1;
        a.mask_value = -1024;       // a masked score lands beyond the last (saturated) entry whatever the maximum
    }
    const int npairs = windows * heads;
    const int grid = (npairs + WPB - 1) / WPB;
    const size_t band_bytes = a.band ? (size_t)WPB * 16 * (band_w + WBAND_PAD) * sizeof(unsigned) : 0;
    const dim3 grd(grid < 8192 ? grid : 8192), blk(NT);
    hipStream_t st = ivit_stream(stream);
    if (a.band) {
        if (a.rq32) hipLaunchKernelGGL((window_attention_kernel<2, true>), grd, blk, band_bytes, st, a);
        else hipLaunchKernelGGL((window_attention_kernel<2, false>), grd, blk, band_bytes, st, a);
    } else if (a.phi) {
        if (a.rq32) hipLaunchKernelGGL((window_attention_kernel<1, true>), grd, blk, 0, st, a);
        else hipLaunchKernelGGL((window_attention_kernel<1, false>), grd, blk, 0, st, a);
    } else {
        if (a.rq32) hipLaunchKernelGGL((window_attention_kernel<0, true>), grd, blk, 0, st, a);
        else hipLaunchKernelGGL((window_attention_kernel<0, false>), grd, blk, 0, st, a);
    }
    IVIT_CHECK_LAUNCH("ivit_window_attention_i8");
}

IVIT_EXPORT int ivit_window_attention_i8_compat(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add,
                                                const uint8_t* mask_region, int mask_value, int windows, int windows_per_image,
                                                int heads, int tokens, int head_dim, uint32_t m_s, int32_t e_s, uint32_t m_b,
                                                int32_t e_b, float s_attn, uint32_t m_o, int32_t e_o, const float* phi,
                                                const float* phi_masked, ivit_stream_t stream)
{
    return window_attention_launch(qkv, out, ldo, bias_add, mask_region, mask_value, windows, windows_per_image, heads, tokens,
                                   head_dim, m_s, e_s, m_b, e_b, s_attn, m_o, e_o, phi, phi_masked, WinMap{0, 0, 0, 0}, stream);
}

IVIT_EXPORT int ivit_window_attention_i8_band(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add,
                                              const uint8_t* mask_region, int windows, int windows_per_image, int heads, int tokens,
                                              int head_dim, uint32_t m_s, int32_t e_s, uint32_t m_b, int32_t e_b, float s_attn,
                                              uint32_t m_o, int32_t e_o, const uint32_t* band, int band_w, int band_rows, int H, int W,
                                              int ws, int shift, ivit_stream_t stream)
{
    IVIT_REQUIRE(band && band_w > 0, "ivit_window_attention_i8_band: no band table");
    if (ws)
        IVIT_REQUIRE(ws > 0 && ws * ws == tokens && H > 0 && W > 0 && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws &&
                         windows_per_image == (H / ws) * (W / ws),
                     "ivit_window_attention_i8_band: H=%d W=%d ws=%d shift=%d do not describe %d windows of %d tokens per image", H, W, ws,
                     shift, windows_per_image, tokens);
    return window_attention_launch(qkv, out, ldo, bias_add, mask_region, 0, windows, windows_per_image, heads, tokens, head_dim, m_s,
                                   e_s, m_b, e_b, s_attn, m_o, e_o, nullptr, nullptr, ws ? WinMap{H, W, ws, shift} : WinMap{0, 0, 0, 0},
                                   stream, band, band_w, band_rows);
}

IVIT_EXPORT int ivit_window_attention_i8_unwindow(const int8_t* qkv, int8_t* out, int64_t ldo, const int16_t* bias_add,
                                                  const uint8_t* mask_region, int mask_value, int windows, int windows_per_image,
                                                  int heads, int tokens, int head_dim, uint32_t m_s, int32_t e_s, uint32_t m_b,
                                                  int32_t e_b, float s_attn, uint32_t m_o, int32_t e_o, const float* phi,
                                                  const float* phi_masked, int H, int W, int ws, int shift, ivit_stream_t stream)
{
    IVIT_REQUIRE(ws > 0 && ws * ws == tokens && H > 0 && W > 0 && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws &&
                 windows_per_image == (H / ws) * (W / ws),
                 "ivit_window_attention_i8_unwindow: H=%d W=%d ws=%d shift=%d do not describe %d windows of %d tokens per image", H, W, ws,
                 shift, windows_per_image, tokens);
    return window_attention_launch(qkv, out, ldo, bias_add, mask_region, mask_value, windows, windows_per_image, heads, tokens,
                                   head_dim, m_s, e_s, m_b, e_b, s_attn, m_o, e_o, phi, phi_masked, WinMap{H, W, ws, shift}, stream);
}
