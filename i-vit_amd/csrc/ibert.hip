// ibert.hip -- the I-BERT operator family of the reference (IBERTIntGELU, IBERTIntSoftmax, IBERTIntLayerNorm,
// /root/reference/models/quantization_utils/ibert_modules.py), module-level kernels.
//
// The reference evaluates these operators in float32 tensors holding (mostly) integers.  Each kernel performs the same
// IEEE float32 operations on the same operands in the same order (build flags -ffp-contract=off -fno-fast-math;
// division and square root are the correctly rounded forms), so results are bit-identical to the reference wherever
// the reference itself is deterministic: the three row sums are taken exactly (integers / float64) and rounded once,
// which equals the reference's float32 reduction whenever that reduction is exact (sum < 2^24; oracle/ibert.py counts
// the rows where it is not).  Scalar constants (b_int, c_int, x0_int, scales) are computed by the caller in float32
// exactly as the reference computes them on the host side of every call.
#include <limits.h>
#include <type_traits>

#include "common.h"
#include "rowsum.h"

namespace {

constexpr int NT = 256;
constexpr int WPB = NT / 64;

static inline int ew_grid(int64_t n)
{
    int64_t b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

static inline int grid_for_rows(int64_t rows)
{
    int64_t blocks = (rows + WPB - 1) / WPB;
    return (int)(blocks < 4096 ? blocks : 4096);
}

// ------------------------------------------------------------------------------------------------ GELU
// ibert_modules.py:203-235 with x_int = k (integer activations):
//   sign * ((min(|x|, -b) + b)^2 + c) -> floor(. / 2^6) = sigmoid_int;  out = x * (sigmoid_int + shift_int)
__global__ __launch_bounds__(NT) void ibert_gelu_kernel(const int32_t* k, int64_t n, float b_int, float c_int,
                                                        float shift_int, int32_t* out)
{
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float x = (float)k[i];
        const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);   // :209
        const float a = fminf(fabsf(x), -b_int);                             // :210-211
        const float t = a + b_int;
        float y = t * t;                                                     // :212 (. ** 2)
        y = y + c_int;
        y = sgn * y;                                                         // :213
        y = floorf(y / 64.0f);                                               // :215  2 ** self.n, n = 6
        out[i] = (int)(x * (y + shift_int));                                 // :231
    }
}

// ------------------------------------------------------------------------------------------------ Softmax
struct IbSoftmaxArgs {
    const int32_t* k;
    int64_t ldx;
    int rows, L;
    float x0_int, b_int, c_int, exp_sf, act_sf;
    double M;          // dyadic multiplier of the internal QuantAct(16): m * 2^-e
    float out_div;     // 2^(32 - output_bit + 1)
    int32_t* out;
    int64_t ldo;
    float* exp_out;    // optional [rows, L]: exp_int before the internal QuantAct (calibration statistics)
};

IVIT_DEV float ib_exp_int(float x, const IbSoftmaxArgs& a)
{
    // int_exp, :285-295 (n = 30)
    x = fmaxf(x, 30.0f * a.x0_int);                       // :288
    const float q = floorf(x / a.x0_int);                 // :290
    const float r = x - a.x0_int * q;                     // :291
    float z = r + a.b_int;                                // :279
    z = r * z;                                            // :280
    z = z + a.c_int;                                      // :281
    float e = floorf(z * ldexpf(1.0f, 30 - (int)q));      // :293  2 ** (n - q): exact power of two
    return fmaxf(e, 0.0f);
}

__global__ __launch_bounds__(NT) void ibert_softmax_kernel(IbSoftmaxArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const int32_t* kr = a.k + (int64_t)row * a.ldx;
        int kmax = INT_MIN;
        for (int i = lane; i < a.L; i += 64) kmax = max(kmax, kr[i]);
        kmax = wave_reduce_max_i32(kmax);
        if (a.exp_out) {   // calibration pass: only exp_int is wanted
            for (int i = lane; i < a.L; i += 64)
                a.exp_out[(int64_t)row * a.L + i] = ib_exp_int((float)(kr[i] - kmax), a);
            continue;
        }
        double sum = 0.0;
        for (int i = lane; i < a.L; i += 64) {
            const float e = ib_exp_int((float)(kr[i] - kmax), a);
            // internal QuantAct(16): fixedpoint_mul(exp_int, exp_sf, 16) (quant_utils.py:220-245)
            const float z_int = rintf(e / a.exp_sf);
            double q16 = __builtin_rint((double)z_int * a.M);
            q16 = fmin(fmax(q16, -32768.0), 32767.0);
            const float exp_int = ((float)q16 * a.act_sf) / a.act_sf;        // :309-310
            sum += (double)exp_int;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long b = __double_as_longlong(sum);
            const int lo = __shfl_xor((int)(b & 0xffffffffll), o), hi = __shfl_xor((int)(b >> 32), o);
            sum += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);   // integer-valued terms: exact in any order
        }
        const float ssum = (float)sum;                                        // :311
        const float factor = floorf(4294967296.0f / ssum);                    // :313
        for (int i = lane; i < a.L; i += 64) {
            const float e = ib_exp_int((float)(kr[i] - kmax), a);
            const float z_int = rintf(e / a.exp_sf);
            double q16 = __builtin_rint((double)z_int * a.M);
            q16 = fmin(fmax(q16, -32768.0), 32767.0);
            const float exp_int = ((float)q16 * a.act_sf) / a.act_sf;
            const float o = floorf((exp_int * factor) / a.out_div);          // :314
            a.out[(int64_t)row * a.ldo + i] = (int)o;   // in [0, 2^(output_bit-1)]: the upper end is reachable (a one-hot row)
        }
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
struct IbLnArgs {
    const int32_t* k;
    int64_t ldx;
    int rows, C;
    const float* bias_int;
    const float* s_out;
    float shift_pow2;   // 2 ** self.shift
    float* out;
    int64_t ldo;
};

__global__ __launch_bounds__(NT) void ibert_layernorm_kernel(IbLnArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const int32_t* kr = a.k + (int64_t)row * a.ldx;
        long long sum = 0;
        for (int c = lane; c < C; c += 64) sum += kr[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int lo = __shfl_xor((int)(sum & 0xffffffffll), o), hi = __shfl_xor((int)(sum >> 32), o);
            sum += ((long long)hi << 32) | (unsigned)lo;
        }
        const float mean_int = rintf((float)sum / (float)C);                  // :127
        double var = 0.0;
        for (int c = lane; c < C; c += 64) {
            const float y = (float)kr[c] - mean_int;                          // :128
            const float ys = floorf(y / a.shift_pow2);                        // :129
            const float sq = ys * ys;                                         // :130 float32 square
            var += (double)sq;                                                // :131 (sum of integer-valued terms)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long b = __double_as_longlong(var);
            const int lo = __shfl_xor((int)(b & 0xffffffffll), o), hi = __shfl_xor((int)(b >> 32), o);
            var += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        }
        const float var_int = (float)var;
        const float std_int = floorf(sqrtf(var_int)) * a.shift_pow2;          // :142
        const float factor = floorf(2147483648.0f / std_int);                 // :143
        float* orow = a.out + (int64_t)row * a.ldo;
        for (int c = lane; c < C; c += 64) {
            const float y = (float)kr[c] - mean_int;
            float v = floorf((y * factor) / 2.0f);                            // :144
            v = v + a.bias_int[c];                                            // :151
            orow[c] = v * a.s_out[c];                                         // :153
        }
    }
}

// ================================================================================================
// Literal (float view) forms: the module receives x = q * s (quant_modules.py:387) and starts with x / s
// (ibert_modules.py:126, 226, 303), which is the integer q only for a power-of-two s.  These kernels take the float view and
// its scale and run the reference's float32 sequence on x / s itself -- any scale -- with the row sums in torch's CPU reduction
// order (rowsum.h).  For a power-of-two s they equal the integer-input kernels above.
// ================================================================================================
__global__ __launch_bounds__(NT) void ibert_gelu_f32_kernel(const float* x, int64_t n, float s, float b_int, float c_int,
                                                            float shift_int, float s_out, float* out)
{
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float xi = x[i] / s;                                           // :226
        const float sgn = (xi > 0.0f) ? 1.0f : ((xi < 0.0f) ? -1.0f : 0.0f); // :209
        const float a = fminf(fabsf(xi), -b_int);                            // :210-211
        const float t = a + b_int;
        float y = t * t;                                                     // :212
        y = y + c_int;
        y = sgn * y;                                                         // :213
        y = floorf(y / 64.0f);                                               // :215
        const float p = xi * (y + shift_int);                                // :231
        // :234; a zero product keeps the sign the reference's result has (its fixtures: -0.0 for the negative output scale)
        out[i] = (p == 0.0f ? 0.0f : p) * s_out;
    }
}

struct IbSoftmaxLitArgs {
    const float* x;
    int64_t ldx;
    int rows, L;
    float s;
    IbSoftmaxArgs c;     // constants (k / out / exp_out of it unused)
    float* out;          // float view exp_int * (2 / 2^output_bit)
    float out_sf;
    int64_t ldo;
    float* exp_out;
};

__global__ __launch_bounds__(NT) void ibert_softmax_f32_kernel(IbSoftmaxLitArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const float* xr = a.x + (int64_t)row * a.ldx;
        float xmax = -__builtin_inff();
        for (int i = lane; i < a.L; i += 64) xmax = fmaxf(xmax, xr[i] / a.s);                 // :303-305
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o));
        if (a.exp_out) {
            for (int i = lane; i < a.L; i += 64) a.exp_out[(int64_t)row * a.L + i] = ib_exp_int(xr[i] / a.s - xmax, a.c);
            continue;
        }
        auto ex = [&](int i) {
            const float e = ib_exp_int(xr[i] / a.s - xmax, a.c);                              // :306-307
            const float z_int = rintf(e / a.c.exp_sf);                                        // internal QuantAct(16), :308
            double q16 = __builtin_rint((double)z_int * a.c.M);
            q16 = fmin(fmax(q16, -32768.0), 32767.0);
            return ((float)q16 * a.c.act_sf) / a.c.act_sf;                                    // :309-310
        };
        const float ssum = torch_rowsum(ex, a.L, lane);                                       // :311
        const float factor = floorf(4294967296.0f / ssum);                                    // :313
        for (int i = lane; i < a.L; i += 64)
            a.out[(int64_t)row * a.ldo + i] = floorf((ex(i) * factor) / a.c.out_div) * a.out_sf;   // :314, 319
    }
}

// ------------------------------------------------------------------------------------------------ fused-engine forms
// The integer engine (engine.py, family "ibert") keeps int8 activations between kernels.  Every value the reference's float
// tensors hold at an operator's input is fl(q * s) for the int8 q the engine carries, so each operator below performs the
// literal float32 sequence of the *_f32 kernels above on fl(q * s) -- any scale, power of two or as calibrated -- and fuses
// the QuantAct that follows (quant_utils.py:220-245: z = round(x / s_pre), RNE(float64(z) * m / 2^e), clamp).

// GELU + mlp.qact1 is a function of q alone: 256 entries, written as all 256 rows of the (row max, q) table that
// ivit_shiftgelu_lut_i8 reads, so the same gather kernel applies it.
__global__ __launch_bounds__(NT) void ibert_gelu_lut_kernel(float s, float b_int, float c_int, float shift_int, float s_out, double Mq,
                                                            int8_t* lut)
{
    const int q = (int)threadIdx.x - 128;                 // NT == 256
    const float x = (float)q * s;                         // the float the reference's tensor holds
    const float xi = x / s;                               // :226
    const float sgn = (xi > 0.0f) ? 1.0f : ((xi < 0.0f) ? -1.0f : 0.0f);
    const float a = fminf(fabsf(xi), -b_int);
    const float t = a + b_int;
    float y = t * t;
    y = y + c_int;
    y = sgn * y;
    y = floorf(y / 64.0f);
    const float p = xi * (y + shift_int);
    const float g = (p == 0.0f ? 0.0f : p) * s_out;       // IBERTIntGELU output (float view)
    const float z = rintf(g / s_out);                     // QuantAct: quant_utils.py:220
    // s_out is negative (coeff[0] < 0, :213): requant(z, s) == requant(-z, -s) exactly, Mq is the multiplier of |s_out|
    double r = __builtin_rint((s_out < 0.0f ? -(double)z : (double)z) * Mq);
    r = fmin(fmax(r, -128.0), 127.0);
    const int8_t v = (int8_t)(int)r;
    for (int row = 0; row < 256; ++row) lut[row * 256 + threadIdx.x] = v;
}

// Softmax: exp_int after the internal QuantAct(16), as the float the reference sums and multiplies (:306-310), for every
// (row max qm, q <= qm): table[(qm + 128) * 256 + (q + 128)]; 0 for q > qm.
__global__ __launch_bounds__(NT) void ibert_softmax_table_kernel(float s, IbSoftmaxArgs c, float* table)
{
    const int idx = blockIdx.x * NT + threadIdx.x;     // 65536 entries
    const int qm = (idx >> 8) - 128, q = (idx & 255) - 128;
    float ef = 0.0f;
    if (q <= qm) {
        const float xm = ((float)qm * s) / s, xq = ((float)q * s) / s;       // :303
        const float e = ib_exp_int(xq - xm, c);                              // :305-307
        const float z_int = rintf(e / c.exp_sf);                             // :308
        double q16 = __builtin_rint((double)z_int * c.M);
        q16 = fmin(fmax(q16, -32768.0), 32767.0);
        ef = ((float)q16 * c.act_sf) / c.act_sf;                             // :309-310
    }
    table[idx] = ef;
}

// LayerNorm (ibert_modules.py:126-153) on int8 rows at scale s_in + the QuantAct behind it; one wave per row.
struct IbLnI8Args {
    const int8_t* x;
    int64_t ldx;
    int rows, C;
    float s_in;
    const float* bias_int;
    const float* s_out;     // sf * gamma[c]
    float shift_pow2;
    const uint32_t* m;      // QuantAct: dyadic(s_out[c] / s_next)
    const int32_t* e;
    int8_t* out;
    int64_t ldo;
    int out_blocks;
};

// one row, literally (whole wave)
template <typename TX = int8_t>
IVIT_DEV void ib_ln_row_literal(const IbLnI8Args& a, int row, int lane)
{
    const int C = a.C;
    {
        const TX* xr = reinterpret_cast<const TX*>(a.x) + (int64_t)row * a.ldx;
        auto xint = [&](int c) { return ((float)xr[c] * a.s_in) / a.s_in; };                  // :126 on fl(q * s)
        const float mean_int = rintf(torch_rowsum(xint, C, lane) / (float)C);                 // :127
        auto sq = [&](int c) {
            const float ys = floorf((xint(c) - mean_int) / a.shift_pow2);                     // :128-129
            return ys * ys;                                                                   // :130
        };
        const float var_int = torch_rowsum(sq, C, lane);                                      // :131
        const float std_int = floorf(sqrtf(var_int)) * a.shift_pow2;                          // :142
        const float factor = floorf(2147483648.0f / std_int);                                 // :143
        const BlockRow brow = block_row(row, C);
        for (int c = lane; c < C; c += 64) {
            const float y = xint(c) - mean_int;
            float v = floorf((y * factor) / 2.0f);                                            // :144
            v = v + a.bias_int[c];                                                            // :151
            const float so = a.s_out[c];
            const float xo = v * so;                                                          // :153
            const float z = rintf(xo / so);                                                   // quant_utils.py:220
            double r = __builtin_rint((double)z * dyadic_mult(a.m[c], a.e[c]));               // :229-230
            r = fmin(fmax(r, -128.0), 127.0);
            const int8_t o = (int8_t)(int)r;
            if (a.out_blocks) a.out[block_off(brow, block_col(c))] = o;
            else a.out[(int64_t)row * a.ldo + c] = o;
        }
    }
}

template <typename TX>
__global__ __launch_bounds__(NT) void ibert_layernorm_i8_kernel(IbLnI8Args a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) ib_ln_row_literal<TX>(a, row, lane);
}

// The same result without evaluating the two float32 row sums term by term.  Both sums only feed a rounding:
//   mean_int = rint(fl(S1 / C)):  S1 (torch order) lies within 0.3 of the real sum R = sum q + sum (phi(q) - q) (32 partials of
//     <= C / 32 terms below 2^12, then ~40 additions below 2^17: worst-case rounding 0.26 for C <= 1024), so rint(R / C) is the
//     answer unless frac(R / C) is within 2e-3 of 0.5 (C >= 192); sum (phi(q) - q) <= C 2^-16 only widens that band;
//   std_int = floor(sqrt(S2)) * 2^shift:  the terms floor(y / 2^shift)^2 are integers; V = their exact integer sum.  V < 2^24: every
//     partial sum is exact in float32, S2 == V in any order.  Otherwise |S2 - V| <= 3e-6 V (non-negative terms, < 48 roundings
//     deep): floor(sqrt) is decided unless sqrt(V) is that close to an integer.
// Undecided rows (and std = 0) take ib_ln_row_literal.  Element steps are the literal float32 operations on phi(q) from a
// 256-entry table; the QuantAct's z = round(fl(fl(v * s) / s)) equals v for |v| < 2^21 (two roundings: error < 0.25).
template <int NJ>
__global__ __launch_bounds__(NT, 4) void ibert_layernorm_i8_fast_kernel(IbLnI8Args a)
{
    __shared__ float tphi[256];
    const int C = a.C, nd = C >> 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a lane always works on the same channels (dwords lane + 64 j): their requantisers and biases stay in registers
    // The QuantAct tail -- x = v * s, z = round(x / s), RNE(float64(z) * M) -- by the bracket certificate of the I-ViT LayerNorm
    // (rowops.hip): z = v (1 + eps), |eps| <= 2^-22 (two float32 roundings; v is an integer up to ~2^31), so
    // the number the reference rounds lies between v * lo and v * hi for float32 lo <= M (1 - 1.25 * 2^-22), hi >= M (1 + 1.25 * 2^-22);
    // fma(v, lo, 1.5 * 2^23) == fma(v, hi, 1.5 * 2^23) certifies the result (|v * M| < 2^22: M <= 2^-9 is the loader's contract).
    // A row with an uncertified element is redone literally.
    float lo_r[NJ][4], hi_r[NJ][4], breg[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int d = min(lane + 64 * j, nd - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double M = dyadic_mult(a.m[4 * d + k], a.e[4 * d + k]);
            const double lod = M * (1.0 - 1.25 / 4194304.0), hid = M * (1.0 + 1.25 / 4194304.0);
            float lf = (float)lod, hf = (float)hid;
            if ((double)lf > lod) lf = __int_as_float(__float_as_int(lf) - 1);   // largest float32 <= lod (lod > 0)
            if ((double)hf < hid) hf = __int_as_float(__float_as_int(hf) + 1);   // smallest float32 >= hid
            const float sl = a.s_out[4 * d + k];
            const bool okc = fabsf(sl) >= 1e-30f && fabsf(sl) <= 1e30f && lod > 1e-35 && hid < 1e30;   // see layernorm_i8_kernel
            lo_r[j][k] = okc ? lf : 0.0f;
            hi_r[j][k] = okc ? hf : __builtin_inff();
            breg[j][k] = a.bias_int[4 * d + k];
        }
    }
    {
        const float qf = (float)(tid - 128);              // NT == 256
        const float ph = (qf * a.s_in) / a.s_in;          // :126 on fl(q * s)
        tphi[tid] = ph;
    }
    __syncthreads();
    const bool shift1 = a.shift_pow2 == 1.0f;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const int* xr = reinterpret_cast<const int*>(a.x + (int64_t)row * a.ldx);
        int w[NJ];
        int sq = 0;
        // all loads of the row first (unconditional, clamped address + select: a branch around a load serialises load -> use ->
        // next load, three memory latencies per row instead of one)
#pragma unroll
        for (int j = 0; j < NJ; ++j) w[j] = xr[min(lane + 64 * j, nd - 1)];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            w[j] = (lane + 64 * j < nd) ? w[j] : 0;
            sq = __builtin_amdgcn_sdot4(w[j], 0x01010101, sq, false);
        }
        sq = wave_reduce_sum_i32(sq);
        // R / C = (sum q + sum (phi(q) - q)) / C: |phi(q) - q| <= ulp(128) = 2^-16, so the second term moves the quotient by at most
        // 1.6e-5 and is not evaluated; the float32 quotient of the exact integer sum adds 8e-6.  Both go into the undecided band.
        const float m0 = (float)sq / (float)C;
        const float fr = m0 - floorf(m0);
        if (fabsf(fr - 0.5f) < 2.2e-3f) {                 // wave-uniform
            ib_ln_row_literal(a, row, lane);
            continue;
        }
        const float mean_int = rintf(m0);
        float y0[NJ][4];
        int V = 0;
        auto variance_terms = [&](auto shift1_tag) {      // two copies: the division by 2^shift (~10 instructions) only where shift > 0
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned u = (unsigned)w[j] ^ 0x80808080u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float y = tphi[(u >> (8 * k)) & 255] - mean_int;        // :128
                    y0[j][k] = y;
                    const float ys = decltype(shift1_tag)::value ? floorf(y) : floorf(y / a.shift_pow2);   // :129 (x / 1 == x)
                    const int yi = (lane + 64 * j < nd) ? (int)ys : 0;
                    V += yi * yi;                                                 // :130-131, exact
                }
            }
        };
        if (shift1) variance_terms(std::true_type{});
        else variance_terms(std::false_type{});
        V = wave_reduce_sum_i32(V);
        bool ok = V > 0;
        if (V >= (1 << 24)) {
            const double r = __builtin_sqrt((double)V);
            ok = __builtin_floor(r * (1.0 - 3e-6)) == __builtin_floor(r * (1.0 + 3e-6));
        }
        if (!ok) {
            ib_ln_row_literal(a, row, lane);
            continue;
        }
        const float std_int = floorf(sqrtf((float)V)) * a.shift_pow2;             // :142
        const float factor = floorf(2147483648.0f / std_int);                    // :143
        const BlockRow brow = block_row(row, C);
        unsigned unc = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = lane + 64 * j;
            int o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = floorf((y0[j][k] * factor) / 2.0f);                     // :144
                v = v + breg[j][k];                                               // :151
                const int tl = __float_as_int(__builtin_fmaf(v, lo_r[j][k], 12582912.0f));
                const int th = __float_as_int(__builtin_fmaf(v, hi_r[j][k], 12582912.0f));
                o[k] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);         // low byte = the int8 result
                unc |= (d < nd && tl != th) ? 1u : 0u;                            // uncertified: the row is redone literally
            }
            const int pw = (o[0] & 0xff) | ((o[1] & 0xff) << 8) | ((o[2] & 0xff) << 16) | ((o[3] & 0xff) << 24);
            if (d < nd) {
                if (a.out_blocks) *reinterpret_cast<int*>(a.out + block_off(brow, block_col(4 * d))) = pw;
                else *reinterpret_cast<int*>(a.out + (int64_t)row * a.ldo + 4 * d) = pw;
            }
        }
        if (__builtin_amdgcn_ballot_w64(unc != 0) != 0) ib_ln_row_literal(a, row, lane);   // wave-uniform, rare
    }
}

// ---- int16 rows (the 16-bit residual stream).  No table of phi(q) = fl(fl(q s) / s) fits 65536 inputs, and with terms up to 2^15
// the float32 row sums cannot be decided from the exact integers (the reduction's rounding error reaches the spacing of the
// means): the row sums ARE evaluated in torch's order, but from registers.  A lane holds the elements c = lane + 64 k.  Torch's
// kernel (rowsum.h) keeps 32 accumulator lanes, lane l adding the elements l + 32 i in sequence: i = 2 k of this layout sit in
// wave lane l, i = 2 k + 1 in wave lane l + 32 -- one v_permlane32_swap per register brings them together, then the cascade of
// rowsum.h runs unrolled on registers (C % 64 == 0: no leftover vectors, no scalar tail).  Element steps are the literal float32
// operations; the QuantAct tail takes the bracket certificate of ibert_layernorm_i8_fast_kernel, an uncertified row (and
// std = 0) is redone by ib_ln_row_literal.
template <int NK>
IVIT_DEV float torch_rowsum_regs(const float (&own)[NK], int lane)
{
    float oth[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const unsigned u = (unsigned)__float_as_int(own[k]);
        const v2u r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // lanes < 32: (own, lane + 32's)
        oth[k] = __int_as_float((int)r.y);
    }
    constexpr int size_ilp = 2 * NK;
    constexpr int lg = size_ilp <= 1 ? 0 : size_ilp <= 2 ? 1 : size_ilp <= 4 ? 2 : size_ilp <= 8 ? 3 : size_ilp <= 16 ? 4 : size_ilp <= 32 ? 5 : 6;
    static_assert(size_ilp <= 64, "C <= 2048");
    constexpr int lp = lg / 4 > 4 ? lg / 4 : 4, step = 1 << lp, mask = step - 1;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
    for (int gq = 0; gq < size_ilp / step; ++gq) {
#pragma unroll
        for (int j = 0; j < step; ++j) {
            const int i = gq * step + j;
            acc0 += (i & 1) ? oth[i >> 1] : own[i >> 1];
        }
        const int i = (gq + 1) * step;
        acc1 += acc0; acc0 = 0.f;
        if ((i & (mask << lp)) == 0) {
            acc2 += acc1; acc1 = 0.f;
            if ((i & (mask << (2 * lp))) == 0) { acc3 += acc2; acc2 = 0.f; }
        }
    }
#pragma unroll
    for (int i = size_ilp / step * step; i < size_ilp; ++i) acc0 += (i & 1) ? oth[i >> 1] : own[i >> 1];
    acc0 += acc1; acc0 += acc2; acc0 += acc3;
    const float p1 = __shfl(acc0, (lane + 8) & 63), p2 = __shfl(acc0, (lane + 16) & 63), p3 = __shfl(acc0, (lane + 24) & 63);
    const float v = ((acc0 + p1) + p2) + p3;
    float fin = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) fin += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    return fin;
}

// FASTDIV: x / s_in by the 3-instruction quotient q0 = x * r, e = fma(-s, q0, x), fma(e, r, q0) with r = RN(1 / s_in) -- correctly
// rounded for every 16-bit q at this s_in, which the host checked exhaustively (prepare.markstein_division_ok) before asking for it.
template <int NK, bool FASTDIV>
__global__ __launch_bounds__(NT) void ibert_layernorm_i16_fast_kernel(IbLnI8Args a, float r_in)
{
    const int C = a.C;     // == 64 NK
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lo_r[NK], hi_r[NK], breg[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int c = lane + 64 * k;
        const double M = dyadic_mult(a.m[c], a.e[c]);
        const double lod = M * (1.0 - 1.25 / 4194304.0), hid = M * (1.0 + 1.25 / 4194304.0);
        float lf = (float)lod, hf = (float)hid;
        if ((double)lf > lod) lf = __int_as_float(__float_as_int(lf) - 1);
        if ((double)hf < hid) hf = __int_as_float(__float_as_int(hf) + 1);
        const float sl = a.s_out[c];
        const bool okc = fabsf(sl) >= 1e-30f && fabsf(sl) <= 1e30f && lod > 1e-35 && hid < 1e30;
        lo_r[k] = okc ? lf : 0.0f;
        hi_r[k] = okc ? hf : __builtin_inff();
        breg[k] = a.bias_int[c];
    }
    const float inv_shift = 1.0f / a.shift_pow2;      // a power of two: y / 2^shift == y * 2^-shift exactly
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const int16_t* xr = reinterpret_cast<const int16_t*>(a.x) + (int64_t)row * a.ldx;
        int xq[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) xq[k] = xr[lane + 64 * k];
        float ph[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float x = (float)xq[k] * a.s_in;                                             // fl(q * s)
            if constexpr (FASTDIV) {
                const float q0 = x * r_in;
                const float e = __builtin_fmaf(-a.s_in, q0, x);
                ph[k] = __builtin_fmaf(e, r_in, q0);                                           // :126
            } else {
                ph[k] = x / a.s_in;                                                            // :126
            }
        }
        const float mean_int = rintf(torch_rowsum_regs<NK>(ph, lane) / (float)C);            // :127
        float sq[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            ph[k] = ph[k] - mean_int;                                                          // :128
            const float ys = floorf(ph[k] * inv_shift);                                        // :129
            sq[k] = ys * ys;                                                                   // :130
        }
        const float var_int = torch_rowsum_regs<NK>(sq, lane);                                // :131
        const float std_int = floorf(sqrtf(var_int)) * a.shift_pow2;                           // :142
        if (!(std_int > 0.0f)) {                           // wave-uniform
            ib_ln_row_literal<int16_t>(a, row, lane);
            continue;
        }
        const float factor = floorf(2147483648.0f / std_int);                                 // :143
        unsigned unc = 0;
        int8_t* orow = a.out + (int64_t)row * a.ldo;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float v = floorf((ph[k] * factor) / 2.0f);                                         // :144
            v = v + breg[k];                                                                   // :151
            const int tl = __float_as_int(__builtin_fmaf(v, lo_r[k], 12582912.0f));
            const int th = __float_as_int(__builtin_fmaf(v, hi_r[k], 12582912.0f));
            unc |= (tl != th) ? 1u : 0u;                                                     // uncertified: the row is redone literally
            orow[lane + 64 * k] = (int8_t)clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);
        }
        if (__builtin_amdgcn_ballot_w64(unc != 0) != 0) ib_ln_row_literal<int16_t>(a, row, lane);   // wave-uniform, rare
    }
}

struct IbLnLitArgs {
    const float* x;
    int64_t ldx;
    int rows, C;
    const float* s_in;
    int n_s;
    const float* bias_int;
    const float* s_out;
    float shift_pow2;
    float* out;
    int64_t ldo;
};

__global__ __launch_bounds__(NT) void ibert_layernorm_f32_kernel(IbLnLitArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const float* xr = a.x + (int64_t)row * a.ldx;
        auto xint = [&](int c) { return xr[c] / a.s_in[a.n_s == 1 ? 0 : c]; };              // :126
        const float mean_int = rintf(torch_rowsum(xint, C, lane) / (float)C);                 // :127
        auto sq = [&](int c) {
            const float ys = floorf((xint(c) - mean_int) / a.shift_pow2);                     // :128-129
            return ys * ys;                                                                   // :130
        };
        const float var_int = torch_rowsum(sq, C, lane);                                      // :131
        const float std_int = floorf(sqrtf(var_int)) * a.shift_pow2;                          // :142
        const float factor = floorf(2147483648.0f / std_int);                                 // :143
        float* orow = a.out + (int64_t)row * a.ldo;
        for (int c = lane; c < C; c += 64) {
            const float y = xint(c) - mean_int;
            float v = floorf((y * factor) / 2.0f);                                            // :144
            v = v + a.bias_int[c];                                                            // :151
            orow[c] = v * a.s_out[c];                                                         // :153
        }
    }
}

}  // namespace

// ================================================================================================
IVIT_EXPORT int ivit_ibert_gelu_i32(const int32_t* k, int64_t n, float b_int, float c_int, float shift_int, int32_t* out,
                                    ivit_stream_t stream)
{
    IVIT_REQUIRE(k && out && n > 0, "ivit_ibert_gelu_i32: bad operand");
    IVIT_REQUIRE(b_int < 0.0f, "ivit_ibert_gelu_i32: b_int must be negative (floor(-1.769 / (s / 1.4142)))");
    hipLaunchKernelGGL(ibert_gelu_kernel, dim3(ew_grid(n)), dim3(NT), 0, ivit_stream(stream), k, n, b_int, c_int,
                       shift_int, out);
    IVIT_CHECK_LAUNCH("ivit_ibert_gelu_i32");
}

IVIT_EXPORT int ivit_ibert_softmax_i32(const int32_t* k, int64_t ldx, int rows, int L, float x0_int, float b_int,
                                          float c_int, float exp_sf, float act_sf, uint32_t m_act, int32_t e_act,
                                          int output_bit, int32_t* out, int64_t ldo, float* exp_out, ivit_stream_t stream)
{
    IVIT_REQUIRE(k && (out || exp_out) && rows > 0 && L > 0 && ldx >= L, "ivit_ibert_softmax_i32: bad operand");
    IVIT_REQUIRE(!out || ldo >= L, "ivit_ibert_softmax_i32: ldo < L");
    IVIT_REQUIRE(x0_int < 0.0f && exp_sf > 0.0f && act_sf > 0.0f, "ivit_ibert_softmax_i32: bad scalar constants");
    IVIT_REQUIRE(output_bit >= 2 && output_bit <= 16, "ivit_ibert_softmax_i32: output_bit=%d unsupported", output_bit);
    IbSoftmaxArgs a{};
    a.k = k; a.ldx = ldx; a.rows = rows; a.L = L;
    a.x0_int = x0_int; a.b_int = b_int; a.c_int = c_int; a.exp_sf = exp_sf; a.act_sf = act_sf;
    a.M = ivit_dyadic_to_double(m_act, e_act);
    a.out_div = __builtin_ldexpf(1.0f, 32 - output_bit + 1);
    a.out = out; a.ldo = ldo; a.exp_out = exp_out;
    hipLaunchKernelGGL(ibert_softmax_kernel, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_ibert_softmax_i32");
}

IVIT_EXPORT int ivit_ibert_layernorm_i32_f32(const int32_t* k, int64_t ldx, int rows, int C, const float* bias_int,
                                             const float* s_out, float shift_pow2, float* out, int64_t ldo,
                                             ivit_stream_t stream)
{
    IVIT_REQUIRE(k && out && bias_int && s_out && rows > 0 && C > 0 && ldx >= C && ldo >= C,
                 "ivit_ibert_layernorm_i32_f32: bad operand");
    IVIT_REQUIRE(shift_pow2 >= 1.0f, "ivit_ibert_layernorm_i32_f32: shift_pow2 = 2^shift must be >= 1");
    IbLnArgs a{k, ldx, rows, C, bias_int, s_out, shift_pow2, out, ldo};
    hipLaunchKernelGGL(ibert_layernorm_kernel, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_ibert_layernorm_i32_f32");
}

IVIT_EXPORT int ivit_ibert_gelu_f32_f32(const float* x, int64_t n, float s, float b_int, float c_int, float shift_int,
                                        float s_out, float* out, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && n > 0 && s != 0.0f, "ivit_ibert_gelu_f32_f32: bad operand");
    IVIT_REQUIRE(b_int < 0.0f, "ivit_ibert_gelu_f32_f32: b_int must be negative (floor(-1.769 / (s / 1.4142)))");
    hipLaunchKernelGGL(ibert_gelu_f32_kernel, dim3(ew_grid(n)), dim3(NT), 0, ivit_stream(stream), x, n, s, b_int, c_int,
                       shift_int, s_out, out);
    IVIT_CHECK_LAUNCH("ivit_ibert_gelu_f32_f32");
}

IVIT_EXPORT int ivit_ibert_softmax_f32_f32(const float* x, int64_t ldx, int rows, int L, float s, float x0_int, float b_int,
                                           float c_int, float exp_sf, float act_sf, uint32_t m_act, int32_t e_act,
                                           int output_bit, float* out, int64_t ldo, float* exp_out, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && (out || exp_out) && rows > 0 && L > 0 && ldx >= L && s > 0.0f, "ivit_ibert_softmax_f32_f32: bad operand");
    IVIT_REQUIRE(!out || ldo >= L, "ivit_ibert_softmax_f32_f32: ldo < L");
    IVIT_REQUIRE(x0_int < 0.0f && exp_sf > 0.0f && act_sf > 0.0f, "ivit_ibert_softmax_f32_f32: bad scalar constants");
    IVIT_REQUIRE(output_bit >= 2 && output_bit <= 16, "ivit_ibert_softmax_f32_f32: output_bit=%d unsupported", output_bit);
    IbSoftmaxLitArgs a{};
    a.x = x; a.ldx = ldx; a.rows = rows; a.L = L; a.s = s;
    a.c.x0_int = x0_int; a.c.b_int = b_int; a.c.c_int = c_int; a.c.exp_sf = exp_sf; a.c.act_sf = act_sf;
    a.c.M = ivit_dyadic_to_double(m_act, e_act);
    a.c.out_div = __builtin_ldexpf(1.0f, 32 - output_bit + 1);
    a.out = out; a.ldo = ldo; a.exp_out = exp_out;
    a.out_sf = 2.0f / __builtin_ldexpf(1.0f, output_bit);           // :317
    hipLaunchKernelGGL(ibert_softmax_f32_kernel, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_ibert_softmax_f32_f32");
}

IVIT_EXPORT int ivit_ibert_layernorm_f32_f32(const float* x, int64_t ldx, int rows, int C, const float* s_in, int n_s,
                                             const float* bias_int, const float* s_out, float shift_pow2, float* out,
                                             int64_t ldo, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && s_in && out && bias_int && s_out && rows > 0 && C > 0 && ldx >= C && ldo >= C && (n_s == 1 || n_s == C),
                 "ivit_ibert_layernorm_f32_f32: bad operand");
    IVIT_REQUIRE(shift_pow2 >= 1.0f, "ivit_ibert_layernorm_f32_f32: shift_pow2 = 2^shift must be >= 1");
    IbLnLitArgs a{x, ldx, rows, C, s_in, n_s, bias_int, s_out, shift_pow2, out, ldo};
    hipLaunchKernelGGL(ibert_layernorm_f32_kernel, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_ibert_layernorm_f32_f32");
}

IVIT_EXPORT int ivit_ibert_gelu_build_lut(float s, float b_int, float c_int, float shift_int, float s_out, uint32_t m_q, int32_t e_q,
                                          int8_t* lut, ivit_stream_t stream)
{
    IVIT_REQUIRE(lut && s != 0.0f && s_out != 0.0f && b_int < 0.0f, "ivit_ibert_gelu_build_lut: bad operand");
    hipLaunchKernelGGL(ibert_gelu_lut_kernel, dim3(1), dim3(NT), 0, ivit_stream(stream), s, b_int, c_int, shift_int, s_out,
                       ivit_dyadic_to_double(m_q, e_q), lut);
    IVIT_CHECK_LAUNCH("ivit_ibert_gelu_build_lut");
}

IVIT_EXPORT int ivit_ibert_softmax_build_table(float s, float x0_int, float b_int, float c_int, float exp_sf, float act_sf,
                                               uint32_t m_act, int32_t e_act, float* table, ivit_stream_t stream)
{
    IVIT_REQUIRE(table && s > 0.0f && x0_int < 0.0f && exp_sf > 0.0f && act_sf > 0.0f, "ivit_ibert_softmax_build_table: bad operand");
    IbSoftmaxArgs c{};
    c.x0_int = x0_int; c.b_int = b_int; c.c_int = c_int; c.exp_sf = exp_sf; c.act_sf = act_sf;
    c.M = ivit_dyadic_to_double(m_act, e_act);
    hipLaunchKernelGGL(ibert_softmax_table_kernel, dim3(65536 / NT), dim3(NT), 0, ivit_stream(stream), s, c, table);
    IVIT_CHECK_LAUNCH("ivit_ibert_softmax_build_table");
}

IVIT_EXPORT int ivit_ibert_layernorm_i8(const int8_t* x, int64_t ldx, int rows, int C, float s_in, const float* bias_int,
                                        const float* s_out, float shift_pow2, const uint32_t* m, const int32_t* e, int8_t* out,
                                        int64_t ldo, int out_blocks, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && bias_int && s_out && m && e && rows > 0 && C > 0 && ldx >= C && ldo >= C && s_in > 0.0f,
                 "ivit_ibert_layernorm_i8: bad operand");
    IVIT_REQUIRE(shift_pow2 >= 1.0f, "ivit_ibert_layernorm_i8: shift_pow2 = 2^shift must be >= 1");
    IVIT_REQUIRE(out_blocks == 0 || (out_blocks == 1 && C % 64 == 0 && ldo == C && ((int64_t)rows + 15) * C < 2147483648ll),
                 "ivit_ibert_layernorm_i8: block-layout output needs C %% 64 == 0, ldo == C and a buffer below 2 GiB");
    IbLnI8Args a{x, ldx, rows, C, s_in, bias_int, s_out, shift_pow2, m, e, out, ldo, out_blocks};
    // the fast form (sums decided without term-by-term float32 additions) needs dword rows and C <= 1024 (its error bound)
    const bool fast = C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 4 == 0) && ((uintptr_t)out % 4 == 0);
    if (fast) {
        const size_t lds = 0;
        const int nj = (C / 4 + 63) / 64;
        // one resident set of workgroups (4 per CU at 121 VGPRs): the per-lane constants are set up once per wave
        const int want = grid_for_rows(rows);
        const dim3 grid(want < 1024 ? want : 1024), blk(NT);
        hipStream_t st = ivit_stream(stream);
        if (nj <= 1) hipLaunchKernelGGL(ibert_layernorm_i8_fast_kernel<1>, grid, blk, lds, st, a);
        else if (nj <= 2) hipLaunchKernelGGL(ibert_layernorm_i8_fast_kernel<2>, grid, blk, lds, st, a);
        else if (nj <= 3) hipLaunchKernelGGL(ibert_layernorm_i8_fast_kernel<3>, grid, blk, lds, st, a);
        else hipLaunchKernelGGL(ibert_layernorm_i8_fast_kernel<4>, grid, blk, lds, st, a);
        IVIT_CHECK_LAUNCH("ivit_ibert_layernorm_i8");
    }
    hipLaunchKernelGGL(ibert_layernorm_i8_kernel<int8_t>, dim3(grid_for_rows(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_ibert_layernorm_i8");
}

IVIT_EXPORT int ivit_ibert_layernorm_i16_i8(const int16_t* x, int64_t ldx, int rows, int C, float s_in, const float* bias_int,
                                            const float* s_out, float shift_pow2, const uint32_t* m, const int32_t* e, int8_t* out,
                                            int64_t ldo, ivit_stream_t stream)
{
    return ivit_ibert_layernorm_i16_i8_ex(x, ldx, rows, C, s_in, bias_int, s_out, shift_pow2, m, e, out, ldo, 0, stream);
}

IVIT_EXPORT int ivit_ibert_layernorm_i16_i8_ex(const int16_t* x, int64_t ldx, int rows, int C, float s_in, const float* bias_int,
                                               const float* s_out, float shift_pow2, const uint32_t* m, const int32_t* e, int8_t* out,
                                               int64_t ldo, int fast_division, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && bias_int && s_out && m && e && rows > 0 && C > 0 && ldx >= C && ldo >= C && s_in > 0.0f,
                 "ivit_ibert_layernorm_i16_i8: bad operand");
    IVIT_REQUIRE(shift_pow2 >= 1.0f, "ivit_ibert_layernorm_i16_i8: shift_pow2 = 2^shift must be >= 1");
    IbLnI8Args a{reinterpret_cast<const int8_t*>(x), ldx, rows, C, s_in, bias_int, s_out, shift_pow2, m, e, out, ldo, 0};
    const dim3 grid(grid_for_rows(rows)), blk(NT);
    hipStream_t st = ivit_stream(stream);
    const float r_in = 1.0f / s_in;
#define IB_LN16(NKv)                                                                                                   \
    do {                                                                                                               \
        if (fast_division) hipLaunchKernelGGL((ibert_layernorm_i16_fast_kernel<NKv, true>), grid, blk, 0, st, a, r_in); \
        else hipLaunchKernelGGL((ibert_layernorm_i16_fast_kernel<NKv, false>), grid, blk, 0, st, a, r_in);              \
    } while (0)
    if (C == 192) IB_LN16(3);
    else if (C == 384) IB_LN16(6);
    else if (C == 768) IB_LN16(12);
    else if (C == 1024) IB_LN16(16);
    else hipLaunchKernelGGL(ibert_layernorm_i8_kernel<int16_t>, grid, blk, 0, st, a);
#undef IB_LN16
    IVIT_CHECK_LAUNCH("ivit_ibert_layernorm_i16_i8");
}
