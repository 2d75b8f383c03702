// literal.hip -- the module-level (float view in, float / int8 out) forms of I-LayerNorm and Shiftmax that restate the
// reference's float32 sequence step by step, for ANY input scale.
//
// The reference's operators receive x = q * s (float32, quant_modules.py:387) and start with x / s (ivit_modules.py:36,
// 165).  For a power-of-two s that quotient is the integer q and the integer kernels of rowops.hip apply; for a scale "as
// calibrated" it is phi = fl(fl(q*s)/s), a float32 next to q, and
//   * IVITIntLayerNorm takes the MEAN over the phi values in float32 (torch's CPU reduction order decides exact .5
//     ties) before truncating them with .to(int32) (:37-38);
//   * IVITIntSoftmax discards its .to(int32) (:166) and runs the whole float32 sequence on phi.
// The module mirror (quantization_utils/ivit_modules.py) calls these kernels; the fused engine carries the same effect as
// 256-entry tables in front of its integer kernels (ivit_layernorm_i8_compat, ivit_attention_fused_i8_compat).
// One wave per row; every arithmetic step is one IEEE float32 operation (-ffp-contract=off, correctly rounded division).
#include "common.h"
#include "rowsum.h"

namespace {

constexpr int NT = 256;
constexpr int WPB = NT / 64;

struct LnLitArgs {
    const float* x;
    int64_t ldx;
    int rows, C;
    const float* s_in;
    int n_s;
    const float* bias_int;
    const float* s_ln;
    float* out;
    int64_t ldo;
    int outer;     // > 0: the mean is taken over a transposed view whose contiguous extent is `outer` (row = .. * outer + column):
                   // torch's outer-reduction order (rowsum.h)
};

// IVITIntLayerNorm.forward, ivit_modules.py:36-63, line by line
__global__ __launch_bounds__(NT) void layernorm_f32_f32_kernel(LnLitArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const float* xr = a.x + (int64_t)row * a.ldx;
        auto xint = [&](int c) { return xr[c] / a.s_in[a.n_s == 1 ? 0 : c]; };   // :36  x / scaling_factor
        const float S = a.outer ? torch_outer_rowsum(xint, C, row % a.outer >= (a.outer & ~31)) : torch_rowsum(xint, C, lane);
        const float mean = S / (float)C;                                          // :37  mean = sum / C
        const int mean_int = (int)rintf(mean);                                    //      round_ste
        long long var = 0;
        for (int c = lane; c < C; c += 64) {
            const long long d = (long long)(int)truncf(xint(c)) - mean_int;       // :38-40  .to(int32); y = x - mean
            var += d * d;                                                         // :41-42  int32 squares, int64 sum
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int vlo = __shfl_xor((int)(var & 0xffffffffll), o);
            const int vhi = __shfl_xor((int)(var >> 32), o);
            var += ((long long)vhi << 32) | (unsigned)vlo;
        }
        const float varf = (float)var;
        float t = 65536.0f;                                                        // :45-49
#pragma unroll 1
        for (int it = 0; it < 10; ++it) t = floorf((t + floorf(varf / t)) * 0.5f);
        const float factor = floorf((1.0f / t) * 2147483648.0f);                  // :51
        for (int c = lane; c < C; c += 64) {
            const float dl = (float)((int)truncf(xint(c)) - mean_int);
            const float v = floorf((dl * factor) * 0.5f);                         // :52
            const float y = v + a.bias_int[c];                                    // :61
            a.out[(int64_t)row * a.ldo + c] = y * a.s_ln[c];                      // :63
        }
    }
}

struct SmLitArgs {
    const float* x;
    int64_t ldx;
    int rows, L;
    float s, x0;
    void* out;       // int8 (output_bit = 8) or int16 (output_bit 9..16)
    int64_t ldo;
    float two_sh;    // 2^(31 - output_bit + 1), ivit_modules.py:175
};

// int_exp_shift on a float32 argument, ivit_modules.py:150-162 verbatim (n = 15)
IVIT_DEV float shiftexp_lit(float d, float x0)
{
    float x = (d + floorf(d / 2.0f)) - floorf(d / 16.0f);     // :151
    x = fmaxf(x, 15.0f * x0);                                  // :155
    const float q = floorf(x / x0);                            // :157
    const float r = x - x0 * q;                                // :158
    float ex = r / 2.0f - x0;                                  // :159
    ex = floorf(ex * ldexpf(1.0f, 15 - (int)q));               // :160
    return fmaxf(ex, 0.0f);
}

// IVITIntSoftmax.forward, ivit_modules.py:164-176, line by line
template <typename TO>
__global__ __launch_bounds__(NT) void shiftmax_f32_kernel(SmLitArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * WPB + wave; row < a.rows; row += gridDim.x * WPB) {
        const float* xr = a.x + (int64_t)row * a.ldx;
        float xmax = -__builtin_inff();
        for (int i = lane; i < a.L; i += 64) xmax = fmaxf(xmax, xr[i] / a.s);     // :165, :167
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o));
        auto ex = [&](int i) { return shiftexp_lit(xr[i] / a.s - xmax, a.x0); };  // :168-170
        float S = torch_rowsum(ex, a.L, lane);                                    // :171
        S = fminf(S, 2147483648.0f);                                              // :173
        const float factor = floorf((1.0f / S) * 2147483648.0f);                  // :174
        for (int i = lane; i < a.L; i += 64)
            reinterpret_cast<TO*>(a.out)[(int64_t)row * a.ldo + i] = (TO)floorf((ex(i) * factor) / a.two_sh);   // :175
    }
}

static inline int rows_grid(int64_t rows)
{
    int64_t b = (rows + WPB - 1) / WPB;
    return (int)(b < 8192 ? b : 8192);
}

}  // namespace

IVIT_EXPORT int ivit_layernorm_f32_f32_ex(const float* x, int64_t ldx, int rows, int C, const float* s_in, int n_s,
                                          const float* bias_int, const float* s_ln, float* out, int64_t ldo, int outer_mean,
                                          ivit_stream_t stream)
{
    IVIT_REQUIRE(x && s_in && bias_int && s_ln && out, "ivit_layernorm_f32_f32: NULL operand");
    IVIT_REQUIRE(rows > 0 && C > 0 && C <= 16384 && ldx >= C && ldo >= C && (n_s == 1 || n_s == C) && outer_mean >= 0 && (outer_mean == 0 || rows % outer_mean == 0),
                 "ivit_layernorm_f32_f32: bad shape rows=%d C=%d n_s=%d", rows, C, n_s);
    LnLitArgs a{x, ldx, rows, C, s_in, n_s, bias_int, s_ln, out, ldo, outer_mean};
    hipLaunchKernelGGL(layernorm_f32_f32_kernel, dim3(rows_grid(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_layernorm_f32_f32");
}

IVIT_EXPORT int ivit_layernorm_f32_f32(const float* x, int64_t ldx, int rows, int C, const float* s_in, int n_s,
                                       const float* bias_int, const float* s_ln, float* out, int64_t ldo,
                                       ivit_stream_t stream)
{
    return ivit_layernorm_f32_f32_ex(x, ldx, rows, C, s_in, n_s, bias_int, s_ln, out, ldo, 0, stream);
}

IVIT_EXPORT int ivit_shiftmax_f32_i8(const float* x, int64_t ldx, int rows, int L, float s, int8_t* out, int64_t ldo,
                                     ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && rows > 0 && L > 0 && ldx >= L && ldo >= L, "ivit_shiftmax_f32_i8: bad operand");
    IVIT_REQUIRE(s > 0.0f, "ivit_shiftmax_f32_i8: scale must be positive");
    const float x0 = __builtin_floorf((1.0f / s) * -1.0f);      // ivit_modules.py:154
    IVIT_REQUIRE(x0 <= -1.0f && x0 >= -1048576.0f, "ivit_shiftmax_f32_i8: x0=%g out of range", (double)x0);
    SmLitArgs a{x, ldx, rows, L, s, x0, out, ldo, 16777216.0f};
    hipLaunchKernelGGL(shiftmax_f32_kernel<int8_t>, dim3(rows_grid(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_shiftmax_f32_i8");
}

IVIT_EXPORT int ivit_shiftmax_f32_i16(const float* x, int64_t ldx, int rows, int L, float s, int output_bit, int16_t* out,
                                      int64_t ldo, ivit_stream_t stream)
{
    IVIT_REQUIRE(x && out && rows > 0 && L > 0 && ldx >= L && ldo >= L, "ivit_shiftmax_f32_i16: bad operand");
    IVIT_REQUIRE(s > 0.0f && output_bit >= 2 && output_bit <= 16, "ivit_shiftmax_f32_i16: scale / output_bit out of range");
    const float x0 = __builtin_floorf((1.0f / s) * -1.0f);      // ivit_modules.py:154
    IVIT_REQUIRE(x0 <= -1.0f && x0 >= -1048576.0f, "ivit_shiftmax_f32_i16: x0=%g out of range", (double)x0);
    SmLitArgs a{x, ldx, rows, L, s, x0, out, ldo, __builtin_ldexpf(1.0f, 31 - output_bit + 1)};
    hipLaunchKernelGGL(shiftmax_f32_kernel<int16_t>, dim3(rows_grid(rows)), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_shiftmax_f32_i16");
}
