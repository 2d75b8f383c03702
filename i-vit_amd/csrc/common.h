// common.h -- shared host/device helpers of libivit_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ivit_hip.h"
#include "../../include/ivit_hip_debug.h"

#define IVIT_EXPORT extern "C" __attribute__((visibility("default")))

// IVIT_LAB = 0: libivit_hip.so, the product (stateless, no measurement hooks); 1: libivit_hip_lab.so (gemm_common.h)
#ifndef IVIT_LAB
#define IVIT_LAB 0
#endif

// ---- host side -------------------------------------------------------------------------------
void ivit_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define IVIT_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ivit_set_error(__VA_ARGS__);   \
            return IVIT_ERR_INVALID;       \
        }                                  \
    } while (0)

#define IVIT_CHECK_LAUNCH(name)                                                       \
    do {                                                                              \
        hipError_t err__ = hipGetLastError();                                         \
        if (err__ != hipSuccess) {                                                    \
            ivit_set_error("%s: launch failed: %s", name, hipGetErrorString(err__));  \
            return IVIT_ERR_LAUNCH;                                                   \
        }                                                                             \
        return IVIT_OK;                                                               \
    } while (0)

static inline hipStream_t ivit_stream(ivit_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// A dyadic requantiser passed by value: M = m * 2^-e as an exact double.
static inline double ivit_dyadic_to_double(uint32_t m, int32_t e) { return __builtin_ldexp((double)m, -e); }

// ---- device side -----------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define IVIT_DEV __device__ __forceinline__

// 2^52 + 2^51: adding it to |v| < 2^51 rounds v to an integer (RNE) and leaves that integer,
// two's complement, in the low mantissa bits.
#define IVIT_MAGIC 6755399441055744.0

IVIT_DEV double dyadic_mult(uint32_t m, int32_t e) { return __builtin_ldexp((double)m, -e); }

// Combine a value over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the four 16-lane rows of the wave), every lane ends with the result:
// v_permlane16_swap / v_permlane32_swap of the value with itself instead of two ds_bpermute_b32 round trips (__shfl_xor).
template <typename Op>
IVIT_DEV unsigned rows_allreduce_u32(unsigned v, Op op)
{
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    const v2u_ a = __builtin_amdgcn_permlane16_swap(v, v, false, false);     // .x: rows 0 0 2 2, .y: rows 1 1 3 3
    v = op(a.x, a.y);
    const v2u_ b = __builtin_amdgcn_permlane32_swap(v, v, false, false);     // .x: lower half twice, .y: upper half twice
    return op(b.x, b.y);
}
IVIT_DEV int rows_allmin_i32(int v) { return (int)rows_allreduce_u32((unsigned)v, [](unsigned x, unsigned y) { return (unsigned)min((int)x, (int)y); }); }
IVIT_DEV int rows_allmax_i32(int v) { return (int)rows_allreduce_u32((unsigned)v, [](unsigned x, unsigned y) { return (unsigned)max((int)x, (int)y); }); }
IVIT_DEV unsigned rows_allsum_u32(unsigned v) { return rows_allreduce_u32(v, [](unsigned x, unsigned y) { return x + y; }); }

// All-reduce butterflies without LDS (a __shfl_xor with a constant offset compiles to ds_bpermute_b32: an LDS round trip per step).
// Steps 1 / 2: DPP quad permutes; 4 / 8: row_half_mirror / row_mirror (the partner lies in the other half of the group, whose lanes
// all hold that half's total by then); 16 / 32: v_permlane16_swap / v_permlane32_swap of the value with itself.  LPR = lanes that
// share a result (power of two, groups aligned to LPR).
#define IVIT_DPP_U32(x, CTRL) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), CTRL, 0xf, 0xf, false))
template <int LPR>
IVIT_DEV int lanes_allsum_i32(int v)
{
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    if constexpr (LPR >= 2) v += (int)IVIT_DPP_U32(v, 0xB1);
    if constexpr (LPR >= 4) v += (int)IVIT_DPP_U32(v, 0x4E);
    if constexpr (LPR >= 8) v += (int)IVIT_DPP_U32(v, 0x141);
    if constexpr (LPR >= 16) v += (int)IVIT_DPP_U32(v, 0x140);
    if constexpr (LPR >= 32) { const v2u_ r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = (int)(r.x + r.y); }
    if constexpr (LPR >= 64) { const v2u_ r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = (int)(r.x + r.y); }
    return v;
}
template <int LPR>
IVIT_DEV unsigned long long lanes_allsum_u64(unsigned long long v)
{
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
#define IVIT_STEP64(CTRL) { const unsigned lo_ = IVIT_DPP_U32((unsigned)v, CTRL), hi_ = IVIT_DPP_U32((unsigned)(v >> 32), CTRL); v += ((unsigned long long)hi_ << 32) | lo_; }
    if constexpr (LPR >= 2) IVIT_STEP64(0xB1)
    if constexpr (LPR >= 4) IVIT_STEP64(0x4E)
    if constexpr (LPR >= 8) IVIT_STEP64(0x141)
    if constexpr (LPR >= 16) IVIT_STEP64(0x140)
#undef IVIT_STEP64
    if constexpr (LPR >= 32) {
        const v2u_ a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        const v2u_ b = __builtin_amdgcn_permlane16_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
        v = (((unsigned long long)b.x << 32) | a.x) + (((unsigned long long)b.y << 32) | a.y);
    }
    if constexpr (LPR >= 64) {
        const v2u_ a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        const v2u_ b = __builtin_amdgcn_permlane32_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
        v = (((unsigned long long)b.x << 32) | a.x) + (((unsigned long long)b.y << 32) | a.y);
    }
    return v;
}
IVIT_DEV int wave_allmax_i32(int v)
{
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    v = max(v, (int)IVIT_DPP_U32(v, 0xB1));
    v = max(v, (int)IVIT_DPP_U32(v, 0x4E));
    v = max(v, (int)IVIT_DPP_U32(v, 0x141));
    v = max(v, (int)IVIT_DPP_U32(v, 0x140));
    const v2u_ r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = max((int)r.x, (int)r.y);
    const v2u_ q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)q.x, (int)q.y);
}

// The per-channel constants of the int8-output LayerNorm kernels, once per workgroup into LDS: bias_int and the float32 bracket
// [lo, hi] of the output requantiser's multiplier (the certificate of layernorm_i8_kernel, rowops.hip).  ALL global loads of a pass
// (two channels per thread: C <= 2 x NTHREADS is one pass) are issued before the first use -- written as one loop with the loads where
// they are consumed, the compiler waited for m / e / s_ln, did the float64 arithmetic, then requested bias_int and waited again: two
// global latencies per channel and thread, 3-4 us of an 8.6 us launch at 12 608 rows of 384 channels (DeiT-S, late round 4).
template <int NTHREADS>
IVIT_DEV void ln_build_table(const uint32_t* m, const int32_t* e, const float* s_ln, const float* bias_int, int C, float* t_bias, float* t_lo,
                             float* t_hi)
{
    for (int c0 = threadIdx.x; c0 < C; c0 += 2 * NTHREADS) {
        uint32_t mm[2];
        int32_t ee[2];
        float sl[2], bb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = min(c0 + u * NTHREADS, C - 1);      // unconditional, clamped
            mm[u] = m[c];
            ee[u] = e[c];
            sl[u] = s_ln[c];
            bb[u] = bias_int[c];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = c0 + u * NTHREADS;
            const double M = dyadic_mult(mm[u], ee[u]);
            const double lod = M * (1.0 - 1.25 / 4194304.0), hid = M * (1.0 + 1.25 / 4194304.0);
            float lf = (float)lod, hf = (float)hid;
            if ((double)lf > lod) lf = __int_as_float(__float_as_int(lf) - 1);   // largest float32 <= lod (lod > 0)
            if ((double)hf < hid) hf = __int_as_float(__float_as_int(hf) + 1);   // smallest float32 >= hid
            const bool ok = fabsf(sl[u]) >= 1e-30f && fabsf(sl[u]) <= 1e30f && lod > 1e-35 && hid < 1e30;   // see layernorm_i8_kernel
            if (c < C) {
                t_bias[c] = bb[u];
                t_lo[c] = ok ? lf : 0.0f;
                t_hi[c] = ok ? hf : __builtin_inff();
            }
        }
    }
}

IVIT_DEV int clamp_i32(int v, int lo, int hi) { return min(max(v, lo), hi); }

// ---- "block" operand layout (IVIT_LAYOUT_BLOCKS, include/ivit_hip.h): an int8 matrix X[rows][K], K % 64 == 0, stored as
// 1 KB blocks of 16 rows x 64 bytes, block (rb, kb) at ((rb * (K / 64)) + kb) * 1024, and inside a block the 16-byte chunk
// (r, c) at position 4r + (c ^ ((r >> 2) & 3)): exactly the order in which one global_load_lds_dwordx4 of the GEMM main
// loop lays a piece into its swizzled LDS stage, so that the instruction reads 1 KB CONTIGUOUS (8 full cache lines)
// instead of 16 half lines -- the LDS-DMA acceptance rate is per cache line touched.
// Rows are padded to a multiple of 16 (the buffer holds ceil(rows / 16) * 16 * K bytes).
// Producers address it as row part + column part + one XOR (32-bit byte offsets: buffers below 4 GB):
//   block_off(block_row(r, K), block_col(c)) == ivit_block_offset(r, c, K)
struct BlockRow { unsigned base, rs; };   // per row: block-row origin + 64 * (row & 15); swizzle term (row >> 2) & 3
struct BlockCol { unsigned base, cc; };   // per byte column: column-block origin + (c & 15); chunk index (c >> 4) & 3
__device__ __forceinline__ BlockRow block_row(int r, int K)
{
    const int rl = r & 15;
    return BlockRow{(unsigned)((r >> 4) * (K >> 6)) * 1024u + (unsigned)(rl << 6), (unsigned)((rl >> 2) & 3)};
}
__device__ __forceinline__ BlockCol block_col(int c) { return BlockCol{(unsigned)(c >> 6) * 1024u + (unsigned)(c & 15), (unsigned)((c >> 4) & 3)}; }
__device__ __forceinline__ unsigned block_off(BlockRow r, BlockCol c) { return r.base + c.base + ((c.cc ^ r.rs) << 4); }

__host__ __device__ __forceinline__ int64_t ivit_block_offset(int64_t r, int c, int K)
{
    const int rl = (int)(r & 15);
    return (((r >> 4) * (K >> 6)) + (c >> 6)) * 1024 + ((((rl << 2) + (((c >> 4) & 3) ^ ((rl >> 2) & 3)))) << 4) + (c & 15);
}

// out = RNE(z * m / 2^e) for |z*m| < 2^53 and |z*M| < 2^31 (GEMM accumulators, int8 operands):
// the reference's double product (quant_utils.py:229) is exact in that range, so one fused
// multiply-add against the magic constant performs the single RNE rounding of :230.
IVIT_DEV int requant_exact(int z, double M)
{
    double t = __builtin_fma((double)z, M, IVIT_MAGIC);
    return (int)(unsigned)__double_as_longlong(t);
}

// General form, bit-faithful to quant_utils.py:229-230 for any float-representable z:
// the product rounds to 53 bits first (as the reference's float64 multiply does), then RNE.
IVIT_DEV double requant_double(double z, double M)
{
    double p = z * M;  // M = m * 2^-e: same rounding as z*m, then exact scaling
    return __builtin_rint(p);
}

IVIT_DEV int wave_reduce_max_i32(int v) { return wave_allmax_i32(v); }      // every lane ends with the result (DPP / permlane swaps, no LDS)

IVIT_DEV int wave_reduce_sum_i32(int v) { return lanes_allsum_i32<64>(v); }

// int_exp_shift of the reference (ivit_modules.py:89-103 / :150-162) for one integer d <= 0,
// x0 = floor(-1/s) (negative), n = 23 (GELU) or 15 (softmax).  Every intermediate is an exact
// small integer or half-integer in the reference's float32 arithmetic, so integer ops reproduce it:
//   x = d + floor(d/2) - floor(d/16); x = max(x, n*x0); q = floor(x/x0); r = x - x0*q;
//   e = floor((r/2 - x0) * 2^(n-q))
IVIT_DEV unsigned shiftexp_int(int d, int x0, int n)
{
    int x = d + (d >> 1) - (d >> 4);
    x = max(x, n * x0);
    int ax = -x, a0 = -x0;            // both >= 0
    int q = ax / a0;                  // floor(x/x0) for x<=0, x0<0
    int r = x - x0 * q;               // in (x0, 0]
    // (r/2 - x0) * 2^(n-q) = (r - 2*x0) * 2^(n-q-1); n-q >= 0
    int t = r - 2 * x0;               // > 0
    int sh = n - q - 1;
    unsigned e = (sh >= 0) ? ((unsigned)t << sh) : ((unsigned)t >> 1);  // floor for sh = -1
    return e;
}

// The same for any sign of d, as a float32 value (ShiftGELU evaluates exp_int(-max) whose argument
// is positive for rows with a negative maximum, ivit_modules.py:115; the value may exceed 2^32 and
// overflows to +inf exactly where the reference's float32 2**(n-q) does).
IVIT_DEV float shiftexp_f32(int d, int x0, int n)
{
    int x = d + (d >> 1) - (d >> 4);
    x = max(x, n * x0);
    const int a0 = -x0;
    const int q = (x <= 0) ? ((-x) / a0) : -((x + a0 - 1) / a0);  // floor(x / x0)
    const int r = x - x0 * q;                                       // in (x0, 0]
    const int t = r - 2 * x0;                                       // 2 * (r/2 - x0) > 0
    return ldexpf((float)t, n - q - 1);
}
