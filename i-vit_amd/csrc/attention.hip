// attention.hip -- fused integer attention core, one workgroup per (image, head).
// Replaces, for one head, the chain of /root/reference/models/vit_quant.py:72-85
//   matmul_1 (QuantMatMul, quant_modules.py:404-409) -> * scale -> qact_attn1 (fixedpoint_mul)
//   -> IVITIntSoftmax (Shiftmax, ivit_modules.py:150-179) -> matmul_2 -> qact2
// without materialising the [B,H,T,T] score tensor.
//
// MFMA formulation (v_mfma_i32_32x32x32_i8, 64-lane waves):
//   S^T tile = K_tile . Q_tile^T : keys are the "A" rows, queries the "B" columns, so a lane owns ONE
//   query (col = lane&31) and 16 of the tile's 32 keys in its registers
//   (key = 32kt + (r&3) + 8(r>>2) + 4(lane>>5)); the other 16 live in lane^32.  A whole softmax row
//   (<= 224 keys) is therefore 7x16 registers in two lanes: the row max / row sum are register
//   reductions plus ONE cross-lane exchange.
//   O^T tile = Vt . P^T : the packed int8 probabilities are used directly as the "B" operand (byte
//   4(r>>2)+(r&3) of k-step kt); V is transposed once per workgroup into LDS with its keys stored in
//   exactly that byte order, so the "A" fragment is a single ds_read_b128.
// Shiftmax's integer exponential depends only on (row max - k) in [0,255]: it is tabulated once per
// workgroup in LDS (256 x u32) with the reference's arithmetic (shiftexp_int), the row sum is an
// exact u32 sum rounded once to float32 (= the reference's float32 sum whenever that is exact).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int HD = 64;              // head dim
constexpr int NKT = 7;              // key tiles of 32 (tokens <= 224)
constexpr int KPAD = NKT * 32;      // 224
constexpr int VT_STRIDE = KPAD + 16;  // 240: ds_read_b128 of 16 lanes with distinct d hit 16 distinct slots
constexpr int K_BYTES = KPAD * HD;          // 14336
constexpr int VT_BYTES = HD * VT_STRIDE;    // 15360
constexpr int LUT_OFF = K_BYTES + VT_BYTES; // 29696
constexpr int SMEM_BYTES = LUT_OFF + 256 * 4;

struct AttnArgs {
    const int8_t* qkv;
    int8_t* out;
    int batch, heads, tokens;
    double Ms, Mo;
    int x0;  // floor(-1/s_attn)
};

IVIT_DEV int kswz(int r, int c) { return r * HD + ((c ^ ((r >> 2) & 3)) << 4); }

__global__ __launch_bounds__(NT) void attention_kernel(AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int T = a.tokens;
    const int bh = blockIdx.x;
    const int b = bh / a.heads, hh = bh - b * a.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t plane = (int64_t)a.batch * a.heads * T * HD;
    const int8_t* qg = a.qkv + (int64_t)bh * T * HD;
    const int8_t* kg = qg + plane;
    const int8_t* vg = qg + 2 * plane;

    // ---- Shiftmax exponent table: lut[i] = int_exp_shift(-i), i = kmax - k in [0,255]
    reinterpret_cast<unsigned*>(smem + LUT_OFF)[tid] = shiftexp_int(-tid, a.x0, 15);

    // ---- K tile [key][64] (swizzled 16-byte chunks); rows >= T are never consumed unmasked
    for (int q = tid; q < T * 4; q += NT) {
        int r = q >> 2, c = q & 3;
        v4i v = *reinterpret_cast<const v4i*>(kg + (int64_t)r * HD + 16 * c);
        *reinterpret_cast<v4i*>(smem + kswz(r, c)) = v;
    }
    // ---- V transposed: Vt[d][32kt + 16h' + 4g + j] = V[key = 32kt + 8g + 4h' + j][d]
    for (int q = tid; q < T * 4; q += NT) {
        int key = q >> 2, c = q & 3;
        v4i v = *reinterpret_cast<const v4i*>(vg + (int64_t)key * HD + 16 * c);
        int kap = key & 31;
        int pos = (key & ~31) + 16 * ((kap >> 2) & 1) + 4 * (kap >> 3) + (kap & 3);
        char* dst = smem + K_BYTES + (16 * c) * VT_STRIDE + pos;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) dst[(4 * w + bb) * VT_STRIDE] = (char)(v[w] >> (8 * bb));
    }
    __syncthreads();

    const unsigned* lut = reinterpret_cast<const unsigned*>(smem + LUT_OFF);
    const int nqt = (T + 31) >> 5;

    for (int qt = wave; qt < nqt; qt += 4) {
        const int qrow = qt * 32 + l31;           // this lane's query
        const int qld = min(qrow, T - 1);
        v4i qf[2];
        qf[0] = *reinterpret_cast<const v4i*>(qg + (int64_t)qld * HD + 16 * h);
        qf[1] = *reinterpret_cast<const v4i*>(qg + (int64_t)qld * HD + 32 + 16 * h);

        // ---- S^T = K . Q^T, requantised to the 8-bit Shiftmax input (qact_attn1)
        int s[NKT][16];
        int rmax = -1000;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            v16i acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                v4i kf = *reinterpret_cast<const v4i*>(smem + kswz(32 * kt + l31, 2 * ks + h));
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // |S| <= 64*128*128 = 2^20, m < 2^32: the product is exact in float64
                int ka = clamp_i32(requant_exact(acc[r], a.Ms), -128, 127);
                if (kt == NKT - 1) {
                    int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    ka = (key < T) ? ka : -1000;
                }
                s[kt][r] = ka;
                rmax = max(rmax, ka);
            }
        }
        rmax = max(rmax, __shfl_xor(rmax, 32));

        // ---- Shiftmax (ivit_modules.py:164-175): e = exp_int(k - max), sum, factor, e*factor >> 24
        unsigned esum = 0;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned e = lut[(rmax - s[kt][r]) & 255];
                if (kt == NKT - 1) e = (s[kt][r] == -1000) ? 0u : e;
                s[kt][r] = (int)e;
                esum += e;
            }
        esum += __shfl_xor(esum, 32);
        float S = (float)esum;                            // exp_int.sum (:171)
        S = fminf(S, 2147483648.0f);                      // clamp_max_(2**31-1) in float32 (:173)
        const float factor = floorf((1.0f / S) * 2147483648.0f);  // (:174)

        v4i pk[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                unsigned w = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pr = (float)(unsigned)s[kt][4 * g4 + j] * factor;  // float32 product (:175)
                    unsigned p = ((unsigned)pr) >> 24;                        // floor(. / 2^24)
                    w |= (p & 0xffu) << (8 * j);
                }
                pk[kt][g4] = (int)w;
            }

        // ---- O^T = Vt . P^T, requantised (attn.qact2), 4 consecutive d per dword
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            v16i acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                v4i vf = *reinterpret_cast<const v4i*>(smem + K_BYTES + (32 * dt + l31) * VT_STRIDE + 32 * kt + 16 * h);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(vf, pk[kt], acc, 0, 0, 0);
            }
            if (qrow < T) {
                int8_t* orow = a.out + ((int64_t)b * T + qrow) * ((int64_t)a.heads * HD) + hh * HD + 32 * dt + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    unsigned w = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // |O| <= 224*127*128 < 2^22: exact float64 product
                        int o = clamp_i32(requant_exact(acc[4 * g4 + j], a.Mo), -128, 127);
                        w |= ((unsigned)o & 0xffu) << (8 * j);
                    }
                    *reinterpret_cast<unsigned*>(orow + 8 * g4) = w;
                }
            }
        }
    }
}

}  // namespace

IVIT_EXPORT int ivit_attention_fused_i8(const int8_t* qkv, int8_t* out, int batch, int heads, int tokens,
                                        int head_dim, uint32_t m_s, int32_t e_s, float s_attn, uint32_t m_o,
                                        int32_t e_o, ivit_stream_t stream)
{
    IVIT_REQUIRE(qkv && out, "ivit_attention_fused_i8: NULL operand");
    IVIT_REQUIRE(batch > 0 && heads > 0, "ivit_attention_fused_i8: empty batch");
    if (head_dim != HD || tokens <= 32 * (NKT - 1) || tokens > KPAD) {
        ivit_set_error("ivit_attention_fused_i8: unsupported geometry head_dim=%d tokens=%d (need 64, 193..224)",
                       head_dim, tokens);
        return IVIT_ERR_UNSUPPORTED;
    }
    IVIT_REQUIRE(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 4 == 0) && ((heads * head_dim) % 4 == 0),
                 "ivit_attention_fused_i8: misaligned operand");
    IVIT_REQUIRE(s_attn > 0.0f, "ivit_attention_fused_i8: scale must be positive");
    AttnArgs a;
    a.qkv = qkv; a.out = out; a.batch = batch; a.heads = heads; a.tokens = tokens;
    a.Ms = ivit_dyadic_to_double(m_s, e_s);
    a.Mo = ivit_dyadic_to_double(m_o, e_o);
    const float x0f = __builtin_floorf((1.0f / s_attn) * -1.0f);  // ivit_modules.py:154
    IVIT_REQUIRE(x0f <= -1.0f && x0f >= -4096.0f, "ivit_attention_fused_i8: x0=%g outside [-4096,-1]", (double)x0f);
    a.x0 = (int)x0f;
    // exact u32 row sum: tokens * |x0| * 2^15 must stay below 2^32
    IVIT_REQUIRE((double)tokens * (double)(-a.x0) * 32768.0 < 4294967296.0,
                 "ivit_attention_fused_i8: Shiftmax row sum could overflow 32 bits (x0=%d)", a.x0);
    IVIT_REQUIRE(a.Ms < 2048.0 && a.Mo < 512.0, "ivit_attention_fused_i8: requant multiplier too large");
    hipLaunchKernelGGL(attention_kernel, dim3(batch * heads), dim3(NT), 0, ivit_stream(stream), a);
    IVIT_CHECK_LAUNCH("ivit_attention_fused_i8");
}
