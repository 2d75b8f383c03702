// valu_rate_probe.hip -- cycles per wave64 VALU instruction on gfx950, by instruction and by waves per SIMD.
// Every kernel runs a loop of 64 independent instructions of one kind (8 destination registers round-robin) and times it with
// s_memtime; one workgroup per CU of 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD.  Reported: cycles per instruction
// per WAVE and per SIMD (= per wave / waves per SIMD), median over waves.  The row kernels of this repository (LayerNorm,
// attention's Shiftmax) are VALU-bound: this is their price list.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip ; run: ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define DEF_KERNEL(NAME, ASMLINE)                                                                                     \
    __global__ __launch_bounds__(1024) void NAME(int iters, unsigned long long* cyc, float* sink)                      \
    {                                                                                                                  \
        float r0 = threadIdx.x, r1 = 1.5f, r2 = 2.5f, r3 = 3.5f, r4 = 4.5f, r5 = 5.5f, r6 = 6.5f, r7 = 7.5f;            \
        float a = 1.0001f + threadIdx.x * 1e-6f, b = 0.5f;                                                              \
        double d0 = 1.0, d1 = 2.0;                                                                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                     \
        for (int it = 0; it < iters; ++it) {                                                                            \
            BODY64(ASMLINE)                                                                                             \
        }                                                                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                     \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(d0 + d1) == 123.456f) sink[0] = r0;                         \
    }

#define A_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r##i) : "v"(a), "v"(b));
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d##i##_), "+v"(d1) : );
#define A_MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(r##i));
#define A_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r##i));
#define A_MED3(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define A_SAD(i) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(r##i) : "v"(a), "v"(b));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define A_DOT4(i) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(r##i) : "v"(a), "v"(b));
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r##i));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(a));
#define A_DPP(i) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r##i));
#define A_XOR(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(r##i) : "v"(a));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##i) : "v"(a));
#define A_FMA64(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d0) : "v"(d1));
#define A_CVTI(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r##i));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r##i));
#define A_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r##i) : "v"(a) : "s10", "s11");
#define A_CMPCND(i) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(a) : "vcc");
#define A_CMP(i) asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(r##i), "v"(a) : "vcc");
#define A_CMP64(i) asm volatile("v_cmp_gt_f32_e64 s[10:11], %1, %0" : : "v"(r##i), "v"(a) : "s10", "s11");
#define A_MAX(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_MAXI(i) asm volatile("v_max_i32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_LSHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r##i));
#define A_ADDU(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_SUBU(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_CVTIF(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r##i));
#define A_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(r##i));
#define A_FMAAK(i) asm volatile("v_fmaak_f32 %0, %1, %0, 0x4b400000" : "+v"(r##i) : "v"(a));
#define A_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define A_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r##i) : "v"(a));
#define A_READLANE(i) asm volatile("v_readlane_b32 s10, %0, 3" : : "v"(r##i) : "s10");
#define A_SUBF3(i) asm volatile("v_sub_f32_e64 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_MIN3(i) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define A_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(r##i) : "v"(a));
#define A_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define A_SDWA(i) asm volatile("v_add_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(r##i) : "v"(a));
#define A_CVTF64(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d0) : "v"(r##i));
#define A_CVTPKU8(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r##i) : "v"(a));
#define A_CVTU32(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(r##i));
#define A_CVTSDWAB(i) asm volatile("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r##i) : "v"(a));
#define A_ADDLSHL(i) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(r##i) : "v"(a));
#define A_MININT(i) asm volatile("v_min_i32 %0, %1, %0" : "+v"(r##i) : "v"(a));
#define A_CVTSDWA(i) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r##i) : "v"(a));

// v_pk_fma_f32 needs 64-bit register pairs: a separate kernel body
__global__ __launch_bounds__(1024) void k_pkfma(int iters, unsigned long long* cyc, float* sink)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f r0 = {1.f, 2.f}, r1 = {1.5f, 2.f}, r2 = {2.5f, 1.f}, r3 = {3.5f, 1.f}, r4 = {4.5f, 1.f}, r5 = {5.5f, 1.f}, r6 = {6.5f, 1.f}, r7 = {7.5f, 1.f};
    v2f a = {1.0001f + threadIdx.x * 1e-6f, 0.999f}, b = {0.5f, 0.25f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define A_PK(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r##i) : "v"(a), "v"(b));
        BODY64(A_PK)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.y == 123.456f) sink[0] = r0.x;
}
__global__ __launch_bounds__(1024) void k_pkmul(int iters, unsigned long long* cyc, float* sink)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f r0 = {1.f, 2.f}, r1 = {1.5f, 2.f}, r2 = {2.5f, 1.f}, r3 = {3.5f, 1.f}, r4 = {4.5f, 1.f}, r5 = {5.5f, 1.f}, r6 = {6.5f, 1.f}, r7 = {7.5f, 1.f};
    v2f a = {1.0001f + threadIdx.x * 1e-6f, 0.999f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define A_PKM(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r##i) : "v"(a));
        BODY64(A_PKM)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.y == 123.456f) sink[0] = r0.x;
}

DEF_KERNEL(k_fma, A_FMA)
DEF_KERNEL(k_mul, A_MUL)
DEF_KERNEL(k_add, A_ADD)
DEF_KERNEL(k_floor, A_FLOOR)
DEF_KERNEL(k_cvtub, A_CVTUB)
DEF_KERNEL(k_med3, A_MED3)
DEF_KERNEL(k_sad, A_SAD)
DEF_KERNEL(k_perm, A_PERM)
DEF_KERNEL(k_dot4, A_DOT4)
DEF_KERNEL(k_rcp, A_RCP)
DEF_KERNEL(k_cndmask, A_CNDMASK)
DEF_KERNEL(k_dpp, A_DPP)
DEF_KERNEL(k_xor, A_XOR)
DEF_KERNEL(k_lshlor, A_LSHLOR)
DEF_KERNEL(k_add3, A_ADD3)
DEF_KERNEL(k_mullo, A_MULLO)
DEF_KERNEL(k_fma64, A_FMA64)
DEF_KERNEL(k_cvti, A_CVTI)
DEF_KERNEL(k_bfe, A_BFE)
DEF_KERNEL(k_cnd64, A_CND64)
DEF_KERNEL(k_cmpcnd, A_CMPCND)
DEF_KERNEL(k_cmp, A_CMP)
DEF_KERNEL(k_cmp64, A_CMP64)
DEF_KERNEL(k_max, A_MAX)
DEF_KERNEL(k_maxi, A_MAXI)
DEF_KERNEL(k_and, A_AND)
DEF_KERNEL(k_lshl, A_LSHL)
DEF_KERNEL(k_addu, A_ADDU)
DEF_KERNEL(k_subu, A_SUBU)
DEF_KERNEL(k_cvtif, A_CVTIF)
DEF_KERNEL(k_rndne, A_RNDNE)
DEF_KERNEL(k_fmaak, A_FMAAK)
DEF_KERNEL(k_fmac, A_FMAC)
DEF_KERNEL(k_madu24, A_MADU24)
DEF_KERNEL(k_mov, A_MOV)
DEF_KERNEL(k_readlane, A_READLANE)
DEF_KERNEL(k_subf3, A_SUBF3)
DEF_KERNEL(k_min3, A_MIN3)
DEF_KERNEL(k_alignbit, A_ALIGNBIT)
DEF_KERNEL(k_or3, A_OR3)
DEF_KERNEL(k_sdwa, A_SDWA)
DEF_KERNEL(k_cvtsdwa, A_CVTSDWA)
DEF_KERNEL(k_cvtf64, A_CVTF64)
DEF_KERNEL(k_cvtpku8, A_CVTPKU8)
DEF_KERNEL(k_cvtu32, A_CVTU32)
DEF_KERNEL(k_cvtsdwab, A_CVTSDWAB)
DEF_KERNEL(k_addlshl, A_ADDLSHL)
DEF_KERNEL(k_minint, A_MININT)

typedef void (*kern_t)(int, unsigned long long*, float*);

int main()
{
    unsigned long long* cyc;
    float* sink;
    CHECK(hipMalloc(&cyc, 512 * 16 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&sink, 64));
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_floor_f32", k_floor},
        {"v_cvt_f32_ubyte1", k_cvtub}, {"v_cvt_f32_i32", k_cvti}, {"v_med3_i32", k_med3}, {"v_sad_u32", k_sad}, {"v_perm_b32", k_perm},
        {"v_dot4_i32_i8", k_dot4}, {"v_rcp_f32", k_rcp}, {"v_cndmask_b32", k_cndmask}, {"v_add_u32_dpp", k_dpp}, {"v_xor_b32", k_xor},
        {"v_lshl_or_b32", k_lshlor}, {"v_add3_u32", k_add3}, {"v_bfe_u32", k_bfe}, {"v_mul_lo_u32", k_mullo}, {"v_fma_f64", k_fma64},
        {"v_cndmask_e64 sgpr", k_cnd64}, {"v_cmp+v_cndmask vcc", k_cmpcnd}, {"v_cmp_gt_f32 vcc", k_cmp}, {"v_cmp_gt_f32 e64", k_cmp64},
        {"v_max_f32", k_max}, {"v_max_i32", k_maxi}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_add_u32", k_addu}, {"v_sub_u32", k_subu},
        {"v_cvt_i32_f32", k_cvtif}, {"v_rndne_f32", k_rndne}, {"v_fmaak_f32", k_fmaak}, {"v_fmac_f32", k_fmac}, {"v_mad_u32_u24", k_madu24},
        {"v_mov_b32", k_mov}, {"v_readlane_b32", k_readlane}, {"v_sub_f32_e64", k_subf3}, {"v_min3_i32", k_min3}, {"v_alignbit_b32", k_alignbit},
        {"v_or3_b32", k_or3}, {"v_add_f32_sdwa", k_sdwa}, {"v_cvt_f32_u32_sdwa b1", k_cvtsdwa},
        {"v_cvt_f64_i32", k_cvtf64}, {"v_cvt_pk_u8_f32", k_cvtpku8}, {"v_cvt_u32_f32", k_cvtu32}, {"v_cvt_u32_f32_sdwa B2 keep", k_cvtsdwab},
        {"v_add_lshl_u32", k_addlshl}, {"v_min_i32", k_minint},
    };
    const int iters = 200;
    printf("%-22s %28s %28s %28s %28s\n", "instruction", "1 wave/SIMD: cyc/instr/wave", "2 waves/SIMD: wave | SIMD", "4 waves/SIMD: wave | SIMD",
           "8 waves/SIMD: wave | SIMD");
    for (auto& e : ks) {
        printf("%-22s", e.name);
        for (int wps : {1, 2, 4, 8}) {      // 8: two workgroups of 1024 threads per CU
            const int threads = 256 * (wps > 4 ? 4 : wps), nw = 256 * 4 * wps, grid = wps > 4 ? 512 : 256;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(e.k, dim3(grid), dim3(threads), 0, 0, iters, cyc, sink);
                CHECK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> h(nw);
            CHECK(hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double per = (double)h[nw / 2] / (iters * 64.0);
            if (wps == 1) printf(" %27.2f", per);
            else printf(" %19.2f | %5.2f", per, per / wps);
        }
        printf("\n");
    }
    return 0;
}
