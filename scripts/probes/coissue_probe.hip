// coissue_probe.hip -- how do an MFMA stream and a VALU stream of two different waves share one SIMD on gfx950?
// The weights-in-registers GEMM keeps two waves per SIMD (one of each co-resident workgroup): while one wave is in its
// requantising epilogue (VALU) the other is in its main loop (v_mfma_i32_16x16x64_i8 back to back).  This probe puts exactly that
// pair on every SIMD of every CU: one workgroup of 8 waves per CU, waves 0-3 play M (the kernel's MFMA loop: 32 accumulator
// tiles of 4 registers, 32 independent MFMAs per pass), waves 4-7 play V (a stream of one kind of VALU instruction, or the
// epilogue's mix cvt / fma / fma / v_sad / v_med3).  Experiments:
//   M alone, V alone                  the two streams by themselves (one wave per SIMD)
//   M timed | V spins                 M runs a fixed count while V issues VALU until M is done: MFMA cost beside VALU, and the
//                                     VALU instructions V got through in that time
//   V timed | M spins                 the other way round
//   V + V                             two VALU waves per SIMD (what both workgroups' epilogues side by side would cost)
//   M + M                             two MFMA waves per SIMD
// Times in shader cycles (s_memtime), median over the 256 x 4 waves of a role.  SIMD placement is read from HW_ID and checked.
// build: hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip ; run: ./coissue_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { R_IDLE = 0, R_FIXED = 1, R_SPIN = 2 };
enum { K_FMA = 0, K_CVT, K_MED3, K_SAD, K_PERM, K_PKFMA, K_MOV64, K_MIX, K_MIXDEP, K_N };
static const char* KN[K_N] = {"v_fma_f32", "v_cvt_f32_i32", "v_med3_i32", "v_sad_u32", "v_perm_b32", "v_pk_fma_f32", "v_mov_b64",
                              "epilogue mix (40 instr / 8 outputs)", "epilogue mix, 2 outputs interleaved"};

struct Args {
    const v4i* data;
    int m_role, v_role;      // R_*
    int v_is_mfma;           // the V waves run the MFMA body instead (M + M)
    int m_is_valu;           // the M waves run the VALU body instead (V + V)
    int v_prio;              // s_setprio of the V waves
    int m_inter;             // VALU instructions of the epilogue mix issued by the M wave ITSELF behind every MFMA (0..4)
    int m_passes, v_passes;  // fixed counts (passes of 32 MFMAs / of 320 VALU instructions)
    unsigned long long* out; // [cu][wave][4]: cycles, passes done, hw simd id, realtime ticks
    int* sink;
};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__device__ __forceinline__ void valu_pass(float (&r)[8], float (&s)[8], float a, float b, double& d0)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    // 320 instructions per pass (8 x 40)
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
        if constexpr (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
            REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_CVT) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));
            REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_MED3) {
#define X(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_SAD) {
#define X(i) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
            REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_PKFMA) {
            v2f* p = reinterpret_cast<v2f*>(r);
            v2f aa = {a, b};
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i & 3]) : "v"(aa));
            REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_MOV64) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d0) : "v"(d0));
            REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_MIX) {   // stage-wise over 8 outputs: dependent instructions are 8 apart
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(s[i]) : "v"(r[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(s[i]), "v"(a), "v"(b));
            REP8(X)
#undef X
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(a));
            REP8(X)
#undef X
#define X(i) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a) : "v"(s[i]), "v"(r[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(b));
            REP8(X)
#undef X
        } else {   // K_MIXDEP: two outputs interleaved, dependent instructions 2 apart (what a naive schedule would do)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(s[i]) : "v"(r[i]));
                X(0) X(1)
#undef X
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(s[i]), "v"(a), "v"(b));
                X(0) X(1)
#undef X
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(a));
                X(0) X(1)
#undef X
#define X(i) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a) : "v"(s[i]), "v"(r[i]));
                X(0) X(1)
#undef X
#define X(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(b));
                X(0) X(1)
#undef X
            }
        }
    }
}

template <int KIND, int INTER>
__global__ __launch_bounds__(512, 1) void probe(Args g)
{
    __shared__ volatile int flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool is_m = wave < 4;
    if (tid == 0) flag = 0;
    __syncthreads();
    const int role = is_m ? g.m_role : g.v_role;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned long long* o = g.out + ((size_t)blockIdx.x * 8 + wave) * 4;
    if (lane == 0) { o[0] = 0; o[1] = 0; o[2] = (hwid >> 4) & 3; o[3] = 0; }
    if (role == R_IDLE) return;
    const bool mfma_body = is_m ? !g.m_is_valu : (g.v_is_mfma != 0);
    const int fixed = is_m ? g.m_passes : g.v_passes;
    unsigned long long passes = 0;
    unsigned long long t0, t1, rt0, rt1;
    if (mfma_body) {
        v4i A[4], B[8], acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) A[i] = g.data[(blockIdx.x % 61) * 1024 + i * 256 + (tid & 255)];
#pragma unroll
        for (int j = 0; j < 8; ++j) B[j] = g.data[65536 + (blockIdx.x % 53) * 2048 + j * 256 + (tid & 255)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (v4i){0, 0, 0, 0};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime();
        float r[8], s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { r[i] = (float)(tid + i); s[i] = 0.f; }
        float fa = 1.0001f + tid * 1e-6f, fb = 0.5f;
        constexpr int inter = INTER;
        for (;;) {
            if constexpr (inter == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(A[i]), "v"(B[j]));
            } else {
                // the epilogue mix, stage-wise over 8 outputs (dependent instructions 8 apart), `inter` of them behind every MFMA
#define MF(i, j) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(A[i]), "v"(B[j]));
#define V0(k) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(s[k]) : "v"(r[k]));
#define V1(k) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[k]) : "v"(s[k]), "v"(fa), "v"(fb));
#define V2(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[k]) : "v"(fb), "v"(fa));
#define V3(k) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(fa) : "v"(s[k]), "v"(r[k]));
#define V4(k) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(fb), "v"(fb));
#define ROW(i, VA, VB, VC, VD) \
    MF(i, 0) VA(0) if constexpr (inter > 1) { VB(0) } if constexpr (inter > 2) { VC(0) } if constexpr (inter > 3) { VD(0) } \
    MF(i, 1) VA(1) if constexpr (inter > 1) { VB(1) } if constexpr (inter > 2) { VC(1) } if constexpr (inter > 3) { VD(1) } \
    MF(i, 2) VA(2) if constexpr (inter > 1) { VB(2) } if constexpr (inter > 2) { VC(2) } if constexpr (inter > 3) { VD(2) } \
    MF(i, 3) VA(3) if constexpr (inter > 1) { VB(3) } if constexpr (inter > 2) { VC(3) } if constexpr (inter > 3) { VD(3) } \
    MF(i, 4) VA(4) if constexpr (inter > 1) { VB(4) } if constexpr (inter > 2) { VC(4) } if constexpr (inter > 3) { VD(4) } \
    MF(i, 5) VA(5) if constexpr (inter > 1) { VB(5) } if constexpr (inter > 2) { VC(5) } if constexpr (inter > 3) { VD(5) } \
    MF(i, 6) VA(6) if constexpr (inter > 1) { VB(6) } if constexpr (inter > 2) { VC(6) } if constexpr (inter > 3) { VD(6) } \
    MF(i, 7) VA(7) if constexpr (inter > 1) { VB(7) } if constexpr (inter > 2) { VC(7) } if constexpr (inter > 3) { VD(7) }
                // (the four rows use different stage orders so that a stage's producer is a row back)
                ROW(0, V0, V3, V4, V1)
                ROW(1, V1, V0, V3, V2)
                ROW(2, V2, V1, V0, V3)
                ROW(3, V4, V2, V1, V0)
#undef ROW
            }
            ++passes;
            if (role == R_FIXED ? passes >= (unsigned long long)fixed : flag != 0) break;
        }
        t1 = __builtin_amdgcn_s_memtime(); rt1 = __builtin_amdgcn_s_memrealtime();
        int x = (int)(fa + r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x ^= acc[i][j][0] ^ acc[i][j][3];
        if (x == 0x12345677) g.sink[tid] = x;
    } else {
        float r[8], s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { r[i] = (float)(tid + i); s[i] = 0.f; }
        float a = 1.0001f + tid * 1e-6f, b = 0.5f;
        double d0 = 1.0;
        if (g.v_prio == 1) __builtin_amdgcn_s_setprio(1);
        if (g.v_prio == 2) __builtin_amdgcn_s_setprio(2);
        if (g.v_prio == 3) __builtin_amdgcn_s_setprio(3);
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            valu_pass<KIND>(r, s, a, b, d0);
            ++passes;
            if (role == R_FIXED ? passes >= (unsigned long long)fixed : flag != 0) break;
        }
        t1 = __builtin_amdgcn_s_memtime(); rt1 = __builtin_amdgcn_s_memrealtime();
        float x = a + (float)d0;
#pragma unroll
        for (int i = 0; i < 8; ++i) x += r[i] + s[i];
        if (x == 123.456f) g.sink[tid] = 1;
    }
    if (role == R_FIXED) flag = 1;      // any fixed wave finishing releases the spinning ones
    if (lane == 0) { o[0] = t1 - t0; o[1] = passes; o[3] = rt1 - rt0; }
}

struct Res { double cyc, passes, mhz; };
static unsigned long long* d_out;
static int* d_sink;
static const v4i* d_data;

template <int KIND>
static void launch(const Args& a)
{
    if (KIND == K_FMA && a.m_inter == 1) hipLaunchKernelGGL((probe<K_FMA, 1>), dim3(256), dim3(512), 0, 0, a);
    else if (KIND == K_FMA && a.m_inter == 2) hipLaunchKernelGGL((probe<K_FMA, 2>), dim3(256), dim3(512), 0, 0, a);
    else if (KIND == K_FMA && a.m_inter == 3) hipLaunchKernelGGL((probe<K_FMA, 3>), dim3(256), dim3(512), 0, 0, a);
    else if (KIND == K_FMA && a.m_inter == 4) hipLaunchKernelGGL((probe<K_FMA, 4>), dim3(256), dim3(512), 0, 0, a);
    else hipLaunchKernelGGL((probe<KIND, 0>), dim3(256), dim3(512), 0, 0, a);
}

static void run(int kind, Args a, Res& m, Res& v, bool& simd_ok)
{
    a.data = d_data; a.out = d_out; a.sink = d_sink;
    for (int rep = 0; rep < 2; ++rep) {
        switch (kind) {
            case K_FMA: launch<K_FMA>(a); break;
            case K_CVT: launch<K_CVT>(a); break;
            case K_MED3: launch<K_MED3>(a); break;
            case K_SAD: launch<K_SAD>(a); break;
            case K_PERM: launch<K_PERM>(a); break;
            case K_PKFMA: launch<K_PKFMA>(a); break;
            case K_MOV64: launch<K_MOV64>(a); break;
            case K_MIX: launch<K_MIX>(a); break;
            default: launch<K_MIXDEP>(a); break;
        }
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(256 * 8 * 4);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mc, mp, vc, vp, mh;
    simd_ok = true;
    for (int cu = 0; cu < 256; ++cu)
        for (int w = 0; w < 8; ++w) {
            const unsigned long long* o = &h[((size_t)cu * 8 + w) * 4];
            if ((int)o[2] != (w & 3)) simd_ok = false;
            if (o[0] == 0) continue;
            (w < 4 ? mc : vc).push_back((double)o[0]);
            (w < 4 ? mp : vp).push_back((double)o[1]);
            if (o[3]) mh.push_back((double)o[0] / (double)o[3] * 100.0);
        }
    auto med = [](std::vector<double>& x) { if (x.empty()) return 0.0; std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    m = {med(mc), med(mp), med(mh)};
    v = {med(vc), med(vp), med(mh)};
}

int main()
{
    std::vector<int> hd((65536 + 53 * 2048 + 4096) * 4);
    srand(1);
    for (auto& x : hd) x = rand() ^ (rand() << 16);
    v4i* dd;
    CHECK(hipMalloc(&dd, hd.size() * 4));
    CHECK(hipMemcpy(dd, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    d_data = dd;
    CHECK(hipMalloc(&d_out, 256 * 8 * 4 * 8));
    CHECK(hipMalloc(&d_sink, 4096));
    const int MP = 400, VP = 60;          // 400 x 32 MFMAs ~ 205 K cycles; 60 x 320 VALU instructions
    Res m, v;
    bool ok;
    Args a{};
    a.m_role = R_FIXED; a.v_role = R_IDLE; a.m_passes = MP;
    run(K_FMA, a, m, v, ok);
    {
        std::vector<unsigned long long> h(8 * 4);
        a.m_role = R_FIXED; a.v_role = R_FIXED; a.v_is_mfma = 1; a.v_passes = 1; a.m_passes = 1; a.data = d_data; a.out = d_out; a.sink = d_sink;
        launch<K_FMA>(a);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
        printf("SIMD id (HW_ID[5:4]) of waves 0..7 of workgroup 0:");
        for (int w = 0; w < 8; ++w) printf(" %d", (int)h[w * 4 + 2]);
        printf("\n");
        a = Args{}; a.m_role = R_FIXED; a.v_role = R_IDLE; a.m_passes = MP;
    }
    printf("SIMD of wave w == w & 3: %s\n", ok ? "yes" : "NO (see the ids above)");
    printf("M alone: %.2f cycles per MFMA (16x16x64 i8), clock %.0f MHz\n", m.cyc / (MP * 32.0), m.mhz);
    a = Args{}; a.m_role = R_FIXED; a.v_role = R_FIXED; a.v_is_mfma = 1; a.m_passes = MP; a.v_passes = MP;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    a.data = d_data; a.out = d_out; a.sink = d_sink;
    launch<K_FMA>(a);
    CHECK(hipEventRecord(e0, 0));
    launch<K_FMA>(a);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("M + M wall clock: %.3f ms for 256 x 8 x %d MFMAs -> %.0f TOPS\n", ms, MP * 32, 256.0 * 8 * MP * 32 * 32768.0 / (ms * 1e-3) / 1e12);
    run(K_FMA, a, m, v, ok);
    printf("M + M  : waves 0-3 %.2f, waves 4-7 %.2f cycles per MFMA and wave (both from the common start), clock %.0f MHz\n", m.cyc / (MP * 32.0),
           v.cyc / (MP * 32.0), m.mhz);
    printf("\n%-38s %9s %9s | %21s | %21s\n", "VALU stream of V", "V alone", "V + V", "M timed, V spins", "V timed, M spins");
    printf("%-38s %9s %9s | %10s %10s | %10s %10s\n", "", "cyc/instr", "per SIMD", "cyc/MFMA", "cyc/VALU", "cyc/VALU", "cyc/MFMA");
    for (int kind = 0; kind < K_N; ++kind) {
        a = Args{}; a.m_role = R_IDLE; a.v_role = R_FIXED; a.v_passes = VP;
        run(kind, a, m, v, ok);
        const double alone = v.cyc / (VP * 320.0);
        a = Args{}; a.m_role = R_FIXED; a.m_is_valu = 1; a.v_role = R_FIXED; a.m_passes = VP; a.v_passes = VP;
        run(kind, a, m, v, ok);
        const double vv = v.cyc / (VP * 640.0);
        a = Args{}; a.m_role = R_FIXED; a.v_role = R_SPIN; a.m_passes = MP;
        run(kind, a, m, v, ok);
        const double mt_m = m.cyc / (MP * 32.0), mt_v = v.cyc / (v.passes * 320.0);
        a = Args{}; a.m_role = R_SPIN; a.v_role = R_FIXED; a.v_passes = 4 * VP;
        run(kind, a, m, v, ok);
        const double vt_v = v.cyc / (4 * VP * 320.0), vt_m = m.cyc / (m.passes * 32.0);
        printf("%-38s %9.2f %9.2f | %10.2f %10.2f | %10.2f %10.2f\n", KN[kind], alone, vv, mt_m, mt_v, vt_v, vt_m);
    }
    printf("\nV at a raised priority (epilogue mix): M timed, V spins\n");
    for (int pr = 0; pr <= 3; ++pr) {
        a = Args{}; a.m_role = R_FIXED; a.v_role = R_SPIN; a.m_passes = MP; a.v_prio = pr;
        run(K_MIX, a, m, v, ok);
        printf("  s_setprio %d: %.2f cycles per MFMA, %.2f per VALU instruction of V\n", pr, m.cyc / (MP * 32.0), v.cyc / (v.passes * 320.0));
    }
    printf("\nONE wave per SIMD issuing both: n epilogue-mix VALU instructions behind every MFMA (software-pipelined epilogue)\n");
    for (int n = 0; n <= 4; ++n) {
        a = Args{}; a.m_role = R_FIXED; a.v_role = R_IDLE; a.m_passes = MP; a.m_inter = n;
        run(K_FMA, a, m, v, ok);
        printf("  %d VALU per MFMA: %.2f cycles per MFMA, clock %.0f MHz\n", n, m.cyc / (MP * 32.0), m.mhz);
    }
    printf("\nTWO such waves per SIMD\n");
    for (int n = 0; n <= 4; ++n) {
        a = Args{}; a.m_role = R_FIXED; a.v_role = R_FIXED; a.v_is_mfma = 1; a.m_passes = MP; a.v_passes = MP; a.m_inter = n;
        run(K_FMA, a, m, v, ok);
        printf("  %d VALU per MFMA: %.2f cycles per MFMA and SIMD (last wave's end / all MFMAs), clock %.0f MHz\n", n, v.cyc / (MP * 64.0), m.mhz);
    }
    return 0;
}
