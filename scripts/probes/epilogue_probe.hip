// epilogue_probe.hip -- what does the requantising epilogue's phase 1 cost per output, and which instruction carries it?
// The body is gemm_common.h epilogue_i8_16's batch: 16 outputs = cvt, two fmas against the bracket (lo, hi), v_sad_u32 into the
// certificate, v_med3 clamp, then v_perm packing and one ds_write_b32 per four outputs; 128 outputs per lane and pass (8 batches),
// as one tile of the weights-in-registers GEMM.  Variants drop one ingredient at a time; 1 or 2 waves per SIMD; cycles per
// pass from s_memtime (median over waves).  No MFMA runs beside it: this is the epilogue ALONE on its SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o epilogue_probe epilogue_probe.hip ; run: ./epilogue_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// V bits: 1 no certificate (one fma, no sad), 2 no clamp, 4 no cvt (float accumulators), 8 no pack / LDS write (xor into a sink),
//         16 certificate by xor/or instead of sad
template <int V>
__global__ __launch_bounds__(256, 2) void probe(const int* src, const float2* lh, int passes, unsigned long long* cyc, int* sink)
{
    __shared__ unsigned cs[128 * 65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int acc[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) acc[i] = src[(i * 256 + tid) & 65535];
    float lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { lo[r] = lh[(tid + r) & 255].x; hi[r] = lh[(tid + r) & 255].y; }
    unsigned fold = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int p = 0; p < passes; ++p) {
#pragma unroll
        for (int i = 0; i < 128; ++i) asm volatile("" : "+v"(acc[i]));     // opaque: every pass recomputes every output
#pragma unroll
        for (int bt = 0; bt < 8; ++bt) {
            int b[4][4];
            unsigned unc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int z = acc[16 * bt + 4 * j + r];
                    const float a = (V & 4) ? __int_as_float(z) : (float)z;
                    const int tl = __float_as_int(__builtin_fmaf(a, lo[r], 12582912.0f));
                    if constexpr (!(V & 1)) {
                        const int th = __float_as_int(__builtin_fmaf(a, hi[r], 12582912.0f));
                        if constexpr (V & 16) unc |= (unsigned)(tl ^ th);
                        else asm("v_sad_u32 %0, %1, %2, %3" : "=v"(unc) : "v"(tl), "v"(th), "v"(unc));
                    }
                    b[j][r] = (V & 2) ? tl : min(max(tl, 0x4B400000 - 128), 0x4B400000 + 127);
                }
            if (__builtin_amdgcn_ballot_w64(unc > 1000000u) != 0) fold += 12345;     // never taken: keeps the certificate alive
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (V & 8) {
                    fold ^= (unsigned)(b[j][0] ^ b[j][1] ^ b[j][2] ^ b[j][3]);
                } else {
                    const unsigned w01 = __builtin_amdgcn_perm((unsigned)b[j][1], (unsigned)b[j][0], 0x0c0c0400u);
                    const unsigned w23 = __builtin_amdgcn_perm((unsigned)b[j][3], (unsigned)b[j][2], 0x04000c0cu);
                    cs[(16 * (bt & 1) + 4 * j + (lane & 15) + 32 * wave) * 65 + (lane >> 4) + 4 * (bt >> 1)] = w01 | w23;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    if (fold == 0x7fffffff) sink[tid] = (int)fold + (int)cs[tid];
}

template <int V>
static void run(const char* name, int wgs_per_cu, const int* src, const float2* lh, unsigned long long* cyc, int* sink)
{
    const int grid = 256 * wgs_per_cu, passes = 64;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 0, 0, src, lh, passes, cyc, sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid * 4);
    CHECK(hipMemcpy(h.data(), cyc, grid * 32, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double c = (double)h[h.size() / 2] / passes;
    printf("%-44s %d wave(s)/SIMD: %7.0f cycles per pass of 128 outputs = %5.2f cycles per output\n", name, wgs_per_cu, c, c / 128);
}

int main()
{
    int* src;
    float2* lh;
    unsigned long long* cyc;
    int* sink;
    std::vector<int> hs(65536);
    srand(3);
    for (auto& v : hs) v = (rand() % 400001) - 200000;
    std::vector<float2> hl(256);
    for (auto& v : hl) { v.x = 3.1e-4f; v.y = 3.1000002e-4f; }
    CHECK(hipMalloc(&src, 65536 * 4));
    CHECK(hipMalloc(&lh, 256 * 8));
    CHECK(hipMalloc(&cyc, 512 * 4 * 8));
    CHECK(hipMalloc(&sink, 1024));
    CHECK(hipMemcpy(src, hs.data(), 65536 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(lh, hl.data(), 256 * 8, hipMemcpyHostToDevice));
    for (int w = 1; w <= 2; ++w) {
        run<0>("full (cvt, 2 fma, sad, med3, perm, ds_write)", w, src, lh, cyc, sink);
        run<16>("certificate by xor + or instead of sad", w, src, lh, cyc, sink);
        run<1>("no certificate (one fma)", w, src, lh, cyc, sink);
        run<2>("no clamp", w, src, lh, cyc, sink);
        run<4>("no cvt", w, src, lh, cyc, sink);
        run<8>("no pack / LDS write", w, src, lh, cyc, sink);
        run<1 | 2 | 4 | 8>("one fma per output only", w, src, lh, cyc, sink);
    }
    return 0;
}
