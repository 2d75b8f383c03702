// lds_pattern_probe.hip -- cycles per ds_read_b128 / ds_read_b64 / ds_read_b32 wave-instruction for lane->address patterns that a
// per-channel constants table produces when several lanes (rows) read the SAME entry.  One workgroup per CU, 1 or 4 waves per
// SIMD; s_memtime around a loop of 32 reads (8 in flight between waits).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_pattern_probe lds_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W>
__global__ __launch_bounds__(1024) void k(int pattern, int iters, unsigned long long* cyc, float* sink)
{
    __shared__ __attribute__((aligned(16))) float tab[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int slot;   // which W-byte entry this lane reads
    switch (pattern) {
    case 0: slot = lane; break;                         // all distinct, contiguous
    case 1: slot = lane & 15; break;                    // 4 row groups of 16 lanes read the same 16 entries
    case 2: slot = lane & 31; break;
    case 3: slot = lane & 7; break;
    case 4: slot = lane & 3; break;
    case 5: slot = lane >> 2; break;                    // quads read the same entry, 16 entries
    case 6: slot = lane >> 4; break;                    // 16 consecutive lanes read the same entry
    case 7: slot = 0; break;                            // full broadcast
    case 8: slot = ((lane & 15) + 4 * (lane >> 4)) & 15; break;   // rotated per row group: same 16 entries, different order
    default: slot = (lane & 15) * 3; break;             // stride 3 entries
    }
    const unsigned addr = (unsigned)(size_t)tab + slot * W;   // LDS byte address (low 32 bits of the generic pointer are the LDS offset)
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (W == 16) {
                float4 r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"(j * 1024));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += r[j].x;
            } else if constexpr (W == 8) {
                float2 r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"(j * 1024));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += r[j].x;
            } else {
                float r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"(j * 1024));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += r[j];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

int main()
{
    unsigned long long* cyc;
    float* sink;
    CHECK(hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&sink, 64));
    const char* names[] = {"lane (distinct)", "lane&15", "lane&31", "lane&7", "lane&3", "lane>>2", "lane>>4", "0 (broadcast)", "rotated &15", "(lane&15)*3"};
    const int iters = 100;
    printf("cycles per wave-instruction per CU (= per wave / waves per CU ... reported: per wave, and per CU)\n");
    printf("%-18s | %-26s | %-26s | %-26s\n", "pattern", "b128: 4 waves/CU | 16", "b64: 4 | 16", "b32: 4 | 16");
    for (int p = 0; p < 10; ++p) {
        printf("%-18s", names[p]);
        for (int W : {16, 8, 4}) {
            printf(" |");
            for (int wps : {1, 4}) {
                const int threads = 256 * wps, nw = 256 * 4 * wps;
                for (int rep = 0; rep < 2; ++rep) {
                    if (W == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(threads), 0, 0, p, iters, cyc, sink);
                    else if (W == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(threads), 0, 0, p, iters, cyc, sink);
                    else hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, p, iters, cyc, sink);
                    CHECK(hipDeviceSynchronize());
                }
                std::vector<unsigned long long> h(nw);
                CHECK(hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                const double per = (double)h[nw / 2] / (iters * 32.0);
                printf(" %6.2f (CU %5.2f)", per, per / (4 * wps));
            }
        }
        printf("\n");
    }
    return 0;
}
