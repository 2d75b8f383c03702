// load_path_probe.hip -- per-CU throughput of the LDS-DMA path (global_load_lds_dwordx4) that gemm_i8_pers_kernel uses
// to bring an L2-resident tile into a CU, as a function of how the 1 KB of one wave instruction is laid out in memory:
//   mode 0  1 KB contiguous (8 full 128-byte lines)                       -- a pre-tiled operand
//   mode 1  16 row segments of 64 B, row stride 768 B (16 half lines)     -- what the kernel reads today (BK = 64, K = 768)
//   mode 2   8 row segments of 128 B, row stride 768 B (8 full lines)     -- BK = 128 on a row-major operand
// Each workgroup (256 threads, 4 waves) issues "steps" of 24 KB = 6 instructions per wave, three steps in flight, with a
// counted wait and a barrier per step like the kernel's main loop, and nothing else (no MFMA, no ds_read).
// The 32 workgroups of an XCD share a window of 384 rows x 768 B (L2-resident after the first pass).
// build: hipcc --offload-arch=gfx950 -O3 -w -o load_path_probe load_path_probe.hip ; run: ./load_path_probe [wgs_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int STEP = 24 * 1024, ROWS = 384, LD = 768, WINDOW = ROWS * LD, NSTEP = 2400;

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const char* src, long long* cycles)
{
    __shared__ __attribute__((aligned(16))) char smem[3 * STEP];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const char* base = src + (size_t)(blockIdx.x & 7) * WINDOW;
    auto issue = [&](int st, int slot) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = wave + 4 * i;   // 0..23
            const char* g;
            if (MODE == 0) g = base + (st % 12) * STEP + piece * 1024 + lane * 16;
            else if (MODE == 1) g = base + (size_t)(16 * piece + (lane >> 2)) * LD + (st % 12) * 64 + (lane & 3) * 16;
            else g = base + (size_t)(16 * (piece >> 1) + 8 * (piece & 1) + (lane >> 3)) * LD + (st % 6) * 128 + (lane & 7) * 16;
            char* l = smem + slot * STEP + piece * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
        }
    };
    issue(0, 0);
    issue(1, 1);
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int st = 0; st < NSTEP; st += 3) {
        issue(st + 2, 2);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(st + 3, 0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(st + 4, 1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int wgs, const char* src, long long* cyc)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, src, cyc);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) {
        printf("%s: launch failed: %s\n", name, hipGetErrorString(err));
        fflush(stdout);
        exit(1);
    }
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, src, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(wgs);
    hipMemcpy(h.data(), cyc, wgs * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= wgs;
    printf("%-52s wgs=%4d  %7.3f ms  %7.1f shader cycles / 24 KB step / workgroup  %6.2f TB/s chip-wide\n", name, wgs, ms,
           mean / NSTEP, (double)NSTEP * STEP * wgs / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int per_cu = argc > 1 ? atoi(argv[1]) : 1;
    const int wgs = 256 * per_cu;
    char* src;
    long long* cyc;
    hipMalloc(&src, (size_t)8 * WINDOW + STEP);
    hipMemset(src, 1, (size_t)8 * WINDOW + STEP);
    hipMalloc(&cyc, wgs * sizeof(long long));
    run<0>("0: 1 KB contiguous per instruction", wgs, src, cyc);
    run<1>("1: 16 x 64 B row segments per instruction (today)", wgs, src, cyc);
    run<2>("2: 8 x 128 B row segments per instruction", wgs, src, cyc);
    return 0;
}
