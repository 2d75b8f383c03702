// mfma_shape_probe.hip -- which INT8 MFMA shape sustains more ops/s under the clock the chip holds on real data?
// (MI355X_MICROARCH.md, DVFS give-back items 6-7: for bf16 the 16x16 shape held 1.12-1.15x the rate of 32x32 at equal cycles.)
//
// Bare loops at the wave tile of gemm_i8_wreg_kernel (64 channels x 128 tokens per wave, 128 accumulator registers,
// K step 64 = 512 MFMA cycles per wave either way):
//   shape 0  v_mfma_i32_32x32x32_i8   8 accumulator tiles of 16 registers, 16 instructions of 32 cycles per K step
//   shape 1  v_mfma_i32_16x16x64_i8  32 accumulator tiles of  4 registers, 32 instructions of 16 cycles per K step
// operand source:
//   lds 0    all fragments stay in registers (the bare loop)
//   lds 1    the token fragments (8 x 16 bytes per lane and K step) are re-read from LDS every K step with ds_read_b128,
//            as the kernel does; the weight fragments stay in registers
// data: random int8 (or zeros with argv[1] = zero): the clock is data-dependent.
// Every workgroup stamps s_memtime / s_memrealtime around its loop: in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip ; run: ./mfma_shape_probe [random|zero]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE, int LDS>
__global__ __launch_bounds__(256, 2) void loop_kernel(const v4i* data, int iters, unsigned long long* stamps, int* sink)
{
    __shared__ __attribute__((aligned(16))) char smem[4 * 8 * 1024];   // 8 KB of token fragments per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    v4i A[4], B[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) A[i] = data[(blockIdx.x % 61) * 1024 + i * 256 + tid];
#pragma unroll
    for (int j = 0; j < 8; ++j) B[j] = data[65536 + (blockIdx.x % 53) * 2048 + j * 256 + tid];
    v4i* mine = reinterpret_cast<v4i*>(smem + wave * 8192);
#pragma unroll
    for (int j = 0; j < 8; ++j) mine[j * 64 + lane] = B[j];
    __syncthreads();
    v16i acc32[8];
    v4i acc16[32];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[t][r] = 0;
#pragma unroll
    for (int t = 0; t < 32; ++t) acc16[t] = (v4i){0, 0, 0, 0};
    const unsigned la = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void*)(mine + lane);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (LDS) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(B[0]) : "v"(la));
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(B[1]) : "v"(la));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(B[2]) : "v"(la));
            asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(B[3]) : "v"(la));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(B[4]) : "v"(la));
            asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(B[5]) : "v"(la));
            asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(B[6]) : "v"(la));
            asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(B[7]) : "v"(la));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]), "+v"(B[5]), "+v"(B[6]), "+v"(B[7]));
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SHAPE == 0) {
            // K half ks: weights A[2 i + ks] (32 channels x 32 k), tokens B[4 ks + j] (32 k x 32 tokens)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc32[4 * i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2 * i + ks], B[4 * ks + j], acc32[4 * i + j], 0, 0, 0);
        } else {
            // weights A[i] (16 channels x 64 k), tokens B[j] (64 k x 16 tokens)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc16[8 * i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i], B[j], acc16[8 * i + j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s ^= acc32[t][r];
    } else {
#pragma unroll
        for (int t = 0; t < 32; ++t) s ^= acc16[t][0] ^ acc16[t][1] ^ acc16[t][2] ^ acc16[t][3];
    }
    if (s == 0x1234567) sink[tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

struct Variant {
    const char* name;
    void (*fn)(const v4i*, int, unsigned long long*, int*);
    std::vector<double> us, mhz;
};

int main(int argc, char** argv)
{
    const bool zero = argc > 1 && !strcmp(argv[1], "zero");
    const int wgs_per_cu = argc > 2 ? atoi(argv[2]) : 2;
    const int grid = 256 * wgs_per_cu, iters = 4096;
    const size_t nbytes = (65536 + 53 * 2048 + 4096) * 16;
    std::vector<int8_t> h(nbytes);
    srand(1);
    for (auto& b : h) b = zero ? 0 : (int8_t)(rand() >> 7);
    v4i* d;
    unsigned long long* stamps;
    int* sink;
    CHECK(hipMalloc(&d, nbytes));
    CHECK(hipMalloc(&stamps, grid * 16));
    CHECK(hipMalloc(&sink, 1024));
    CHECK(hipMemcpy(d, h.data(), nbytes, hipMemcpyHostToDevice));
    Variant vs[4] = {{"32x32x32 registers", loop_kernel<0, 0>}, {"16x16x64 registers", loop_kernel<1, 0>},
                     {"32x32x32 lds-reads", loop_kernel<0, 1>}, {"16x16x64 lds-reads", loop_kernel<1, 1>}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // >= 2 s of back-to-back launches first: the clock under sustained load is what is compared
    for (int w = 0; w < 40; ++w)
        for (auto& v : vs) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, d, iters, stamps, sink);
    CHECK(hipDeviceSynchronize());
    const int ROUNDS = 9, REP = 8;
    std::vector<unsigned long long> hs(2 * grid);
    for (int r = 0; r < ROUNDS; ++r)
        for (int k = 0; k < 4; ++k) {
            Variant& v = vs[(r % 2) ? 3 - k : k];
            hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, d, iters, stamps, sink);
            CHECK(hipEventRecord(e0));
            for (int q = 0; q < REP; ++q) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, d, iters, stamps, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            v.us.push_back(ms * 1e3 / REP);
            CHECK(hipMemcpy(hs.data(), stamps, grid * 16, hipMemcpyDeviceToHost));
            std::vector<double> c;
            for (int b = 0; b < grid; ++b) c.push_back(100.0 * (double)hs[2 * b] / (double)hs[2 * b + 1]);
            std::sort(c.begin(), c.end());
            v.mhz.push_back(c[c.size() / 2]);
        }
    const double ops = 2.0 * 64 * 128 * 64 * (double)iters * 4 * grid;   // per launch
    printf("data=%s workgroups/CU=%d grid=%d iters=%d (K steps of 64 per wave)\n", zero ? "zero" : "random", wgs_per_cu, grid, iters);
    for (auto& v : vs) {
        std::sort(v.us.begin(), v.us.end());
        std::sort(v.mhz.begin(), v.mhz.end());
        const double med = v.us[v.us.size() / 2], mhz = v.mhz[v.mhz.size() / 2];
        // cycles per K step and wave = in-kernel cycles / iters
        printf("%-20s median %8.1f us  %7.1f TOPS  (min %8.1f us)  in-kernel clock %6.0f MHz  cycles per K step %6.1f\n", v.name, med,
               ops / med / 1e6, v.us[0], mhz, med * mhz / iters);
    }
    return 0;
}
