// xcc_map_probe.hip -- which XCD does workgroup b of a one-dimensional grid run on?  The persistent GEMMs assume b & 7 (the dispatcher
// deals workgroups to the 8 XCDs round-robin) when they give all channel tiles of a token panel to one XCD's L2.
// Every workgroup records XCC_ID and HW_ID; a workgroup occupies half a CU as the GEMMs do (256 threads, 63 KB LDS).
// build: hipcc --offload-arch=gfx950 -O3 -o xcc_map_probe xcc_map_probe.hip ; run: ./xcc_map_probe [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin)
{
    __shared__ char blocker[63 * 1024];
    blocker[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const unsigned hw_id = __builtin_amdgcn_s_getreg((16 - 1) << 11 | (0 << 6) | 4);
    const unsigned xcc_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | (0 << 6) | 20);
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);     // stay resident until the whole grid has been dealt out
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc_id & 15u) | (hw_id << 8) | ((unsigned)blocker[3] << 30 & 0u);
}

int main(int argc, char** argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 512;
    unsigned* d;
    CHECK(hipMalloc(&d, grid * 4));
    std::vector<unsigned> h(grid);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, 200);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost));
        int agree = 0, per[16] = {0};
        for (int b = 0; b < grid; ++b) {
            agree += (int)(h[b] & 15u) == (b & 7);
            per[h[b] & 15u]++;
        }
        printf("launch %d, grid %d: XCC_ID == b & 7 for %d workgroups; per XCD:", rep, grid, agree);
        for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
        printf("\n  first 32: ");
        for (int b = 0; b < 32 && b < grid; ++b) printf("%u ", h[b] & 15u);
        printf("\n  b = 256..271: ");
        for (int b = 256; b < 272 && b < grid; ++b) printf("%u ", h[b] & 15u);
        printf("\n");
        // which workgroups share a CU?  key = (XCC_ID, HW_ID.se_id, sh_id, cu_id)
        auto cu_key = [&](int b) { const unsigned hw = h[b] >> 8; return ((h[b] & 15u) << 16) | (hw & 0xff00u); };
        int pair_half = 0, pair_next = 0;
        for (int b = 0; b + grid / 2 < grid; ++b) pair_half += cu_key(b) == cu_key(b + grid / 2);
        for (int b = 0; b + 8 < grid; b += 16) pair_next += cu_key(b) == cu_key(b + 8);
        printf("  workgroups b and b + grid/2 on the same CU: %d of %d;  b and b + 8 on the same CU: %d of %d\n", pair_half, grid / 2, pair_next, grid / 16);
    }
    return 0;
}
