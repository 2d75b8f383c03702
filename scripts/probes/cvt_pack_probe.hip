// cvt_pack_probe.hip -- semantics of two ways to get floor(x) (0 <= x < 256) as a BYTE of a packed dword in one instruction:
//   v_cvt_pk_u8_f32 d, x, sel, old      (VOP3: converts x to u8 and packs it into byte `sel` of `old`)  -- which rounding?
//   v_cvt_u32_f32_sdwa d, x dst_sel:BYTE_n dst_unused:UNUSED_PRESERVE     (truncation; writes byte n, keeps the rest?)
// prints the results for a set of x around integers and halves, and checks both against floorf over 1 M random values.
// build: hipcc --offload-arch=gfx950 -O3 -o cvt_pack_probe cvt_pack_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k(const float* x, unsigned* pk, unsigned* sd, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    unsigned a = 0xAABBCCDDu, b = 0xAABBCCDDu;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a) : "v"(v));          // byte 1
    asm volatile("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(b) : "v"(v));   // byte 2
    pk[i] = a;
    sd[i] = b;
}

int main()
{
    const int n = 1 << 20;
    std::vector<float> h(n);
    const float special[] = {0.f, 0.25f, 0.5f, 0.75f, 0.99999994f, 1.f, 1.5f, 2.5f, 3.5f, 126.5f, 127.f, 127.5f, 127.99999f, 128.f, 200.7f, 254.5f, 255.f,
                             255.5f, 256.f, 300.f, 1e9f, -0.25f, -0.5f, -0.75f, -1.f, -3.7f};
    const int ns = sizeof(special) / sizeof(float);
    for (int i = 0; i < ns; ++i) h[i] = special[i];
    srand(1);
    for (int i = ns; i < n; ++i) h[i] = (float)rand() / (float)RAND_MAX * 255.99f;
    float* dx; unsigned *dp, *ds;
    CHECK(hipMalloc(&dx, n * 4)); CHECK(hipMalloc(&dp, n * 4)); CHECK(hipMalloc(&ds, n * 4));
    CHECK(hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dp, ds, n);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> p(n), s(n);
    CHECK(hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < ns; ++i)
        printf("x = %14.8f  v_cvt_pk_u8_f32 -> %08x (byte1 = %3u)   v_cvt_u32_f32_sdwa BYTE_2 preserve -> %08x (byte2 = %3u)\n", h[i], p[i],
               (p[i] >> 8) & 255, s[i], (s[i] >> 16) & 255);
    long bad_pk_floor = 0, bad_pk_rne = 0, bad_sd = 0, bad_keep = 0;
    for (int i = ns; i < n; ++i) {
        const unsigned fl = (unsigned)floorf(h[i]), rn = (unsigned)nearbyintf(h[i]);
        if (((p[i] >> 8) & 255) != fl) ++bad_pk_floor;
        if (((p[i] >> 8) & 255) != (rn > 255 ? 255 : rn)) ++bad_pk_rne;
        if (((s[i] >> 16) & 255) != fl) ++bad_sd;
        if ((p[i] & 0xFFFF00FFu) != 0xAABB00DDu || (s[i] & 0xFF00FFFFu) != 0xAA00CCDDu) ++bad_keep;
    }
    printf("random x in [0, 256): v_cvt_pk_u8_f32 != floor in %ld, != round-to-nearest-even in %ld; sdwa cvt != floor in %ld; other bytes disturbed in %ld (of %d)\n",
           bad_pk_floor, bad_pk_rne, bad_sd, bad_keep, n - ns);
    return 0;
}
