// Issue rate of 32-bit integer multiplies on gfx950 against adds and 24-bit multiplies (the word kernel's hash uses six
// v_mul_lo_u32 per word).  build: hipcc --offload-arch=gfx950 -O3 tools/micro/mulrate.hip -o tools/mulrate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned a, unsigned b, int iters)
{
    unsigned x0 = threadIdx.x + a, x1 = x0 ^ b, x2 = x0 + 7u, x3 = x1 + 11u, x4 = x0 * 3u, x5 = x1 * 5u, x6 = x2 ^ 0x55u, x7 = x3 ^ 0x33u;
    for (int i = 0; i < iters; ++i) {
#define STEP(x) if (OP == 0) x = x * a + 0u; else if (OP == 1) x = __umul24(x, a); else if (OP == 2) x = (x + a) ^ b; else x = __umulhi(x, a);
        STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
template <int OP> static void run(const char* name, unsigned* d)
{
    const int iters = 4096, blocks = 256 * 8;                                   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B1u, 0x85EBCA6Bu, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B1u, 0x85EBCA6Bu, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 8 * (OP == 2 ? 2 : 1);   // wave-instructions
    printf("%-14s %.3f ms  %.2f wave-instructions per ns chip-wide = %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms,
           winstr / (ms * 1e6), 1024.0 * 2.4 / (winstr / (ms * 1e6)));
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<2>("add+xor", d); run<0>("mul_lo_u32", d); run<1>("mul_u32_u24", d); run<3>("mul_hi_u32", d);
    return 0;
}
