// Do kernels from two HIP streams overlap on this box?  small+small, big+small, big+big.
// hipcc --offload-arch=gfx950 -O3 tools/concurrency.hip -o tools/concurrency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(unsigned* out, int iters)
{
    unsigned x = threadIdx.x + blockIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 0x12345678u) *out = x;
}
static double run(hipStream_t a, hipStream_t b, int ga, int gb, int ia, int ib, unsigned* o, bool both)
{
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    spin<<<ga, 256, 0, a>>>(o, ia);
    if (both) spin<<<gb, 256, 0, b>>>(o, ib);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main()
{
    hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
    unsigned* o; hipMalloc(&o, 4);
    run(a, b, 64, 64, 1000, 1000, o, true);
    struct { const char* name; int ga, gb, ia, ib; } cases[] = {
        {"small(64 WG) + small(64 WG)", 64, 64, 400000, 400000},
        {"big(70000 WG) + small(64 WG)", 70000, 64, 1500, 400000},
        {"big(70000 WG) + big(70000 WG)", 70000, 70000, 1500, 1500},
        {"full(2048 WG, 1/CU-slot) + small(64 WG)", 2048, 64, 50000, 400000},
    };
    for (auto& c : cases) {
        double ta = run(a, b, c.ga, c.gb, c.ia, c.ib, o, false);
        double tb = run(b, a, c.gb, c.ga, c.ib, c.ia, o, false);
        double tab = run(a, b, c.ga, c.gb, c.ia, c.ib, o, true);
        printf("%-42s A alone %.3f ms, B alone %.3f ms, A||B %.3f ms  (sum %.3f, max %.3f)\n", c.name, ta, tb, tab, ta + tb, ta > tb ? ta : tb);
    }
    return 0;
}
