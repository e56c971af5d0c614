// Calibration: do host-to-device and device-to-host copies on two streams run at the same time on this box (the PCIe-inclusive host
// path of the headline moves 2.87 GB in and 0.94 GB out)?    hipcc --offload-arch=gfx950 -O3 tools/micro/pcie_duplex.hip -o tools/pcie_duplex
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
int main()
{
    const size_t n = 1ull << 30;
    void *h0, *h1, *d0, *d1;
    (void)hipHostMalloc(&h0, n, hipHostMallocDefault); (void)hipHostMalloc(&h1, n, hipHostMallocDefault);
    (void)hipMalloc(&d0, n); (void)hipMalloc(&d1, n);
    hipStream_t a, b; (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    auto run = [&](int mode, size_t chunk) {
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t o = 0; o < n; o += chunk) {
            if (mode & 1) (void)hipMemcpyAsync((char*)d0 + o, (char*)h0 + o, chunk, hipMemcpyHostToDevice, a);
            if (mode & 2) (void)hipMemcpyAsync((char*)h1 + o, (char*)d1 + o, chunk, hipMemcpyDeviceToHost, b);
        }
        (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double gb = ((mode & 1) ? 1.0 : 0.0) + ((mode & 2) ? 1.0 : 0.0);
        printf("%-22s chunks of %4zu MB: %.1f ms, %.1f GB/s in total\n", mode == 1 ? "H2D alone" : mode == 2 ? "D2H alone" : "H2D and D2H together", chunk >> 20, s * 1e3, gb * 1.073741824 / s);
    };
    for (size_t chunk : {(size_t)32 << 20, (size_t)256 << 20}) { run(1, chunk); run(2, chunk); run(3, chunk); run(3, chunk); }
    return 0;
}
