// Calibration: what a plain streaming kernel reaches on this box (fill, read, copy), to put the pipeline kernels'
// bandwidth in context.  hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned __attribute__((ext_vector_type(4))) v4u;
__global__ void fill_nt(v4u* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    v4u v = {1, 1, 1, 1};
    for (; i < n; i += st) __builtin_nontemporal_store(v, p + i);
}
__global__ void fill(v4u* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    v4u v = {1, 1, 1, 1};
    for (; i < n; i += st) p[i] = v;
}
__global__ void rd(const v4u* p, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    unsigned a = 0;
    for (; i < n; i += st) { v4u v = __builtin_nontemporal_load(p + i); a += v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345) *out = a;
}
__global__ void cp(const v4u* s, v4u* d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
int main() {
    const size_t bytes = 2048ull << 20, n = bytes / 16;
    v4u *a, *b; unsigned* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768, 131072}) {
        for (int which = 0; which < 4; ++which) {
            float best = 1e9;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                if (which == 0) fill_nt<<<grid, 256>>>(a, n);
                if (which == 1) fill<<<grid, 256>>>(a, n);
                if (which == 2) rd<<<grid, 256>>>(a, n, o);
                if (which == 3) cp<<<grid, 256>>>(a, b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const char* nm[] = {"fill_nt", "fill", "read_nt", "copy_nt"};
            printf("grid %6d %-8s %.3f ms  %.0f GB/s%s\n", grid, nm[which], best, (which == 3 ? 2.0 : 1.0) * bytes / best / 1e6, which == 3 ? " (r+w)" : "");
        }
    }
    return 0;
}
