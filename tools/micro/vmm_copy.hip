// Do HIP's copies and memsets honour INTERIOR pointers of virtual-memory mappings (hipMemAddressReserve + hipMemCreate + hipMemMap)?
// The diagnostic build's guard-granule allocator (gz_api.cpp, diag_guard) hands out such pointers; round 6 saw wrong rows -- no fault --
// in guard mode only.  Every operation below is checked byte by byte through a second, independent path.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/vmm_copy tools/micro/vmm_copy.hip && tools/micro/vmm_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void read_back(const unsigned char* p, unsigned char* out, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) out[i] = p[i]; }
__global__ void fill_k(unsigned char* p, size_t n, unsigned char v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned char)(v + i); }

struct Map { void* base; size_t reserved; unsigned char* at; size_t mapped; hipMemGenericAllocationHandle_t h; };
static int make(Map& m, size_t bytes, size_t gran, const hipMemAllocationProp& prop)
{
    m.mapped = (bytes + gran - 1) / gran * gran; m.reserved = m.mapped + 2 * gran;
    CK(hipMemAddressReserve(&m.base, m.reserved, gran, nullptr, 0));
    m.at = (unsigned char*)m.base + gran;
    CK(hipMemCreate(&m.h, m.mapped, &prop, 0));
    CK(hipMemMap(m.at, m.mapped, 0, m.h, 0));
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(m.at, m.mapped, &acc, 1));
    return 0;
}

int main()
{
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu\n", gran);
    Map A, B; if (make(A, 4096, gran, prop) || make(B, 4096, gran, prop)) return 1;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned char *pin, *chk_d; CK(hipHostMalloc((void**)&pin, 1 << 16)); CK(hipMalloc((void**)&chk_d, 1 << 16));
    std::vector<unsigned char> got(1 << 16);
    int bad = 0;
    auto verify = [&](const char* what, unsigned char* dptr, size_t n, auto expect) -> int {
        hipLaunchKernelGGL(read_back, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dptr, chk_d, n);
        if (hipStreamSynchronize(s) != hipSuccess) { printf("%s: kernel read faulted\n", what); return 1; }
        if (hipMemcpy(got.data(), chk_d, n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        size_t wrong = 0; for (size_t i = 0; i < n; ++i) wrong += got[i] != expect(i);
        printf("%-70s %s (%zu of %zu bytes differ)\n", what, wrong ? "WRONG" : "ok", wrong, n);
        bad += wrong != 0;
        return 0;
    };
    for (int layout = 0; layout < 2; ++layout) {
        // layout 0: the buffer starts at the mapping's first byte; 1: it ends at its last byte
        const size_t n = 1000;
        unsigned char* a = layout ? A.at + A.mapped - 1008 : A.at;
        unsigned char* b = layout ? B.at + B.mapped - 1008 : B.at;
        printf("-- layout %d: buffers at %s of their mappings\n", layout, layout ? "the END" : "the START");
        // kernel fill, kernel read (no copy engine involved)
        hipLaunchKernelGGL(fill_k, dim3(4), dim3(256), 0, s, a, n, (unsigned char)3);
        verify("kernel write, kernel read", a, n, [](size_t i) { return (unsigned char)(3 + i); });
        for (size_t off : {(size_t)0, (size_t)16, (size_t)100}) {
            char what[128];
            for (size_t i = 0; i < n; ++i) pin[i] = (unsigned char)(17 * i + off);
            CK(hipMemsetAsync(a, 0xEE, n, s));
            CK(hipMemcpyAsync(a + off, pin, n - off, hipMemcpyHostToDevice, s));
            snprintf(what, sizeof what, "hipMemcpyAsync H2D (pinned) to buffer + %zu", off);
            verify(what, a, n, [&](size_t i) { return i < off ? (unsigned char)0xEE : (unsigned char)(17 * (i - off) + off); });
            CK(hipMemsetAsync(b, 0x11, n, s));
            CK(hipMemcpyAsync(b + off, a + off, n - off, hipMemcpyDeviceToDevice, s));
            snprintf(what, sizeof what, "hipMemcpyAsync D2D mapping -> mapping, both + %zu", off);
            verify(what, b, n, [&](size_t i) { return i < off ? (unsigned char)0x11 : (unsigned char)(17 * (i - off) + off); });
            CK(hipMemsetAsync(b + off, 0x5A, 64, s));
            snprintf(what, sizeof what, "hipMemsetAsync 64 bytes at buffer + %zu", off);
            verify(what, b, n, [&](size_t i) { return i < off ? (unsigned char)0x11 : i < off + 64 ? (unsigned char)0x5A : (unsigned char)(17 * (i - off) + off); });
            memset(pin + 4096, 0, n);
            CK(hipMemcpyAsync(pin + 4096, b + off, n - off, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            size_t wrong = 0;
            for (size_t i = off; i < n; ++i) wrong += pin[4096 + i - off] != (i < off + 64 ? (unsigned char)0x5A : (unsigned char)(17 * (i - off) + off));
            printf("hipMemcpyAsync D2H (pinned) from buffer + %-35zu %s (%zu differ)\n", off, wrong ? "WRONG" : "ok", wrong);
            bad += wrong != 0;
            // a small copy into ordinary hipMalloc memory from the mapping, and back (the tiny-text staging of the library)
            CK(hipMemsetAsync(chk_d + 8192, 0, 64, s));
            CK(hipMemcpyAsync(chk_d + 8192 + 16, a + off, 5, hipMemcpyDeviceToDevice, s));
            CK(hipStreamSynchronize(s));
            unsigned char five[64]; CK(hipMemcpy(five, chk_d + 8192, 64, hipMemcpyDeviceToHost));
            wrong = 0; for (int i = 0; i < 5; ++i) wrong += five[16 + i] != (unsigned char)(17 * i + off);
            printf("hipMemcpyAsync D2D 5 bytes mapping + %zu -> hipMalloc + 16 %26s (%zu differ)\n", off, wrong ? "WRONG" : "ok", wrong);
            bad += wrong != 0;
        }
    }
    printf("%s\n", bad ? "SOME OPERATIONS ON INTERIOR POINTERS OF MAPPINGS ARE WRONG" : "all operations on interior pointers of mappings are right");
    return 0;
}
