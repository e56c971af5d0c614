// Calibration: what the LDS accesses of gz_words2_kernel's round cost the CU's LDS pipe (cycles per wave instruction with the
// CU's 32 waves all issuing the same access shape).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ldsrate.hip -o tools/ldsrate
// Shapes (lane i, "text" = consecutive words ~ 6.5 bytes apart, like a round of running text):
//   0  ds_read_u16   consecutive (the start list)
//   1  ds_read_b64   byte address text(i) / 8 (the end bitmap window; byte-misaligned)
//   2  ds_read_b128  byte address text(i) (the key: byte-misaligned)
//   3  ds_read_b128  text(i) rounded down to 16 (aligned, same neighbourhood)
//   4  2 x ds_read_b64 byte address text(i), text(i) + 8
//   5  ds_read_b128  random 16-byte entries of a 16-KB table (the hot words)
//   6  ds_read_u16   random entries of a 32-KB table (the displacements)
//   7  ds_read_b96+b32 ... not used
//   8  2 x ds_read_b128 aligned: text(i) & ~15 and + 16 (the key as two aligned halves)
//   9  ds_read_b64   random 8-byte entries of a 8-KB table
//  10  ds_read_b32   byte-misaligned text(i)
//  11  4 x ds_read_b32 aligned: text(i) & ~3 + 0, 4, 8, 12
// A "group" is what one lane-per-word round would issue for that shape (1, 2 or 4 instructions); four groups are in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
typedef unsigned __attribute__((ext_vector_type(4))) v4u;
typedef unsigned __attribute__((ext_vector_type(2))) v2u;

template <int SHAPE>
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned seed)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (unsigned i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u + seed;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned r = (threadIdx.x * 2654435761u) ^ seed;
    unsigned acc = 0;
    // the wave's own window of "text": 1 KiB + slack at wave * 2 KiB
    const unsigned base = wave * 2048u;
    for (int it = 0; it < ITERS; ++it) {
        r = r * 1664525u + 1013904223u;
        const unsigned text = base + ((lane * 13u) >> 1) + ((unsigned)it & 511u);       // ~6.5 bytes per lane, window slides
        unsigned a;
        // four independent accesses in flight per wave (offsets are multiples of 16: the alignment case stays), then one wait
#define RD4(INS, T, A, X) { T v0, v1, v2, v3; asm volatile(INS " %0, %4\n " INS " %1, %4 offset:208\n " INS " %2, %4 offset:416\n " INS " %3, %4 offset:624\n s_waitcnt lgkmcnt(0)" \
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(A)); acc ^= X; }
        if (SHAPE == 0) { a = base + 2u * lane + 2u * ((unsigned)it & 127u); RD4("ds_read_u16", unsigned, a, v0 ^ v1 ^ v2 ^ v3) }
        if (SHAPE == 1) { a = base + (text >> 3); RD4("ds_read_b64", v2u, a, v0.x ^ v1.y ^ v2.x ^ v3.y) }
        if (SHAPE == 2) { a = text; RD4("ds_read_b128", v4u, a, v0.x ^ v1.y ^ v2.z ^ v3.w) }
        if (SHAPE == 3) { a = text & ~15u; RD4("ds_read_b128", v4u, a, v0.x ^ v1.y ^ v2.z ^ v3.w) }
        if (SHAPE == 4) { a = text; RD4("ds_read_b64", v2u, a, v0.x ^ v1.y ^ v2.x ^ v3.y) a = text + 8u; RD4("ds_read_b64", v2u, a, v0.x ^ v1.y ^ v2.x ^ v3.y) }
        if (SHAPE == 5) { a = (r >> 18) & 0x3FF0u; RD4("ds_read_b128", v4u, a, v0.x ^ v1.y ^ v2.z ^ v3.w) }
        if (SHAPE == 6) { a = (r >> 17) & 0x7FFEu; RD4("ds_read_u16", unsigned, a, v0 ^ v1 ^ v2 ^ v3) }
        if (SHAPE == 8) { a = text & ~15u; RD4("ds_read_b128", v4u, a, v0.x ^ v1.y ^ v2.z ^ v3.w) a += 16u; RD4("ds_read_b128", v4u, a, v0.x ^ v1.y ^ v2.z ^ v3.w) }
        if (SHAPE == 9) { a = (r >> 19) & 0x1FF8u; RD4("ds_read_b64", v2u, a, v0.x ^ v1.y ^ v2.x ^ v3.y) }
        if (SHAPE == 10) { a = text; RD4("ds_read_b32", unsigned, a, v0 ^ v1 ^ v2 ^ v3) }
        if (SHAPE == 11) { a = text & ~3u; RD4("ds_read_b32", unsigned, a, v0 ^ v1 ^ v2 ^ v3) a += 4u; RD4("ds_read_b32", unsigned, a, v0 ^ v1 ^ v2 ^ v3)
                           a += 4u; RD4("ds_read_b32", unsigned, a, v0 ^ v1 ^ v2 ^ v3) a += 4u; RD4("ds_read_b32", unsigned, a, v0 ^ v1 ^ v2 ^ v3) }
        if (SHAPE == 7) { acc ^= r; }
    }
    if (acc == 0x1234567u) *out = acc;
}

template <int SHAPE> static void run(const char* name, unsigned* o, int cus, double mhz)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<SHAPE>, dim3(cus * 2), dim3(1024), 0, 0, o, (unsigned)it);      // two workgroups of 16 waves per CU (64 KB each)
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double cycles = best * 1e-3 * mhz * 1e6;
    printf("%-44s %.3f ms  %.1f cycles of the CU per access group of one wave (32 waves per CU, %d groups each)\n", name, best, cycles / (32.0 * ITERS * 4), ITERS * 4);
}

int main()
{
    unsigned* o; (void)hipMalloc(&o, 4);
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount; const double mhz = pr.clockRate / 1e3;
    run<7>("(loop alone: no access)", o, cus, mhz);
    run<0>("u16 consecutive (start list)", o, cus, mhz);
    run<1>("b64 at text/8 (end bits, misaligned)", o, cus, mhz);
    run<2>("b128 at text (key, misaligned)", o, cus, mhz);
    run<3>("b128 at text & ~15 (aligned)", o, cus, mhz);
    run<4>("2 x b64 at text, +8 (misaligned)", o, cus, mhz);
    run<8>("2 x b128 aligned at text & ~15, +16", o, cus, mhz);
    run<10>("b32 at text (misaligned)", o, cus, mhz);
    run<11>("4 x b32 at text & ~3 (aligned dwords)", o, cus, mhz);
    run<5>("b128 random of 16 KB (hot words)", o, cus, mhz);
    run<9>("b64 random of 8 KB", o, cus, mhz);
    run<6>("u16 random of 32 KB (displacements)", o, cus, mhz);
    return 0;
}
