// Calibration: issue rate of the vector instructions the hashes are made of (wave instructions per cycle and SIMD), to price
// the multiplies of gz_word1_h0 / gz_pair_ha / gz_ph_slot.  Eight independent chains per lane, 8 waves per SIMD, whole chip.
//   hipcc --offload-arch=gfx950 -O3 tools/valurate.hip -o tools/valurate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed)
{
    unsigned v[CHAINS];
    for (int i = 0; i < CHAINS; ++i) v[i] = threadIdx.x * 2654435761u + seed + i;
    const unsigned c = seed | 1u;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(v[i]));
            if (OP == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 6) asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long*)&v[i & ~1]) : "v"(v[i]), "v"(c) : "vcc");
            if (OP == 9) asm volatile("v_bfe_u32 %0, %0, 3, 24" : "+v"(v[i]));
            if (OP == 10) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(*(unsigned long long*)&v[i & ~1]));
            if (OP == 11) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
        }
    }
    unsigned a = 0;
    for (int i = 0; i < CHAINS; ++i) a ^= v[i];
    if (a == 0x1234567u) *out = a;
}

int main()
{
    unsigned* o; (void)hipMalloc(&o, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, grid = cus * 8;          // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    const double mhz = pr.clockRate / 1e3;
    const char* nm[] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_xor_b32", "v_mad_u32_u24", "v_alignbit_b32", "v_mul_hi_u32", "v_xad_u32",
                        "v_lshl_add_u32", "v_mad_u64_u32", "v_bfe_u32", "v_lshlrev_b64", "v_perm_b32"};
    for (int op = 0; op < 12; ++op) {
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            (void)hipEventRecord(e0);
            switch (op) {
            case 0: k<0><<<grid, 256>>>(o, it); break;   case 1: k<1><<<grid, 256>>>(o, it); break;
            case 2: k<2><<<grid, 256>>>(o, it); break;   case 3: k<3><<<grid, 256>>>(o, it); break;
            case 4: k<4><<<grid, 256>>>(o, it); break;   case 5: k<5><<<grid, 256>>>(o, it); break;
            case 6: k<6><<<grid, 256>>>(o, it); break;   case 7: k<7><<<grid, 256>>>(o, it); break;
            case 8: k<8><<<grid, 256>>>(o, it); break;   case 9: k<9><<<grid, 256>>>(o, it); break;
            case 10: k<10><<<grid, 256>>>(o, it); break; case 11: k<11><<<grid, 256>>>(o, it); break;
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // wave instructions per SIMD: 8 waves x CHAINS x ITERS; cycles = ms * clock
        const double insts = 8.0 * CHAINS * ITERS, cycles = best * 1e-3 * mhz * 1e6;
        printf("%-16s %.3f ms   %.2f cycles per wave instruction (clock %.0f MHz as reported)\n", nm[op], best, cycles / insts, mhz);
    }
    return 0;
}
