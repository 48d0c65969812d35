// int_rate.hip -- issue rates of the integer / byte VALU forms the frame metrics could be built from (gfx950).
// hipcc --offload-arch=gfx950 -O2 tools/ubench/int_rate.hip -o tools/ubench/int_rate && ./int_rate
// Eight independent accumulators per lane, 64 instructions per loop trip, W waves per SIMD on every SIMD of the device.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define KERNEL3(name, ins)                                                                                                   \
    __global__ void name(unsigned* out, int trips, unsigned x, unsigned y)                                                   \
    {                                                                                                                        \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int t = 0; t < trips; ++t) {                                                                                    \
            REP8(asm volatile(ins " %0, %8, %9, %0\n" ins " %1, %8, %9, %1\n" ins " %2, %8, %9, %2\n" ins " %3, %8, %9, %3\n" \
                              ins " %4, %8, %9, %4\n" ins " %5, %8, %9, %5\n" ins " %6, %8, %9, %6\n" ins " %7, %8, %9, %7\n" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));) \
        }                                                                                                                    \
        const unsigned s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                            \
        if (s == 0x12345u) out[0] = s;                                                                                       \
    }
#define KERNEL2(name, ins)                                                                                                   \
    __global__ void name(unsigned* out, int trips, unsigned x, unsigned y)                                                   \
    {                                                                                                                        \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int t = 0; t < trips; ++t) {                                                                                    \
            REP8(asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n"                 \
                              ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n"                 \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));) \
        }                                                                                                                    \
        const unsigned s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                            \
        if (s == 0x12345u) out[0] = s;                                                                                       \
    }
KERNEL3(k_sad_u8, "v_sad_u8")
KERNEL3(k_sad_u16, "v_sad_u16")
KERNEL3(k_sad_u32, "v_sad_u32")
KERNEL3(k_msad_u8, "v_msad_u8")
KERNEL3(k_lerp_u8, "v_lerp_u8")
KERNEL3(k_dot4_u8, "v_dot4_u32_u8")
KERNEL3(k_mad_u24, "v_mad_u32_u24")
KERNEL3(k_perm, "v_perm_b32")
KERNEL3(k_add3, "v_add3_u32")
KERNEL3(k_and_or, "v_and_or_b32")
KERNEL3(k_bfe, "v_bfe_u32")
KERNEL2(k_add_u32, "v_add_u32")
KERNEL2(k_and, "v_and_b32")
KERNEL2(k_pk_sub_u16, "v_pk_sub_u16")
KERNEL2(k_pk_max_u16, "v_pk_max_u16")
KERNEL2(k_pk_add_u16, "v_pk_add_u16")
KERNEL2(k_sub_u32, "v_sub_u32")
KERNEL2(k_lshr, "v_lshrrev_b32")

template <typename K> static void run(const char* name, K kern, int waves_per_simd, int cus, unsigned* d)
{
    const int trips = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(cus * waves_per_simd), block(256);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, d, 16, 0x01020304u, 0x05060708u);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, d, trips, 0x01020304u, 0x05060708u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double n = 64.0 * trips * waves_per_simd;
    printf("%-16s waves/SIMD=%d  %.3f ms -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, waves_per_simd, ms, ms * 1e6 / n * 2.4);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    unsigned* d;
    hipMalloc((void**)&d, 1024);
    for (int w = 1; w <= 2; ++w) {
        run("v_sad_u8", k_sad_u8, w, cus, d);
        run("v_sad_u16", k_sad_u16, w, cus, d);
        run("v_sad_u32", k_sad_u32, w, cus, d);
        run("v_msad_u8", k_msad_u8, w, cus, d);
        run("v_lerp_u8", k_lerp_u8, w, cus, d);
        run("v_dot4_u32_u8", k_dot4_u8, w, cus, d);
        run("v_mad_u32_u24", k_mad_u24, w, cus, d);
        run("v_perm_b32", k_perm, w, cus, d);
        run("v_add3_u32", k_add3, w, cus, d);
        run("v_and_or_b32", k_and_or, w, cus, d);
        run("v_bfe_u32", k_bfe, w, cus, d);
        run("v_add_u32", k_add_u32, w, cus, d);
        run("v_and_b32", k_and, w, cus, d);
        run("v_pk_sub_u16", k_pk_sub_u16, w, cus, d);
        run("v_pk_max_u16", k_pk_max_u16, w, cus, d);
        run("v_pk_add_u16", k_pk_add_u16, w, cus, d);
        run("v_sub_u32", k_sub_u32, w, cus, d);
        run("v_lshrrev_b32", k_lshr, w, cus, d);
    }
    return 0;
}
