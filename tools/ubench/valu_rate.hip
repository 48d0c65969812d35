// valu_rate.hip -- issue-rate microbenchmarks for the packed fp32 VALU forms the fused evaluation kernel uses.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && ./valu_rate
// Each kernel runs N unrolled instructions per loop trip on W waves per SIMD (grid = CUs * 4 * W waves of 64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// independent plain v_pk_mul_f32 on 8 accumulators
__global__ void k_pk_indep(float* out, int trips)
{
    f2 a0 = {1.0f, 1.0f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, m = {1.0000001f, 0.9999999f};
    for (int t = 0; t < trips; ++t) {
        REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                          "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x == 12345.0f) out[0] = s.y;
}
// the same with op_sel broadcast of the second source
__global__ void k_pk_opsel(float* out, int trips)
{
    f2 a0 = {1.0f, 1.0f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, m = {1.0000001f, 0.9999999f};
    for (int t = 0; t < trips; ++t) {
        REP8(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                          "v_pk_add_f32 %2, %2, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                          "v_pk_add_f32 %4, %4, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                          "v_pk_add_f32 %6, %6, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x == 12345.0f) out[0] = s.y;
}
// dependent chains: DIST independent accumulators round-robin (DIST=1: every instruction depends on the previous)
template <int DIST> __global__ void k_pk_dep(float* out, int trips)
{
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{1.0f, 1.0f};
    f2 m = {1.0000001f, 0.9999999f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i % DIST]) : "v"(m));
    }
    f2 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
    if (s.x == 12345.0f) out[0] = s.y;
}
template <int DIST> __global__ void k_f32_dep(float* out, int trips)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0f;
    float m = 1.0000001f;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i % DIST]) : "v"(m));
    }
    float s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
    if (s == 12345.0f) out[0] = s;
}

template <typename K> static void run(const char* name, K kern, int instr_per_trip, int waves_per_simd, int cus, float* d)
{
    const int trips = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(cus * waves_per_simd), block(256);          // 4 waves per block: one per SIMD
    hipLaunchKernelGGL(kern, grid, block, 0, 0, d, 16);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, d, trips);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)instr_per_trip * trips * waves_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    float* d;
    hipMalloc(&d, 1024);
    for (int w = 1; w <= 2; ++w) {
        run("v_pk_mul_f32 independent", k_pk_indep, 64, w, cus, d);
        run("v_pk_add_f32 op_sel/neg", k_pk_opsel, 64, w, cus, d);
        run("v_pk_mul_f32 dep dist 1", k_pk_dep<1>, 64, w, cus, d);
        run("v_pk_mul_f32 dep dist 2", k_pk_dep<2>, 64, w, cus, d);
        run("v_pk_mul_f32 dep dist 3", k_pk_dep<3>, 64, w, cus, d);
        run("v_pk_mul_f32 dep dist 4", k_pk_dep<4>, 64, w, cus, d);
        run("v_add_f32 dep dist 1", k_f32_dep<1>, 64, w, cus, d);
        run("v_add_f32 dep dist 2", k_f32_dep<2>, 64, w, cus, d);
        run("v_add_f32 dep dist 4", k_f32_dep<4>, 64, w, cus, d);
    }
    return 0;
}
