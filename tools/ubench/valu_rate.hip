// VALU issue-rate microbenchmark for gfx950: plain fp32 add / mul / fma vs packed fp32, wave64.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.0f}; }
    const float c = seed * 1.0001f;
    const f2 pc = f2{c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
                if (MODE == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int elems_per_inst, int waves_per_simd)
{
    float* d;
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 4000;   // 256 threads = 1 wave per SIMD per block
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * (threads / 64) * iters * 32.0;     // wave-instructions
    const double per_simd_per_s = insts / (ms * 1e-3) / 1024.0;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.2f G wave-inst/s/SIMD  -> cycles/inst @2.4GHz = %.2f   lane-ops %.1f T/s\n", name, waves_per_simd, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, insts * 64 * elems_per_inst / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f32", 1, w); run<1>("v_mul_f32", 1, w); run<6>("v_sub_f32", 1, w); run<2>("v_fma_f32", 1, w);
        run<3>("v_pk_add_f32", 2, w); run<4>("v_pk_mul_f32", 2, w); run<5>("v_pk_fma_f32", 2, w);
    }
    return 0;
}
