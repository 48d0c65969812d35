// eval_ordered_sum.h -- the reference-order (strictly sequential) sum of a row of per-pixel terms held in LDS
// (CorrelationScore's `result += score`, LogoScan.hpp:295-315), shared by the exact evaluation kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace amt {

constexpr int kSumChunk = 16;          // floats a summing lane keeps in flight

// Adds row[0..n) to acc strictly front to back.  Two register sets alternate (no copies): while one set's 16
// dependent adds retire, the other set's four ds_read_b128 are in flight.  row is 16-byte aligned and readable up to
// 2*kSumChunk floats past n.
__device__ __forceinline__ void sum_load(float4 (&v)[kSumChunk / 4], const float* p)
{
#pragma unroll
    for (int j = 0; j < kSumChunk / 4; ++j) v[j] = *reinterpret_cast<const float4*>(p + 4 * j);
}
__device__ __forceinline__ float sum_add(const float4 (&v)[kSumChunk / 4], float acc)
{
#pragma unroll
    for (int j = 0; j < kSumChunk / 4; ++j) { acc += v[j].x; acc += v[j].y; acc += v[j].z; acc += v[j].w; }
    return acc;
}
__device__ __forceinline__ float sum_add_n(const float4 (&v)[kSumChunk / 4], float acc, int rem)
{
#pragma unroll
    for (int j = 0; j < kSumChunk / 4; ++j) {
        if (4 * j + 0 < rem) acc += v[j].x;
        if (4 * j + 1 < rem) acc += v[j].y;
        if (4 * j + 2 < rem) acc += v[j].z;
        if (4 * j + 3 < rem) acc += v[j].w;
    }
    return acc;
}
__device__ __forceinline__ float ordered_row_sum(const float* row, int n, float acc)
{
    float4 A[kSumChunk / 4], B[kSumChunk / 4];
    sum_load(A, row);
    int q = 0;
    // sched_barrier: keep each set's reads ahead of the other set's adds (the scheduler otherwise sinks them and
    // the chain waits a full LDS round trip per chunk)
    for (; q + 2 * kSumChunk <= n; q += 2 * kSumChunk) {
        sum_load(B, row + q + kSumChunk);
        __builtin_amdgcn_sched_barrier(0);
        acc = sum_add(A, acc);
        __builtin_amdgcn_sched_barrier(0);
        sum_load(A, row + q + 2 * kSumChunk);
        __builtin_amdgcn_sched_barrier(0);
        acc = sum_add(B, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    sum_load(B, row + q + kSumChunk);
    acc = sum_add_n(A, acc, n - q);
    return sum_add_n(B, acc, n - q - kSumChunk);
}

} // namespace amt
