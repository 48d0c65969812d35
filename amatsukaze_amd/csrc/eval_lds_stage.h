// eval_lds_stage.h -- small device helpers shared by the one-pixel-per-thread evaluation kernels (eval_linear_kernels.hip,
// eval_pair_kernels.hip, through eval_tile_stage.h): vector types, global loads by byte offset, packed helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "eval_plan.h"
#include "exact_math.h"

namespace amt {

namespace lin {

typedef const __attribute__((address_space(1))) char* gptr_t;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 __attribute__((aligned(4))) f4u;
template <typename T> __device__ __forceinline__ T gld(gptr_t base, unsigned byteoff)
{
    return *reinterpret_cast<const __attribute__((address_space(1))) T*>(base + byteoff);
}
__device__ __forceinline__ f2 bc_lo(f2 v) { return __builtin_shufflevector(v, v, 0, 0); }
__device__ __forceinline__ f2 bc_hi(f2 v) { return __builtin_shufflevector(v, v, 1, 1); }
__device__ __forceinline__ int score_bin_dev(float mean)
{
    const int bin = (int)__builtin_amdgcn_fmed3f(mean, 0.0f, 255.0f) >> 3;     // == exact_math.h score_bin below 2^31
    return mean >= 2147483648.0f ? 0 : bin;
}
__device__ __forceinline__ f2 div25_pk(f2 x)
{
    const f2 z = {0.04f, 0.04f};
    const f2 q = x * z;
    const f2 r = __builtin_elementwise_fma(f2{-25.0f, -25.0f}, q, x);
    return __builtin_elementwise_fma(r, z, q);
}

} // namespace lin

} // namespace amt
