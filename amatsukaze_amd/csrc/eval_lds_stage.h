// eval_lds_stage.h -- helpers shared by the one-pixel-per-thread evaluation kernels (eval_linear_kernels.hip,
// eval_pair_kernels.hip): raw sample groups and their LDS-direct loads, {s, bg} pair planes, window reads.
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

template <typename pix_t> struct Raw4;
template <> struct Raw4<uint8_t> {
    unsigned v;
    __device__ __forceinline__ void load(gptr_t base, unsigned byteoff)
    {
        typedef unsigned __attribute__((aligned(1))) ua_t;
        v = *reinterpret_cast<const __attribute__((address_space(1))) ua_t*>(base + byteoff);
    }
    // buffer_load ... lds: the four samples go straight to LDS (lane l to dst[l]), no register is held while they travel
    static __device__ __forceinline__ void request_lds(__amdgpu_buffer_rsrc_t r, unsigned* dst, unsigned voff, int soff, int)
    {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 4, voff, soff, 0, 0);
    }
    __device__ __forceinline__ void from_lds(const unsigned* src, int lane, int) { v = src[lane]; }
    static constexpr int kDwordsPerLane = 1;
    __device__ __forceinline__ int get(int k) const { return (int)((v >> (8 * k)) & 0xFFu); }
};
template <> struct Raw4<uint16_t> {
    u2 v;
    __device__ __forceinline__ void load(gptr_t base, unsigned byteoff)
    {
        typedef u2 __attribute__((aligned(2))) ua_t;
        v = *reinterpret_cast<const __attribute__((address_space(1))) ua_t*>(base + byteoff);
    }
    static __device__ __forceinline__ void request_lds(__amdgpu_buffer_rsrc_t r, unsigned* dst, unsigned voff, int soff, int nl)
    {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 4, voff, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + nl), 4, voff + 4u, soff, 0, 0);
    }
    __device__ __forceinline__ void from_lds(const unsigned* src, int lane, int nl) { v = u2{src[lane], src[nl + lane]}; }
    static constexpr int kDwordsPerLane = 2;
    __device__ __forceinline__ int get(int k) const { return (int)((v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu); }
};

constexpr int kWaves = kLinThreads / 64;

// the 5x5 window of one pixel, element (r, c) at W[r*5+c] = {s, bg}
__device__ __forceinline__ void load_window(const f2* plane, int woff, int lp, f2 (&W)[25])
{
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) W[r * 5 + c] = plane[woff + r * lp + c];
}
// {mean(s), mean(bg)} in the reference's order: column sums ((r0+r1)+(r2+r3))+r4, hsum5, /25
__device__ __forceinline__ f2 window_means(const f2 (&W)[25])
{
    f2 c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = ((W[i] + W[5 + i]) + (W[10 + i] + W[15 + i])) + W[20 + i];
    return div25_pk(((c[0] + c[4]) + c[2]) + (c[1] + c[3]));
}
// {corr(k, s), corr(k, bg)} = sum_i k_i (w_i - mean) = sum_i k_i w_i - mean * sum_i k_i: 25 packed FMAs + one (this path is not
// the reference's evaluation order; its rounding is covered by EvalEngine::linear_error_bound).  Taps as pairs
// Kp[j] = {k[2j], k[2j+1]} with Kp[12].y = sum_i k_i, broadcast per use.
__device__ __forceinline__ f2 window_corr(const f2 (&Kp)[13], const f2 (&W)[25], f2 M)
{
    f2 acc0 = {0.0f, 0.0f}, acc1 = acc0;         // two chains: the FMAs of one depend on each other
#pragma unroll
    for (int e = 0; e < 25; ++e) {
        const f2 kk = (e & 1) ? bc_hi(Kp[e >> 1]) : bc_lo(Kp[e >> 1]);
        if (e & 1) acc1 = __builtin_elementwise_fma(kk, W[e], acc1);
        else acc0 = __builtin_elementwise_fma(kk, W[e], acc0);
    }
    return __builtin_elementwise_fma(-bc_hi(Kp[12]), M, acc0 + acc1);
}
// mean of the blended window exactly as EvaluateLogo + CalcCorrelation5x5_AVX produce it (LogoScan.hpp:244-251)
__device__ __forceinline__ float exact_blend_mean(const f2* plane, int woff, int lp, float fade, float omf)
{
    float c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float v[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const f2 e = plane[woff + r * lp + i];
            v[r] = fade_mix(fade, e.y, e.x);
        }
        c[i] = ((v[0] + v[1]) + (v[2] + v[3])) + v[4];
    }
    return div25(hsum5(c[0], c[1], c[2], c[3], c[4]));
}

// Staging of ONE row of an evaluation logo's band for one frame ("unit"), shared by the pair and the linear kernel.  A lane stages
// four adjacent columns (w <= 256); a ragged right edge (w % 4 == 2) is covered by shifting the last lane group left, its two
// duplicated columns are written twice with the same values.  Everything except the column offset is wave-uniform.
template <typename pix_t> struct RowStager {
    static constexpr unsigned ES = sizeof(pix_t);
    static constexpr int kD = Raw4<pix_t>::kDwordsPerLane;
    int w, h, lp, lane, nl, sx, pitchB, row_step, deint;
    bool slane;
    float maxv;
    __device__ __forceinline__ RowStager(const EvalLogoDev& L, int lane_, int pitch, float maxv_)
        : w(L.w), h(L.h), lp(L.lp), lane(lane_), nl((L.w + 3) >> 2), sx(min(4 * lane_, L.w - 4)), pitchB(pitch * (int)ES),
          row_step(L.row_step), deint(L.deint), slane(4 * lane_ < L.w), maxv(maxv_) {}

    // LDS-direct request (buffer_load ... lds: no registers held while the samples travel) of the raw source rows of logo row y --
    // y-1, y, y+1 (clamped) under DeintY's [1 2 1] blend (LogoScan.hpp:763-780), the row itself for field logos -- into the plane
    // row they will be converted into (3 * nl * sizeof(sample) * 4 bytes <= one plane row of lp pairs).  One multiply per unit,
    // the neighbours by adding the pitch.
    __device__ __forceinline__ void request(const __amdgpu_buffer_rsrc_t rS, int y, f2* prow) const
    {
        if (!slane) return;
        unsigned* dst = reinterpret_cast<unsigned*>(prow);
        if (deint) {
            const int o1 = y * pitchB;
            Raw4<pix_t>::request_lds(rS, dst, (unsigned)sx * ES, y > 0 ? o1 - pitchB : o1, nl);
            Raw4<pix_t>::request_lds(rS, dst + nl * kD, (unsigned)sx * ES, o1, nl);
            Raw4<pix_t>::request_lds(rS, dst + 2 * nl * kD, (unsigned)sx * ES, y < h - 1 ? o1 + pitchB : o1, nl);
        } else {
            Raw4<pix_t>::request_lds(rS, dst, (unsigned)sx * ES, y * row_step * pitchB, nl);
        }
    }
    // raw samples (landed in prow) -> {s, bg = a*s + b*maxv} pairs in place (LogoScan.hpp:247); bmv holds b*maxv -- the same two
    // roundings.  Byte-wise conversion; the [1 2 1] blend on floats: every intermediate is an integer below 2^24, so
    // (r0 + 2 r1 + r2 + 2) * 0.25 equals the reference's (float)(int sum) / 4.0f bit for bit.
    __device__ __forceinline__ void convert(f2* prow, int y, const f4& av, const f4& bmv) const
    {
        if (!slane) return;
        int sxl = sx;
        asm volatile("" : "+v"(sxl));      // LDS addresses hoisted out of the iteration loop would be spilled
        const unsigned* src = reinterpret_cast<const unsigned*>(prow);
        f4 sv;
        if (deint && y != 0 && y != h - 1) {
            Raw4<pix_t> r0, r1, r2;
            r0.from_lds(src, lane, nl);
            r1.from_lds(src + nl * kD, lane, nl);
            r2.from_lds(src + 2 * nl * kD, lane, nl);
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = (((float)r0.get(k) + 2.0f * (float)r1.get(k)) + ((float)r2.get(k) + 2.0f)) * 0.25f;
        } else {
            Raw4<pix_t> r1;
            r1.from_lds(src + (deint ? nl * kD : 0), lane, nl);
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = (float)r1.get(k);
        }
        f2* dst = prow + sxl;
        reinterpret_cast<f4*>(dst)[0] = f4{sv[0], av[0] * sv[0] + bmv[0], sv[1], av[1] * sv[1] + bmv[1]};
        reinterpret_cast<f4*>(dst)[1] = f4{sv[2], av[2] * sv[2] + bmv[2], sv[3], av[3] * sv[3] + bmv[3]};
    }
    // the logo coefficients of row y from memory: a and b*maxv (the product is rounded once, exactly as in a*s + b*maxv)
    __device__ __forceinline__ void load_ab(const __amdgpu_buffer_rsrc_t rA, const __amdgpu_buffer_rsrc_t rB, int y, f4& av, f4& bmv) const
    {
        const int ro = min(y, h - 1) * w * 4;
        av = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rA, (unsigned)sx * 4u, ro, 0));
        const f4 bv = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rB, (unsigned)sx * 4u, ro, 0));
        bmv = bv * maxv;
    }
    // ... kept in LDS as {a, b*maxv} pairs across the frames of a workgroup (abrow = the row's pairs)
    __device__ __forceinline__ void ab_to_lds(f2* abrow, const f4& av, const f4& bmv) const
    {
        if (!slane) return;
        int sxl = sx;
        asm volatile("" : "+v"(sxl));
        f4* d = reinterpret_cast<f4*>(abrow + sxl);
        d[0] = f4{av[0], bmv[0], av[1], bmv[1]};
        d[1] = f4{av[2], bmv[2], av[3], bmv[3]};
    }
    __device__ __forceinline__ void ab_from_lds(const f2* abrow, f4& av, f4& bmv) const
    {
        int sxl = sx;
        asm volatile("" : "+v"(sxl));
        const f4* d = reinterpret_cast<const f4*>(abrow + min(sxl, lp - 4));
        const f4 lo = d[0], hi = d[1];
        av = f4{lo[0], lo[2], hi[0], hi[2]};
        bmv = f4{lo[1], lo[3], hi[1], hi[3]};
    }
};

} // namespace lin

} // namespace amt
