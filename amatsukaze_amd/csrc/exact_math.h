// exact_math.h -- fp32 arithmetic shared by host setup code and HIP kernels.
//
// The reference's scores feed discontinuous decisions (the 8-level bin select in CorrelationScore,
// LogoScan.hpp:304, argmin over fades in CalcFade2, :1290), so the kernels reproduce the reference's
// fp32 evaluation ORDER, not just its value to a tolerance.  On any AVX x86 the order is the one of
// CalcCorrelation5x5_AVX (ComputeKernel.cpp:77-121): per column ((r0+r1)+(r2+r3))+r4, then across
// the five columns ((c0+c4)+(c2+0))+((c1+0)+(c3+0)) (hsum256_ps, :54-74, with lanes 5..7 zeroed).
// Everything here must be compiled with -ffp-contract=off.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AMT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define AMT_HD inline
#endif

namespace amt {

// horizontal sum of five column values in the reference's AVX lane order:
// (x0+x4, x1+x5, x2+x6, x3+x7) -> ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7)) with x5=x6=x7=+0.
// The "+0" adds only turn a -0 into +0; a zero's sign cannot reach a score (v-m, k*0, sum+0 are the same
// value either way and the running total starts at +0), so they are dropped.
AMT_HD float hsum5(float c0, float c1, float c2, float c3, float c4)
{
    return ((c0 + c4) + c2) + (c1 + c3);
}

// x / 25.0f, correctly rounded.  On the device: q = x*z, r = fma(-25, q, x) (exact remainder), q' = fma(r, z, q)
// with z = RN(1/25) -- verified exhaustively over all 2^32 floats to equal the IEEE quotient bit for bit
// (only the sign of -0/25 differs, see above), 3 instructions instead of the ~10 of a generic fp32 divide.
AMT_HD float div25(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float z = 0.04f;
    const float q = x * z;
    const float r = __builtin_fmaf(-25.0f, q, x);
    return __builtin_fmaf(r, z, q);
#else
    return x / 25.0f;
#endif
}

// x86 cvttss2si semantics for (int)f: out-of-range and NaN give INT_MIN
AMT_HD int trunc_x86(float f)
{
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (-2147483647 - 1);
}

// window element (row r, col c) at v[r*STRIDE + c], kernel k[row*5+col]; returns the correlation, *mean gets the
// window average.  STRIDE > 5 lets horizontally adjacent mask pixels share one wider register window.
template <int STRIDE>
AMT_HD float corr5x5_strided(const float* k, const float* v, float* mean)
{
    float c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = ((v[i] + v[STRIDE + i]) + (v[2 * STRIDE + i] + v[3 * STRIDE + i])) + v[4 * STRIDE + i];
    const float m = div25(hsum5(c[0], c[1], c[2], c[3], c[4]));
    float p[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float t0 = k[0 + i] * (v[i] - m);
        const float t1 = k[5 + i] * (v[STRIDE + i] - m);
        const float t2 = k[10 + i] * (v[2 * STRIDE + i] - m);
        const float t3 = k[15 + i] * (v[3 * STRIDE + i] - m);
        const float t4 = k[20 + i] * (v[4 * STRIDE + i] - m);
        p[i] = ((t0 + t1) + (t2 + t3)) + t4;
    }
    *mean = m;
    return hsum5(p[0], p[1], p[2], p[3], p[4]);
}
AMT_HD float corr5x5(const float* k, const float v[5][5], float* mean) { return corr5x5_strided<5>(k, &v[0][0], mean); }

// one mask pixel's contribution (LogoScan.hpp:302-308); scale/scale2 already selected by bin
AMT_HD int score_bin(float mean)
{
    int iv = trunc_x86(mean);
    iv = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
    return iv >> 3;
}
AMT_HD float score_term(float corr, float scale, float scale2)
{
    float t = corr * scale;
    float lo = (t < 1.0f) ? t : 1.0f;          // std::min(1.0f, t)
    float nm = (-1.0f < lo) ? lo : -1.0f;      // std::max(-1.0f, lo)
    return nm * scale2;
}

// unblend one pixel (LogoScan.hpp:244-249 / :1253-1257)
AMT_HD float unblend_bg(float a, float b, float maxv, float s) { return a * s + b * maxv; }
AMT_HD float fade_mix(float fade, float bg, float s) { return fade * bg + (1 - fade) * s; }

} // namespace amt
