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

// horizontal sum of five column values in the reference's AVX lane order
AMT_HD float hsum5(float c0, float c1, float c2, float c3, float c4)
{
    // (x0+x4, x1+x5, x2+x6, x3+x7) -> ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7)), x5=x6=x7=+0
    float q0 = c0 + c4, q1 = c1 + 0.0f, q2 = c2 + 0.0f, q3 = c3 + 0.0f;
    return (q0 + q2) + (q1 + q3);
}

// x86 cvttss2si semantics for (int)f: out-of-range and NaN give INT_MIN
AMT_HD int trunc_x86(float f)
{
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (-2147483647 - 1);
}

// window v[row][col], kernel k[row*5+col]; returns the correlation, *mean gets the window average
AMT_HD float corr5x5(const float* k, const float v[5][5], float* mean)
{
    float c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = ((v[0][i] + v[1][i]) + (v[2][i] + v[3][i])) + v[4][i];
    float m = hsum5(c[0], c[1], c[2], c[3], c[4]) / 25.0f;
    float p[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float t0 = k[0 + i] * (v[0][i] - m);
        float t1 = k[5 + i] * (v[1][i] - m);
        float t2 = k[10 + i] * (v[2][i] - m);
        float t3 = k[15 + i] * (v[3][i] - m);
        float t4 = k[20 + i] * (v[4][i] - m);
        p[i] = ((t0 + t1) + (t2 + t3)) + t4;
    }
    *mean = m;
    return hsum5(p[0], p[1], p[2], p[3], p[4]);
}

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
