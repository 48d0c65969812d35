// eval_linear_kernels.hip -- many-fade logo evaluation through the linearity of the correlation (decision-guarded mode).
//
// AMTAnalyzeLogo evaluates 11 blends  W_f = f*bg + (1-f)*s  of every frame (LogoScan.hpp:1151-1155).  CalcCorrelation5x5
// (ComputeKernel.cpp:77-121) is linear in the window: in real arithmetic  mean(W_f) = f*mean(bg) + (1-f)*mean(s)  and
// corr(k, W_f) = f*corr(k, bg) + (1-f)*corr(k, s).  This kernel evaluates the window of s and of bg ONCE per mask pixel and
// forms all fades from the two (sum, mean) pairs; what is not linear -- the 8-level bin select, the scale / clamp
// (LogoScan.hpp:302-308) -- is applied per fade as the reference does.  Results differ from the reference's fp32
// evaluation order by rounding only (bounded by EvalEngine::linear_error_bound(), ~1e-6 typical), which is inside the 1e-4
// the north star allows for the float scores; the INTEGER decisions taken from them are protected separately:
//   * the bin select is discontinuous: when the interpolated mean lies within `bin_delta` of a bin edge, that fade's mean is
//     re-computed in the reference's exact order (blend, column sums, hsum, /25) and the bin comes from the exact value;
//   * argmin over fades (CalcFade2, :1288-1314): analysis_mark_kernel lists every frame whose best / second-best margin is
//     below twice the error bound, and the exact kernel (eval_fused_kernels.hip) re-evaluates just those frames.
//
// Shape: workgroup (512 threads) = (logo, G frames) walking the logo's pixel bands (<= 512 raster-consecutive mask pixels and
// the rows their windows touch); thread = ONE mask pixel.  LDS holds the band's rows as interleaved {s, bg} pairs, so a
// window element arrives as one 8-byte read with s in the low and bg in the high half: the window mean and the correlation
// of BOTH operands are computed by the same packed fp32 instructions (v_pk_*_f32, the 25 taps broadcast to both halves).
// No ordered sum: per-pixel terms go through an LDS row per fade and a fixed-order tree -- deterministic.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <algorithm>

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
    __device__ __forceinline__ int get(int k) const { return (int)((v >> (8 * k)) & 0xFFu); }
};
template <> struct Raw4<uint16_t> {
    u2 v;
    __device__ __forceinline__ void load(gptr_t base, unsigned byteoff)
    {
        typedef u2 __attribute__((aligned(2))) ua_t;
        v = *reinterpret_cast<const __attribute__((address_space(1))) ua_t*>(base + byteoff);
    }
    __device__ __forceinline__ int get(int k) const { return (int)((v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu); }
};

constexpr int kWaves = kLinThreads / 64;
constexpr int kStageRows = 2;                  // rows a wave stages per trip
constexpr int kPartPitch = kLinThreads + 4;    // one LDS row of per-pixel terms per fade

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
// {corr(k, s), corr(k, bg)} (CalcCorrelation5x5_AVX order); taps as pairs Kp[j] = {k[2j], k[2j+1]}, broadcast per use
__device__ __forceinline__ f2 window_corr(const f2 (&Kp)[13], const f2 (&W)[25], f2 M)
{
    f2 p[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        f2 t[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int e = r * 5 + i;
            const f2 kk = (e & 1) ? bc_hi(Kp[e >> 1]) : bc_lo(Kp[e >> 1]);
            t[r] = kk * (W[e] - M);
        }
        p[i] = ((t[0] + t[1]) + (t[2] + t[3])) + t[4];
    }
    return ((p[0] + p[4]) + p[2]) + (p[1] + p[3]);
}
// is v within delta of a bin edge that matters: multiples of 8 in [8, 248] ((int)avg clamped to 0..255, >> 3)
__device__ __forceinline__ bool near_bin_edge(float v, float delta)
{
    const float t = v * 0.125f;
    const float e = __builtin_rintf(t);
    return fabsf(t - e) * 8.0f < delta && e >= 1.0f && e <= 31.0f;
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

} // namespace lin

using namespace lin;

template <typename pix_t>
__global__ __launch_bounds__(kLinThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void logo_eval_linear_kernel(const EvalLogoDev* __restrict__ logos, const LinLogoDev* __restrict__ lins, const EvalBand* __restrict__ bands,
                             const float* __restrict__ fades, int nfades, int fade0, const pix_t* __restrict__ Y,
                             const int* __restrict__ frame_map, long long frame_stride, int pitch, float maxv, int nframes, int G,
                             int ngroups, float* __restrict__ out, int out_frame_stride, int take_abs, int plane_cap, float bin_delta)
{
    extern __shared__ float lds[];
    f2* const plane = reinterpret_cast<f2*>(lds);         // [plane_cap] {s, bg = a*s + b*maxv} of the band's rows, current frame
    float* const part = lds + 2 * plane_cap;              // [kLinMaxFades][kPartPitch] per-pixel terms of the current (band, frame)
    float* const accs = part + kLinMaxFades * kPartPitch; // [G][nfades] running sums

    const int logo = blockIdx.x / ngroups;
    const int grp = blockIdx.x - logo * ngroups;
    const int F0 = grp * G;
    const int gcount = min(G, nframes - F0);
    const EvalLogoDev L = logos[logo];
    const LinLogoDev X = lins[logo];
    const gptr_t gA = (gptr_t)L.a, gB = (gptr_t)L.b, gScales = (gptr_t)L.scales, gK = (gptr_t)X.kpix, gPos = (gptr_t)X.pos;
    const unsigned cpad = (unsigned)L.count_pad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = L.w, lp = L.lp;
    constexpr unsigned ES = sizeof(pix_t);

    if (tid < G * nfades) accs[tid] = 0.0f;
    const int fade_bits = __builtin_bit_cast(int, fades[fade0 + min(lane, nfades - 1)]);     // lane f holds fade f

    // tree sum of the previous iteration's per-pixel terms: wave w owns fades w, w+8; fixed order -> deterministic
    auto reduce_part = [&](int g) {
        for (int f = wave; f < nfades; f += kWaves) {
            const float4 v0 = *reinterpret_cast<const float4*>(part + f * kPartPitch + 4 * lane);
            const float4 v1 = *reinterpret_cast<const float4*>(part + f * kPartPitch + 256 + 4 * lane);
            float s = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) accs[g * nfades + f] += s;
        }
    };

    int it = 0, prev_g = 0;
    for (int bi = 0; bi < X.nbands; ++bi) {
        const EvalBand B = bands[X.band0 + bi];
        const bool act = tid < B.npix;
        const unsigned m = (unsigned)(B.m0 + (act ? tid : 0));
        const unsigned pos = gld<unsigned>(gPos, m * 4u);
        const int woff = ((int)(pos >> 16) - 2 - B.y0) * lp + (int)(pos & 0xFFFFu) - 2;   // plane offset of the window's top-left element
        f2 Kp[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) Kp[j] = gld<f2>(gK, ((unsigned)j * cpad + m) * 8u);

        for (int g = 0; g < gcount; ++g, ++it) {
            const int frame = F0 + g;
            const int srcFrame = frame_map ? frame_map[frame] : frame;
            const gptr_t src = (gptr_t)(Y + (long long)srcFrame * frame_stride + (long long)(L.imgy + L.row0) * pitch + L.imgx);
            // ---- 1. {s, bg} of the band's rows -> LDS: a wave stages 2 consecutive rows x 256 columns per trip, a lane four
            //      adjacent columns; the [1 2 1] vertical blend of DeintY (LogoScan.hpp:763-780) re-uses the 4 raw rows it loads ----
            for (int rg = wave * kStageRows; rg < B.nrows; rg += kStageRows * kWaves) {
                const int y = B.y0 + rg;
                for (int xg = 0; xg < w; xg += 256) {
                    const int x = xg + 4 * lane;
                    const int nv = min(4, w - x);
                    const int xl = nv >= 4 ? x : 0;
                    Raw4<pix_t> raw[kStageRows + 2];
                    f4 av[kStageRows], bv[kStageRows];
                    if (L.deint) {
#pragma unroll
                        for (int j = 0; j < kStageRows + 2; ++j)
                            raw[j].load(src, (unsigned)(min(max(y - 1 + j, 0), L.h - 1) * pitch + xl) * ES);
                    } else {
#pragma unroll
                        for (int j = 0; j < kStageRows; ++j)
                            raw[j + 1].load(src, (unsigned)(min(y + j, L.h - 1) * L.row_step * pitch + xl) * ES);
                        raw[0] = raw[1]; raw[kStageRows + 1] = raw[kStageRows];
                    }
#pragma unroll
                    for (int j = 0; j < kStageRows; ++j) {
                        const unsigned o = (unsigned)(min(y + j, L.h - 1) * w + xl) * 4u;
                        av[j] = gld<f4u>(gA, o);
                        bv[j] = gld<f4u>(gB, o);
                    }
                    if (nv >= 4) {
#pragma unroll
                        for (int j = 0; j < kStageRows; ++j) {
                            const int yy = y + j;
                            if (rg + j < B.nrows) {
                                const bool blend = L.deint && yy != 0 && yy != L.h - 1;
                                f4 lo, hi;           // {s0,bg0,s1,bg1} {s2,bg2,s3,bg3}
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const float s1 = blend ? (float)(raw[j].get(k) + 2 * raw[j + 1].get(k) + raw[j + 2].get(k) + 2) / 4.0f
                                                           : (float)raw[j + 1].get(k);
                                    const float g1 = unblend_bg(av[j][k], bv[j][k], maxv, s1);
                                    if (k < 2) { lo[2 * k] = s1; lo[2 * k + 1] = g1; } else { hi[2 * k - 4] = s1; hi[2 * k - 3] = g1; }
                                }
                                f4* dst = reinterpret_cast<f4*>(plane + (rg + j) * lp + x);
                                dst[0] = lo;
                                dst[1] = hi;
                            }
                        }
                    } else if (nv > 0) {
                        for (int j = 0; j < kStageRows && rg + j < B.nrows; ++j) {
                            const int yy = y + j;
                            const bool blend = L.deint && yy != 0 && yy != L.h - 1;
                            for (int k = 0; k < nv; ++k) {
                                int q0, q1, q2;
                                if (L.deint) {
                                    q0 = gld<pix_t>(src, (unsigned)(max(yy - 1, 0) * pitch + x + k) * ES);
                                    q1 = gld<pix_t>(src, (unsigned)(yy * pitch + x + k) * ES);
                                    q2 = gld<pix_t>(src, (unsigned)(min(yy + 1, L.h - 1) * pitch + x + k) * ES);
                                } else {
                                    q0 = q2 = 0;
                                    q1 = gld<pix_t>(src, (unsigned)(yy * L.row_step * pitch + x + k) * ES);
                                }
                                const float s1 = blend ? (float)(q0 + 2 * q1 + q2 + 2) / 4.0f : (float)q1;
                                plane[(rg + j) * lp + x + k] = f2{s1, unblend_bg(gld<float>(gA, (unsigned)(yy * w + x + k) * 4u),
                                                                                 gld<float>(gB, (unsigned)(yy * w + x + k) * 4u), maxv, s1)};
                            }
                        }
                    }
                }
            }
            // ---- 2. fold the previous iteration's terms into the running sums (its rows are complete since barrier B0) ----
            if (it > 0) reduce_part(prev_g);
            __syncthreads();                         // B1: plane complete, part rows free again

            // ---- 3. ONE window evaluation for both operands: R = {corr(s), corr(bg)}, M = {mean(s), mean(bg)} ----
            f2 R = {0.0f, 0.0f}, M = R;
            if (act) {
                f2 W[25];
                load_window(plane, woff, lp, W);
                M = window_means(W);
                R = window_corr(Kp, W, M);
            }
            // ---- 4. all fades from the two pairs.  First every fade's bin (a rolled loop: the exact-mean path exists once in
            //      the code); it passes through the thread's own cell of the still unused part row so that the gathers below
            //      can sit in statically indexed registers ----
            if (act) {
                for (int f = 0; f < nfades; ++f) {
                    const float fade = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fade_bits, f));
                    const float omf = 1 - fade;
                    float mf = fade * M.y + omf * M.x;
                    if (near_bin_edge(mf, bin_delta)) mf = exact_blend_mean(plane, woff, lp, fade, omf);   // the reference's own value decides
                    reinterpret_cast<int*>(part)[f * kPartPitch + tid] = score_bin_dev(mf);
                }
            }
            // ---- the scale gathers, all in flight together ----
            f2 sc[kLinMaxFades];
#pragma unroll
            for (int f = 0; f < kLinMaxFades; ++f) {
                if (f < nfades && act) {
                    const int bin = reinterpret_cast<const int*>(part)[f * kPartPitch + tid];
                    sc[f] = gld<f2>(gScales, (__umul24((unsigned)bin, cpad) + m) * 8u);
                }
            }
            // ---- the correlations and the per-pixel terms (LogoScan.hpp:305-308) ----
#pragma unroll
            for (int f = 0; f < kLinMaxFades; ++f) {
                if (f < nfades) {
                    float t = 0.0f;
                    if (act) {
                        const float fade = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fade_bits, f));
                        const float omf = 1 - fade;
                        t = score_term(fade * R.y + omf * R.x, sc[f].x, sc[f].y);
                    }
                    part[f * kPartPitch + tid] = t;
                }
            }
            prev_g = g;
            __syncthreads();                         // B0: part rows complete, plane consumed
        }
    }
    if (it > 0) reduce_part(prev_g);
    __syncthreads();
    if (tid < gcount * nfades) {
        const int g = tid / nfades, f = tid - g * nfades;
        float r = accs[tid] / L.blackScore;
        if (take_abs) r = fabsf(r);
        out[(long long)(F0 + g) * out_frame_stride + L.out_off + fade0 + f] = r;
    }
}

hipError_t launch_logo_eval_linear(hipStream_t st, int bits, const EvalLogoDev* dlogos, const LinLogoDev* dlins, int nlogos,
                                   const EvalBand* dbands, const float* dfades, int nfades, int fade0, const void* dY,
                                   const int* dframe_map, long long frame_stride_elems, int pitch, int nframes, int G, float* dout,
                                   int out_frame_stride, int take_abs, int plane_cap, float bin_delta)
{
    if (nframes <= 0 || nlogos <= 0 || nfades <= 0) return hipSuccess;
    if (nfades > kLinMaxFades || G * nfades > kLinThreads || plane_cap > kLinPlaneCap) return hipErrorInvalidValue;
    const int ngroups = (nframes + G - 1) / G;
    const float maxv = (float)((1 << bits) - 1);
    const size_t lds = ((size_t)2 * plane_cap + (size_t)kLinMaxFades * kPartPitch + (size_t)G * nfades) * sizeof(float);
    dim3 grid((unsigned)((long long)ngroups * nlogos));
    if (bits <= 8)
        hipLaunchKernelGGL(logo_eval_linear_kernel<uint8_t>, grid, dim3(kLinThreads), lds, st, dlogos, dlins, dbands, dfades, nfades, fade0,
                           (const uint8_t*)dY, dframe_map, frame_stride_elems, pitch, maxv, nframes, G, ngroups, dout, out_frame_stride,
                           take_abs, plane_cap, bin_delta);
    else
        hipLaunchKernelGGL(logo_eval_linear_kernel<uint16_t>, grid, dim3(kLinThreads), lds, st, dlogos, dlins, dbands, dfades, nfades, fade0,
                           (const uint16_t*)dY, dframe_map, frame_stride_elems, pitch, maxv, nframes, G, ngroups, dout, out_frame_stride,
                           take_abs, plane_cap, bin_delta);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// Decision guard.  A consumer of the analysis record only ever takes argmin over the 11 fades of p, t or b
// (AMTEraseLogo::CalcFade2, LogoScan.hpp:1288-1314: std::min_element = first minimum).  With every score within err[k] of
// the reference's, the argmin is the reference's whenever the smallest value beats every other by more than 2*err[k];
// frames where it does not (ties included, NaN included) are listed for exact re-evaluation.
// rec: [nframes][stride] with the groups of nfades floats at offsets k*nfades; eps_k = guard margin of group k.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void analysis_mark_kernel(const float* __restrict__ rec, int stride, int nframes, int ngroups, int nfades, float eps0, float eps1, float eps2,
                          int* __restrict__ list, int* __restrict__ count)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nframes) return;
    bool amb = false;
    for (int k = 0; k < ngroups; ++k) {
        const float eps = k == 0 ? eps0 : (k == 1 ? eps1 : eps2);
        const float* p = rec + (long long)n * stride + k * nfades;
        float lo = INFINITY, lo2 = INFINITY;
        bool bad = false;
        for (int f = 0; f < nfades; ++f) {
            const float v = p[f];
            bad |= !(v == v);
            if (v < lo) { lo2 = lo; lo = v; }
            else if (v < lo2) lo2 = v;
        }
        amb |= bad || !(lo2 - lo > eps);
    }
    if (amb) list[atomicAdd(count, 1)] = n;
}

hipError_t launch_analysis_mark(hipStream_t st, const float* drec, int stride, int nframes, int ngroups, int nfades, const float* eps3,
                                int* dlist, int* dcount)
{
    if (nframes <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(dcount, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(analysis_mark_kernel, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, st, drec, stride, nframes, ngroups, nfades,
                       eps3[0], eps3[1], eps3[2], dlist, dcount);
    return hipGetLastError();
}

} // namespace amt
