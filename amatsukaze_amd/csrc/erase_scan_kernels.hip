// erase_scan_kernels.hip -- logo erase (Delogo) and logo generation (LogoScan accumulate) kernels.
//
// Both are elementwise / reduction passes over the logo rectangle only (w*h luma + 2*wUV*hUV chroma
// samples per frame): HBM-bound, a few ops per byte.  One workgroup row = one rectangle row so that a
// wave reads/writes one contiguous run of the frame; frames and rows give >> 256 workgroups.
#include "build_knobs.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>

#include "exact_math.h"

namespace amt {

// ------------------------------------------------------------------------------------------------
// AMTEraseLogo::Delogo (LogoScan.hpp:1248-1261) applied as GetFrameT mode 0 does (:1374-1397):
// frame mode when fadeT == fadeB, otherwise per field with chroma row parity (imgy/2)%2.  In field mode
// the reference processes hUV/2 chroma rows per field, so an odd last chroma row is left untouched.
// ------------------------------------------------------------------------------------------------
struct EraseGeom {
    int w, h, wUV, hUV;
    int imgx, imgy, cx, cy;       // rectangle origin in luma / chroma planes
    int uvparity;
};

// One workgroup = kDelogoRows consecutive rectangle rows (luma rows first, then U, then V) of one frame; a
// thread owns PAIRS of horizontally adjacent samples (the rectangle origin and width are even, LogoScan.hpp:69,
// so a luma pair is one aligned 2*sizeof(pix_t) access; chroma pairs are used when the chroma origin, width and
// pitch are even too, otherwise that plane goes sample by sample), for kDelogoFrames consecutive frames: the pass is a
// read-modify-write of 48 KiB per frame against 384 KiB of logo coefficients, which are therefore loaded once per group.
#ifndef AMT_DELOGO_ROWS
#define AMT_DELOGO_ROWS 16
#endif
#ifndef AMT_DELOGO_FRAMES
#define AMT_DELOGO_FRAMES 8
#endif
constexpr int kDelogoRows = AMT_DELOGO_ROWS;
constexpr int kDelogoThreads = 256;
constexpr int kDelogoFrames = AMT_DELOGO_FRAMES;       // frames a workgroup walks through with the row's logo coefficients in registers

// four adjacent samples in one access (rows whose width is a multiple of 4: one wave moves 256 / 512 contiguous bytes per row)
template <typename pix_t> struct PixQuad;
template <> struct PixQuad<uint8_t> {
    typedef uint32_t __attribute__((aligned(2))) type;
    static __device__ __forceinline__ float get(uint32_t v, int k) { return (float)((v >> (8 * k)) & 0xFFu); }
    static __device__ __forceinline__ uint32_t pack(float r0, float r1, float r2, float r3)
    {
        return (uint32_t)(uint8_t)r0 | ((uint32_t)(uint8_t)r1 << 8) | ((uint32_t)(uint8_t)r2 << 16) | ((uint32_t)(uint8_t)r3 << 24);
    }
};
template <> struct PixQuad<uint16_t> {
    typedef uint64_t __attribute__((aligned(4))) type;
    static __device__ __forceinline__ float get(uint64_t v, int k) { return (float)((v >> (16 * k)) & 0xFFFFu); }
    static __device__ __forceinline__ uint64_t pack(float r0, float r1, float r2, float r3)
    {
        return (uint64_t)(uint16_t)r0 | ((uint64_t)(uint16_t)r1 << 16) | ((uint64_t)(uint16_t)r2 << 32) | ((uint64_t)(uint16_t)r3 << 48);
    }
};
template <typename pix_t> struct PixPair;
template <> struct PixPair<uint8_t> { typedef uint16_t type; };
template <> struct PixPair<uint16_t> { typedef uint32_t type; };

__device__ __forceinline__ float delogo_px(float s, float a, float b, float maxv, float fade)
{
    const float bg = unblend_bg(a, b, maxv, s);
    const float t = fade_mix(fade, bg, s) + 0.5f;
    const float lo = (t < 0.0f) ? 0.0f : t;            // std::max(t, 0.0f)
    return (maxv < lo) ? maxv : lo;                    // std::min(lo, maxv)
}

template <typename pix_t>
__global__ __launch_bounds__(kDelogoThreads)
void delogo_kernel(const pix_t* sY, const pix_t* sU, const pix_t* sV, pix_t* Y, pix_t* U, pix_t* V, long long strideY,
                   long long strideUV, int pitchY, int pitchUV, const float* __restrict__ planes, EraseGeom g,
                   float maxv, const float2* __restrict__ fades, int pairY, int pairUV, int nframes, int quadY, int quadUV,
                   int zero_identity)
{
    typedef typename PixPair<pix_t>::type pair_t;
    typedef typename PixQuad<pix_t>::type quad_t;
    constexpr int SH = 8 * sizeof(pix_t);
    const int f0 = blockIdx.y * kDelogoFrames;
    const int f1 = min(nframes, f0 + kDelogoFrames);
    const size_t ysz = (size_t)g.w * g.h, csz = (size_t)g.wUV * g.hUV;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (zero_identity) {
        // a group whose frames all carry fade {0, 0} has nothing to rewrite (half of a clip whose logo comes and goes): leave before the
        // coefficient rows are fetched
        bool any = false;
        for (int f = f0; f < f1; ++f) { const float2 fd = fades[f]; any = any || fd.x != 0.0f || fd.y != 0.0f; }
        if (!any) return;
    }
    const int rend = min(g.h + 2 * g.hUV, (int)(blockIdx.x + 1) * kDelogoRows);
    for (int r = blockIdx.x * kDelogoRows + wv; r < rend; r += kDelogoThreads / 64) {
        // the row's geometry and logo coefficients are the same for every frame of the group: the logo planes are
        // 8 B per sample against 2 B of frame traffic, so they are read once and kept in registers
        pix_t* row0;             // the row in the destination batch
        const pix_t* srow0;      // ... in the source batch (the same pointer for the in-place call)
        const float *A, *B;
        int roww, paired, quad, y, pl;
        long long stride;
        if (r < g.h) {
            y = r; pl = 0;
            roww = g.w; paired = pairY; quad = quadY; stride = strideY;
            row0 = Y + (long long)(g.imgy + y) * pitchY + g.imgx;
            srow0 = sY + (long long)(g.imgy + y) * pitchY + g.imgx;
            A = planes + (size_t)y * g.w; B = planes + ysz + (size_t)y * g.w;
        } else {
            pl = (r - g.h) >= g.hUV ? 2 : 1;
            y = r - g.h - (pl - 1) * g.hUV;
            roww = g.wUV; paired = pairUV; quad = quadUV; stride = strideUV;
            row0 = (pl == 2 ? V : U) + (long long)(g.cy + y) * pitchUV + g.cx;
            srow0 = (pl == 2 ? sV : sU) + (long long)(g.cy + y) * pitchUV + g.cx;
            const float* base = planes + 2 * ysz + (size_t)(pl - 1) * 2 * csz;
            A = base + (size_t)y * g.wUV; B = base + csz + (size_t)y * g.wUV;
        }
        auto fade_of = [&](float2 fd, bool& skip) {
            const bool frameMode = fd.x == fd.y;
            skip = pl != 0 && !frameMode && y >= 2 * (g.hUV / 2);   // field mode leaves an odd last chroma row alone
            // fade 0 writes back what it read (0*bg + 1*s + 0.5 truncates to s) whenever bg is finite, which the host has checked
            // for this logo (zero_identity): frames without a logo cost no traffic
            skip = skip || (zero_identity && fd.x == 0.0f && fd.y == 0.0f);
            if (frameMode) return fd.x;
            return pl == 0 ? ((y & 1) ? fd.y : fd.x) : (((y & 1) == g.uvparity) ? fd.x : fd.y);
        };
        if (quad) {
            for (int x = 4 * lane; x < roww; x += 256) {
                const float4 a = *reinterpret_cast<const float4*>(A + x);
                const float4 b = *reinterpret_cast<const float4*>(B + x);
                float fd[kDelogoFrames];
                bool live[kDelogoFrames];
#pragma unroll
                for (int k = 0; k < kDelogoFrames; ++k) {
                    bool skip = true;
                    fd[k] = f0 + k < f1 ? fade_of(fades[f0 + k], skip) : 0.0f;
                    live[k] = !skip;
                }
                typename PixQuad<pix_t>::type v[kDelogoFrames];          // all live frames' loads in flight before the first use
#pragma unroll
                for (int k = 0; k < kDelogoFrames; ++k)
                    if (live[k]) v[k] = *reinterpret_cast<const quad_t*>(srow0 + (long long)(f0 + k) * stride + x);
#pragma unroll
                for (int k = 0; k < kDelogoFrames; ++k) {
                    if (!live[k]) continue;
                    typedef PixQuad<pix_t> Q;
                    const float r0 = delogo_px(Q::get(v[k], 0), a.x, b.x, maxv, fd[k]);
                    const float r1 = delogo_px(Q::get(v[k], 1), a.y, b.y, maxv, fd[k]);
                    const float r2 = delogo_px(Q::get(v[k], 2), a.z, b.z, maxv, fd[k]);
                    const float r3 = delogo_px(Q::get(v[k], 3), a.w, b.w, maxv, fd[k]);
                    *reinterpret_cast<quad_t*>(row0 + (long long)(f0 + k) * stride + x) = Q::pack(r0, r1, r2, r3);
                }
            }
        } else if (paired) {
            for (int x = 2 * lane; x < roww; x += 128) {
                const float2 a = *reinterpret_cast<const float2*>(A + x);
                const float2 b = *reinterpret_cast<const float2*>(B + x);
                pair_t v[kDelogoFrames];                                 // all frames' loads in flight before the first use
#pragma unroll
                for (int k = 0; k < kDelogoFrames; ++k)
                    v[k] = *reinterpret_cast<const pair_t*>(srow0 + (long long)min(f0 + k, f1 - 1) * stride + x);
#pragma unroll
                for (int k = 0; k < kDelogoFrames; ++k) {
                    const int f = f0 + k;
                    if (f >= f1) break;
                    bool skip;
                    const float fade = fade_of(fades[f], skip);
                    if (skip) continue;
                    const float r0 = delogo_px((float)(pix_t)v[k], a.x, b.x, maxv, fade);
                    const float r1 = delogo_px((float)(pix_t)(v[k] >> SH), a.y, b.y, maxv, fade);
                    *reinterpret_cast<pair_t*>(row0 + (long long)f * stride + x) = (pair_t)((pair_t)(pix_t)r0 | ((pair_t)(pix_t)r1 << SH));
                }
            }
        } else {
            for (int x = lane; x < roww; x += 64) {
                const float a = A[x], b = B[x];
                for (int f = f0; f < f1; ++f) {
                    bool skip;
                    const float fade = fade_of(fades[f], skip);
                    if (skip) continue;
                    row0[(long long)f * stride + x] = (pix_t)delogo_px((float)srow0[(long long)f * stride + x], a, b, maxv, fade);
                }
            }
        }
    }
}

hipError_t launch_delogo(hipStream_t st, int bits, const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV, long long strideY,
                         long long strideUV, int pitchY, int pitchUV, const float* dplanes, EraseGeom g, int nframes, const float2* dfades,
                         int zero_identity)
{
    if (nframes <= 0) return hipSuccess;
    const int rows = g.h + 2 * g.hUV;
    dim3 grid((unsigned)((rows + kDelogoRows - 1) / kDelogoRows), (unsigned)((nframes + kDelogoFrames - 1) / kDelogoFrames)), block(kDelogoThreads);
    const float maxv = (float)((1 << bits) - 1);
    const int es = bits <= 8 ? 1 : 2;
    auto even = [](long long v) { return (v & 1) == 0; };
    // a pair access needs 2*es alignment of every row start and an even width (plane bases come from hipMalloc /
    // AviSynth's 64-byte aligned planes; a caller handing odd byte offsets falls back to single samples)
    const int pairY = even(g.w) && even(g.imgx) && even(pitchY) && even(strideY) && ((uintptr_t)dY % (2 * es) == 0) && ((uintptr_t)sY % (2 * es) == 0);
    const int pairUV = even(g.wUV) && even(g.cx) && even(pitchUV) && even(strideUV) && ((uintptr_t)dU % (2 * es) == 0) &&
                       ((uintptr_t)dV % (2 * es) == 0) && ((uintptr_t)sU % (2 * es) == 0) && ((uintptr_t)sV % (2 * es) == 0);
    // four samples per lane where a row is a multiple of 4 wide (coefficient rows are then 16-byte aligned float4s)
    const int quadY = pairY && g.w % 4 == 0;
    const int quadUV = pairUV && g.wUV % 4 == 0;
    if (bits <= 8)
        hipLaunchKernelGGL(delogo_kernel<uint8_t>, grid, block, 0, st, (const uint8_t*)sY, (const uint8_t*)sU, (const uint8_t*)sV, (uint8_t*)dY, (uint8_t*)dU, (uint8_t*)dV, strideY,
                           strideUV, pitchY, pitchUV, dplanes, g, maxv, dfades, pairY, pairUV, nframes, quadY, quadUV, zero_identity);
    else
        hipLaunchKernelGGL(delogo_kernel<uint16_t>, grid, block, 0, st, (const uint16_t*)sY, (const uint16_t*)sU, (const uint16_t*)sV, (uint16_t*)dY, (uint16_t*)dU, (uint16_t*)dV, strideY,
                           strideUV, pitchY, pitchUV, dplanes, g, maxv, dfades, pairY, pairUV, nframes, quadY, quadUV, zero_identity);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// AMTEraseLogo::CalcFade / CalcFade2 (LogoScan.hpp:1263-1341) for a whole batch: one thread per frame.  The decision reads
// nine analysis records around frame n (the argmin over the 11 fades of p for each; t and b of the centre one) -- 99 + 22 floats
// of a table that is L2 resident -- so this is a latency-sized kernel whose point is that the fades never leave the device:
// analysis -> fades -> Delogo is one stream-ordered chain without a host synchronise in the middle.
// Index arithmetic as decisions.cpp fade_from_analysis: k = clamp(n + i) + i, analyze frame q = clamp(k >> 3), source frame
// clamp(8 q + (k & 7)) -- the AviSynth cache's clamping of the analyze clip's frame number, negative k included.
// ------------------------------------------------------------------------------------------------
// i / 10.0f for i = 0..10 and the two threshold tests of (sum of four argmins) / 40 against 0.3 and 0.7 for every possible sum 0..40,
// all evaluated on the host exactly as the reference writes them (float division, comparison against double literals): the
// kernel itself does no floating-point arithmetic, so nothing depends on the device's division
struct FadeTable { float v[11]; unsigned long long below03, above07; };

__device__ __forceinline__ int argmin11_dev(const float* __restrict__ v)
{
    int best = 0;
    float bv = v[0];
#pragma unroll
    for (int i = 1; i < 11; ++i) {
        const float x = v[i];
        if (x < bv) { bv = x; best = i; }       // strict <: the first minimum wins, NaN never does (as the host loop)
    }
    return best;
}

__global__ __launch_bounds__(128)
void calc_fades_kernel(const float* __restrict__ analysis, int analysis_first, int analysis_count, int num_frames, int first, int nframes,
                       const uint8_t* __restrict__ frame_state /* null = no logoframe file */, int half, FadeTable tab,
                       float2* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nframes) return;
    const int n = first + i;
    const int last = num_frames - 1;
    auto clampf = [&](int v) { return max(0, min(last, v)); };
    if (frame_state) {
        // CalcFade (:1317-1341): a window of maxfade frames in one state -> fade 1 if that state is "logo on", else 0
        const int st0 = frame_state[clampf(n - half)];
        bool uniform = true;
        for (int d = -half + 1; d <= half; ++d) uniform = uniform && frame_state[clampf(n + d)] == st0;
        if (uniform) {
            const float f = frame_state[clampf(n)] == 2 ? 1.0f : 0.0f;
            out[i] = make_float2(f, f);
            return;
        }
    }
    const int nAnalyze = (num_frames + 7) / 8;
    int best[9];
    const float* centre = analysis;
#pragma unroll
    for (int d = -4; d <= 4; ++d) {
        const int k = clampf(n + d) + d;
        const int q = max(0, min(nAnalyze - 1, k >> 3));
        const int src = clampf(q * 8 + (k & 7));
        // the host has checked that every record read lies inside [analysis_first, analysis_first + analysis_count)
        const int rel = max(0, min(analysis_count - 1, src - analysis_first));
        const float* rec = analysis + (size_t)rel * 33;
        best[d + 4] = argmin11_dev(rec);
        if (d == 0) centre = rec;
    }
    const int before = best[3] + best[2] + best[1] + best[0], after = best[5] + best[6] + best[7] + best[8];     // 0..40 each
    auto bit = [](unsigned long long m, int s) { return ((m >> s) & 1ull) != 0; };
    const bool abrupt = (bit(tab.below03, before) && bit(tab.above07, after)) || (bit(tab.above07, before) && bit(tab.below03, after));
    if (abrupt) out[i] = make_float2(tab.v[argmin11_dev(centre + 11)], tab.v[argmin11_dev(centre + 22)]);
    else out[i] = make_float2(tab.v[best[4]], tab.v[best[4]]);
}

hipError_t launch_calc_fades(hipStream_t st, const float* danalysis, int analysis_first, int analysis_count, int num_frames, int first,
                             int nframes, const uint8_t* dstate, int half, float2* dout)
{
    if (nframes <= 0) return hipSuccess;
    FadeTable tab;
    for (int i = 0; i < 11; ++i) tab.v[i] = (float)i / 10.0f;
    tab.below03 = tab.above07 = 0;
    for (int s = 0; s <= 40; ++s) {
        float q = 0;
        q += s;                 // (the sum of four small integers is exact in fp32 whatever the order)
        q /= 40;
        if (q < 0.3) tab.below03 |= 1ull << s;
        if (q > 0.7) tab.above07 |= 1ull << s;
    }
    hipLaunchKernelGGL(calc_fades_kernel, dim3((unsigned)((nframes + 127) / 128)), dim3(128), 0, st, danalysis, analysis_first,
                       analysis_count, num_frames, first, nframes, dstate, half, tab, dout);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LogoScan::AddFrame, first half (LogoScan.hpp:604-653): the rectangle's border samples of each plane,
// reject when max-min > thy, else background = med_average (:414-428) = (int)((sum of the middle half of
// the sorted samples + nn/2) / nn).  Sorting is replaced by a histogram (exact: the sorted sequence is
// the histogram read out in order).  One workgroup per frame.
// out[frame] = {valid, bgY, bgU, bgV}
// ------------------------------------------------------------------------------------------------
constexpr int kBorderThreads = 256;

template <typename pix_t>
__device__ void border_plane(const pix_t* __restrict__ p, int pitch, int w, int h, int nbins, int* hist,
                             long long* red, int* redi, int& vmin, int& vmax, int& bg)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < nbins; i += kBorderThreads) hist[i] = 0;
    __syncthreads();
    // samples above the declared depth (a 10-bit clip in uint16 containers is not guaranteed to stay below 1 << bits) are
    // counted in the top bin instead of indexing past the histogram
    const int top = nbins - 1;
    for (int x = tid; x < w; x += kBorderThreads) {
        atomicAdd(&hist[min((int)p[x], top)], 1);
        atomicAdd(&hist[min((int)p[x + (long long)(h - 1) * pitch], top)], 1);
    }
    for (int y = 1 + tid; y < h - 1; y += kBorderThreads) {
        atomicAdd(&hist[min((int)p[(long long)y * pitch], top)], 1);
        atomicAdd(&hist[min((int)p[w - 1 + (long long)y * pitch], top)], 1);
    }
    __syncthreads();
    const int n = 2 * w + 2 * max(0, h - 2);     // a one-row plane (chroma of a 2-row rectangle) pushes its row twice (:616-635)
    const int lo = n / 4, hi = n - n / 4;
    // each thread owns nbins/256 consecutive bins; exclusive prefix of the per-thread counts through LDS
    const int per = (nbins + kBorderThreads - 1) / kBorderThreads;
    const int b0 = tid * per, b1 = min(nbins, b0 + per);
    int cnt = 0;
    for (int b = b0; b < b1; ++b) cnt += hist[b];
    redi[tid] = cnt;
    __syncthreads();
    int before = 0;
    for (int t = 0; t < tid; ++t) before += redi[t];
    long long part = 0;
    int mn = 0x7FFFFFFF, mx = -1;
    int cum = before;
    for (int b = b0; b < b1; ++b) {
        const int c = hist[b];
        if (c) {
            mn = min(mn, b);
            mx = max(mx, b);
            const int s = max(cum, lo), e = min(cum + c, hi);
            if (e > s) part += (long long)(e - s) * b;
        }
        cum += c;
    }
    __syncthreads();
    red[tid] = part;
    redi[tid] = mn;
    redi[kBorderThreads + tid] = mx;
    __syncthreads();
    for (int s = kBorderThreads / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] += red[tid + s];
            redi[tid] = min(redi[tid], redi[tid + s]);
            redi[kBorderThreads + tid] = max(redi[kBorderThreads + tid], redi[kBorderThreads + tid + s]);
        }
        __syncthreads();
    }
    vmin = redi[0];
    vmax = redi[kBorderThreads];
    const int nn = hi - lo;
    double t = (double)red[0];
    t = (t + nn / 2) / nn;
    bg = (int)t;
    __syncthreads();
}

template <typename pix_t>
__global__ __launch_bounds__(kBorderThreads)
void scan_border_kernel(const pix_t* __restrict__ Y, const pix_t* __restrict__ U, const pix_t* __restrict__ V,
                        long long strideY, long long strideUV, int pitchY, int pitchUV, int imgx, int imgy, int cx, int cy,
                        int w, int h, int wUV, int hUV, int nbins, int thy, int4* __restrict__ out)
{
    extern __shared__ int sh[];
    int* hist = sh;                                        // nbins
    int* redi = hist + nbins;                              // 2*256
    long long* red = (long long*)(redi + 2 * kBorderThreads);   // 256 (8-byte aligned: nbins is a multiple of 2)
    const int frame = blockIdx.x;
    int mn, mx, bgY, bgU, bgV;
    bool ok = true;
    border_plane(Y + (long long)frame * strideY + (long long)imgy * pitchY + imgx, pitchY, w, h, nbins, hist, red, redi, mn, mx, bgY);
    ok = ok && (abs(mn - mx) <= thy);
    border_plane(U + (long long)frame * strideUV + (long long)cy * pitchUV + cx, pitchUV, wUV, hUV, nbins, hist, red, redi, mn, mx, bgU);
    ok = ok && (abs(mn - mx) <= thy);
    border_plane(V + (long long)frame * strideUV + (long long)cy * pitchUV + cx, pitchUV, wUV, hUV, nbins, hist, red, redi, mn, mx, bgV);
    ok = ok && (abs(mn - mx) <= thy);
    if (threadIdx.x == 0) out[frame] = make_int4(ok ? 1 : 0, bgY, bgU, bgV);
}

hipError_t launch_scan_border(hipStream_t st, int bits, const void* dY, const void* dU, const void* dV, long long strideY,
                              long long strideUV, int pitchY, int pitchUV, int imgx, int imgy, int cx, int cy, int w, int h,
                              int wUV, int hUV, int thy, int nframes, int4* dout)
{
    if (nframes <= 0) return hipSuccess;
    const int nbins = 1 << bits;
    const size_t lds = (size_t)nbins * 4 + 2 * kBorderThreads * 4 + kBorderThreads * 8;
    dim3 grid((unsigned)nframes), block(kBorderThreads);
    if (bits <= 8)
        hipLaunchKernelGGL(scan_border_kernel<uint8_t>, grid, block, lds, st, (const uint8_t*)dY, (const uint8_t*)dU,
                           (const uint8_t*)dV, strideY, strideUV, pitchY, pitchUV, imgx, imgy, cx, cy, w, h, wUV, hUV, nbins, thy, dout);
    else
        hipLaunchKernelGGL(scan_border_kernel<uint16_t>, grid, block, lds, st, (const uint16_t*)dY, (const uint16_t*)dU,
                           (const uint16_t*)dV, strideY, strideUV, pitchY, pitchUV, imgx, imgy, cx, cy, w, h, wUV, hUV, nbins, thy, dout);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LogoScan::AddScanFrame (LogoScan.hpp:568-592) over the accepted frames of a batch.  The reference
// keeps five double sums per pixel; F, F*F and F*B of integer samples are exact integers, so int64
// accumulation gives the same values in any order (and lets shards be all-reduced exactly).  sumB and
// sumB2 are per-plane constants of the accepted set and are kept by the host.
// acc: 3 int64 per pixel {sumF, sumF2, sumFB}; pixels = Y rows, then U rows, then V rows.
// ------------------------------------------------------------------------------------------------
// Frames are split over blockIdx.y so that the launch has a few thousand workgroups, no more: every workgroup ends with
// three 64-bit atomics per pixel, and with a fixed 32 frames per workgroup those atomics (not the 1 B/pixel/frame of
// reads) were the whole cost of a 20000-frame accumulate.
constexpr int kAccMinFramesPerBlock = 32;
constexpr int kAccFrameChunks = 16;

template <typename pix_t>
__global__ __launch_bounds__(256)
void scan_accumulate_kernel(const pix_t* __restrict__ Y, const pix_t* __restrict__ U, const pix_t* __restrict__ V,
                            long long strideY, long long strideUV, int pitchY, int pitchUV, int imgx, int imgy, int cx, int cy,
                            int w, int h, int wUV, int hUV, const int4* __restrict__ accepted /* {frame, bgY, bgU, bgV} */,
                            int naccepted, int frames_per_block, unsigned long long* __restrict__ acc)
{
    const int r = blockIdx.x;
    const int g0 = blockIdx.y * frames_per_block;
    const int g1 = min(naccepted, g0 + frames_per_block);
    const pix_t* base;
    long long stride;
    int roww, pl;
    size_t pix0;
    if (r < h) {
        pl = 0; roww = w; stride = strideY;
        base = Y + (long long)(imgy + r) * pitchY + imgx;
        pix0 = (size_t)r * w;
    } else {
        pl = (r - h) >= hUV ? 2 : 1;
        const int y = r - h - (pl - 1) * hUV;
        roww = wUV; stride = strideUV;
        base = (pl == 1 ? U : V) + (long long)(cy + y) * pitchUV + cx;
        pix0 = (size_t)w * h + (size_t)(pl - 1) * wUV * hUV + (size_t)y * wUV;
    }
    for (int x = threadIdx.x; x < roww; x += blockDim.x) {
        long long sF = 0, sF2 = 0, sFB = 0;
#pragma unroll 8
        for (int i = g0; i < g1; ++i) {
            const int4 a = accepted[i];
            const int f = base[(long long)a.x * stride + x];
            const int b = pl == 0 ? a.y : (pl == 1 ? a.z : a.w);
            sF += f;
            sF2 += f * f;
            sFB += f * b;
        }
        unsigned long long* o = acc + (pix0 + x) * 3;
        atomicAdd(o + 0, (unsigned long long)sF);
        atomicAdd(o + 1, (unsigned long long)sF2);
        atomicAdd(o + 2, (unsigned long long)sFB);
    }
}

hipError_t launch_scan_accumulate(hipStream_t st, int bits, const void* dY, const void* dU, const void* dV, long long strideY,
                                  long long strideUV, int pitchY, int pitchUV, int imgx, int imgy, int cx, int cy, int w, int h,
                                  int wUV, int hUV, const int4* daccepted, int naccepted, unsigned long long* dacc)
{
    if (naccepted <= 0) return hipSuccess;
#ifdef AMT_SCAN_ACC_FIXED32        // instrumented build: round 1's fixed 32 frames per workgroup
    const int per = kAccMinFramesPerBlock;
#else
    const int per = std::max(kAccMinFramesPerBlock, (naccepted + kAccFrameChunks - 1) / kAccFrameChunks);
#endif
    dim3 grid((unsigned)(h + 2 * hUV), (unsigned)((naccepted + per - 1) / per)), block(256);
    if (bits <= 8)
        hipLaunchKernelGGL(scan_accumulate_kernel<uint8_t>, grid, block, 0, st, (const uint8_t*)dY, (const uint8_t*)dU,
                           (const uint8_t*)dV, strideY, strideUV, pitchY, pitchUV, imgx, imgy, cx, cy, w, h, wUV, hUV, daccepted, naccepted, per, dacc);
    else
        hipLaunchKernelGGL(scan_accumulate_kernel<uint16_t>, grid, block, 0, st, (const uint16_t*)dY, (const uint16_t*)dU,
                           (const uint16_t*)dV, strideY, strideUV, pitchY, pitchUV, imgx, imgy, cx, cy, w, h, wUV, hUV, daccepted, naccepted, per, dacc);
    return hipGetLastError();
}

} // namespace amt
