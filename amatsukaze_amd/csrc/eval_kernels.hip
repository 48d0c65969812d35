// eval_kernels.hip -- logo correlation on CDNA4 (gfx950).
//
// Replaces the inner loops of LogoDataParam::EvaluateLogo / CorrelationScore (LogoScan.hpp:231-255,
// 288-318) + DeintY / CopyY (:763-790) as they are driven by LogoFrame::ScanFrame (:1543-1568),
// AMTAnalyzeLogo::GetFrameT (:1119-1161) and LogoAnalyzer::ReMakeLogo (:955-982).
//
// Shape of the work: every mask pixel m of an evaluation logo owns a private 25-tap kernel k_m and is
// evaluated on `nfades` blends of every frame.  There is no operand reuse across m (not a GEMM, no
// MFMA); the reuse that exists is k_m across fades and the 5x5 windows overlapping in the rectangle.
// So: one workgroup = (frame, band of <= 256 run slots); a run slot is up to PXT horizontally adjacent mask pixels
// owned by one thread, which keeps their kernel taps in VGPRs for all fades and reads ONE 5 x (4+PXT) window
// per fade for all of them (mask pixels come in runs along logo edges: 97 % pair up, which halves the LDS
// traffic per pixel).  The unblended rectangle rows of the band live in LDS (double buffered per fade, one
// barrier per fade), the source pixels and their background estimate stay in registers between fades.  fp32 VALU-bound, ~115 non-fusable ops per mask pixel per
// fade; the per-pixel order of operations is the reference's (exact_math.h).
//
// The cross-pixel sum is sequential in the reference (result += score, raster order).  To return the
// reference's bits rather than a re-associated sum, per-pixel terms go to a scratch row and
// ordered_sum_kernel adds each row front to back, 64 rows per wave (transposed through LDS so that the
// global reads stay coalesced).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

#include "eval_plan.h"
#include "exact_math.h"

namespace amt {

// Table pointers reach the kernel inside a struct read from memory, so the compiler cannot tell their address
// space and would emit FLAT loads (which also tick the LDS counter and serialise against the window reads).
// They are all hipMalloc'ed: say so.
// They are all hipMalloc'ed: say so, and address them as uniform base + 32-bit byte offset so that the loads
// take the SGPR-base form instead of 64-bit per-lane address arithmetic.
typedef const __attribute__((address_space(1))) char* gbase_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T gload(gbase_t base, unsigned byteoff)
{
    return *reinterpret_cast<const __attribute__((address_space(1))) T*>(base + byteoff);
}

// PXT = mask pixels per thread (kernel taps held in VGPRs), STG = staged rectangle pixels per thread
// (plane floats <= kEvalThreads * STG).  {4,16}: 198 VGPRs, 2 waves/SIMD; {2,12}: ~128 VGPRs, 4 waves/SIMD.
template <typename pix_t, int PXT, int STG, int NT>
__global__ __launch_bounds__(NT)
void logo_corr_kernel(const EvalLogoDev* __restrict__ logos, const EvalBand* __restrict__ bands,
                      int nbands, int nbands8, const float* __restrict__ fades, int nfades,
                      const pix_t* __restrict__ Y, const int* __restrict__ frame_map, long long frame_stride, int pitch,
                      float maxv, float* __restrict__ scores, long long scores_per_frame, int plane_cap)
{
    extern __shared__ float lds[];                 // W0[plane_cap], W1[plane_cap]
    // XCD-aware block -> (frame, band): blocks are dealt to XCDs round-robin (b % 8), so band slot
    // (b % 8) + 8*k keeps every band's tables (kernels, scales: ~350 B per mask pixel) in ONE XCD's L2.
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int j = id >> 3;
    const int slot = j % nbands8;
    const int frame = j / nbands8;
    const int band = slot * 8 + xcd;
    if (band >= nbands) return;

    const EvalBand B = bands[band];
    const EvalLogoDev L = logos[B.logo];
    const gbase_t gA = (gbase_t)L.a, gB = (gbase_t)L.b, gPos = (gbase_t)L.pos, gKern = (gbase_t)L.kern, gScales = (gbase_t)L.scales;
    const unsigned cpad = (unsigned)L.count_pad;
    const int tid = threadIdx.x;
    const int w = L.w;
    const int lp = L.lp;                           // LDS row pitch: rows 8 banks apart
    const int nplane = B.nrows * lp;

    // ---- stage: source pixel s and background estimate bg = a*s + b*maxv, kept in registers ----
    float sreg[STG], bgreg[STG];
    {
        const int srcFrame = frame_map ? frame_map[frame] : frame;      // optional gather of non-contiguous frames
        const gbase_t src = (gbase_t)(Y + (long long)srcFrame * frame_stride + (long long)(L.imgy + L.row0) * pitch + L.imgx);
        constexpr unsigned ES = sizeof(pix_t);
#pragma unroll
        for (int q = 0; q < STG; ++q) {
            const int i = tid + q * NT;
            float s = 0.0f, bg = 0.0f;
            if (i < nplane) {
                const int r = (int)__umulhi((unsigned)i, L.lp_magic);   // i / lp
                const int x = i - r * lp;
                if (x < w) {
                    const int y = B.y0 + r;
                    if (L.deint) {
                        if (y == 0 || y == L.h - 1) {
                            s = (float)gload<pix_t>(src, (unsigned)(x + y * pitch) * ES);
                        } else {
                            const int p0 = gload<pix_t>(src, (unsigned)(x + (y - 1) * pitch) * ES);
                            const int p1 = gload<pix_t>(src, (unsigned)(x + y * pitch) * ES);
                            const int p2 = gload<pix_t>(src, (unsigned)(x + (y + 1) * pitch) * ES);
                            s = (float)(p0 + 2 * p1 + p2 + 2) / 4.0f;
                        }
                    } else {
                        s = (float)gload<pix_t>(src, (unsigned)(x + y * L.row_step * pitch) * ES);
                    }
                    const float a = gload<float>(gA, (unsigned)(x + y * w) * 4u);
                    const float b = gload<float>(gB, (unsigned)(x + y * w) * 4u);
                    bg = unblend_bg(a, b, maxv, s);
                }
            }
            sreg[q] = s;
            bgreg[q] = bg;
        }
    }

    // ---- this thread's run slot: up to PXT horizontally adjacent mask pixels, their kernel taps in registers
    //      for the whole fade loop, one shared 5 x (4+PXT) window per fade ----
    constexpr int WW = 4 + PXT;
    const bool act = tid < B.nslots;
    const uint32_t sl = gload<uint32_t>((gbase_t)L.slots, (unsigned)(B.s0 + (act ? tid : 0)) * 4u);
    const unsigned m0 = sl & 0x0FFFFFFFu;
    const int npx = act ? (int)(sl >> 28) : 0;
    int woff;
    {
        const uint32_t ps = gload<uint32_t>(gPos, m0 * 4u);
        const int x = ps & 0xFFFF, y = ps >> 16;
        woff = (y - 2 - B.y0) * lp + (x - 2);
    }
    float k[PXT][25];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const unsigned m = min(m0 + (unsigned)j, (unsigned)L.count - 1u);
#pragma unroll
        for (int t = 0; t < 25; ++t) k[j][t] = gload<float>(gKern, ((unsigned)t * cpad + m) * 4u);
    }

    const int pcap = plane_cap;
    float* wbuf0 = lds;
    float* wbuf1 = lds + pcap;
    auto mix = [&](float* dst, float fade) {
        const float omf = 1 - fade;
#pragma unroll
        for (int q = 0; q < STG; ++q) {
            const int i = tid + q * NT;
            if (i < nplane) dst[i] = fade * bgreg[q] + omf * sreg[q];
        }
    };

    mix(wbuf0, fades[0]);
    __syncthreads();
    __attribute__((address_space(1))) char* out =
        (__attribute__((address_space(1))) char*)(scores + (long long)frame * scores_per_frame + L.score_off);
    for (int f = 0; f < nfades; ++f) {
        const float* cur = (f & 1) ? wbuf1 : wbuf0;
        float* nxt = (f & 1) ? wbuf0 : wbuf1;
        if (f + 1 < nfades) mix(nxt, fades[f + 1]);
        if (act) {
            float v[5 * WW];
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int c = 0; c < WW; ++c) v[r * WW + c] = cur[woff + r * lp + c];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                if (j < npx) {
                    float mean;
                    const float corr = corr5x5_strided<WW>(k[j], v + j, &mean);
                    const f32x2_t sc = gload<f32x2_t>(gScales, ((unsigned)score_bin(mean) * cpad + m0 + j) * 8u);
                    *reinterpret_cast<__attribute__((address_space(1))) float*>(out + ((unsigned)f * cpad + m0 + j) * 4u) =
                        score_term(corr, sc.x, sc.y);
                }
            }
        }
        __syncthreads();
    }
}

// Sequential (reference-order) sum of each score row.
// rows of one logo: row e = frame*nfades + f, at scores + frame*scores_per_frame + score_off + f*count_pad.
// One wave owns 16 rows: all 64 lanes stream the rows in 128-column chunks (16 B per lane, two rows per
// instruction, next chunk in flight while the current one is consumed), lanes 0..15 then add their row's chunk
// front to back out of LDS.  The chain of dependent adds (count x ~5 cycles) is the floor per row; thousands of
// rows run side by side.
constexpr int kSumRows = 16, kSumCols = 128, kSumPitch = 132;   // pitch 132: the 16 row readers hit 64 distinct banks

__global__ __launch_bounds__(64)
void ordered_sum_kernel(const EvalLogoDev* __restrict__ logos, int nfades, int nframes,
                        const float* __restrict__ scores, long long scores_per_frame,
                        float* __restrict__ out, int out_frame_stride, int take_abs)
{
    __shared__ float tile[2][kSumRows][kSumPitch];
    const EvalLogoDev L = logos[blockIdx.y];
    const int lane = threadIdx.x;
    const int nrows = nframes * nfades;
    const int half = lane >> 5, c4 = (lane & 31) * 4;
    long long offs[kSumRows / 2];
#pragma unroll
    for (int j = 0; j < kSumRows / 2; ++j) {
        const int row = min(nrows - 1, (int)blockIdx.x * kSumRows + 2 * j + half);
        const int frame = row / nfades, f = row - frame * nfades;
        offs[j] = (long long)frame * scores_per_frame + L.score_off + (long long)f * L.count_pad + c4;
    }
    const int nchunks = (L.count + kSumCols - 1) / kSumCols;
    float4 v[kSumRows / 2];
#pragma unroll
    for (int j = 0; j < kSumRows / 2; ++j) v[j] = *reinterpret_cast<const float4*>(scores + offs[j]);
    float acc = 0;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
#pragma unroll
        for (int j = 0; j < kSumRows / 2; ++j) *reinterpret_cast<float4*>(&tile[buf][2 * j + half][c4]) = v[j];
        if (c + 1 < nchunks) {
#pragma unroll
            for (int j = 0; j < kSumRows / 2; ++j)
                v[j] = *reinterpret_cast<const float4*>(scores + offs[j] + (long long)(c + 1) * kSumCols);
        }
        __syncthreads();
        if (lane < kSumRows) {
            const int ncol = min(kSumCols, L.count - c * kSumCols);
            const float* rowp = tile[buf][lane];
            int q = 0;
            for (; q + 4 <= ncol; q += 4) {
                const float4 t = *reinterpret_cast<const float4*>(rowp + q);
                acc += t.x; acc += t.y; acc += t.z; acc += t.w;
            }
            for (; q < ncol; ++q) acc += rowp[q];
        }
    }
    const int row = blockIdx.x * kSumRows + lane;
    if (lane < kSumRows && row < nrows) {
        const int frame = row / nfades, f = row - frame * nfades;
        float r = acc / L.blackScore;
        if (take_abs) r = fabsf(r);
        out[(long long)frame * out_frame_stride + L.out_off + f] = r;
    }
}

// ---- launch helpers (called from the host engine) ----
size_t corr_lds_bytes(int plane_cap) { return (size_t)plane_cap * 2 * sizeof(float); }

template <typename pix_t, int PXT, int STG, int NT>
static void launch_corr_t(hipStream_t st, dim3 grid, size_t lds, const EvalLogoDev* dlogos, const EvalBand* dbands, int nbands,
                          int nbands8, const float* dfades, int nfades, const void* dY, const int* dframe_map,
                          long long frame_stride_elems, int pitch, float maxv, float* dscores, long long scores_per_frame, int plane_cap)
{
    hipLaunchKernelGGL((logo_corr_kernel<pix_t, PXT, STG, NT>), grid, dim3(NT), lds, st, dlogos, dbands, nbands, nbands8, dfades,
                       nfades, (const pix_t*)dY, dframe_map, frame_stride_elems, pitch, maxv, dscores, scores_per_frame, plane_cap);
}

// variant = (mask pixels per thread, threads per workgroup); see eval_variant() in eval_plan.h
hipError_t launch_logo_corr(hipStream_t st, int bits, int pxt, int nt, const EvalLogoDev* dlogos, const EvalBand* dbands, int nbands,
                            const float* dfades, int nfades, const void* dY, const int* dframe_map, long long frame_stride_elems,
                            int pitch, int nframes, float* dscores, long long scores_per_frame, int plane_cap)
{
    const int nbands8 = (nbands + 7) / 8;
    const long long nblocks = (long long)nframes * nbands8 * 8;
    if (nblocks <= 0) return hipSuccess;
    const float maxv = (float)((1 << bits) - 1);
    dim3 grid((unsigned)nblocks);
    const size_t lds = corr_lds_bytes(plane_cap);
#define AMT_CORR(T, P, S, N) launch_corr_t<T, P, S, N>(st, grid, lds, dlogos, dbands, nbands, nbands8, dfades, nfades, dY, dframe_map, \
                                                    frame_stride_elems, pitch, maxv, dscores, scores_per_frame, plane_cap)
#define AMT_CORR_T(T)                                                    \
    do {                                                                 \
        if (pxt == 1 && nt == 256) AMT_CORR(T, 1, 8, 256);               \
        else if (pxt == 1 && nt == 512) AMT_CORR(T, 1, 8, 512);          \
        else if (pxt == 1 && nt == 1024) AMT_CORR(T, 1, 4, 1024);        \
        else if (pxt == 2 && nt == 256) AMT_CORR(T, 2, 12, 256);         \
        else if (pxt == 2 && nt == 512) AMT_CORR(T, 2, 8, 512);          \
        else if (pxt == 4 && nt == 256) AMT_CORR(T, 4, 16, 256);         \
        else return hipErrorInvalidValue;                                \
    } while (0)
    if (bits <= 8) AMT_CORR_T(uint8_t); else AMT_CORR_T(uint16_t);
#undef AMT_CORR_T
#undef AMT_CORR
    return hipGetLastError();
}

hipError_t launch_ordered_sum(hipStream_t st, const EvalLogoDev* dlogos, int nlogos, int nfades, int nframes,
                              const float* dscores, long long scores_per_frame, float* dout, int out_frame_stride, int take_abs)
{
    if (nframes <= 0 || nlogos <= 0) return hipSuccess;
    dim3 grid((unsigned)((nframes * nfades + kSumRows - 1) / kSumRows), (unsigned)nlogos), block(64);
    hipLaunchKernelGGL(ordered_sum_kernel, grid, block, 0, st, dlogos, nfades, nframes, dscores, scores_per_frame, dout,
                       out_frame_stride, take_abs);
    return hipGetLastError();
}

} // namespace amt
