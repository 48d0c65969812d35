// eval_pair_kernels.hip -- the two-fade evaluation of LogoFrame::ScanFrame (LogoScan.hpp:1543-1568): corr0 = EvaluateLogo(fade 0)
// and corr1 = EvaluateLogo(fade 1) of every logo on every frame, in the reference's fp32 evaluation order (bit-exact records).
//
// With fade 0 the blended window  fade*bg + (1-fade)*s  (LogoScan.hpp:244-251) IS s, with fade 1 it IS bg = a*s + b*maxv
// (0*x + y == y for finite x; the host checks that every logo coefficient is finite and small enough for bg to stay finite,
// and launches the generic kernel of eval_fused_kernels.hip otherwise).  So the two evaluations of a mask pixel are the SAME
// instruction stream on two operands: LDS holds the band's rows as interleaved {s, bg} pairs, a window element arrives as one
// 8-byte read, and every add / sub / mul of CalcCorrelation5x5_AVX's order (ComputeKernel.cpp:77-121, exact_math.h) is one
// packed fp32 instruction whose low half evaluates fade 0 and whose high half evaluates fade 1 -- no blend arithmetic, no
// FMA contraction (-ffp-contract=off), the 25 taps broadcast to both halves through op_sel.
//
// Shape: workgroup = (logo, G frames), 8 evaluation waves + 1 summing wave, walking the logo's pixel bands (<= 512
// raster-consecutive mask pixels and the <= 16 rows their windows touch; the tables of the linear kernel); two frames per
// iteration.  An evaluation thread owns ONE mask pixel.  Pipeline, ONE barrier per (band, frame pair) iteration:
//   * the raw rows of the next iteration are requested at the top with buffer_load ... lds (no registers held) into the wave's
//     own rows of the other half of a double-buffered plane, and converted to {s, bg} in place after the evaluation;
//   * the band's logo coefficients stay in LDS across the frames of the workgroup;
//   * per-pixel terms go to an LDS row per (frame, fade); the ninth wave adds the PREVIOUS iteration's rows front to back -- one
//     lane per row, the reference's order (`result += score`, LogoScan.hpp:295-315) -- while the others evaluate the current one.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <algorithm>

#include "eval_plan.h"
#include "exact_math.h"
#include "eval_lds_stage.h"
#include "eval_ordered_sum.h"

namespace amt {

using namespace lin;

constexpr int kPairEvalWaves = kLinThreads / 64;                 // 8: one mask pixel per evaluation thread
constexpr int kPairThreads = kLinThreads + 64;                   // + the summing wave
constexpr int kPairFPI = 2;                                      // frames per iteration
constexpr int kPairRows = 2 * kPairFPI;                          // score rows per iteration: (frame, fade)
constexpr int kPairRowPitch = kLinBandPix + kEvalScorePad;       // floats; the sum reads ahead of the row's end

// {corr(k, s), corr(k, bg)} and the two window means in the reference's order (exact_math.h corr5x5_strided), both halves at once
__device__ __forceinline__ f2 window_corr_exact(const f2 (&Kp)[13], const f2 (&W)[25], f2& M)
{
    f2 c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = ((W[i] + W[5 + i]) + (W[10 + i] + W[15 + i])) + W[20 + i];
    M = div25_pk(((c[0] + c[4]) + c[2]) + (c[1] + c[3]));
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

template <typename pix_t>
__global__ __launch_bounds__(kPairThreads)
void logo_eval_pair_kernel(const EvalLogoDev* __restrict__ logos, const LinLogoDev* __restrict__ lins, const EvalBand* __restrict__ bands,
                           const pix_t* __restrict__ Y, const int* __restrict__ frame_map, long long frame_stride, int pitch, float maxv,
                           int nframes, int G, int ngroups, float* __restrict__ out, int out_frame_stride, int take_abs, int plane_cap)
{
    extern __shared__ float lds[];
    f2* const planes = reinterpret_cast<f2*>(lds);                       // [2][kPairFPI][plane_cap] {s, bg} of a band's rows
    f2* const abp = planes + 2 * kPairFPI * plane_cap;                   // [plane_cap] {a, b} of the current band's rows
    float* const rows = lds + (2 * kPairFPI + 1) * 2 * plane_cap;        // [2][kPairRows][kPairRowPitch] per-pixel terms
    float* const accs = rows + 2 * kPairRows * kPairRowPitch;            // [G][2] running sums

    const int logo = blockIdx.x / ngroups;
    const int grp = blockIdx.x - logo * ngroups;
    const int F0 = grp * G;
    const int gcount = min(G, nframes - F0);
    const EvalLogoDev L = logos[logo];
    const LinLogoDev X = lins[logo];
    const gptr_t gScales = (gptr_t)L.scales, gK = (gptr_t)X.kpix, gPos = (gptr_t)X.pos;
    const unsigned cpad = (unsigned)L.count_pad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = L.w, lp = L.lp;
    constexpr unsigned ES = sizeof(pix_t);

    if (tid < 2 * G) accs[tid] = 0.0f;
    const int npairs = (gcount + kPairFPI - 1) / kPairFPI;
    const int niter = X.nbands * npairs;

    if (wave == kPairEvalWaves) {
        // ---------------- the summing wave: iteration it adds the rows written during iteration it - 1 ----------------
        __syncthreads();                                  // the prologue's barrier
        int bi = 0, pr = 0;
        int prev_npix = 0, prev_g = 0, prev_rows = 0;
        for (int it = 0; it < niter; ++it) {
#ifndef AMT_PAIR_NO_SUM
            if (it > 0 && lane < prev_rows) {
#else
            if (it > 0 && lane < prev_rows && prev_npix > 100000) {
#endif
                float* a = accs + prev_g * 2 + lane;      // row fr*2 + fade belongs to frame prev_g + fr
                *a = ordered_row_sum(rows + (((it - 1) & 1) * kPairRows + lane) * kPairRowPitch, prev_npix, *a);
            }
            prev_npix = bands[X.band0 + bi].npix;
            prev_g = pr * kPairFPI;
            prev_rows = 2 * min(kPairFPI, gcount - prev_g);
            if (++pr == npairs) { pr = 0; ++bi; }
            __syncthreads();
        }
        if (niter > 0 && lane < prev_rows) {
            float* a = accs + prev_g * 2 + lane;
            *a = ordered_row_sum(rows + (((niter - 1) & 1) * kPairRows + lane) * kPairRowPitch, prev_npix, *a);
        }
        __syncthreads();
    } else {
        // ---------------- evaluation waves ----------------
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.a), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.b), 0, 0x7FFFFFFF, 0x00020000);
        // staging of one (band, frame): a wave owns rows 2*wave, 2*wave+1 of the band, a lane four adjacent columns (w <= 256);
        // a ragged right edge (w % 4 == 2) is covered by shifting the last lane group left
        const bool slane = 4 * lane < w;
        const int sx = min(4 * lane, w - 4);
        const int nl = (w + 3) >> 2;
        auto frame_rsrc = [&](int g) {
            const int frame = F0 + min(g, gcount - 1);            // the second frame of a ragged last pair repeats the first
            const int srcFrame = frame_map ? frame_map[frame] : frame;
            const pix_t* src = Y + (long long)srcFrame * frame_stride + (long long)(L.imgy + L.row0) * pitch + L.imgx;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<pix_t*>(src), 0, 0x7FFFFFFF, 0x00020000);
        };
        auto src_row = [&](int y, int j) {
            return L.deint ? min(max(y - 1 + j, 0), L.h - 1) : min(y + max(j - 1, 0), L.h - 1) * L.row_step;
        };
        auto load_raw = [&](const __amdgpu_buffer_rsrc_t rS, int y0, Raw4<pix_t> (&raw)[kStageRows + 2]) {
            const int y = y0 + kStageRows * wave;
#pragma unroll
            for (int j = 0; j < kStageRows + 2; ++j) raw[j].load_buf(rS, (unsigned)sx * ES, src_row(y, j) * pitch * (int)ES);
        };
        // LDS-direct request of the next iteration's raw rows into the first of the wave's own rows of the plane they will be
        // converted into (4 * nl * sizeof(sample) * 4 <= one plane row); collected by pickup_raw after the evaluation
        auto request_raw = [&](const __amdgpu_buffer_rsrc_t rS, int y0, int nrows, f2* plane) {
            const int rg = kStageRows * wave;
            if (rg >= nrows || !slane) return;
            const int y = y0 + rg;
            unsigned* dst = reinterpret_cast<unsigned*>(plane + rg * lp);
#pragma unroll
            for (int j = 0; j < kStageRows + 2; ++j)
                Raw4<pix_t>::request_lds(rS, dst + j * nl * Raw4<pix_t>::kDwordsPerLane, (unsigned)sx * ES, src_row(y, j) * pitch * (int)ES, nl);
        };
        auto pickup_raw = [&](int nrows, const f2* plane, Raw4<pix_t> (&raw)[kStageRows + 2]) {
            const int rg = kStageRows * wave;
            if (rg >= nrows || !slane) return;
            const unsigned* src = reinterpret_cast<const unsigned*>(plane + rg * lp);
#pragma unroll
            for (int j = 0; j < kStageRows + 2; ++j) raw[j].from_lds(src + j * nl * Raw4<pix_t>::kDwordsPerLane, lane, nl);
        };
        auto load_ab = [&](int y0, f4 (&av)[kStageRows], f4 (&bv)[kStageRows]) {
            const int y = y0 + kStageRows * wave;
#pragma unroll
            for (int j = 0; j < kStageRows; ++j) {
                const int ro = min(y + j, L.h - 1) * w * 4;
                av[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rA, (unsigned)sx * 4u, ro, 0));
                bv[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rB, (unsigned)sx * 4u, ro, 0));
            }
        };
        // the band's coefficients as {a, b*maxv} pairs in LDS (the product is rounded once, exactly as in a*s + b*maxv), written and
        // read by the wave that stages those rows (no barrier involved); the column offset is made opaque where LDS addresses are
        // formed (hoisted, they would be spilled)
        auto ab_to_lds = [&](int nrows, const f4 (&av)[kStageRows], const f4 (&bv)[kStageRows]) {
            if (!slane) return;
            int sxl = sx;
            asm volatile("" : "+v"(sxl));
#pragma unroll
            for (int j = 0; j < kStageRows; ++j) {
                if (kStageRows * wave + j >= nrows) break;
                f4* d = reinterpret_cast<f4*>(abp + (kStageRows * wave + j) * lp + sxl);
                d[0] = f4{av[j][0], bv[j][0] * maxv, av[j][1], bv[j][1] * maxv};
                d[1] = f4{av[j][2], bv[j][2] * maxv, av[j][3], bv[j][3] * maxv};
            }
        };
        auto ab_from_lds = [&](int nrows, f4 (&av)[kStageRows], f4 (&bv)[kStageRows]) {
            int sxl = sx;
            asm volatile("" : "+v"(sxl));
#pragma unroll
            for (int j = 0; j < kStageRows; ++j) {
                if (kStageRows * wave + j >= nrows) break;
                const f4* d = reinterpret_cast<const f4*>(abp + (kStageRows * wave + j) * lp + min(sxl, lp - 4));
                const f4 lo = d[0], hi = d[1];
                av[j] = f4{lo[0], lo[2], hi[0], hi[2]};
                bv[j] = f4{lo[1], lo[3], hi[1], hi[3]};
            }
        };
        // byte-wise conversion; the [1 2 1] blend of DeintY (LogoScan.hpp:763-780) on floats: every intermediate is an integer below
        // 2^24, so (r0 + 2 r1 + r2 + 2) * 0.25 equals the reference's (float)(int sum) / 4.0f bit for bit
        auto convert_store = [&](f2* plane, int y0, int nrows, const Raw4<pix_t> (&raw)[kStageRows + 2], const f4 (&av)[kStageRows],
                                 const f4 (&bv)[kStageRows]) {
            const int rg = kStageRows * wave;
            if (!slane) return;
            int sxl = sx;
            asm volatile("" : "+v"(sxl));
            f4 fr[kStageRows + 2];
#pragma unroll
            for (int j = 0; j < kStageRows + 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) fr[j][k] = (float)raw[j].get(k);
#pragma unroll
            for (int j = 0; j < kStageRows; ++j) {
                const int yy = y0 + rg + j;
                if (rg + j < nrows) {
                    f4 sv;
                    if (L.deint && yy != 0 && yy != L.h - 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) sv[k] = ((fr[j][k] + 2.0f * fr[j + 1][k]) + (fr[j + 2][k] + 2.0f)) * 0.25f;
                    } else {
                        sv = fr[j + 1];
                    }
                    // bv holds b*maxv: bg = a*s + b*maxv (LogoScan.hpp:247), the same two roundings
                    f2* dst = plane + (rg + j) * lp + sxl;
                    reinterpret_cast<f4*>(dst)[0] = f4{sv[0], av[j][0] * sv[0] + bv[j][0], sv[1], av[j][1] * sv[1] + bv[j][1]};
                    reinterpret_cast<f4*>(dst)[1] = f4{sv[2], av[j][2] * sv[2] + bv[j][2], sv[3], av[j][3] * sv[3] + bv[j][3]};
                }
            }
        };

        EvalBand B = bands[X.band0];
        // prologue: the first iteration's rows
        {
            f4 av[kStageRows], bv[kStageRows];
            load_ab(B.y0, av, bv);
            ab_to_lds(B.nrows, av, bv);
            ab_from_lds(B.nrows, av, bv);                 // bv = b*maxv from here on
#pragma unroll
            for (int fr = 0; fr < kPairFPI; ++fr) {
                Raw4<pix_t> raw[kStageRows + 2];
                load_raw(frame_rsrc(fr), B.y0, raw);
                convert_store(planes + fr * plane_cap, B.y0, B.nrows, raw, av, bv);
            }
        }
        bool act = false;
        unsigned m8 = 0;
        int woff = 0;
        const unsigned cpad8 = cpad * 8u;
        f2 Kp[13];
        __syncthreads();

        int bi = 0, pr = 0;
        for (int it = 0; it < niter; ++it) {
            const int cur = it & 1;
            f2* const plane = planes + cur * kPairFPI * plane_cap;
            f2* const nplane = planes + (cur ^ 1) * kPairFPI * plane_cap;
            if (pr == 0) {
                // ---- a new band: this thread's mask pixel, its window offset and its taps ----
                act = tid < B.npix;
                const unsigned m = (unsigned)(B.m0 + (act ? tid : 0));
                const unsigned pos = gld<unsigned>(gPos, m * 4u);
                woff = ((int)(pos >> 16) - 2 - B.y0) * lp + (int)(pos & 0xFFFFu) - 2;
                m8 = m * 8u;
#pragma unroll
                for (int j = 0; j < 13; ++j) Kp[j] = gld<f2>(gK, ((unsigned)j * cpad + m) * 8u);
            }
            const bool has_next = it + 1 < niter;
            const bool next_band = pr + 1 == npairs;
            const int npr = next_band ? 0 : pr + 1;
            EvalBand Bn = B;
            if (has_next && next_band) {
                const EvalBand* nb = bands + X.band0 + bi + 1;
                Bn.m0 = nb->m0; Bn.npix = nb->npix; Bn.y0 = nb->y0; Bn.nrows = nb->nrows;
            }
            // ---- 1. request the next iteration's raw rows ----
#ifndef AMT_PAIR_NO_STAGE
            if (has_next) {
#else
            if (false) {
#endif
#pragma unroll
                for (int fr = 0; fr < kPairFPI; ++fr) request_raw(frame_rsrc(npr * kPairFPI + fr), Bn.y0, Bn.nrows, nplane + fr * plane_cap);
            }
            // ---- 2. both fades of both frames: one packed window evaluation per frame ----
            // (the taps are loop-invariant: LICM would hoist their {k,k} broadcasts and keep 50 registers of copies; the empty asm
            //  makes them opaque per iteration and the broadcast folds into the multiply's op_sel)
#pragma unroll
            for (int j = 0; j < 13; ++j) asm volatile("" : "+v"(Kp[j]));
            float* const myrows = rows + cur * kPairRows * kPairRowPitch + tid;
            f2 R[kPairFPI], sc0[kPairFPI], sc1[kPairFPI];
#pragma unroll
            for (int fr = 0; fr < kPairFPI; ++fr) {
                f2 W[25], M;
#ifdef AMT_PAIR_NO_EVAL
                M = plane[fr * plane_cap + woff]; R[fr] = Kp[fr] * M;
#else
                load_window(plane + fr * plane_cap, woff, lp, W);        // surplus threads read pixel B.m0's window: never written out
                R[fr] = window_corr_exact(Kp, W, M);
#endif
#ifdef AMT_PAIR_NO_GATHER
                sc0[fr] = f2{1e-4f * (float)score_bin_dev(M.x), 0.5f}; sc1[fr] = f2{1e-4f * (float)score_bin_dev(M.y), 0.5f};
#else
                sc0[fr] = gld<f2>(gScales, __umul24((unsigned)score_bin_dev(M.x), cpad8) + m8);
                sc1[fr] = gld<f2>(gScales, __umul24((unsigned)score_bin_dev(M.y), cpad8) + m8);
#endif
            }
            f4 av[kStageRows], bv[kStageRows];
            if (has_next && next_band) load_ab(Bn.y0, av, bv);           // once per band, from memory
            // ---- 3. per-pixel terms (LogoScan.hpp:305-308) -> the score rows ----
#pragma unroll
            for (int fr = 0; fr < kPairFPI; ++fr) {
                if (act) {
                    myrows[(fr * 2 + 0) * kPairRowPitch] = score_term(R[fr].x, sc0[fr].x, sc0[fr].y);
                    myrows[(fr * 2 + 1) * kPairRowPitch] = score_term(R[fr].y, sc1[fr].x, sc1[fr].y);
                }
            }
            // ---- 4. the next iteration's rows: raw -> {s, bg} in place (holding the terms back until after the conversion so that it
            //      covers the scale gathers' trip was tried: 36 registers spilled, 5.4 -> 8.2 ms per 10 000 frames) ----
#ifdef AMT_PAIR_NO_STAGE
            if (false) {
#else
            if (has_next) {
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the LDS-direct loads are counted with the vector-memory loads
                if (next_band) ab_to_lds(Bn.nrows, av, bv);
                ab_from_lds(Bn.nrows, av, bv);
#pragma unroll
                for (int fr = 0; fr < kPairFPI; ++fr) {
                    Raw4<pix_t> raw[kStageRows + 2];
                    pickup_raw(Bn.nrows, nplane + fr * plane_cap, raw);
                    convert_store(nplane + fr * plane_cap, Bn.y0, Bn.nrows, raw, av, bv);
                }
            }
            __syncthreads();                     // next planes and this iteration's score rows complete; current planes consumed
            if (next_band) { B.m0 = Bn.m0; B.npix = Bn.npix; B.y0 = Bn.y0; B.nrows = Bn.nrows; ++bi; }
            pr = npr;
        }
        __syncthreads();                         // the summing wave's last rows
    }
    if (tid < gcount * 2) {
        const int gg = tid >> 1, f = tid & 1;
        float r = accs[tid] / L.blackScore;
        if (take_abs) r = fabsf(r);
        out[(long long)(F0 + gg) * out_frame_stride + L.out_off + f] = r;
    }
}

hipError_t launch_logo_eval_pair(hipStream_t st, int bits, const EvalLogoDev* dlogos, const LinLogoDev* dlins, int nlogos,
                                 const EvalBand* dbands, const void* dY, const int* dframe_map, long long frame_stride_elems, int pitch,
                                 int nframes, int G, float* dout, int out_frame_stride, int take_abs, int plane_cap)
{
    if (nframes <= 0 || nlogos <= 0) return hipSuccess;
    if (2 * G > kLinThreads || plane_cap > kLinPlaneCap) return hipErrorInvalidValue;
    const int ngroups = (nframes + G - 1) / G;
    const float maxv = (float)((1 << bits) - 1);
    const size_t lds = ((size_t)(2 * kPairFPI + 1) * 2 * plane_cap + (size_t)2 * kPairRows * kPairRowPitch + (size_t)2 * G) * sizeof(float);
    dim3 grid((unsigned)((long long)ngroups * nlogos));
    if (bits <= 8)
        hipLaunchKernelGGL(logo_eval_pair_kernel<uint8_t>, grid, dim3(kPairThreads), lds, st, dlogos, dlins, dbands, (const uint8_t*)dY, dframe_map,
                           frame_stride_elems, pitch, maxv, nframes, G, ngroups, dout, out_frame_stride, take_abs, plane_cap);
    else
        hipLaunchKernelGGL(logo_eval_pair_kernel<uint16_t>, grid, dim3(kPairThreads), lds, st, dlogos, dlins, dbands, (const uint16_t*)dY, dframe_map,
                           frame_stride_elems, pitch, maxv, nframes, G, ngroups, dout, out_frame_stride, take_abs, plane_cap);
    return hipGetLastError();
}

} // namespace amt
