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

// {corr(k, s), corr(k, bg)} around the window means M = window_means(W) in the reference's order (exact_math.h corr5x5_strided),
// both halves at once
__device__ __forceinline__ f2 window_corr_exact(const f2 (&Kp)[13], const f2 (&W)[25], f2 M)
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

#ifdef AMT_PAIR_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#define AMT_PTICK(k) do { const long long t_ = clock64(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define AMT_PTICK(k) do { } while (0)
#endif
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
            AMT_PTICK(2);
            __syncthreads();
            AMT_PTICK(6);
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
        // Staging unit = one row of the band of one of the iteration's frames; the nrows * kPairFPI units of an iteration are dealt
        // round-robin to the 8 waves (unit u -> wave u % 8: a 10-row band gives every wave 2 or 3 units; whole rows per wave left
        // three waves idle and the rest with 4).  A lane stages four adjacent columns (w <= 256); a ragged right edge (w % 4 == 2)
        // is covered by shifting the last lane group left.
        constexpr int kMaxUnits = (kLinBandRows * kPairFPI + kPairEvalWaves - 1) / kPairEvalWaves;      // 4
        auto frame_rsrc = [&](int g) {
            const int frame = F0 + min(g, gcount - 1);            // the second frame of a ragged last pair repeats the first
            const int srcFrame = frame_map ? frame_map[frame] : frame;
            const pix_t* src = Y + (long long)srcFrame * frame_stride + (long long)(L.imgy + L.row0) * pitch + L.imgx;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<pix_t*>(src), 0, 0x7FFFFFFF, 0x00020000);
        };
        const RowStager<pix_t> st(L, lane, pitch, maxv);
        // this thread's mask pixel of a band: window offset, table index, taps
        auto load_pixel = [&](const EvalBand& Bd, bool& act_, int& woff_, unsigned& m8_, f2 (&K)[13]) {
            act_ = tid < Bd.npix;
            const unsigned m = (unsigned)(Bd.m0 + (act_ ? tid : 0));
            const unsigned pos = gld<unsigned>(gPos, m * 4u);
            woff_ = ((int)(pos >> 16) - 2 - Bd.y0) * lp + (int)(pos & 0xFFFFu) - 2;
            m8_ = m * 8u;
#pragma unroll
            for (int j = 0; j < 13; ++j) K[j] = gld<f2>(gK, ((unsigned)j * cpad + m) * 8u);
        };

        EvalBand B = bands[X.band0];
        bool act = false;
        unsigned m8 = 0;
        int woff = 0;
        const unsigned cpad8 = cpad * 8u;
        f2 Kp[13];
        load_pixel(B, act, woff, m8, Kp);
        // prologue: the first iteration's rows, straight from memory
        {
            const int U = B.nrows * kPairFPI;
#pragma unroll
            for (int k = 0; k < kMaxUnits; ++k) {
                const int u = wave + kPairEvalWaves * k;
                if (u >= U) break;
                const int fr = u >= B.nrows ? 1 : 0, r = u - fr * B.nrows;
                f2* prow = planes + fr * plane_cap + r * lp;
                st.request(frame_rsrc(fr), B.y0 + r, prow);
                f4 av, bmv;
                st.load_ab(rA, rB, B.y0 + r, av, bmv);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (fr == 0) st.ab_to_lds(abp + r * lp, av, bmv);
                st.convert(prow, B.y0 + r, av, bmv);
            }
        }
        __syncthreads();

        int bi = 0, pr = 0;
        for (int it = 0; it < niter; ++it) {
            const int cur = it & 1;
            f2* const plane = planes + cur * kPairFPI * plane_cap;
            f2* const nplane = planes + (cur ^ 1) * kPairFPI * plane_cap;
            const bool has_next = it + 1 < niter;
            const bool next_band = pr + 1 == npairs;
            const int npr = next_band ? 0 : pr + 1;
            EvalBand Bn = B;
            if (has_next && next_band) {
                const EvalBand* nb = bands + X.band0 + bi + 1;
                Bn.m0 = nb->m0; Bn.npix = nb->npix; Bn.y0 = nb->y0; Bn.nrows = nb->nrows;
            }
            const int Un = Bn.nrows * kPairFPI;
            AMT_PTICK(0);
            // ---- 1. request the next iteration's raw rows ----
#ifndef AMT_PAIR_NO_STAGE
            if (has_next) {
#else
            if (false) {
#endif
                const __amdgpu_buffer_rsrc_t rs0 = frame_rsrc(npr * kPairFPI), rs1 = frame_rsrc(npr * kPairFPI + 1);
#pragma unroll
                for (int k = 0; k < kMaxUnits; ++k) {
                    const int u = wave + kPairEvalWaves * k;
                    if (u >= Un) break;
                    const int fr = u >= Bn.nrows ? 1 : 0, r = u - fr * Bn.nrows;
                    st.request(fr ? rs1 : rs0, Bn.y0 + r, nplane + fr * plane_cap + r * lp);
                }
            }
            AMT_PTICK(1);
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
                M = window_means(W);
                R[fr] = window_corr_exact(Kp, W, M);
#endif
                // (issuing the two scale gathers right after the means, ahead of the correlation's 75 packed ops, measured slower:
                //  5.30 vs 5.16 ms per 10 000 frames)
#ifdef AMT_PAIR_NO_GATHER
                sc0[fr] = f2{1e-4f * (float)score_bin_dev(M.x), 0.5f}; sc1[fr] = f2{1e-4f * (float)score_bin_dev(M.y), 0.5f};
#else
                sc0[fr] = gld<f2>(gScales, __umul24((unsigned)score_bin_dev(M.x), cpad8) + m8);
                sc1[fr] = gld<f2>(gScales, __umul24((unsigned)score_bin_dev(M.y), cpad8) + m8);
#endif
            }
#ifdef AMT_PAIR_TIMING
            if (R[0].x == 123456.0f && R[1].y == 123456.0f) tacc[7] += 1;
#endif
            AMT_PTICK(2);
            // ---- 3. per-pixel terms (LogoScan.hpp:305-308) -> the score rows ----
            const bool act_now = act;
#pragma unroll
            for (int fr = 0; fr < kPairFPI; ++fr) {
                if (act_now) {
                    myrows[(fr * 2 + 0) * kPairRowPitch] = score_term(R[fr].x, sc0[fr].x, sc0[fr].y);
                    myrows[(fr * 2 + 1) * kPairRowPitch] = score_term(R[fr].y, sc1[fr].x, sc1[fr].y);
                }
            }
            AMT_PTICK(3);
            // ---- 4. the next iteration's rows: raw -> {s, bg} in place (holding the terms back until after the conversion so that it
            //      covers the scale gathers' trip was tried: 36 registers spilled, 5.4 -> 8.2 ms per 10 000 frames) ----
#ifdef AMT_PAIR_NO_STAGE
            if (false) {
#else
            if (has_next) {
#endif
                f4 av[kMaxUnits], bmv[kMaxUnits];
                if (next_band) {
#pragma unroll
                    for (int k = 0; k < kMaxUnits; ++k) {
                        const int u = wave + kPairEvalWaves * k;
                        if (u >= Un) break;
                        const int fr = u >= Bn.nrows ? 1 : 0, r = u - fr * Bn.nrows;
                        st.load_ab(rA, rB, Bn.y0 + r, av[k], bmv[k]);     // once per band, from memory
                    }
                    // the band's last evaluation is done: the next band's pixel and taps (14 loads, issued after everything the
                    // conversion needs) travel while the rows are converted
                    asm volatile("" ::: "memory");
                    load_pixel(Bn, act, woff, m8, Kp);
                    asm volatile("" ::: "memory");
                }
                // the LDS-direct loads were issued before everything else and vector-memory loads return in order: what may stay
                // outstanding here are the next band's 14 pixel / tap loads (when one starts), never the raw rows or the coefficients
                if (next_band) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                AMT_PTICK(4);
#pragma unroll
                for (int k = 0; k < kMaxUnits; ++k) {
                    const int u = wave + kPairEvalWaves * k;
                    if (u >= Un) break;
                    const int fr = u >= Bn.nrows ? 1 : 0, r = u - fr * Bn.nrows;
                    if (next_band) { if (fr == 0) st.ab_to_lds(abp + r * lp, av[k], bmv[k]); }
                    else st.ab_from_lds(abp + r * lp, av[k], bmv[k]);
                    st.convert(nplane + fr * plane_cap + r * lp, Bn.y0 + r, av[k], bmv[k]);
                }
            }
            AMT_PTICK(5);
            __syncthreads();                     // next planes and this iteration's score rows complete; current planes consumed
            AMT_PTICK(6);
            if (next_band) { B.m0 = Bn.m0; B.npix = Bn.npix; B.y0 = Bn.y0; B.nrows = Bn.nrows; ++bi; }
            pr = npr;
        }
        __syncthreads();                         // the summing wave's last rows
    }
#ifdef AMT_PAIR_TIMING
    if (lane == 0 && blockIdx.x == gridDim.x / 2 && (wave == 0 || wave == 3 || wave == 7 || wave == 8)) {
        long long* tb = reinterpret_cast<long long*>(out + (long long)nframes * out_frame_stride);   // host reserves room
        const int slot = wave == 0 ? 0 : (wave == 3 ? 1 : (wave == 7 ? 2 : 3));
        for (int k = 0; k < 8; ++k) tb[slot * 8 + k] = tacc[k];
    }
#endif
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
