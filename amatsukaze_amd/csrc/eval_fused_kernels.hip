// eval_fused_kernels.hip -- logo correlation on CDNA4 (gfx950), fused with the reference-order sum.
//
// Replaces LogoDataParam::EvaluateLogo / CorrelationScore (LogoScan.hpp:231-255, 288-318) + DeintY / CopyY
// (:763-790) as driven by LogoFrame::ScanFrame (:1543-1568), AMTAnalyzeLogo::GetFrameT (:1119-1161) and
// LogoAnalyzer::ReMakeLogo (:955-982): out[frame][logo][fade] = (|.|) CorrelationScore(blend(fade)) / blackScore.
//
// Shape of the work: every mask pixel m of an evaluation logo owns a private 25-tap kernel k_m and is evaluated on
// `nfades` blends  fade*bg + (1-fade)*s  of every frame; the per-pixel terms are then added in raster order
// (result += score, :295-315 -- a strictly sequential fp32 chain whose rounding the outputs inherit).  There is no
// operand reuse across mask pixels (not a GEMM, no MFMA): the kernel is bound by fp32 VALU issue, so the design
// minimises instructions per evaluation and keeps everything else off the inner loop:
//
//   * one workgroup = (logo, group of G frames); it walks the logo's BANDS (<= 256 run slots = <= 512 raster-
//     consecutive mask pixels and the <= 11 rectangle rows their windows touch) in order, because the sum is
//     sequential over all mask pixels of the logo;
//   * a thread owns one run slot: up to two horizontally adjacent mask pixels (97 % pair up along logo edges).  Its
//     2 x 25 kernel taps, the 5x6 source window S and the 5x6 background-estimate window BG stay in VGPRs for the
//     whole fade loop; the blend is recomputed in registers per fade, so the fade loop has NO LDS traffic and NO
//     barrier -- only packed fp32 math (v_pk_mul/add_f32: the two pixels of a slot ride in the two halves of the
//     64-bit operands; window columns are paired (1,2) (3,4) (0,5) so that every add/sub/mul of the reference's
//     evaluation order is a packed op without register shuffles) and one 8-byte scale gather per pixel, whose
//     latency is hidden behind the next fade's math;
//   * LDS holds the band's rows once per (band, frame): s and bg = a*s + b*maxv, each computed by one thread;
//   * per-pixel terms go to an LDS row per fade; one wave (rotating) adds each row front to back -- one lane per
//     fade, the reference's order -- while the workgroup's next global loads are in flight.  Nothing but the final
//     per-frame results touches HBM.
#include "build_knobs.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <algorithm>
#include <cstdlib>

#include "eval_plan.h"
#include "exact_math.h"
#include "eval_ordered_sum.h"

namespace amt {

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
// exact_math.h score_bin in 5 instructions: clamping to [0,255] BEFORE the truncation gives the same bin as cvttss2si + clamp
// for everything below 2^31 (negative and NaN -> 0: v_med3_f32 returns the minimum when an input is NaN); at and above 2^31
// cvttss2si yields INT_MIN, i.e. bin 0, not 31
__device__ __forceinline__ int score_bin_dev(float mean)
{
    const int bin = (int)__builtin_amdgcn_fmed3f(mean, 0.0f, 255.0f) >> 3;
    return mean >= 2147483648.0f ? 0 : bin;
}
// x / 25 for both halves (exact_math.h div25, packed)
__device__ __forceinline__ f2 div25_pk(f2 x)
{
    const f2 z = {0.04f, 0.04f};
    const f2 q = x * z;
    const f2 r = __builtin_elementwise_fma(f2{-25.0f, -25.0f}, q, x);
    return __builtin_elementwise_fma(r, z, q);
}

// four adjacent samples of a frame row in one load
template <typename pix_t> struct Raw4;
template <> struct Raw4<uint8_t> {
    unsigned v;
    __device__ __forceinline__ void load(gptr_t base, unsigned byteoff);
    __device__ __forceinline__ int get(int k) const { return (int)((v >> (8 * k)) & 0xFFu); }
};
template <> struct Raw4<uint16_t> {
    u2 v;
    __device__ __forceinline__ void load(gptr_t base, unsigned byteoff);
    __device__ __forceinline__ int get(int k) const { return (int)((v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu); }
};

__device__ __forceinline__ void Raw4<uint8_t>::load(gptr_t base, unsigned byteoff)
{
    typedef unsigned __attribute__((aligned(1))) ua_t;     // rectangle origins are even, not 4-byte aligned
    v = *reinterpret_cast<const __attribute__((address_space(1))) ua_t*>(base + byteoff);
}
__device__ __forceinline__ void Raw4<uint16_t>::load(gptr_t base, unsigned byteoff)
{
    typedef u2 __attribute__((aligned(2))) ua_t;
    v = *reinterpret_cast<const __attribute__((address_space(1))) ua_t*>(base + byteoff);
}

// AMT_FUSED_BG_LDS: the background-estimate window is NOT kept in registers across the fade loop; every fade reads it from the LDS
// plane straight into the registers its blend is formed in (15 two-dword reads).  30 VGPRs less -> three waves per SIMD instead of
// two; the price is LDS traffic inside the loop.
#ifndef AMT_FUSED_BG_LDS
#define AMT_FUSED_BG_LDS 0
#endif
constexpr int kFusedWaves = kEvalThreads / 64;
constexpr int kStageRows = 4;         // rows a staging wave handles per trip

// FPI = frames per iteration: a band-frame costs ~8-10 k cycles besides its fade loop (staging round trip, window reads,
// taps, two barriers) whatever the number of fades; passes with few fades (the 2-fade scan) stage and evaluate two frames
// per iteration -- one staging latency, one pair of barriers, FPI*nfades score rows summed by as many lanes.
template <typename pix_t, int FPI>
__device__ __forceinline__
void logo_eval_fused_body(const EvalLogoDev* __restrict__ logos, const EvalBand* __restrict__ bands,
                          const float* __restrict__ fades, int nfades, int fade0,
                          const pix_t* __restrict__ Y, const int* __restrict__ frame_map, long long frame_stride, int pitch,
                          float maxv, int nframes, int G, int ngroups, float* __restrict__ out, int out_frame_stride,
                          int take_abs, int plane_cap, int sc_pitch, int dbg_in, int bid, int scatter)
{
#ifdef AMT_EXPERIMENT
    const int dbg = dbg_in;      // timing ablations of instrumented builds only (amatsukaze_amd/build.py build_variant)
#else
    constexpr int dbg = 0;
    (void)dbg_in;
#endif
    extern __shared__ float lds[];
    float* const planes = lds;                       // [FPI][S: source pixels | BG: a*s + b*maxv][plane_cap]  the band's rows
    float* const sc = lds + 2 * FPI * plane_cap;     // [FPI][nfades][sc_pitch]  per-pixel terms of the current (band, frames)
    float* const accs = sc + FPI * nfades * sc_pitch;  // [G][nfades]  running sums

    const int logo = bid / ngroups;
    const int grp = bid - logo * ngroups;
    const int F0 = grp * G;
    const int gcount = min(G, nframes - F0);
    const EvalLogoDev L = logos[logo];
    const gptr_t gA = (gptr_t)L.a, gB = (gptr_t)L.b, gScales = (gptr_t)L.scales, gK = (gptr_t)L.kslot, gSlot = (gptr_t)L.slot2;
    const unsigned cpad = (unsigned)L.count_pad;
    const unsigned spad = (unsigned)L.nslots_pad;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: the staging rows' address math goes to the scalar unit
    const int w = L.w;
    constexpr unsigned ES = sizeof(pix_t);

    if (tid < G * nfades) accs[tid] = 0.0f;          // G * nfades <= 256 (host)
    const int fade_bits = __builtin_bit_cast(int, fades[fade0 + min(lane, nfades - 1)]);   // lane f holds fade f (nfades <= 64)

#ifdef AMT_FUSED_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#define AMT_TICK(k) do { const long long t_ = clock64(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define AMT_TICK(k) do { } while (0)
#endif
    int it = 0;                                      // (band, frame) iterations done
    int prev_npix = 0, prev_g = 0, prev_rows = 0;    // the iteration waiting to be summed: its band's pixels, first frame, score rows
    for (int bi = 0; bi < L.nbands; ++bi) {
        const EvalBand B = bands[L.band0 + bi];
        const int lp = B.lp, bx0 = B.x0, bx1 = B.x0 + B.bw;       // the band's LDS row pitch and the logo columns [bx0, bx1) it stages
        // ---- this thread's run slot: kernel taps as packed pairs, resident for all frames and fades of the band ----
        const bool act = tid < B.nslots;
        const unsigned slot = (unsigned)(B.s0 + (act ? tid : 0));
        const u2 sl = gld<u2>(gSlot, slot * 8u);
        const unsigned m0 = sl.x & 0x0FFFFFFFu;
        const int npx = act ? (int)(sl.x >> 28) : 0;
        const int woff = (int)sl.y;                  // LDS float offset of the window's top-left element
        const int p0 = (int)m0 - B.m0;               // index of the slot's first pixel in the band's score rows
        f2 K[25];
#pragma unroll
        for (int t = 0; t < 25; ++t) K[t] = gld<f2>(gK, ((unsigned)t * spad + slot) * 8u);
        const unsigned m1 = min(m0 + 1u, (unsigned)L.count - 1u);
#ifdef AMT_FUSED_TIMING
        if (K[24].x == 123456.0f) tacc[7] += 1;      // force the tap loads to land before the tick
#endif
        AMT_TICK(0);

        for (int g = 0; g < gcount; g += FPI, ++it) {
            const int nfr = min(FPI, gcount - g);
            // ---- 1. s and bg of the band's rows -> LDS.  A wave stages 4 consecutive rows x 256 columns at a time, a
            //      lane four adjacent columns (one 4*sizeof(pix_t) load per raw row, 16-byte loads of a and b, 16-byte
            //      LDS stores); the row is wave-uniform, so the address math is scalar, and the [1 2 1] vertical
            //      blend of DeintY re-uses the 6 raw rows it loads ----
            gptr_t src[FPI];
#pragma unroll
            for (int fr = 0; fr < FPI; ++fr) {
                const int frame = F0 + min(g + fr, gcount - 1);
                const int srcFrame = frame_map ? frame_map[frame] : frame;
                src[fr] = (gptr_t)(Y + (long long)srcFrame * frame_stride + (long long)(L.imgy + L.row0) * pitch + L.imgx);
            }
            // Three waves stage (4 rows each per trip); the fourth -- the one that adds the previous iteration's
            // score rows below -- stages nothing, so that the sum runs beside the staging loads' latency.
            const int sumwave = (it + kFusedWaves - 1) & (kFusedWaves - 1);
            const int worker = (wave - sumwave + kFusedWaves - 1) & (kFusedWaves - 1);       // 0..2 stage, 3 = sumwave
            for (int rg = worker * kStageRows; rg < ((dbg & 2) || worker == kFusedWaves - 1 ? 0 : B.nrows); rg += kStageRows * (kFusedWaves - 1)) {
                const int y = B.y0 + rg;                                     // logo row of the group's first row
                for (int xg = bx0; xg < bx1; xg += 256) {
                    const int x = xg + 4 * lane;
                    const int nv = min(4, bx1 - x);                          // valid columns of this lane (<= 0: none)
                    const int xl = nv >= 4 ? x : bx0;                        // lanes at a ragged right edge go column by column
                    Raw4<pix_t> raw[FPI][kStageRows + 2];
                    f4 av[kStageRows], bv[kStageRows];
                    // all frames' loads first (one round trip), then the arithmetic and the LDS stores
#pragma unroll
                    for (int fr = 0; fr < FPI; ++fr) {
                        if (L.deint) {
#pragma unroll
                            for (int j = 0; j < kStageRows + 2; ++j)
                                raw[fr][j].load(src[fr], (unsigned)(min(max(y - 1 + j, 0), L.h - 1) * pitch + xl) * ES);
                        } else {
#pragma unroll
                            for (int j = 0; j < kStageRows; ++j)
                                raw[fr][j + 1].load(src[fr], (unsigned)(min(y + j, L.h - 1) * L.row_step * pitch + xl) * ES);
                            raw[fr][0] = raw[fr][1]; raw[fr][kStageRows + 1] = raw[fr][kStageRows];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < kStageRows; ++j) {
                        const unsigned o = (unsigned)(min(y + j, L.h - 1) * w + xl) * 4u;
                        av[j] = gld<f4u>(gA, o);      // 8-byte aligned (w and x even), not 16
                        bv[j] = gld<f4u>(gB, o);
                    }
#pragma unroll
                    for (int fr = 0; fr < FPI; ++fr) {
                        if (fr >= nfr) break;
                        float* const planeS = planes + fr * 2 * plane_cap;
                        float* const planeB = planeS + plane_cap;
                        if (nv >= 4) {
#pragma unroll
                            for (int j = 0; j < kStageRows; ++j) {
                                const int yy = y + j;
                                if (rg + j < B.nrows) {
                                    const bool blend = L.deint && yy != 0 && yy != L.h - 1;
                                    f4 sv, gv;
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        const float s1 = blend ? (float)(raw[fr][j].get(k) + 2 * raw[fr][j + 1].get(k) + raw[fr][j + 2].get(k) + 2) / 4.0f
                                                               : (float)raw[fr][j + 1].get(k);
                                        sv[k] = s1;
                                        gv[k] = unblend_bg(av[j][k], bv[j][k], maxv, s1);
                                    }
                                    *reinterpret_cast<f4*>(planeS + (rg + j) * lp + (x - bx0)) = sv;
                                    *reinterpret_cast<f4*>(planeB + (rg + j) * lp + (x - bx0)) = gv;
                                }
                            }
                        } else if (nv > 0) {
                            for (int j = 0; j < kStageRows && rg + j < B.nrows; ++j) {
                                const int yy = y + j;
                                const bool blend = L.deint && yy != 0 && yy != L.h - 1;
                                for (int k = 0; k < nv; ++k) {
                                    int q0, q1, q2;
                                    if (L.deint) {
                                        q0 = gld<pix_t>(src[fr], (unsigned)(max(yy - 1, 0) * pitch + x + k) * ES);
                                        q1 = gld<pix_t>(src[fr], (unsigned)(yy * pitch + x + k) * ES);
                                        q2 = gld<pix_t>(src[fr], (unsigned)(min(yy + 1, L.h - 1) * pitch + x + k) * ES);
                                    } else {
                                        q0 = q2 = 0;
                                        q1 = gld<pix_t>(src[fr], (unsigned)(yy * L.row_step * pitch + x + k) * ES);
                                    }
                                    const float s1 = blend ? (float)(q0 + 2 * q1 + q2 + 2) / 4.0f : (float)q1;
                                    planeS[(rg + j) * lp + (x - bx0) + k] = s1;
                                    planeB[(rg + j) * lp + (x - bx0) + k] = unblend_bg(gld<float>(gA, (unsigned)(yy * w + x + k) * 4u),
                                                                               gld<float>(gB, (unsigned)(yy * w + x + k) * 4u), maxv, s1);
                                }
                            }
                        }
                    }
                }
            }
            AMT_TICK(1);
            // ---- 2. one wave adds the previous iteration's per-pixel terms in raster order (the others wait at B1, their
            //         SIMDs run other workgroups' fade loops meanwhile) ----
            if (!(dbg & 1) && it > 0 && wave == sumwave && lane < prev_rows) {
                float* a = accs + prev_g * nfades + lane;            // row fr*nfades + f belongs to frame prev_g + fr
                *a = ordered_row_sum(sc + lane * sc_pitch, prev_npix, *a);
            }
            AMT_TICK(2);
            __syncthreads();                         // B1: planes complete, score rows free again
            AMT_TICK(3);
#pragma unroll
            for (int fr = 0; fr < FPI; ++fr) {
            if (fr >= nfr) break;
            // ---- 4. windows -> registers: per row the column pairs (1,2) (3,4) (0,5) ----
            f2 S[15];
#if !AMT_FUSED_BG_LDS
            f2 BG[15];
#endif
            const float* const planeS = planes + fr * 2 * plane_cap;
            const float* const planeB = planeS + plane_cap;
            unsigned bgrow[5];                                   // LDS byte addresses of the five rows of the bg window
#pragma unroll
            for (int r = 0; r < 5; ++r)
                bgrow[r] = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)(planeB + woff + r * lp);
            if (act) {
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const float* ps = planeS + woff + r * lp;
                    S[3 * r + 0] = f2{ps[1], ps[2]};
                    S[3 * r + 1] = f2{ps[3], ps[4]};
                    S[3 * r + 2] = f2{ps[0], ps[5]};
#if !AMT_FUSED_BG_LDS
                    const float* pb = planeB + woff + r * lp;
                    BG[3 * r + 0] = f2{pb[1], pb[2]};
                    BG[3 * r + 1] = f2{pb[3], pb[4]};
                    BG[3 * r + 2] = f2{pb[0], pb[5]};
#endif
                }
            }
#ifdef AMT_FUSED_TIMING
            if (act && S[14].x == 123456.0f) tacc[7] += 1;
#endif
            AMT_TICK(4);
            // ---- 5. fade loop: registers only.  Two register sets alternate so that the scale gather issued by
            //      evaluation f is consumed after evaluation f+1's math (a full iteration of latency cover, no copies) ----
            if (act) {
                auto eval = [&](int f, f2& R, f2& s0, f2& s1) {
                    const float fade = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fade_bits, f));   // no memory op in the loop
                    const float omf = 1 - fade;
                    f2 W[15];
#if AMT_FUSED_BG_LDS
                    // the bg window lands in the registers its blend is formed in: column pairs (1,2) (3,4) (0,5) of every row
#pragma unroll
                    for (int r = 0; r < 5; ++r)
                        asm volatile("ds_read2_b32 %0, %3 offset0:1 offset1:2\n\tds_read2_b32 %1, %3 offset0:3 offset1:4\n\t"
                                     "ds_read2_b32 %2, %3 offset1:5"
                                     : "=&v"(W[3 * r]), "=&v"(W[3 * r + 1]), "=&v"(W[3 * r + 2]) : "v"(bgrow[r]) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7]), "+v"(W[8]), "+v"(W[9]),
                                   "+v"(W[10]), "+v"(W[11]), "+v"(W[12]), "+v"(W[13]), "+v"(W[14]) : : "memory");
#pragma unroll
                    for (int i = 0; i < 15; ++i) W[i] = W[i] * fade + S[i] * omf;       // fade*bg + (1-fade)*s
#else
#pragma unroll
                    for (int i = 0; i < 15; ++i) W[i] = BG[i] * fade + S[i] * omf;      // fade*bg + (1-fade)*s
#endif
                    // column sums ((r0+r1)+(r2+r3))+r4 for the column pairs
                    const f2 CA = ((W[0] + W[3]) + (W[6] + W[9])) + W[12];               // cols 1,2
                    const f2 CB = ((W[1] + W[4]) + (W[7] + W[10])) + W[13];              // cols 3,4
                    const f2 CE = ((W[2] + W[5]) + (W[8] + W[11])) + W[14];              // cols 0,5
                    f2 M;
                    M.x = hsum5(CE.x, CA.x, CA.y, CB.x, CB.y);                          // pixel 0: cols 0..4
                    M.y = hsum5(CA.x, CA.y, CB.x, CB.y, CE.y);                          // pixel 1: cols 1..5
                    M = div25_pk(M);
                    s0 = gld<f2>(gScales, (__umul24((unsigned)score_bin_dev(M.x), cpad) + m0) * 8u);   // bin < 32, count_pad < 2^24: full-rate multiply
                    s1 = gld<f2>(gScales, (__umul24((unsigned)score_bin_dev(M.y), cpad) + m1) * 8u);
                    f2 P[5];
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        f2 T[5];
#pragma unroll
                        for (int r = 0; r < 5; ++r) {
                            f2 wv;
                            if (c == 0) wv = bc_lo(W[3 * r + 0]);        // window col 1: tap col 1 of px 0, 0 of px 1
                            else if (c == 1) wv = bc_hi(W[3 * r + 0]);   // col 2
                            else if (c == 2) wv = bc_lo(W[3 * r + 1]);   // col 3
                            else if (c == 3) wv = bc_hi(W[3 * r + 1]);   // col 4
                            else wv = W[3 * r + 2];                      // (col 0 for px 0, col 5 for px 1)
                            T[r] = K[c * 5 + r] * (wv - M);
                        }
                        P[c] = ((T[0] + T[1]) + (T[2] + T[3])) + T[4];
                    }
                    R.x = hsum5(P[4].x, P[0].x, P[1].x, P[2].x, P[3].x);
                    R.y = hsum5(P[0].y, P[1].y, P[2].y, P[3].y, P[4].y);
                    __builtin_amdgcn_sched_barrier(0);       // what follows (an older gather's consumer) stays behind this math
                };
                auto consume = [&](int f, const f2& R, const f2& s0, const f2& s1) {
                    float* row = sc + (fr * nfades + f) * sc_pitch + p0;
                    row[0] = score_term(R.x, s0.x, s0.y);
                    if (npx > 1) row[1] = score_term(R.y, s1.x, s1.y);
                    __builtin_amdgcn_sched_barrier(0);
                };
                const int nfe = (dbg & 4) ? 0 : nfades;
                if (nfe > 0) {
                    f2 RA, a0, a1, RB, b0, b1;
                    eval(0, RA, a0, a1);
                    int f = 1;
                    for (; f + 1 < nfe; f += 2) {
                        eval(f, RB, b0, b1);
                        consume(f - 1, RA, a0, a1);
                        eval(f + 1, RA, a0, a1);
                        consume(f, RB, b0, b1);
                    }
                    if (f < nfe) {
                        eval(f, RB, b0, b1);
                        consume(f - 1, RA, a0, a1);
                        consume(f, RB, b0, b1);
                    } else {
                        consume(f - 1, RA, a0, a1);
                    }
                }
            }
            }
            prev_npix = B.npix; prev_g = g; prev_rows = nfr * nfades;
            AMT_TICK(5);
            __syncthreads();                         // B0: score rows complete, windows consumed
            AMT_TICK(6);
        }
    }
    // ---- last iteration's sum, then the results ----
    if (it > 0 && wave == ((it - 1) & (kFusedWaves - 1))) {
        if (lane < prev_rows) {
            float* a = accs + prev_g * nfades + lane;
            *a = ordered_row_sum(sc + lane * sc_pitch, prev_npix, *a);
        }
    }
    __syncthreads();
#ifdef AMT_FUSED_TIMING
    if (lane == 0 && blockIdx.x == gridDim.x / 2) {
        long long* tb = reinterpret_cast<long long*>(out + (long long)nframes * out_frame_stride);   // host reserves room
        for (int k = 0; k < 8; ++k) tb[wave * 8 + k] = tacc[k];
    }
#endif
    if (tid < gcount * nfades) {
        const int g = tid / nfades, f = tid - g * nfades;
        float r = accs[tid] / L.blackScore;
        if (take_abs) r = fabsf(r);
        const int row = scatter ? frame_map[F0 + g] : F0 + g;
        out[(long long)row * out_frame_stride + L.out_off + fade0 + f] = r;
    }
}

// <= 256 VGPRs: two waves per SIMD, nothing spilled (the fade loop alone holds ~200 live registers: taps 50, the two
// windows 60, their blend 30, two alternating result/gather sets).  Three waves per SIMD (<= 168) spills taps to
// scratch inside the fade loop and measured 2.6x slower.
#if AMT_FUSED_BG_LDS
#define AMT_FUSED_OCC 3
#else
#define AMT_FUSED_OCC 2
#endif
template <typename pix_t, int FPI>
__global__ __launch_bounds__(kEvalThreads) __attribute__((amdgpu_waves_per_eu(AMT_FUSED_OCC, AMT_FUSED_OCC)))
void logo_eval_fused_kernel(const EvalLogoDev* __restrict__ logos, const EvalBand* __restrict__ bands, const float* __restrict__ fades,
                            int nfades, int fade0, const pix_t* __restrict__ Y, const int* __restrict__ frame_map,
                            long long frame_stride, int pitch, float maxv, int nframes, int G, int ngroups, float* __restrict__ out,
                            int out_frame_stride, int take_abs, int plane_cap, int sc_pitch, int dbg, const int* __restrict__ nframes_dev,
                            int scatter, int nlogos, int fade_chunk)
{
    if (!nframes_dev) {
        logo_eval_fused_body<pix_t, FPI>(logos, bands, fades, nfades, fade0, Y, frame_map, frame_stride, pitch, maxv, nframes, G, ngroups, out,
                                         out_frame_stride, take_abs, plane_cap, sc_pitch, dbg, (int)blockIdx.x, scatter);
        return;
    }
    // listed re-evaluation (decision guard of the linear mode): the number of frames present sits on the device and is usually a
    // handful, so a small fixed grid walks the (logo, group) pairs that exist instead of launching the worst case
    // The fades are independent of each other (one ordered sum per fade): with fade_chunk > 0 a (logo, group) pair is shared out over
    // ceil(nfades / fade_chunk) workgroups, each walking the bands for its own fades only.
    const int present = min(*nframes_dev, nframes);
    const int groups = (present + G - 1) / G;
    const int nchunks = fade_chunk > 0 ? (nfades + fade_chunk - 1) / fade_chunk : 1;
    for (int b = (int)blockIdx.x; b < groups * nlogos * nchunks; b += (int)gridDim.x) {
        const int pair = b / nchunks, f0 = (b - pair * nchunks) * fade_chunk;
        const int nf = fade_chunk > 0 ? min(fade_chunk, nfades - f0) : nfades;
        logo_eval_fused_body<pix_t, FPI>(logos, bands, fades, nf, fade0 + f0, Y, frame_map, frame_stride, pitch, maxv, present, G, groups, out,
                                         out_frame_stride, take_abs, plane_cap, sc_pitch, dbg, pair, scatter);
        __syncthreads();
    }
}

static size_t fused_lds_bytes(int plane_cap, int nfades, int sc_pitch, int G, int fpi)
{
    return ((size_t)2 * fpi * plane_cap + (size_t)fpi * nfades * sc_pitch + (size_t)G * nfades + 2 * kSumChunk) * sizeof(float);
}

hipError_t launch_logo_eval_fused(hipStream_t st, int bits, const EvalLogoDev* dlogos, int nlogos, const EvalBand* dbands,
                                  const float* dfades, int nfades, int fade0, const void* dY, const int* dframe_map,
                                  long long frame_stride_elems, int pitch, int nframes, int G, float* dout, int out_frame_stride,
                                  int take_abs, int plane_cap, const int* dnframes, int scatter, int fade_chunk)
{
    if (nframes <= 0 || nlogos <= 0 || nfades <= 0) return hipSuccess;
    if (scatter && !dframe_map) return hipErrorInvalidValue;
    if (nfades > kEvalMaxFades || G * nfades > kEvalThreads || plane_cap > kEvalThreads * kEvalStage) return hipErrorInvalidValue;
    // Instrumented builds only (-DAMT_EXPERIMENT, build.py build_variant): AMTGPU_DBG bit 0 skips the sum, 1 the staging,
    // 2 the fade loop (timing ablations, WRONG results); AMTGPU_LDSPAD inflates the LDS request to lower the occupancy;
    // AMTGPU_FPI=1 forces one frame per iteration.  The release library reads no environment variable here.
#ifdef AMT_EXPERIMENT
    static const int dbg = std::getenv("AMTGPU_DBG") ? std::atoi(std::getenv("AMTGPU_DBG")) : 0;
    static const int ldspad = std::getenv("AMTGPU_LDSPAD") ? std::atoi(std::getenv("AMTGPU_LDSPAD")) : 0;
    static const int fpi_env = std::getenv("AMTGPU_FPI") ? std::atoi(std::getenv("AMTGPU_FPI")) : 0;
#else
    constexpr int dbg = 0, ldspad = 0, fpi_env = 0;
#endif
    const int ngroups = (nframes + G - 1) / G;
    const float maxv = (float)((1 << bits) - 1);
    const int sc_pitch = kEvalBandPixels + kEvalScorePad;
    // listed mode: a fixed small grid that strides over the (logo, group) pairs present
    if (fade_chunk < 0 || (fade_chunk > 0 && !dnframes)) return hipErrorInvalidValue;
    const int nchunks = fade_chunk > 0 ? (nfades + fade_chunk - 1) / fade_chunk : 1;
    dim3 grid(dnframes ? (unsigned)std::min<long long>((long long)ngroups * nlogos * nchunks, 1024) : (unsigned)((long long)ngroups * nlogos));
    // two frames per iteration while the planes and score rows of both fit half a CU's LDS (two workgroups per CU)
    int fpi = fpi_env > 0 ? std::min(2, fpi_env) : 2;
    if (G < 2 || fused_lds_bytes(plane_cap, nfades, sc_pitch, G, 2) > 80 * 1024 - 512) fpi = 1;
    const size_t lds = fused_lds_bytes(plane_cap, nfades, sc_pitch, G, fpi) + (size_t)ldspad;
#define AMT_LAUNCH(T, F)                                                                                                          \
    hipLaunchKernelGGL((logo_eval_fused_kernel<T, F>), grid, dim3(kEvalThreads), lds, st, dlogos, dbands, dfades, nfades, fade0,         \
                       (const T*)dY, dframe_map, frame_stride_elems, pitch, maxv, nframes, G, ngroups, dout, out_frame_stride, take_abs, \
                       plane_cap, sc_pitch, dbg, dnframes, scatter, nlogos, fade_chunk)
    if (fpi == 2) { if (bits <= 8) AMT_LAUNCH(uint8_t, 2); else AMT_LAUNCH(uint16_t, 2); }
    else { if (bits <= 8) AMT_LAUNCH(uint8_t, 1); else AMT_LAUNCH(uint16_t, 1); }
#undef AMT_LAUNCH
    return hipGetLastError();
}

} // namespace amt
