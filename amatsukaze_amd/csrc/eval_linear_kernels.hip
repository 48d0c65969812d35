// eval_linear_kernels.hip -- many-fade logo evaluation through the linearity of the correlation (decision-guarded mode).
//
// AMTAnalyzeLogo evaluates 11 blends  W_f = f*bg + (1-f)*s  of every frame (LogoScan.hpp:1151-1155).  CalcCorrelation5x5
// (ComputeKernel.cpp:77-121) is linear in the window: in real arithmetic  mean(W_f) = f*mean(bg) + (1-f)*mean(s)  and
// corr(k, W_f) = f*corr(k, bg) + (1-f)*corr(k, s).  This kernel evaluates the window of s and of bg ONCE per mask pixel and
// forms all fades from the two (sum, mean) pairs; what is not linear -- the 8-level bin select, the scale / clamp
// (LogoScan.hpp:302-308) -- is applied per fade as the reference does.  Results differ from the reference's fp32
// evaluation order by rounding only (bounded by EvalEngine::linear_error_bound(), ~1e-6 typical), which is inside the 1e-4
// the north star allows for the float scores; the INTEGER decisions taken from them are protected separately:
//   * the bin select is discontinuous: when the interpolated mean lies within `bin_eps` (the bound on its distance from the exactly
//     evaluated mean) of a bin edge, that fade's mean is re-computed in the reference's exact order (blend, column sums, hsum, /25)
//     and the bin comes from the exact value;
//   * argmin over fades (CalcFade2, :1288-1314): analysis_mark_kernel lists every frame whose best / second-best margin is
//     below twice the error bound, and the exact kernel (eval_fused_kernels.hip) re-evaluates just those frames.
//
// Shape (eval_tiles.hpp, eval_tile_stage.h): workgroup (256 threads) = (logo, G frames); its four waves share out the logo's
// TILES (64 mask pixels and the bounding box of their windows) round-robin and never meet before the end: a wave stages its
// tile for one frame into its own LDS plane as {s, bg} pairs -- raw samples prefetched into registers an iteration ahead --
// evaluates the window mean and the correlation of BOTH operands with the same packed fp32 instructions (the 25 taps broadcast
// to both halves), forms the fades, adds the terms over each quad of lanes with two DPP adds and keeps the quads' sums per (frame,
// fade) in LDS cells that only it touches.  The summation order of this mode is free (it is not the reference's; the error bound
// covers any order) but fixed: quads by DPP, tiles in turn, the 16 quads and the four waves in order at the end -- deterministic.
// No barrier in the loop.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <algorithm>

#include "eval_tile_stage.h"

namespace amt {

using namespace lin;
using namespace tile;

#ifndef AMT_LIN_WAVES
#define AMT_LIN_WAVES 4
#endif
constexpr int kLinWaves = AMT_LIN_WAVES;     // waves per workgroup: one per SIMD
constexpr int kLinWgThreads = kLinWaves * 64;
// A wave's running sums of one frame: 48 bytes per quad of lanes (twelve floats: the eleven fades' sums over the quad's mask pixels),
// i.e. 16 partial sums per fade that are added up at the very end.  (Tried and not kept: the same sums on the matrix pipe,
// v_mfma_f32_16x16x4_f32 with row selectors -- tools/ubench/mfma_rowsum.hip, profiles/r03_notes.md.)
constexpr int kLinAccFrameBytes = 16 * 48;

// mean of the blended window exactly as EvaluateLogo + CalcCorrelation5x5_AVX produce it (LogoScan.hpp:244-251, ComputeKernel.cpp:88-98):
// the uncommon path of the bin select.  Two LDS round trips (rows 0-2, rows 3-4) instead of one per row: what this path costs
// is mostly the latency of its reads.
__device__ __forceinline__ float exact_blend_mean_2trips(const unsigned (&wrow)[5], float fade)
{
    typedef const __attribute__((address_space(3))) f2* lds_pair;
    f2 e[3][5];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 5; ++i) e[r][i] = ((lds_pair)(unsigned long long)wrow[r])[i];
    float t01[5], v2[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        t01[i] = fade_mix(fade, e[0][i].y, e[0][i].x) + fade_mix(fade, e[1][i].y, e[1][i].x);
        v2[i] = fade_mix(fade, e[2][i].y, e[2][i].x);
    }
    __builtin_amdgcn_sched_barrier(0);                             // (the second round of reads goes into the registers of the first)
    f2 g[2][5];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 5; ++i) g[r][i] = ((lds_pair)(unsigned long long)wrow[3 + r])[i];
    float c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = (t01[i] + (v2[i] + fade_mix(fade, g[0][i].y, g[0][i].x))) + fade_mix(fade, g[1][i].y, g[1][i].x);
    return div25(hsum5(c[0], c[1], c[2], c[3], c[4]));
}

struct LinLaunch {
    const EvalLogoDev* logos;
    const TileLogoDev* tls;
    const float* fades;
    const void* Y;
    const int* frame_map;
    long long frame_stride;      // elements
    int pitch;                   // elements
    float maxv;
    int nfades, fade0;
    int nframes, G, ngroups;
    float* out;
    int out_frame_stride, take_abs;
    float bin_eps;      // bound on |interpolated mean - exactly evaluated mean|, fixed-point rounding included (gray levels)
    int qlog2;          // means are compared with the bin edges in units of 2^-qlog2 gray levels (v * 2^qlog2 < 2^31)
};

#ifndef AMT_LIN_OCC
#define AMT_LIN_OCC 4
#endif
#ifndef AMT_LIN_OCC16
#define AMT_LIN_OCC16 4
#endif
// NF fades (11 for AMTAnalyzeLogo, the only caller of this mode: no per-fade branches)
template <typename pix_t, int NF>
__device__ __forceinline__ void logo_eval_linear_body(const LinLaunch& A)
{
    static_assert(NF >= 3 && NF <= 12, "the running sums hold rows 0..11 of the 16x16 accumulator");
    extern __shared__ float lds[];
    f2* const planes = reinterpret_cast<f2*>(lds);                 // [kLinWaves][kTileCap] a wave's own tile: {s, bg}
    float* const wacc = lds + kLinWaves * kTileCap * 2;            // [kLinWaves][G][48 lanes][4] a wave's running sums (kLinAccFrameBytes per frame)

    const int G = A.G;
    const int logo = blockIdx.x / A.ngroups;
    const int grp = blockIdx.x - logo * A.ngroups;
    const int F0 = grp * G;
    const int gcount = min(G, A.nframes - F0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const EvalLogoDev* const Lp = A.logos + logo;
    const TileLogoDev* const Xp = A.tls + logo;
    const int ntl = Xp->ntlist;                                    // tiles that hold pixels
    const const_int_ptr tlist = (const_int_ptr)Xp->tlist;
    const gptr_t gPq = (gptr_t)Xp->pq;
    const float floorResp = Xp->floorResp;                         // (wave-uniform) L: scale2 = min(1, r / L)
    const const_tile_ptr tiles = (const_tile_ptr)Xp->tiles;
    // Bin edges in fixed point.  Q = 2^qlog2 units per gray level; a bin is 8 gray levels = 2^(qlog2+3) units = one unit of y below:
    // y = (mean * Q + dq) / 2^(qlog2+3) with dq = e + 1, e = ceil(bin_eps * Q).  floor(y) is the bin of mean + dq / Q; an edge within bin_eps
    // ABOVE the mean has been crossed by y (fract(y) * 2^(qlog2+3) in [0, dq]), one within bin_eps BELOW it leaves fract(y) * 2^(qlog2+3)
    // in [dq - 1, dq + e + 1): so "fract(y) < ywin = (dq + e + 1) / 2^(qlog2+3)" flags every (pixel, fade) whose exact mean might fall in
    // another bin, and for all others floor(y) is the bin of the exact mean.  Scaling by powers of two is exact, so y comes straight out
    // of the multiply-add that interpolates the mean (the scale sits in its operands).
    const float qscale = __builtin_amdgcn_ldexpf(1.0f, A.qlog2);
    const int qe = (int)__builtin_ceilf(A.bin_eps * qscale);
    const int dq = qe + 1;
    const float yscale = __builtin_amdgcn_ldexpf(1.0f, -3);                          // mean (gray levels) -> bins
    const float ydq = __builtin_amdgcn_ldexpf((float)dq, -(A.qlog2 + 3));
    const float ywin = __builtin_amdgcn_ldexpf((float)(dq + qe + 1), -(A.qlog2 + 3));

#ifdef AMT_LIN_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#define AMT_LTICK(k) do { const long long t_ = clock64(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define AMT_LTICK(k) do { } while (0)
#endif
    // the fades, wave-uniform (scalar registers)
    float fd[NF];
    {
        typedef const __attribute__((address_space(4))) float* const_float_ptr;
        const const_float_ptr fp = (const_float_ptr)(A.fades + A.fade0);
#pragma unroll
        for (int f = 0; f < NF; ++f) fd[f] = fp[f];
    }
    float* const myacc = wacc + wave * G * (kLinAccFrameBytes / 4);
    for (int i = lane; i < G * (kLinAccFrameBytes / 4); i += 64) myacc[i] = 0.0f;
    const unsigned myacc_base = __builtin_amdgcn_readfirstlane(lds_address(myacc));
    // the wave's own copy of the fades, lane f <-> fade f (read back by the bin fix-up; a wave's LDS operations complete in order)
    float* const myfades = wacc + kLinWaves * G * (kLinAccFrameBytes / 4) + wave * 16;
    if (lane < 16) myfades[lane] = A.fades[A.fade0 + min(lane, NF - 1)];
    const unsigned myfades_base = __builtin_amdgcn_readfirstlane(lds_address(myfades));

    f2* const myplane = planes + wave * kTileCap;
    const unsigned plane_base = lds_address(myplane);
    TileStager<pix_t, false, true> st;
    st.init(Lp, A.pitch, A.maxv, myplane, nullptr);
    TilePixel px;
    f2 PQ;                                                       // this lane's pixel: its response on flat level c is |PQ.x + PQ.y * c|
    TileDesc T;

    // this wave's tiles: entries wave, wave + 4, ... of the logo's list of tiles.  (i, g) = (list position, frame) of an iteration
    auto advance = [&](int& i, int& g) {
        const bool last = g + 1 == gcount;
        g = last ? 0 : g + 1;
        i = last ? i + kLinWaves : i;
    };
    // Pipeline: while iteration i is evaluated, the raw samples of i + 1 sit in registers and those of i + 2 travel.
    //   A. window reads of i from the plane, means and correlations of s and bg
    //   B. the fades' bins (a mean next to a bin edge: the reference's exact mean decides, rare)
    //   T. the 11 terms, their sums over the quads of lanes, added to the wave's running sums of the frame in LDS.  No table is looked
    //      up: the scale of a term is a function of the bin number and two per-pixel constants (see "T." below)
    //   C. raw(i + 1) -> plane; request raw(i + 2)   (the pixel / taps of i + 1, if its tile is a new one, are requested first)
    int i0 = wave, g0 = 0;                                       // iteration i
    int i1 = i0, g1 = 0;                                         // iteration i + 1 (its raw samples are in the registers)
    int iu = -1;                                                 // the list position whose tile the staging units describe
    PQ = f2{0.0f, 0.0f};
    if (i0 < ntl) {
        const int t0 = tlist[i0];
        fetch_tile(T, tiles + t0);
        st.setup_units(T, lane);
        iu = i0;
        st.request(frame_rsrc<pix_t>(A.Y, A.frame_map, A.frame_stride, F0));
        asm volatile("" ::: "memory");                             // (see eval_pair_kernels.hip: the raw loads stay ahead of the tap loads)
        px.load(Xp, (unsigned)t0 * 64u + (unsigned)lane, T, plane_base);
        PQ = gld<f2>(gPq, px.slot8);
        st.convert();
        // (the first taps have arrived before the loop: the wait the compiler places at the loop head is the merge of this path and
        //  the back edge, and on the back edge the taps of a new tile are the OLDEST loads in flight -- see step B')
#pragma unroll
        for (int j = 0; j < 13; ++j) asm volatile("" : "+v"(px.Kp[j]));
        advance(i1, g1);
        if (i1 < ntl) {
            if (i1 != iu) { fetch_tile(T, tiles + tlist[i1]); st.setup_units(T, lane); iu = i1; }
            st.request(frame_rsrc<pix_t>(A.Y, A.frame_map, A.frame_stride, F0 + g1));
        }
    }
    while (i0 < ntl) {
        AMT_LTICK(0);
        // ---- A. ONE window evaluation for both operands: R = {corr(s), corr(bg)}, M = {mean(s), mean(bg)} ----
        // (the taps' {k, k} broadcasts live in the multiply-adds' op_sel: eval_tile_stage.h pk_fma_tap)
        f2 R, M;
        {
            unsigned wrow[5];
            px.rows(wrow);
#ifdef AMT_LIN_NO_EVAL                                          // (ablations of the instrumented builds: wrong results, timing only)
            R = px.Kp[0]; M = px.Kp[1] + f2{100.0f, 120.0f};
#else
            window_eval_streamed(wrow, px.Kp, M, R);         // idle lanes read the tile's first window: finite values, zero taps
#endif
        }
        AMT_LTICK(2);
        // ---- B. bins, as floats: y = position of the interpolated mean in bins (+ the window's half-width), floor(y) its bin, fract(y)
        //      its distance from the bin edge below.  The bin select is discontinuous (LogoScan.hpp:304): for a mean within bin_eps of an
        //      edge -- about 3e-4 of all (pixel, fade) pairs -- the mean is evaluated exactly as the reference does and ITS bin is taken ----
        const float y0 = M.x * yscale, y1 = M.y * yscale;           // (exact: powers of two)
        const float dy = y1 - y0;
        const float y0d = y0 + ydq;                                 // (the window's half-width rides in the multiply-add's addend: one rounding
                                                                    //  for the sum, one for the FMA -- the two the bound counted for FMA and add)
        float emin = 1.0f;
        float binf[NF];                                            // the fade's bin, 0..31
        // Fade 0 blends to s and fade 1 to bg exactly (0 * x + y == y), and M holds their means in the reference's own order (column
        // sums, hsum, /25: window_eval_streamed): those two bins are the reference's without any test.  (The caller guarantees that the
        // first fade is 0 and the last is 1.)  It matters: mean(s) is an integer / 25 and sits exactly ON a bin edge once in 200 pixels.
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const bool end = f == 0 || f == NF - 1;
            const float y = end ? (f == 0 ? y0 : y1) : __builtin_fmaf(fd[f], dy, y0d);
            if (!end) emin = __builtin_fminf(emin, __builtin_amdgcn_fractf(y));
            binf[f] = __builtin_amdgcn_fmed3f(__builtin_floorf(y), 0.0f, 31.0f);       // (int)clamp(mean, 0, 255) >> 3; a NaN mean gives bin 0 like the reference's (int)NaN = INT_MIN
        }
        // (uncommon -- one wave iteration in five has such a pixel -- and kept small in code and registers: a rolled loop; unrolled, the
        //  inlined window re-reads cost the whole kernel its occupancy.  The fades come out of a vector register by v_readlane,
        //  filled from the wave's copy in LDS: a scalar load per fade would put memory latencies in a row.)
#ifdef AMT_LIN_NO_FIXUP
        const bool near_edge = false;
#else
        const bool near_edge = px.act && emin < ywin;
#endif
        if (__builtin_amdgcn_ballot_w64(near_edge) != 0) {         // wave-uniform: every lane reads the fades (v_readlane needs lanes 0..10)
            unsigned wrow[5];
            px.rows(wrow);
            int lane_here = lane;                                  // (opaque: hoisted out of the loop this address would be spilled, and its
            asm volatile("" : "+v"(lane_here));                    //  reload waits for every load in flight)
            const float fadev = *(const __attribute__((address_space(3))) float*)(unsigned long long)(myfades_base + (unsigned)(lane_here & 15) * 4u);
#pragma unroll 1
            for (int f = 1; f < NF - 1; ++f) {
                const float fade = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fadev), f));
                if (near_edge && __builtin_amdgcn_fractf(__builtin_fmaf(fade, dy, y0d)) < ywin) {
                    const float nb = (float)score_bin_dev(exact_blend_mean_2trips(wrow, fade));
#pragma unroll
                    for (int ff = 0; ff < NF; ++ff) binf[ff] = ff == f ? nb : binf[ff];
                }
            }
        }
        AMT_LTICK(3);
        // ---- B'. a new tile next: its pixel and taps are requested BEFORE the raw samples -- vector-memory loads return in order, and
        //      the taps are what the next iteration needs first ----
        f2 PQn = PQ;
        if (i1 < ntl && i1 != i0) {
            const int t1 = tlist[i1];
            TileDesc Tn;
            fetch_tile(Tn, tiles + t1);
            px.load(Xp, (unsigned)t1 * 64u + (unsigned)lane, Tn, plane_base);
            PQn = gld<f2>(gPq, px.slot8);                           // (this iteration's terms below still need the CURRENT pixel's response line)
        }
        // ---- T. the terms.  The reference looks {scale, scale2} = {1 / r, min(1, r / L)} up by bin, r = |response of the pixel's kernel on
        //      that flat level| (LogoScan.hpp:190-207), and forms clamp(x * scale, -1, 1) * scale2 (:305-308) = clamp(x, -r, r) / max(r, L).
        //      r is |P + Q bin| up to the rounding of the table's fp32 evaluation (the composite of a flat level is affine in the level):
        //      one multiply-add instead of an 8-byte gather per fade; what the difference can do to a term is computed per (pixel, bin)
        //      on the host and is part of the error bound (eval_engine.hip ensure_linear).  Then the sum over each quad of lanes by two
        //      DPP steps; lane 0 of the quad adds the sums to the quad's running sums of the frame in LDS (48 bytes per quad: 16 partial
        //      sums per fade, added up at the very end).  A lane's cells are its own and LDS operations of a wave complete in order: no
        //      barrier, no atomics. ----
#ifndef AMT_LIN_NO_FLUSH
        {
            typedef __attribute__((address_space(3))) f4* lds_quad;
            const lds_quad cell = (lds_quad)(unsigned long long)(myacc_base + (unsigned)g0 * (unsigned)kLinAccFrameBytes + (unsigned)(lane >> 2) * 48u);
            const float R0 = R.x, dR = R.y - R.x;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float term[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 4 * q + r;
                    if (f < NF) {
                        const float resp = __builtin_fabsf(__builtin_fmaf(PQ.y, binf[f], PQ.x));
                        const float x = __builtin_fmaf(fd[f], dR, R0);
                        float t = __builtin_amdgcn_fmed3f(x, -resp, resp) * __builtin_amdgcn_rcpf(__builtin_fmaxf(resp, floorResp));
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));   // quad_perm:[1,0,3,2]
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true));   // quad_perm:[2,3,0,1]
                        asm volatile("" : "+v"(t));       // (left alone the add sinks into the lane-0 branch below and its DPP operand stays a v_mov_b32_dpp)
                        term[r] = t;
                    } else term[r] = 0.0f;
                }
                if ((lane & 3) == 0) {
                    const f4 o = cell[q];
                    cell[q] = f4{o[0] + term[0], o[1] + term[1], o[2] + term[2], o[3] + term[3]};
                }
            }
        }
#endif
        AMT_LTICK(4);
        PQ = PQn;
        AMT_LTICK(5);
        // ---- C. the next iteration's tile into the plane, its pixel if the tile changes, the raw samples of the one after ----
        if (i1 < ntl) {
#ifndef AMT_LIN_NO_CONVERT
            st.convert();
#endif
            int i2 = i1, g2 = g1;
            advance(i2, g2);
            if (i2 < ntl) {
                if (i2 != iu) { fetch_tile(T, tiles + tlist[i2]); st.setup_units(T, lane); iu = i2; }
#if defined(AMT_LIN_RAW_SAMEFRAME)                              // (ablation: every request hits the cache -- what the raw loads' LATENCY costs)
                st.request(frame_rsrc<pix_t>(A.Y, A.frame_map, A.frame_stride, 0));
#elif !defined(AMT_LIN_NO_RAW)                                    // (ablation: the samples of the first two requests are converted over and over)
                st.request(frame_rsrc<pix_t>(A.Y, A.frame_map, A.frame_stride, F0 + g2));
#endif
            }
        }
        AMT_LTICK(1);
        i0 = i1; g0 = g1;
        advance(i1, g1);
    }
    __syncthreads();
#ifdef AMT_LIN_TIMING
    if (lane == 0 && blockIdx.x == gridDim.x / 6 && wave < 4) {      // a workgroup of logo 0 (the deint logo)
        long long* tb = reinterpret_cast<long long*>(A.out + (long long)A.nframes * A.out_frame_stride);      // host reserves room
        for (int k = 0; k < 8; ++k) tb[wave * 8 + k] = tacc[k];
    }
#endif
    // the waves' sums, in order: per wave the 16 partial sums of a fade, front to back
    // (the thread index is re-derived: kept across the loop it would be one register too many)
    const int tid_end = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (tid_end < gcount * NF) {
        const int gg = tid_end / NF, f = tid_end - gg * NF;
        float r = 0.0f;
        for (int q = 0; q < kLinWaves; ++q) {
            const float* const cells = wacc + (q * G + gg) * (kLinAccFrameBytes / 4) + f;      // quad j: 12 floats at 12 j
            float rq = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) rq += cells[j * 12];
            r += rq;
        }
        r = r / Lp->blackScore;
        if (A.take_abs) r = fabsf(r);
        A.out[(long long)(F0 + gg) * A.out_frame_stride + Lp->out_off + A.fade0 + f] = r;
    }
}

// Four waves per SIMD (<= 128 registers) for both sample sizes.
// (Inside the loop a spilled register would be reloaded with a wait for EVERY load in flight -- the pipeline's whole point.)
__global__ __launch_bounds__(kLinWgThreads) __attribute__((amdgpu_waves_per_eu(AMT_LIN_OCC, AMT_LIN_OCC)))
void logo_eval_linear_kernel(const LinLaunch A) { logo_eval_linear_body<uint8_t, 11>(A); }
__global__ __launch_bounds__(kLinWgThreads) __attribute__((amdgpu_waves_per_eu(AMT_LIN_OCC16, AMT_LIN_OCC16)))
void logo_eval_linear_kernel16(const LinLaunch A) { logo_eval_linear_body<uint16_t, 11>(A); }

hipError_t launch_logo_eval_linear(hipStream_t st, int bits, const EvalLogoDev* dlogos, const TileLogoDev* dtls, int nlogos,
                                   const float* dfades, int nfades, int fade0, const void* dY,
                                   const int* dframe_map, long long frame_stride_elems, int pitch, int nframes, int G, float* dout,
                                   int out_frame_stride, int take_abs, float bin_eps, int qlog2)
{
    // (the caller guarantees dfades[fade0] == 0 and dfades[fade0 + nfades - 1] == 1: EvalEngine::run_linear)
    if (nframes <= 0 || nlogos <= 0 || nfades <= 0) return hipSuccess;
    if (qlog2 < 4 || qlog2 > 24 || !(bin_eps >= 0.0f) || bin_eps * (float)(1 << qlog2) > 1048576.0f) return hipErrorInvalidValue;
    if (nfades != 11 || G < 1 || G > (bits > 8 ? kLinMaxFrames16 : kLinMaxFrames) || G * nfades > kLinWgThreads) return hipErrorInvalidValue;       // (AMTAnalyzeLogo's fades; anything else keeps the exact kernel)
    LinLaunch A;
    A.logos = dlogos; A.tls = dtls; A.fades = dfades; A.Y = dY; A.frame_map = dframe_map; A.frame_stride = frame_stride_elems; A.pitch = pitch;
    A.maxv = (float)((1 << bits) - 1);
    A.nfades = nfades; A.fade0 = fade0;
    A.nframes = nframes; A.G = G; A.ngroups = (nframes + G - 1) / G;
    A.out = dout; A.out_frame_stride = out_frame_stride; A.take_abs = take_abs; A.bin_eps = bin_eps; A.qlog2 = qlog2;
    const size_t lds = (size_t)kLinWaves * kTileCap * 2 * sizeof(float) + (size_t)kLinWaves * G * kLinAccFrameBytes + (size_t)kLinWaves * 16 * sizeof(float);
    if (lds * (bits > 8 ? AMT_LIN_OCC16 : AMT_LIN_OCC) > 160 * 1024) return hipErrorInvalidValue;      // (the workgroups that share a CU must fit its LDS)
    dim3 grid((unsigned)((long long)A.ngroups * nlogos));
    if (bits <= 8) hipLaunchKernelGGL(logo_eval_linear_kernel, grid, dim3(kLinWgThreads), lds, st, A);
    else hipLaunchKernelGGL(logo_eval_linear_kernel16, grid, dim3(kLinWgThreads), lds, st, A);
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
                          int* __restrict__ list, int* __restrict__ count, const uint8_t* __restrict__ force)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nframes) return;
    bool amb = force != nullptr && force[n] != 0;          // frames the error bound does not cover (out-of-range samples): always exact
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
                                int* dlist, int* dcount, const uint8_t* dforce)
{
    if (nframes <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(dcount, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(analysis_mark_kernel, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, st, drec, stride, nframes, ngroups, nfades,
                       eps3[0], eps3[1], eps3[2], dlist, dcount, dforce);
    return hipGetLastError();
}

// The linear mode's error bound (and with it the margin of the bin test and of the decision guard) assumes every sample <= maxv.  A 10-
// or 12-bit clip lives in 16-bit containers and may carry larger values (the erase path clamps them, LogoScan.hpp:1258; the analysis does
// not): one workgroup per frame looks at the logo rectangle -- all the analysis reads, 64 KB at the bench shape -- and flags the frame
// when a sample exceeds maxv; analysis_mark_kernel then hands it to the exact kernel whatever its scores say.  Not launched at 8 and
// 16 bits, where every container value is in range.
__global__ __launch_bounds__(256)
void rect_range_flag_kernel(const uint16_t* __restrict__ Y, long long frame_stride, int pitch, int imgx, int imgy, int w, int h, unsigned maxv,
                            uint8_t* __restrict__ flag)
{
    const uint16_t* p = Y + (long long)blockIdx.x * frame_stride + (long long)imgy * pitch + imgx;
    unsigned m = 0;
    // rectangle origin and width are even (LogoScan.hpp:69): two samples per 32-bit load
    const int pairs = w >> 1;
    for (int i = threadIdx.x; i < pairs * h; i += 256) {
        const int y = i / pairs, x = i - y * pairs;
        const unsigned v = *reinterpret_cast<const unsigned __attribute__((aligned(2)))*>(p + (long long)y * pitch + 2 * x);
        m = max(m, max(v & 0xFFFFu, v >> 16));
    }
    if (w & 1)
        for (int y = threadIdx.x; y < h; y += 256) m = max(m, (unsigned)p[(long long)y * pitch + w - 1]);
    const int any = __syncthreads_or(m > maxv);
    if (threadIdx.x == 0) flag[blockIdx.x] = any ? 1 : 0;
}

hipError_t launch_rect_range_flag(hipStream_t st, const void* dY, long long frame_stride_elems, int pitch, int imgx, int imgy, int w, int h, int bits,
                                  int nframes, uint8_t* dflag)
{
    if (nframes <= 0) return hipSuccess;
    hipLaunchKernelGGL(rect_range_flag_kernel, dim3((unsigned)nframes), dim3(256), 0, st, (const uint16_t*)dY, frame_stride_elems, pitch, imgx, imgy, w, h,
                       (unsigned)((1 << bits) - 1), dflag);
    return hipGetLastError();
}

} // namespace amt
