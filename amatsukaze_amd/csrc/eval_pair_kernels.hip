// eval_pair_kernels.hip -- the two-fade evaluation of LogoFrame::ScanFrame (LogoScan.hpp:1543-1568): corr0 = EvaluateLogo(fade 0)
// and corr1 = EvaluateLogo(fade 1) of every logo on every frame, in the reference's fp32 evaluation order (bit-exact records).
//
// With fade 0 the blended window  fade*bg + (1-fade)*s  (LogoScan.hpp:244-251) IS s, with fade 1 it IS bg = a*s + b*maxv
// (0*x + y == y for finite x; the host checks that every logo coefficient is finite and small enough for bg to stay below 2^30,
// and launches the generic kernel of eval_fused_kernels.hip otherwise).  So the two evaluations of a mask pixel are the SAME
// instruction stream on two operands: LDS holds {s, bg} pairs, a window element arrives as one 8-byte read, and every
// add / sub / mul of CalcCorrelation5x5_AVX's order (ComputeKernel.cpp:77-121, exact_math.h) is one packed fp32 instruction whose
// low half evaluates fade 0 and whose high half evaluates fade 1 -- no blend arithmetic, no FMA contraction
// (-ffp-contract=off), the 25 taps broadcast to both halves through op_sel.
//
// Shape (eval_tiles.hpp): workgroup = (logo, G <= 8 frames) = kTileWaves (11) evaluation waves + 1 summing wave, walking the logo's
// bands of <= 64 kTileWaves raster-consecutive mask pixels.  Within a band every evaluation wave owns a TILE: 64 of the band's mask pixels (the
// band sorted by column and dealt out 64 at a time) and the bounding box of their 5x5 windows.  The wave stages its tile for
// one frame per iteration into LDS nobody else touches -- raw samples prefetched into registers an iteration ahead, converted to
// {s, bg} with the tile's logo coefficients held in registers for the whole band -- and evaluates its pixels from it.  LDS
// operations of one wave complete in order, so nothing in an iteration needs a barrier.  The per-pixel terms go to an LDS row
// per (frame, fade) at the pixel's raster position; the waves meet ONCE PER BAND, and the last wave then adds the band's
// 2 G rows front to back -- one lane per row, the reference's order (`result += score`, LogoScan.hpp:295-315) -- while the
// others evaluate the next band into the second set of rows.
//
// The loop is bound by the NUMBER of vector instructions a wave issues (two to three waves per SIMD sustain one per ~6 cycles
// whatever their mix: profiles/r03_notes.md), so the code around the 101 packed operations of a window is kept short: raw
// bytes are blended two samples per instruction (16-bit lanes), a tile of at most 64 units is staged in one pass, every
// address that does not change within a band is kept in a register.
#include "build_knobs.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <algorithm>

#include "engine.hpp"
#include "eval_tiles.hpp"
#include "exact_math.h"
#include "eval_tile_stage.h"
#include "eval_ordered_sum.h"

namespace amt {

using namespace lin;
using namespace tile;

constexpr int kPairThreads = (kTileWaves + 1) * 64;              // the evaluation waves + the summing wave
constexpr int kPairRowPitch = kTileBandPix + kEvalScorePad;      // floats; the sum reads ahead of the row's end

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

// bin of CorrelationScore (LogoScan.hpp:304) for a mean below 2^31 (guaranteed by pair_eligible: |bg| < 2^30)
__device__ __forceinline__ unsigned score_bin_bounded(float mean) { return (unsigned)((int)__builtin_amdgcn_fmed3f(mean, 0.0f, 255.0f) >> 3); }

struct PairLaunch {
    const EvalLogoDev* logos;
    const TileLogoDev* tls;
    const void* Y;
    const int* frame_map;
    long long frame_stride;      // elements
    int pitch;                   // elements
    float maxv;
    int nframes, G, ngroups, nlogos;
    float* out;
    int out_frame_stride, take_abs;
};

#ifdef AMT_PAIR_OCC
#define AMT_PAIR_OCC_ATTR __attribute__((amdgpu_waves_per_eu(AMT_PAIR_OCC, AMT_PAIR_OCC)))
#else
#define AMT_PAIR_OCC_ATTR
#endif
template <typename pix_t>
__global__ __launch_bounds__(kPairThreads) AMT_PAIR_OCC_ATTR
void logo_eval_pair_kernel(const PairLaunch A)
{
    extern __shared__ float lds[];
    f2* const planes = reinterpret_cast<f2*>(lds);                       // [kTileWaves][kTileCap] {s, bg}: a wave's own tile
    float* const rows = lds + kTileWaves * kTileCap * 2;                 // [2][2 G][kPairRowPitch] per-pixel terms of a band

#ifdef AMT_PAIR_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#define AMT_PTICK(k) do { const long long t_ = clock64(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#define AMT_PDUMP() do { if (lane == 0 && logo == 0 && grp == A.ngroups / 2) { \
        long long* tb = reinterpret_cast<long long*>(A.out + (long long)A.nframes * A.out_frame_stride);   /* the host reserves room */ \
        for (int k = 0; k < 8; ++k) tb[wave * 8 + k] = tacc[k]; \
        tb[16 * 8 + wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   /* HW_REG_HW_ID: SIMD id in bits 5:4 */ } } while (0)
#else
#define AMT_PTICK(k) do { } while (0)
#define AMT_PDUMP() do { } while (0)
#endif
    const int G = A.G;
    const WgMap wm = wg_map_shared_rows((int)blockIdx.x, A.nlogos, A.ngroups);
    if (wm.grp >= A.ngroups) return;                               // (the grid is whole blocks of eight groups; a whole workgroup leaves: no barrier is missed)
    const int logo = wm.logo, grp = wm.grp;
    const int F0 = grp * G;
    const int gcount = min(G, A.nframes - F0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const EvalLogoDev* const Lp = A.logos + logo;
    const TileLogoDev* const Xp = A.tls + logo;
    const int nbands = Xp->nbands;
    constexpr int ES = (int)sizeof(pix_t);

    if (wave == kTileWaves) {
        // ---------------- the summing wave: after the barrier that ends band b it adds band b's rows ----------------
        // Its chain of dependent adds is short on instructions but long on latency: with the issue priority raised it gets its
        // slot as soon as an add's operand is ready instead of queueing behind the SIMD's evaluation waves for every element
        // (measured: 18.7 -> 10 cycles per element), and the band's rows are free again long before the next barrier.
        __builtin_amdgcn_s_setprio(3);
        float acc = 0.0f;
        typedef const __attribute__((address_space(4))) TileBandDesc* const_band_ptr;
        const const_band_ptr bd = (const_band_ptr)Xp->bands;
        for (int b = 0; b < nbands; ++b) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            AMT_PTICK(5);
            const int npix = bd[b].npix;
#ifndef AMT_PAIR_NO_SUM                                         // (ablations of the instrumented builds: wrong results, timing only)
            if (lane < 2 * gcount) acc = ordered_row_sum(rows + ((b & 1) * 2 * G + lane) * kPairRowPitch, npix, acc);
#endif
#ifdef AMT_PAIR_TIMING
            asm volatile("" : "+v"(acc));
#endif
            AMT_PTICK(0);
        }
        if (lane < 2 * gcount) {
            float r = acc / Lp->blackScore;
            if (A.take_abs) r = fabsf(r);
            A.out[(long long)(F0 + (lane >> 1)) * A.out_frame_stride + Lp->out_off + (lane & 1)] = r;
        }
        AMT_PDUMP();
        return;
    }

    // ---------------- evaluation waves ----------------
    const gptr_t gSc = (gptr_t)Xp->sc;
    const unsigned nslots8 = (unsigned)Xp->nslots * 8u;
    const const_tile_ptr tiles = (const_tile_ptr)(Xp->tiles + wave);
    f2* const myplane = planes + wave * kTileCap;
    const unsigned plane_base = lds_address(myplane);
    TileStager<pix_t> st;
    st.init(Lp, A.pitch, A.maxv, myplane);
    TilePixel px;

    TileDesc T;
    fetch_tile(T, tiles);
    st.setup_units(T, lane);
    st.request(frame_rsrc<pix_t>(A.Y, A.frame_map, A.frame_stride, F0));
    // (the first raw samples must not be the LAST loads issued before the loop: vector-memory loads return in order, and the wait the
    //  compiler places at the loop head for them is the merge of this path and the back edge -- with the raw loads sunk below the 14
    //  pixel / tap loads it became vmcnt(0) in every iteration, which also waits for the scale gathers issued just before)
    asm volatile("" ::: "memory");
    px.load(Xp, (unsigned)wave * 64u + (unsigned)lane, T, plane_base);

    // the terms of an evaluation are formed one iteration later, when its two scale gathers have long arrived
    f2 pR = {0.0f, 0.0f}, psc0 = pR, psc1 = pR;
    int prow = 0;                    // (scalar) the pending terms' first row: 2 * frame in the band's set of rows
    bool pact = false;
    auto flush_terms = [&]() {
        if (pact) {
            float* const pdst = rows + prow * kPairRowPitch + px.ridx;
            pdst[0] = score_term(pR.x, psc0.x, psc0.y);               // LogoScan.hpp:305-308
            pdst[kPairRowPitch] = score_term(pR.y, psc1.x, psc1.y);
        }
    };

    int b = 0, g = 0;
    const int niter = nbands * gcount;
    for (int it = 0; it < niter; ++it) {
        const bool band_end = g + 1 == gcount;
        // ---- 1. the iteration's tile: raw -> {s, bg} pairs in this wave's plane ----
        AMT_PTICK(6);
#ifdef AMT_PAIR_TIMING
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // (timing build: the wait for the raw samples on its own)
        AMT_PTICK(7);
#endif
#ifndef AMT_PAIR_NO_CONVERT
        st.convert();
#endif
        AMT_PTICK(0);
        // ---- 2. the next iteration's raw samples travel during the evaluation (past the last iteration: a repeat nobody reads) ----
        if (band_end && b + 1 < nbands) {
            fetch_tile(T, tiles + (b + 1) * kTileWaves);
            st.setup_units(T, lane);
        }
#if defined(AMT_PAIR_RAW_SAMEFRAME)                             // (ablation: every request hits the cache -- separates the loads' latency from their issue cost)
        st.request(frame_rsrc<pix_t>(A.Y, A.frame_map, A.frame_stride, 0));
#elif !defined(AMT_PAIR_NO_RAW)
        st.request(frame_rsrc<pix_t>(A.Y, A.frame_map, A.frame_stride, F0 + (band_end ? 0 : g + 1)));
#endif
        AMT_PTICK(1);
        // ---- 3. both fades of the frame: one packed window evaluation ----
        // (the taps are loop-invariant: LICM would hoist their {k,k} broadcasts and keep 50 registers of copies; the empty asm
        //  makes them opaque per iteration and the broadcast folds into the multiply's op_sel)
#pragma unroll
        for (int j = 0; j < 13; ++j) asm volatile("" : "+v"(px.Kp[j]));
        f2 W[25];
        unsigned wrow[5];
        px.rows(wrow);
#ifdef AMT_PAIR_NO_EVAL
        const f2 M = px.Kp[1] + f2{100.0f, 120.0f}, R = px.Kp[0];
        (void)W;
#else
        const f2 M = window_load_means(wrow, W);      // (issuing the reads before the next requests instead: 2.630 ms either way, round 5)
        const f2 R = window_corr_exact(px.Kp, W, M);
#endif
#ifdef AMT_PAIR_TIMING
        { f2 Rt = R; asm volatile("" : "+v"(Rt)); }
#endif
        AMT_PTICK(2);
        // ---- 4. the previous iteration's terms -> the band's score rows, at the pixel's raster position; then this iteration's two
        //      scale gathers go straight into the registers the terms were read from (a copy would wait for them here) ----
#ifndef AMT_PAIR_NO_FLUSH
        flush_terms();
#endif
#ifdef AMT_PAIR_NO_GATHER
        psc0 = f2{1e-3f, 1.0f} + M; psc1 = f2{1e-3f, 1.0f} - M;
#else
        psc0 = gld<f2>(gSc, __umul24(score_bin_bounded(M.x), nslots8) + px.slot8);
        psc1 = gld<f2>(gSc, __umul24(score_bin_bounded(M.y), nslots8) + px.slot8);
#endif
        pR = R; pact = px.act;
        AMT_PTICK(3);
        prow = (b & 1) * 2 * G + g * 2;
        if (band_end) {
            flush_terms();                                           // the band's rows are complete before the waves meet
            pact = false;
            // the band's last evaluation is done: the next band's pixel and taps travel across the barrier
            ++b; g = 0;
            if (b < nbands) px.load(Xp, (unsigned)(b * kTileWaves + wave) * 64u + (unsigned)lane, T, plane_base);
            AMT_PTICK(4);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            AMT_PTICK(5);
        } else {
            ++g;
        }
    }
    AMT_PDUMP();
}

hipError_t launch_logo_eval_pair(hipStream_t st, int bits, const EvalLogoDev* dlogos, const TileLogoDev* dtls, int nlogos,
                                 const void* dY, const int* dframe_map, long long frame_stride_elems, int pitch,
                                 int nframes, int G, float* dout, int out_frame_stride, int take_abs)
{
    if (nframes <= 0 || nlogos <= 0) return hipSuccess;
    PairLaunch A;
    A.logos = dlogos; A.tls = dtls; A.Y = dY; A.frame_map = dframe_map; A.frame_stride = frame_stride_elems; A.pitch = pitch;
    A.maxv = (float)((1 << bits) - 1);
    A.nframes = nframes; A.G = G; A.ngroups = (nframes + G - 1) / G; A.nlogos = nlogos;
    A.out = dout; A.out_frame_stride = out_frame_stride; A.take_abs = take_abs;
    const size_t lds = ((size_t)kTileWaves * kTileCap * 2 + (size_t)2 * 2 * G * kPairRowPitch) * sizeof(float);
    if (G < 1 || 2 * G > 64 || lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid((unsigned)wg_grid_shared_rows(A.ngroups, nlogos));
    if (bits <= 8) hipLaunchKernelGGL(logo_eval_pair_kernel<uint8_t>, grid, dim3(kPairThreads), lds, st, A);
    else hipLaunchKernelGGL(logo_eval_pair_kernel<uint16_t>, grid, dim3(kPairThreads), lds, st, A);
    return hipGetLastError();
}

} // namespace amt
