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
// No ordered sum: per-pixel terms are summed per wave with DPP adds and the eight wave sums in a fixed order -- deterministic.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <algorithm>

#include "eval_plan.h"
#include "exact_math.h"
#include "eval_lds_stage.h"

namespace amt {

using namespace lin;

// sum over the 64 lanes of a wave in a fixed order (DPP: every step is one v_add_f32); the total lands in lane 63
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));   // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));   // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));   // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));   // row_shr:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));   // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, true));   // row_bcast:31
    return v;
}

// NF > 0: exactly NF fades (11 for AMTAnalyzeLogo: no per-fade branches); NF == 0: nfades <= kLinMaxFades at run time.
//
// Software pipeline over the (band, frame) iterations of a workgroup, ONE barrier per iteration: the raw rows of the NEXT
// iteration are requested before the current window evaluation (buffer_load ... lds: no registers in flight) straight into the
// other half of a double-buffered LDS plane and converted there, in place, after the fade code -- the global-memory latency of
// staging sits behind the arithmetic instead of in front of a barrier.
template <typename pix_t, int NF>
__global__ __launch_bounds__(kLinThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void logo_eval_linear_kernel(const EvalLogoDev* __restrict__ logos, const LinLogoDev* __restrict__ lins, const EvalBand* __restrict__ bands,
                             const float* __restrict__ fades, int nfades_rt, int fade0, const pix_t* __restrict__ Y,
                             const int* __restrict__ frame_map, long long frame_stride, int pitch, float maxv, int nframes, int G,
                             int ngroups, float* __restrict__ out, int out_frame_stride, int take_abs, int plane_cap, float bin_delta)
{
    constexpr int NFMAX = NF > 0 ? NF : kLinMaxFades;
    const int nfades = NF > 0 ? NF : nfades_rt;
    extern __shared__ float lds[];
    f2* const planes = reinterpret_cast<f2*>(lds);                 // [2][plane_cap] {s, bg = a*s + b*maxv} of a band's rows, one frame each
    f2* const abp = planes + 2 * plane_cap;                        // [plane_cap] {a, b} of the current band's rows (frame-independent)
    float* const wpart = lds + 6 * plane_cap;                      // [2][kWaves][NFMAX] per-wave sums of the terms of an iteration
    float* const accs = wpart + 2 * kWaves * NFMAX;                // [G][nfades] running sums

    const int logo = blockIdx.x / ngroups;
    const int grp = blockIdx.x - logo * ngroups;
    const int F0 = grp * G;
    const int gcount = min(G, nframes - F0);
    const EvalLogoDev L = logos[logo];
    const LinLogoDev X = lins[logo];
    const gptr_t gScales = (gptr_t)L.scales, gK = (gptr_t)X.kpix, gPos = (gptr_t)X.pos;
    const unsigned cpad = (unsigned)L.count_pad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: the staging rows' address math stays scalar
    const int w = L.w, lp = L.lp;
    constexpr unsigned ES = sizeof(pix_t);
    // bin edges in 1/4096 fixed point: a mean within dq of a multiple of 8 takes the exact path
    const int dq = (int)(bin_delta * 4096.0f) + 2;

#ifdef AMT_LIN_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#define AMT_LTICK(k) do { const long long t_ = clock64(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define AMT_LTICK(k) do { } while (0)
#endif
    if (tid < G * nfades) accs[tid] = 0.0f;
    const int fade_bits = __builtin_bit_cast(int, fades[fade0 + min(lane, nfades - 1)]);     // lane f holds fade f
    // buffer descriptors: loads below are (descriptor, per-lane column offset, wave-uniform row offset) -- no per-load address math
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.a), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.b), 0, 0x7FFFFFFF, 0x00020000);

    // ---- staging: unit = one row of the band (RowStager, eval_lds_stage.h); row r belongs to wave r % 8 -- rows 2w, 2w+1 per wave put
    //      twice the conversion work on the SIMD that hosts waves 0 and 4 of a typical 10-row band.  The raw rows of the next iteration
    //      are requested at the top of an iteration with buffer_load ... lds (no registers held) into the plane row they are
    //      converted into after the fade code; the row's logo coefficients {a, b*maxv} stay in LDS across the frames of the
    //      workgroup, written and read by the wave that owns the row (no barrier involved) ----
    constexpr int kMaxUnits = kLinBandRows / kWaves;               // 2
    const RowStager<pix_t> st(L, lane, pitch, maxv);
    auto frame_rsrc = [&](int g) {
        const int frame = F0 + g;
        const int srcFrame = frame_map ? frame_map[frame] : frame;
        const pix_t* src = Y + (long long)srcFrame * frame_stride + (long long)(L.imgy + L.row0) * pitch + L.imgx;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<pix_t*>(src), 0, 0x7FFFFFFF, 0x00020000);
    };

    const int niter = X.nbands * gcount;
    EvalBand B = bands[X.band0];
    // prologue: the first iteration's rows
    {
        const __amdgpu_buffer_rsrc_t rs = frame_rsrc(0);
#pragma unroll
        for (int k = 0; k < kMaxUnits; ++k) {
            const int r = wave + kWaves * k;
            if (r >= B.nrows) break;
            f4 av, bmv;
            st.request(rs, B.y0 + r, planes + r * lp);
            st.load_ab(rA, rB, B.y0 + r, av, bmv);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st.ab_to_lds(abp + r * lp, av, bmv);
            st.convert(planes + r * lp, B.y0 + r, av, bmv);
        }
    }
    bool act = false;
    unsigned m = 0, m8 = 0;
    int woff = 0;
    const unsigned cpad8 = cpad * 8u;
    f2 Kp[13];
    // a band's mask pixel of this thread: window offset, table index, taps (a surplus thread gets zero taps: it evaluates to exactly 0,
    // no branches in the fade code)
    auto load_pixel = [&](const EvalBand& Bd) {
        act = tid < Bd.npix;
        m = (unsigned)(Bd.m0 + (act ? tid : 0));
        const unsigned pos = gld<unsigned>(gPos, m * 4u);
        woff = ((int)(pos >> 16) - 2 - Bd.y0) * lp + (int)(pos & 0xFFFFu) - 2;          // plane offset of the window's top-left element
        m8 = m * 8u;
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            Kp[j] = gld<f2>(gK, ((unsigned)j * cpad + m) * 8u);
            if (!act) Kp[j] = f2{0.0f, 0.0f};
        }
    };
    load_pixel(B);
    __syncthreads();

    int bi = 0, g = 0;
    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        f2* const plane = planes + cur * plane_cap;
        // (a band's pixel and taps are requested behind the previous band's last evaluation, see load_pixel below)
        AMT_LTICK(0);
        // the next iteration: same band / next frame, or the next band / first frame
        const bool has_next = it + 1 < niter;
        const bool next_band = g + 1 == gcount;
        const int ng = next_band ? 0 : g + 1;
        EvalBand Bn = B;
        if (has_next && next_band) {
            const EvalBand* nb = bands + X.band0 + bi + 1;
            Bn.m0 = nb->m0; Bn.npix = nb->npix; Bn.y0 = nb->y0; Bn.nrows = nb->nrows;
        }
        // ---- 1. request the next iteration's raw rows (LDS-direct: no registers held) ----
        if (has_next) {
            const __amdgpu_buffer_rsrc_t rs = frame_rsrc(ng);
#pragma unroll
            for (int k = 0; k < kMaxUnits; ++k) {
                const int r = wave + kWaves * k;
                if (r >= Bn.nrows) break;
                st.request(rs, Bn.y0 + r, planes + (cur ^ 1) * plane_cap + r * lp);
            }
        }
        AMT_LTICK(1);
        // ---- 2. fold the previous iteration's per-wave sums into the running sums (fixed order: deterministic) ----
        if (it > 0 && tid < nfades) {
            const float* wp = wpart + (cur ^ 1) * kWaves * NFMAX + tid;
            float s = 0.0f;
#pragma unroll
            for (int q = 0; q < kWaves; ++q) s += wp[q * NFMAX];
            const int pg = g == 0 ? gcount - 1 : g - 1;        // the previous iteration's frame
            accs[pg * nfades + tid] += s;
        }
        // ---- 3. ONE window evaluation for both operands: R = {corr(s), corr(bg)}, M = {mean(s), mean(bg)} ----
        // The taps are loop-invariant, so LICM would hoist their {k,k} broadcasts out of the loop and keep 50 registers of
        // copies; an empty asm makes them opaque per iteration and the broadcast folds into the multiply's op_sel instead.
#pragma unroll
        for (int j = 0; j < 13; ++j) asm volatile("" : "+v"(Kp[j]));
        f2 R, M;
#ifdef AMT_LIN_NO_EVAL
        M = plane[woff]; R = Kp[0] * M;
#else
        {
            f2 W[25];
            load_window(plane, woff, lp, W);     // surplus threads read pixel B.m0's window: finite values, zero taps
            M = window_means(W);
            R = window_corr(Kp, W, M);
        }
#endif
#ifdef AMT_LIN_TIMING
        if (R.x == 123456.0f && M.y == 123456.0f) tacc[7] += 1;
#endif
        AMT_LTICK(2);
        f4 av[kMaxUnits], bmv[kMaxUnits];
        // ---- 5. all fades from the two pairs: interpolated mean -> bin -> scale gather, all in flight together; while they travel
        //      the next iteration's rows are converted into the other plane (loads return in order: raw rows and coefficients
        //      were requested earlier); then correlation and per-pixel term (LogoScan.hpp:305-308), summed over the wave.
        //      The bin select is discontinuous: a mean within dq of a bin edge is noted, and those (pixel, fade) pairs -- about
        //      1e-4 of all -- are redone below with the mean evaluated exactly as the reference does ----
        const float m0q = M.x * 4096.0f, dMq = (M.y - M.x) * 4096.0f, dR = R.y - R.x;     // mean in 1/4096 units: m0q + fade * dMq
        unsigned emin = 0x7FFFu;                                                          // smallest distance (+dq) to a bin edge
        float term[NFMAX];
#ifdef AMT_LIN_NO_FADES
#pragma unroll
        for (int f = 0; f < NFMAX; ++f) term[f] = R.x + dR * (float)f + m0q + dMq;
#else
        // every fade's scale gather is in flight before the first term is formed: one exposed round trip to L2 per iteration
        constexpr int NA = (NFMAX + 1) / 2;
        auto issue = [&](int f, f2& dst) {
            const float fade = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fade_bits, f));
            const int q = (int)__builtin_fmaf(fade, dMq, m0q);
            emin = min(emin, (unsigned)(q + dq) & 0x7FFFu);
#ifdef AMT_LIN_NO_GATHER
            dst = f2{1e-4f * (float)(q >> 15), 0.5f};
#elif defined(AMT_LIN_GATHER_BIN0)
            dst = gld<f2>(gScales, __umul24((unsigned)min(max(q >> 25, 0), 31), cpad8) + m8);     // every lane reads bin 0: fully coalesced
#else
            dst = gld<f2>(gScales, __umul24((unsigned)min(max(q >> 15, 0), 31), cpad8) + m8);
#endif
        };
        auto finish = [&](int f, const f2& sc) {
            const float fade = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fade_bits, f));
            term[f] = __builtin_amdgcn_fmed3f(__builtin_fmaf(fade, dR, R.x) * sc.x, -1.0f, 1.0f) * sc.y;
        };
        f2 scA[NA], scB[NFMAX - NA];
#pragma unroll
        for (int f = 0; f < NA; ++f) if (NF > 0 || f < nfades) issue(f, scA[f]);
#pragma unroll
        for (int f = NA; f < NFMAX; ++f) if (NF > 0 || f < nfades) issue(f, scB[f - NA]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NA; ++f) if (NF > 0 || f < nfades) finish(f, scA[f]);
#pragma unroll
        for (int f = NA; f < NFMAX; ++f) if (NF > 0 || f < nfades) finish(f, scB[f - NA]);
        // a mean near an edge: exact mean -> the reference's bin; when it differs from the interpolated mean's, replace the term
        if (act && emin <= (unsigned)(2 * dq)) {
#pragma unroll
            for (int f = 0; f < NFMAX; ++f) {
                if (NF > 0 || f < nfades) {
                    const float fade = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fade_bits, f));
                    const int q = (int)__builtin_fmaf(fade, dMq, m0q);
                    if (((unsigned)(q + dq) & 0x7FFFu) <= (unsigned)(2 * dq)) {
                        const int bin_ref = score_bin_dev(exact_blend_mean(plane, woff, lp, fade, 1 - fade));
                        if (bin_ref != min(max(q >> 15, 0), 31)) {
                            const f2 s2 = gld<f2>(gScales, __umul24((unsigned)bin_ref, cpad8) + m8);
                            term[f] = __builtin_amdgcn_fmed3f(__builtin_fmaf(fade, dR, R.x) * s2.x, -1.0f, 1.0f) * s2.y;
                        }
                    }
                }
            }
        }
#endif
#ifdef AMT_LIN_TIMING
        if (term[0] == 123456.0f && term[NFMAX - 1] == 123456.0f) tacc[7] += 1;
#endif
        AMT_LTICK(3);
        __builtin_amdgcn_sched_barrier(0);
        if (has_next && next_band) {                           // once per band: from memory (the wave sums below cover the trip)
#pragma unroll
            for (int k = 0; k < kMaxUnits; ++k) {
                const int r = wave + kWaves * k;
                if (r >= Bn.nrows) break;
                st.load_ab(rA, rB, Bn.y0 + r, av[k], bmv[k]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef AMT_LIN_NO_REDUCE
        if (lane == 63) wpart[(cur * kWaves + wave) * NFMAX] = term[0] + term[NFMAX - 1];
#else
#pragma unroll
        for (int f = 0; f < NFMAX; ++f) {
            if (NF > 0 || f < nfades) {
                const float s = wave_sum_dpp(term[f]);
                if (lane == 63) wpart[(cur * kWaves + wave) * NFMAX + f] = s;
            }
        }
#endif
        AMT_LTICK(4);
#ifndef AMT_LIN_NO_STAGE
        if (has_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the LDS-direct loads are counted with the vector-memory loads
            AMT_LTICK(5);
            // the band's last evaluation is long done: the next band's pixel and taps travel while the rows are converted
            if (next_band) load_pixel(Bn);
#pragma unroll
            for (int k = 0; k < kMaxUnits; ++k) {
                const int r = wave + kWaves * k;
                if (r >= Bn.nrows) break;
                if (next_band) st.ab_to_lds(abp + r * lp, av[k], bmv[k]); else st.ab_from_lds(abp + r * lp, av[k], bmv[k]);
                st.convert(planes + (cur ^ 1) * plane_cap + r * lp, Bn.y0 + r, av[k], bmv[k]);
            }
        }
#endif
        AMT_LTICK(6);
        __syncthreads();                         // next plane and this iteration's wave sums complete; current plane consumed
        AMT_LTICK(7);
        if (next_band) { B.m0 = Bn.m0; B.npix = Bn.npix; B.y0 = Bn.y0; B.nrows = Bn.nrows; ++bi; }
        g = ng;
    }
    // the last iteration's wave sums
    if (niter > 0 && tid < nfades) {
        const float* wp = wpart + ((niter - 1) & 1) * kWaves * NFMAX + tid;
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < kWaves; ++q) s += wp[q * NFMAX];
        accs[(gcount - 1) * nfades + tid] += s;
    }
    __syncthreads();
#ifdef AMT_LIN_TIMING
    if (lane == 0 && blockIdx.x == gridDim.x / 6 && (wave == 0 || wave == 3 || wave == 5 || wave == 7)) {      // a workgroup of logo 0 (the deint logo)
        long long* tb = reinterpret_cast<long long*>(out + (long long)nframes * out_frame_stride);            // host reserves room
        const int slot = wave == 0 ? 0 : (wave == 3 ? 1 : (wave == 5 ? 2 : 3));
        for (int k = 0; k < 8; ++k) tb[slot * 8 + k] = tacc[k];
    }
#endif
    if (tid < gcount * nfades) {
        const int gg = tid / nfades, f = tid - gg * nfades;
        float r = accs[tid] / L.blackScore;
        if (take_abs) r = fabsf(r);
        out[(long long)(F0 + gg) * out_frame_stride + L.out_off + fade0 + f] = r;
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
    const int nfmax = nfades == 11 ? 11 : kLinMaxFades;
    const size_t lds = ((size_t)6 * plane_cap + (size_t)2 * (kLinThreads / 64) * nfmax + (size_t)G * nfades) * sizeof(float);
    dim3 grid((unsigned)((long long)ngroups * nlogos));
#define AMT_LAUNCH(T, N)                                                                                                              \
    hipLaunchKernelGGL((logo_eval_linear_kernel<T, N>), grid, dim3(kLinThreads), lds, st, dlogos, dlins, dbands, dfades, nfades, fade0,   \
                       (const T*)dY, dframe_map, frame_stride_elems, pitch, maxv, nframes, G, ngroups, dout, out_frame_stride, take_abs, \
                       plane_cap, bin_delta)
    if (nfades == 11) { if (bits <= 8) AMT_LAUNCH(uint8_t, 11); else AMT_LAUNCH(uint16_t, 11); }
    else { if (bits <= 8) AMT_LAUNCH(uint8_t, 0); else AMT_LAUNCH(uint16_t, 0); }
#undef AMT_LAUNCH
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
