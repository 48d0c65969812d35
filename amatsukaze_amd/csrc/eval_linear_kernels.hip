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
//     evaluated mean) of a bin edge, the (pixel, frame, fade) is LISTED (a per-wave queue in LDS) and the loop carries on with the
//     tentative bin; after the loop every listed pair gets its mean re-computed from the frame in the reference's exact order (blend,
//     column sums, hsum, /25), the bin from the exact value and, where that differs, the correction t(exact bin) - t(tentative bin);
//     a queue that overflows marks the workgroup's frames for the exact kernel (LinLaunch::force);
//   * the scale of a term is COMPUTED, not looked up: CreateLogoMask's table value for (pixel, bin) is |P + Q*bin| up to rounding (the
//     composite of a flat level is affine in the level), so t = med3(x, -r, r) * rcp(max(r, L)) with r = |fma(Q, bin, P)| -- the
//     deviation from the reference's table-driven clamp(x*scale, -1, 1)*scale2 is evaluated per (pixel, bin) on the host and is part of
//     the error bound (EvalEngine::ensure_linear);
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
#include "build_knobs.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <algorithm>
#include <cmath>

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

// Mean of the blended 5x5 window around logo pixel (x, y) of one frame, exactly as EvaluateLogo + CalcCorrelation5x5_AVX produce it
// (LogoScan.hpp:244-251, ComputeKernel.cpp:88-98) -- from the frame's samples themselves: s = the sample, or DeintY's (a + 2b + c + 2) / 4
// (LogoScan.hpp:763-780; an integer sum below 2^24 times 0.25), bg = a*s + b*maxv, the blend, column sums ((r0 + r1) + (r2 + r3)) + r4,
// hsum256_ps' order, /25.  The uncommon path of the bin select: run once per listed (pixel, frame, fade) after the workgroup's loop.
// five adjacent samples of a source row, p[0..4], with as few loads as their alignment allows: the 4-byte words that hold them (the
// listed pairs sit in different frames and rows -- every load instruction of the wave touches up to 64 cache lines, so the number of
// load INSTRUCTIONS per pair is what the check costs)
__device__ __forceinline__ void load5(const uint8_t* p, unsigned (&out)[5])
{
    const unsigned long long a = (unsigned long long)p;
    const unsigned* q = reinterpret_cast<const unsigned*>(a & ~3ull);
    const unsigned sh = ((unsigned)a & 3u) * 8u;
    const unsigned w0 = q[0], w1 = q[1];
    const unsigned long long v = (((unsigned long long)w1 << 32) | w0) >> sh;          // bytes 0..3 (and 4 when sh < 32)
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    out[0] = lo & 0xFFu; out[1] = (lo >> 8) & 0xFFu; out[2] = (lo >> 16) & 0xFFu; out[3] = lo >> 24; out[4] = hi & 0xFFu;
}
__device__ __forceinline__ void load5(const uint16_t* p, unsigned (&out)[5])
{
    const unsigned long long a = (unsigned long long)p;
    const unsigned* q = reinterpret_cast<const unsigned*>(a & ~3ull);
    const bool odd = (a & 2ull) != 0;
    const unsigned w0 = q[0], w1 = q[1], w2 = q[2];
    if (odd) { out[0] = w0 >> 16; out[1] = w1 & 0xFFFFu; out[2] = w1 >> 16; out[3] = w2 & 0xFFFFu; out[4] = w2 >> 16; }
    else { out[0] = w0 & 0xFFFFu; out[1] = w0 >> 16; out[2] = w1 & 0xFFFFu; out[3] = w1 >> 16; out[4] = w2 & 0xFFFFu; }
}
template <typename pix_t>
__device__ __forceinline__ float exact_blend_mean_from_frame(const EvalLogoDev* Lp, const pix_t* frame, int pitch, float maxv, int x, int y, float fade)
{
    const int w = Lp->w, h = Lp->h, step = Lp->row_step;
    const bool deint = Lp->deint != 0;
    const float* const la = Lp->a + (y - 2) * w + (x - 2);
    const float* const lb = Lp->b + (y - 2) * w + (x - 2);
    // the window's source rows: for a deinterlaced logo the rows above and below as well (seven in all), clamped to the logo's own rows
    // -- the first and the last row are not blended and never look outside.  (The words load5 reads lie inside the frame's rows: the
    // window's columns x - 2 .. x + 2 are at least two samples from the rectangle's edge on either side.)
    const pix_t* const p0 = frame + (long long)(Lp->imgy + Lp->row0) * pitch + Lp->imgx + (x - 2);
    unsigned raw[7][5];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int yy = min(max(y - 3 + r, 0), h - 1);
        if ((r == 0 || r == 6) && !deint) {
#pragma unroll
            for (int i = 0; i < 5; ++i) raw[r][i] = 0u;
        } else {
            load5(p0 + (long long)yy * step * pitch, raw[r]);
        }
    }
    float c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = 0.0f;
    float v[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int yy = y - 2 + r;
        const bool blend = deint && yy > 0 && yy < h - 1;
        float av[5], bv[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { av[i] = la[r * w + i]; bv[i] = lb[r * w + i]; }      // (adjacent: the compiler merges them into wide loads)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float sv = blend ? __builtin_amdgcn_ldexpf((float)(((raw[r + 1][i] << 1) + raw[r][i]) + (raw[r + 2][i] + 2u)), -2) : (float)raw[r + 1][i];
            const float bmv = bv[i] * maxv;
            const float bg = av[i] * sv + bmv;
            v[r][i] = fade_mix(fade, bg, sv);
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = ((v[0][i] + v[1][i]) + (v[2][i] + v[3][i])) + v[4][i];
    return div25(hsum5(c[0], c[1], c[2], c[3], c[4]));
}

// ------------------------------------------------------------------------------------------------------------------------
// The raw-sample loads of the loop, with a HAND-PLACED wait.  The loop's point is to have the next frame's raw samples in flight while
// the current frame is evaluated.  The compiler's wait insertion counts the loads in flight per control-flow path and takes the
// strictest count where paths meet; with loads that are issued on some trips only (a new tile's taps, its response line, its logo
// coefficients) it drained EVERY load -- the raw samples included -- at the loop head or in front of the first conditional consumer
// (rounds 2-5 had that drain in front of their scale gathers).  So:
//   * the raw-sample loads -- issued on every trip, consumed exactly one trip later -- are inline assembly, which the compiler neither
//     counts nor waits for; the wait is written out (LinStager::landed) and names the registers it releases as in/out operands, so the
//     consumers depend on it and nothing can be scheduled across it.  The loads' destinations are in/out operands too ("+v"): the load
//     lands in the register the variable already lives in.  As plain outputs the compiler is free to give the results fresh registers
//     and copy them into the loop-carried ones right behind the load -- it believes the value is there -- and the copy reads what the
//     load has not written yet (seen; tests/test_isa_guards.py walks the code for any read of a register between such a load and the
//     next vmcnt wait);
//   * the per-tile loads stay the compiler's (a value loaded on one path and carried on another IS copied where the paths meet, and
//     only the compiler can protect that copy), but are consumed in the iteration that issues them, behind the terms (LinPixel::landed):
//     no trip reaches the loop head with one of them in flight, so the compiler has nothing to wait for there.
// A load the compiler adds to the loop can only make the hand-placed wait stricter (it waits for "at most N outstanding").
// ------------------------------------------------------------------------------------------------------------------------
typedef unsigned u4 __attribute__((ext_vector_type(4)));
// buffer descriptor of a source frame: base, stride 0, 2 GiB of records, raw dword format (as __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7FFFFFFF, 0x00020000))
template <typename pix_t>
__device__ __forceinline__ u4 frame_desc(const void* Y, const int* frame_map, long long frame_stride, int frame)
{
    const int srcFrame = frame_map ? ((const_int_ptr)frame_map)[frame] : frame;
    const unsigned long long a = (unsigned long long)(reinterpret_cast<const pix_t*>(Y) + (long long)srcFrame * frame_stride);
    return u4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
}
__device__ __forceinline__ void vm_load_raw(Quad<uint8_t>& q, u4 desc, int voff)
{
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(q.v) : "v"(voff), "s"(desc) : "memory");
}
__device__ __forceinline__ void vm_load_raw(Quad<uint16_t>& q, u4 desc, int voff)
{
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "+v"(q.v) : "v"(voff), "s"(desc) : "memory");
}
// a table base as the scalar register pair the saddr form of a global load takes (the value IS wave-uniform -- it hangs off the
// workgroup's logo -- but arrives through a vector load where the compiler cannot prove it)
typedef u2 sbase_t;
__device__ __forceinline__ sbase_t uniform_base(const void* p)
{
    const unsigned long long a = (unsigned long long)p;
    return sbase_t{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32))};
}
__device__ __forceinline__ gptr_t table_base(sbase_t b) { return (gptr_t)(((unsigned long long)b.y << 32) | b.x); }
// at most N vector-memory loads outstanding; releases ...
template <int N, typename Q> __device__ __forceinline__ void vm_wait_raw(Q (&raw)[kTileUnits][3])   // ... the raw samples of a request
{
    static_assert(kTileUnits == 2, "operand list written for two units");
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(raw[0][0].v), "+v"(raw[0][1].v), "+v"(raw[0][2].v), "+v"(raw[1][0].v), "+v"(raw[1][1].v), "+v"(raw[1][2].v)
                 : "n"(N) : "memory");
}

// {s.x, s.y} * {v.y, v.y} + {v.x, v.x}: one packed multiply-add for a PAIR of fades, the multiplier and the addend broadcast out of the
// two halves of one register pair by the instruction's op_sel (no copies).  Every vector instruction costs the SIMD the same four
// cycles, packed or not (profiles/r06_notes.md): what can be said for two fades at once is said once.
__device__ __forceinline__ f2 pk_fma_hi_lo_s(f2 s, f2 v)      // s: a scalar register pair (two fades)
{
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "s"(s), "v"(v));
    return r;
}
__device__ __forceinline__ f2 pk_fma_hi_lo_v(f2 a, f2 v)
{
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(v));
    return r;
}

// One wave's staging state (eval_tile_stage.h TileStager, SLIM form) with the loads above.  BLEND = false: a field logo, whose rows are
// the source rows themselves (CopyY): one row load per unit instead of three, no [1 2 1] sums.
template <typename pix_t, bool BLEND> struct LinStager {
    static constexpr int ES = (int)sizeof(pix_t);
    static constexpr int kRowsPerUnit = BLEND ? 3 : 1;
    static constexpr int kRawLoads = kTileUnits * kRowsPerUnit;      // vector-memory loads of one request()
    static constexpr int kCoefLoads = 2 * kTileUnits;                // ... of one setup_units()
    int pitchB;                      // (wave-uniform)
    float maxv;
    f2* plane;
    bool second_pass;                // the tile has more than 64 units
    bool coef_fresh;                 // ub[] still holds b, not b * maxv (the product -- and with it the wait for the loads -- is left to the
                                     // first conversion, an iteration later)
    int ulds[kTileUnits];            // pair offset in the tile plane; sign bit: this row is a [1 2 1] blend
    int ug[kTileUnits];              // byte offset in a frame of the unit's own row
    f4 ua[kTileUnits], ub[kTileUnits];                             // the unit's logo coefficients: a, b * maxv
    Quad<pix_t> raw[kTileUnits][3];

    __device__ __forceinline__ void init(int pitch, float maxv_, f2* plane_)
    {
        pitchB = pitch * ES; maxv = maxv_; plane = plane_;
        second_pass = false; coef_fresh = false;
#pragma unroll
        for (int k = 0; k < kTileUnits; ++k)                        // (the raw loads' destinations are in/out operands: they need a value)
#pragma unroll
            for (int j = 0; j < 3; ++j) raw[k][j].v = decltype(raw[k][j].v){};
    }
    // (the logo's geometry is read again at every new tile -- scalar loads: the loop is short of scalar registers)
    __device__ __forceinline__ void setup_units(const EvalLogoDev* Lp, const TileLogoDev* Xp, const TileDesc& T, int lane)
    {
        typedef const __attribute__((address_space(4))) EvalLogoDev* logo_ptr;
        typedef const __attribute__((address_space(4))) TileLogoDev* tlogo_ptr;
        const logo_ptr L = (logo_ptr)Lp;
        const tlogo_ptr X = (tlogo_ptr)Xp;
        const int w = L->w, h = L->h, deint = L->deint, srow0 = L->imgy + L->row0, srow_step = L->row_step, scol0 = L->imgx;
        const gptr_t base = table_base(uniform_base(X->lin));
        const unsigned oa = X->lin_a, ob = X->lin_b;
        typedef f4 __attribute__((aligned(8))) f4a8;
        second_pass = T.nrows * T.ncol4 > 64;
#pragma unroll
        for (int k = 0; k < kTileUnits; ++k) {
            const TileUnit U = tile_unit(T, lane + 64 * k, w);
            const bool blend = BLEND && deint && U.y > 0 && U.y < h - 1;       // DeintY copies the first and the last row (LogoScan.hpp:763-780)
            ulds[k] = U.lds | (blend ? (int)0x80000000 : 0);
            ug[k] = (srow0 + U.y * srow_step) * pitchB + (scol0 + U.xs) * ES;
            ua[k] = gld<f4a8>(base, (unsigned)(U.y * w + U.xs) * 4u + oa);
            ub[k] = gld<f4a8>(base, (unsigned)(U.y * w + U.xs) * 4u + ob);
        }
        coef_fresh = true;
    }
    __device__ __forceinline__ void request(const u4 frame)
    {
#pragma unroll
        for (int k = 0; k < kTileUnits; ++k) {
            if (!BLEND) {
                vm_load_raw(raw[k][1], frame, ug[k]);
            } else {
                const int d = (ulds[k] >> 31) & pitchB;           // a blended row: the rows above and below; otherwise the row itself
                vm_load_raw(raw[k][0], frame, ug[k] - d);
                vm_load_raw(raw[k][1], frame, ug[k]);
                vm_load_raw(raw[k][2], frame, ug[k] + d);
            }
        }
    }
    // the samples of the oldest request() have landed when at most N younger loads are outstanding
    template <int N> __device__ __forceinline__ void landed()
    {
        if (!BLEND) {   // (the unused rows must not be operands: they were never written)
            static_assert(kTileUnits == 2, "operand list written for two units");
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(raw[0][1].v), "+v"(raw[1][1].v) : "n"(N) : "memory");
        } else {
            vm_wait_raw<N>(raw);
        }
    }
    __device__ __forceinline__ void convert_unit(int k)
    {
        float sv[4];
        if (BLEND) Quad<pix_t>::blend(raw[k][0], raw[k][1], raw[k][2], (unsigned)(ulds[k] >> 31) & Quad<pix_t>::kBias, sv);
        else Quad<pix_t>::copy(raw[k][1], sv);
        f2* dst = plane + (ulds[k] & 0x7FFFFFFF);
        reinterpret_cast<f4*>(dst)[0] = f4{sv[0], ua[k][0] * sv[0] + ub[k][0], sv[1], ua[k][1] * sv[1] + ub[k][1]};      // {s, bg = a*s + b*maxv} (LogoScan.hpp:247)
        reinterpret_cast<f4*>(dst)[1] = f4{sv[2], ua[k][2] * sv[2] + ub[k][2], sv[3], ua[k][3] * sv[3] + ub[k][3]};
    }
    __device__ __forceinline__ void convert()
    {
        if (coef_fresh) {                                           // b * maxv, rounded once, exactly as in a*s + b*maxv
#pragma unroll
            for (int k = 0; k < kTileUnits; ++k) ub[k] = ub[k] * maxv;
            coef_fresh = false;
        }
        convert_unit(0);
        if (second_pass) {
#pragma unroll
            for (int k = 1; k < kTileUnits; ++k) convert_unit(k);
        }
    }
};

// this lane's mask pixel of a tile: its taps, its response line and its slot word (window offset in the tile, valid bit)
struct LinPixel {
    f2 Kp[13];
    f2 PQ;                           // the pixel's response on flat level c is |PQ.x + PQ.y * c|
    unsigned si;                     // tile_slot_info
    unsigned slot8;
    unsigned tp8;                    // (wave-uniform) the tile's row pitch in bytes
    // the taps, the response line and the slot word of `slot` of the logo's blob (TileLogoDev::lin: kp at 0), requested
    __device__ __forceinline__ void load(gptr_t base, unsigned nslots8, unsigned off_sinfo, unsigned off_pq, unsigned slot, const TileDesc& T)
    {
        tp8 = (unsigned)T.tp * 8u;
        slot8 = slot * 8u;
        si = gld<unsigned>(base, slot * 4u + off_sinfo);
#pragma unroll
        for (int j = 0; j < 13; ++j) Kp[j] = gld<f2>(base, (unsigned)j * nslots8 + slot8);
        PQ = gld<f2>(base, slot8 + off_pq);
    }
    // ... landed: every value is touched, i.e. the compiler places its wait for them HERE (behind the terms, whose duration they had
    // to arrive) and not at the loop head, where it would wait for the raw samples as well
    __device__ __forceinline__ void landed()
    {
#pragma unroll
        for (int j = 0; j < 13; ++j) asm volatile("" : "+v"(Kp[j]));
        asm volatile("" : "+v"(PQ), "+v"(si));
    }
    __device__ __forceinline__ bool act() const { return (si >> 31) != 0; }
    // LDS byte addresses of the five rows of the window (idle lanes: the tile's first window -- zero taps, never written out)
    __device__ __forceinline__ void rows(unsigned plane_base, unsigned (&wrow)[5]) const
    {
        const unsigned w0 = plane_base + (si & 0xFFFu) * 8u;
#pragma unroll
        for (int r = 0; r < 5; ++r) wrow[r] = w0 + (unsigned)r * tp8;
    }
};

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
    float ydq, ywin;    // (dq and dq + e + 1) / 2^(qlog2 + 3): see the kernel's comment on bin edges in fixed point
    int qcap;           // (pixel, frame, fade) pairs a wave can list for the exact bin check
    uint8_t* force;     // [nframes] (optional) a workgroup whose list overflowed marks its frames: the caller re-evaluates them exactly
};

#ifndef AMT_LIN_OCC
#define AMT_LIN_OCC 4
#endif
#ifndef AMT_LIN_OCC16
#define AMT_LIN_OCC16 4
#endif
// NF fades (11 for AMTAnalyzeLogo, the only caller of this mode: no per-fade branches)
template <typename pix_t, int NF, bool BLEND>
__device__ __forceinline__ void logo_eval_linear_body(const LinLaunch& A)
{
    static_assert(NF >= 3 && NF <= 12, "the running sums hold rows 0..11 of the 16x16 accumulator");
    extern __shared__ float lds[];
    f2* const planes = reinterpret_cast<f2*>(lds);                 // [kLinWaves][kTileCap] a wave's own tile: {s, bg}
    float* const wacc = lds + kLinWaves * kTileCap * 2;            // [kLinWaves][G][48 lanes][4] a wave's running sums (kLinAccFrameBytes per frame)

    const int G = A.G;
    const int logo = blockIdx.x / A.ngroups;                       // (logo-major on purpose: eval_tiles.hpp, wg_map_shared_rows)
    const int grp = blockIdx.x - logo * A.ngroups;
    const int F0 = grp * G;
    const int gcount = min(G, A.nframes - F0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const EvalLogoDev* const Lp = A.logos + logo;
    const TileLogoDev* const Xp = A.tls + logo;
    const int ntl = Xp->ntlist;                                    // tiles that hold pixels
    const const_int_ptr tlist = (const_int_ptr)Xp->tlist;
    typedef const __attribute__((address_space(4))) TileLogoDev* tlogo_ptr;
    const sbase_t gLin = uniform_base(((tlogo_ptr)Xp)->lin);       // the logo's blob: taps at 0, then response lines, slot words, a, b
    const unsigned nslots8 = (unsigned)((tlogo_ptr)Xp)->nslots * 8u, off_pq = ((tlogo_ptr)Xp)->lin_pq, off_sinfo = ((tlogo_ptr)Xp)->lin_sinfo;
    const float floorResp = Xp->floorResp;                         // (wave-uniform) L: scale2 = min(1, r / L)
    const const_tile_ptr tiles = (const_tile_ptr)Xp->tiles;
    // Bin edges in fixed point.  Q = 2^qlog2 units per gray level; a bin is 8 gray levels = 2^(qlog2+3) units = one unit of y below:
    // y = (mean * Q + dq) / 2^(qlog2+3) with dq = e + 1, e = ceil(bin_eps * Q).  floor(y) is the bin of mean + dq / Q; an edge within bin_eps
    // ABOVE the mean has been crossed by y (fract(y) * 2^(qlog2+3) in [0, dq]), one within bin_eps BELOW it leaves fract(y) * 2^(qlog2+3)
    // in [dq - 1, dq + e + 1): so "fract(y) < ywin = (dq + e + 1) / 2^(qlog2+3)" flags every (pixel, fade) whose exact mean might fall in
    // another bin, and for all others floor(y) is the bin of the exact mean.  Scaling by powers of two is exact, so y comes straight out
    // of the multiply-add that interpolates the mean (the scale sits in its operands).
    // (dq, e and the two constants below are formed on the host: launch_logo_eval_linear)
    const float yscale = 0.125f;                                                     // mean (gray levels) -> bins
    const float ydq = A.ydq, ywin = A.ywin;

#ifdef AMT_LIN_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#define AMT_LTICK(k) do { const long long t_ = clock64(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define AMT_LTICK(k) do { } while (0)
#endif
    // the fades, wave-uniform: scalar register PAIRS {fade 2j, fade 2j + 1} (operands of the packed multiply-adds) and the last one
    static_assert(NF % 2 == 1, "the fades are walked in pairs plus the last one");
    constexpr int NP = (NF - 1) / 2;
    f2 fdp[NP];
    float fd[NF];
    {
        typedef const __attribute__((address_space(4))) float* const_float_ptr;
        const const_float_ptr fp = (const_float_ptr)(A.fades + A.fade0);
#pragma unroll
        for (int f = 0; f < NF; ++f) fd[f] = fp[f];
#pragma unroll
        for (int j = 0; j < NP; ++j) fdp[j] = f2{fd[2 * j], fd[2 * j + 1]};
    }
    float* const myacc = wacc + wave * G * (kLinAccFrameBytes / 4);
    for (int i = lane; i < G * (kLinAccFrameBytes / 4); i += 64) myacc[i] = 0.0f;
    const unsigned myacc_base = __builtin_amdgcn_readfirstlane(lds_address(myacc));
    // the wave's list of (pixel, frame, fade) pairs whose bin the exact mean must confirm: {slot | frame << 20 | fade << 23 | bin << 27, x}
    u2* const queues = reinterpret_cast<u2*>(wacc + kLinWaves * G * (kLinAccFrameBytes / 4));          // [kLinWaves][qcap]
    int* const qcount = reinterpret_cast<int*>(queues + kLinWaves * A.qcap);                          // [kLinWaves] pairs found (may exceed qcap: overflow)
    u2* const myqueue = queues + wave * A.qcap;
    int qn = 0;                                                  // (wave-uniform)

    f2* const myplane = planes + wave * kTileCap;
    const unsigned plane_base = lds_address(myplane);
    typedef LinStager<pix_t, BLEND> Stager;
    Stager st;
    st.init(A.pitch, A.maxv, myplane);
    LinPixel px;
    TileDesc T;
    const gptr_t gTab = table_base(gLin);

    // this wave's tiles: entries wave, wave + 4, ... of the logo's list of tiles.  (i, g) = (list position, frame) of an iteration
    auto advance = [&](int& i, int& g) {
        const bool last = g + 1 == gcount;
        g = last ? 0 : g + 1;
        i = last ? i + kLinWaves : i;
    };
    // Pipeline: while iteration i is evaluated, the raw samples of i + 1 travel (requested an iteration ago: hand-written loads, see above).
    //   A. window reads of i from the plane, means and correlations of s and bg
    //   B. the fades' bins; means next to a bin edge are listed for the exact check after the loop
    //   B'. a new tile next: its pixel's taps, slot word and response line are requested (the compiler's loads) ...
    //   T. the 11 terms, their sums over the quads of lanes, added to the wave's running sums of the frame in LDS.  No table is looked
    //      up: the scale of a term is a function of the bin number and two per-pixel constants (see "T." below)
    //   T'. ... and consumed: the compiler's wait for them sits here, where nothing else is in flight but the raw samples C needs next
    //   C. raw(i + 1) -> plane; request raw(i + 2)
    int i0 = wave, g0 = 0;                                       // iteration i
    int i1 = i0, g1 = 0;                                         // iteration i + 1 (its raw samples are in flight)
    int iu = -1;                                                 // the list position whose tile the staging units describe
    if (i0 < ntl) {
        const int t0 = tlist[i0];
        fetch_tile(T, tiles + t0);
        st.setup_units(Lp, Xp, T, lane);
        iu = i0;
        st.request(frame_desc<pix_t>(A.Y, A.frame_map, A.frame_stride, F0));
        px.load(gTab, nslots8, off_sinfo, off_pq, (unsigned)t0 * 64u + (unsigned)lane, T);
        px.landed();
        st.template landed<0>();
        st.convert();
        advance(i1, g1);
        {
            const bool more = i1 < ntl;
            if (more && i1 != iu) { fetch_tile(T, tiles + tlist[i1]); st.setup_units(Lp, Xp, T, lane); iu = i1; }
            st.request(frame_desc<pix_t>(A.Y, A.frame_map, A.frame_stride, F0 + (more ? g1 : g0)));
        }
    }
    if (i0 < ntl) for (;;) {
        const f2 PQ = px.PQ;                                       // (B' below overwrites the pixel; the terms are this pixel's)
        AMT_LTICK(0);
        // ---- A. ONE window evaluation for both operands: R = {corr(s), corr(bg)}, M = {mean(s), mean(bg)} ----
        // (the taps' {k, k} broadcasts live in the multiply-adds' op_sel: eval_tile_stage.h pk_fma_tap)
        f2 R, M;
        {
            unsigned wrow[5];
            px.rows(plane_base, wrow);
#ifdef AMT_LIN_NO_EVAL                                          // (ablations of the instrumented builds: wrong results, timing only)
            R = px.Kp[0]; M = px.Kp[1] + f2{100.0f, 120.0f};
#else
            window_eval_streamed<sizeof(pix_t) == 1>(wrow, px.Kp, M, R);         // idle lanes read the tile's first window: finite values, zero taps
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
        const f2 YD = f2{y0d, dy};
        float emin = 1.0f;
        f2 binp[NP];                                               // the bins of fades 2j and 2j + 1, 0..31
        // Fade 0 blends to s and fade 1 to bg exactly (0 * x + y == y), and M holds their means in the reference's own order (column
        // sums, hsum, /25: window_eval_streamed): those two bins are the reference's without any test.  (The caller guarantees that the
        // first fade is 0 and the last is 1.)  It matters: mean(s) is an integer / 25 and sits exactly ON a bin edge once in 200 pixels.
        const float binl = __builtin_amdgcn_fmed3f(__builtin_floorf(y1), 0.0f, 31.0f);          // the last fade's
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const f2 yy = pk_fma_hi_lo_s(fdp[j], YD);               // {fade 2j, fade 2j + 1}: fade * dy + y0d
            const float ya = j == 0 ? y0 : yy.x;                      // (fade 0: the exact mean's own bin; the packed lane is not used)
            if (j != 0) emin = __builtin_fminf(emin, __builtin_amdgcn_fractf(ya));
            emin = __builtin_fminf(emin, __builtin_amdgcn_fractf(yy.y));
            // (int)clamp(mean, 0, 255) >> 3; a NaN mean gives bin 0 like the reference's (int)NaN = INT_MIN
            binp[j] = f2{__builtin_amdgcn_fmed3f(__builtin_floorf(ya), 0.0f, 31.0f), __builtin_amdgcn_fmed3f(__builtin_floorf(yy.y), 0.0f, 31.0f)};
        }
        auto bin_of = [&](int f) -> float { return f == NF - 1 ? binl : ((f & 1) ? binp[f >> 1].y : binp[f >> 1].x); };
        // A mean within bin_eps of a bin edge (3e-4 of all (pixel, fade) pairs; one wave iteration in five has one): the term below is
        // formed with the TENTATIVE bin floor(y), and the pair is put on the wave's list {slot, frame, fade, tentative bin; x}.  When the
        // workgroup has finished, every listed pair is looked at with all threads at once: the mean evaluated exactly as the reference
        // does (from the frame itself), its bin, and -- where that differs -- the difference of the two terms as a correction to the
        // frame's sum.  (Rounds 2-5 re-evaluated the mean on the spot, for one or two lanes while sixty-two waited: 14 % of the kernel
        // and, through the registers its window took, the reason the loop could not carry more.)
#ifndef AMT_LIN_NO_FIXUP
        {
            const bool near_edge = px.act() && emin < ywin;
            if (__builtin_amdgcn_ballot_w64(near_edge) != 0) {         // wave-uniform
                const float R0q = R.x, dRq = R.y - R.x;
                const unsigned code0 = (px.slot8 >> 3) | ((unsigned)g0 << 20);
#pragma unroll
                for (int f = 1; f < NF - 1; ++f) {
                    const bool flag = near_edge && __builtin_amdgcn_fractf(__builtin_fmaf(fd[f], dy, y0d)) < ywin;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
                    if (m != 0) {
                        const unsigned pos = (unsigned)qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (flag && pos < (unsigned)A.qcap)
                            myqueue[pos] = u2{code0 | ((unsigned)f << 23) | ((unsigned)bin_of(f) << 27), __builtin_bit_cast(unsigned, __builtin_fmaf(fd[f], dRq, R0q))};
                        qn += __builtin_popcountll(m);
                    }
                }
            }
        }
#endif
        AMT_LTICK(3);
        // ---- B'. a new tile next: its pixel and taps are requested BEFORE the raw samples -- vector-memory loads return in order, and
        //      the taps are what the next iteration needs first ----
        const bool new_tile = i1 < ntl && i1 != i0;
        if (new_tile) {
            const int t1 = tlist[i1];
            TileDesc Tn;
            fetch_tile(Tn, tiles + t1);
            // (the lane number is re-derived where a tile begins: carried through the loop it is the 16-bit kernel's register 129)
            px.load(gTab, nslots8, off_sinfo, off_pq, (unsigned)t1 * 64u + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), Tn);
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
            const float R0 = R.x, dR = R.y - R.x;
            // Every lane of a quad ends up with all eleven quad sums; lane j < 3 of the quad keeps the four of fades 4j .. 4j + 3 and adds
            // them to its 16 bytes of the quad's cell: ONE 16-byte read-modify-write per iteration for the whole wave (a 16-byte LDS store
            // costs 13 cycles of the CU's store path whatever its exec mask: three of them by lane 0 alone were a fifth of the loop's LDS time)
            f4 mine;
            const f2 RR = f2{R0, dR};
            float term[NF + 1];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const f2 resp = pk_fma_hi_lo_v(binp[j], PQ);        // {Q * bin + P} of the two fades
                const f2 x = pk_fma_hi_lo_s(fdp[j], RR);            // {fade * dR + R0}
                const f2 md = f2{__builtin_amdgcn_fmed3f(x.x, -__builtin_fabsf(resp.x), __builtin_fabsf(resp.x)),
                                 __builtin_amdgcn_fmed3f(x.y, -__builtin_fabsf(resp.y), __builtin_fabsf(resp.y))};
                const f2 rc = f2{__builtin_amdgcn_rcpf(__builtin_fmaxf(__builtin_fabsf(resp.x), floorResp)),
                                 __builtin_amdgcn_rcpf(__builtin_fmaxf(__builtin_fabsf(resp.y), floorResp))};
                const f2 tt = md * rc;
                term[2 * j] = tt.x; term[2 * j + 1] = tt.y;
            }
            {
                const float resp = __builtin_fabsf(__builtin_fmaf(PQ.y, binl, PQ.x));
                const float x = __builtin_fmaf(fd[NF - 1], dR, R0);
                term[NF - 1] = __builtin_amdgcn_fmed3f(x, -resp, resp) * __builtin_amdgcn_rcpf(__builtin_fmaxf(resp, floorResp));
                term[NF] = 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float tq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 4 * q + r;
                    if (f < NF) {
                        float t = term[f];
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));   // quad_perm:[1,0,3,2]
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true));   // quad_perm:[2,3,0,1]
                        asm volatile("" : "+v"(t));       // (left alone the add sinks behind the selects below and its DPP operand stays a v_mov_b32_dpp: one more instruction per fade)
                        tq[r] = t;
                    } else tq[r] = 0.0f;
                }
                if (q == 0) mine = f4{tq[0], tq[1], tq[2], tq[3]};
                else {
                    const bool me = (lane & 3) == q;
                    mine = f4{me ? tq[0] : mine[0], me ? tq[1] : mine[1], me ? tq[2] : mine[2], me ? tq[3] : mine[3]};
                }
                __builtin_amdgcn_sched_barrier(0);          // (four fades at a time: left alone the scheduler forms all eleven sums first -- eleven registers too many)
            }
            if ((lane & 3) != 3) {
                const lds_quad cell = (lds_quad)(unsigned long long)(myacc_base + (unsigned)g0 * (unsigned)kLinAccFrameBytes + (unsigned)(lane >> 2) * 48u + (unsigned)(lane & 3) * 16u);
                cell[0] = cell[0] + mine;
            }
        }
#endif
        AMT_LTICK(4);
        if (new_tile) px.landed();
        AMT_LTICK(5);
        // ---- C. the next iteration's tile into the plane, the raw samples of the one after.  The request is issued on EVERY trip round the
        //      loop (past the wave's last iteration it repeats the previous one): the compiler counts the loads in flight per path, and a
        //      path without the six requests behind the taps would turn its wait for the taps at the loop head into a wait for everything ----
        if (i1 >= ntl) break;
        st.template landed<0>();
#ifndef AMT_LIN_NO_CONVERT
        st.convert();
#endif
        {
            int i2 = i1, g2 = g1;
            advance(i2, g2);
            const bool more = i2 < ntl;
            if (more && i2 != iu) {
                fetch_tile(T, tiles + tlist[i2]);
                st.setup_units(Lp, Xp, T, (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
                iu = i2;
            }
#if defined(AMT_LIN_RAW_SAMEFRAME)                              // (ablation: every request hits the cache -- what the raw loads' LATENCY costs)
            st.request(frame_desc<pix_t>(A.Y, A.frame_map, A.frame_stride, 0));
#else
            st.request(frame_desc<pix_t>(A.Y, A.frame_map, A.frame_stride, F0 + (more ? g2 : g1)));
#endif
        }
        AMT_LTICK(1);
        i0 = i1; g0 = g1;
        advance(i1, g1);
    }
    // the request of the last trip is never converted: its loads must have landed before their registers can mean anything else
    if (wave < ntl) st.template landed<0>();
    if (lane == 0) qcount[wave] = qn;
    __syncthreads();
#ifdef AMT_LIN_TIMING
    if (lane == 0 && logo == 0 && grp == A.ngroups / 2 && wave < 4) {      // a workgroup of logo 0 (the deint logo)
        long long* tb = reinterpret_cast<long long*>(A.out + (long long)A.nframes * A.out_frame_stride);      // host reserves room
        for (int k = 0; k < 8; ++k) tb[wave * 8 + k] = tacc[k];
    }
#endif
    // (the thread index is re-derived: kept across the loop it would be one register too many)
    const int tid_end = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // ---- the listed pairs, all threads at once: exact mean -> exact bin -> where it differs from the tentative one, the correction
    //      t(exact bin) - t(tentative bin) replaces the pair's x in the list ----
    bool overflow = false;
    for (int q = 0; q < kLinWaves; ++q) {
        const int nq = qcount[q];
        overflow = overflow || nq > A.qcap;
        for (int e = tid_end; e < min(nq, A.qcap); e += kLinWgThreads) {
            const u2 ent = queues[q * A.qcap + e];
            const unsigned slot = ent.x & 0xFFFFFu;
            const int g = (int)((ent.x >> 20) & 7u), f = (int)((ent.x >> 23) & 15u), tb = (int)(ent.x >> 27);
            const unsigned xbits = ent.y;              // (a scalar copy first: __builtin_bit_cast applied to the vector ELEMENT ent.y reads element 0 -- clang 19 / ROCm 7.2)
            const float x = __builtin_bit_cast(float, xbits);
            const unsigned pos = Xp->pos[slot];
            const int srcFrame = A.frame_map ? A.frame_map[F0 + g] : F0 + g;
            const pix_t* const frame = reinterpret_cast<const pix_t*>(A.Y) + (long long)srcFrame * A.frame_stride;
            const float mean = exact_blend_mean_from_frame<pix_t>(Lp, frame, A.pitch, A.maxv, (int)(pos & 0xFFFFu), (int)(pos >> 16), A.fades[A.fade0 + f]);
            const int eb = score_bin_dev(mean);
            float delta = 0.0f;
            if (eb != tb) {
                const float2 pq = Xp->pq[slot];
                const float re = __builtin_fabsf(__builtin_fmaf(pq.y, (float)eb, pq.x)), rt = __builtin_fabsf(__builtin_fmaf(pq.y, (float)tb, pq.x));
                delta = __builtin_amdgcn_fmed3f(x, -re, re) * __builtin_amdgcn_rcpf(__builtin_fmaxf(re, floorResp))
                      - __builtin_amdgcn_fmed3f(x, -rt, rt) * __builtin_amdgcn_rcpf(__builtin_fmaxf(rt, floorResp));
            }
            queues[q * A.qcap + e].y = __builtin_bit_cast(unsigned, delta);
        }
    }
    __syncthreads();
    // the waves' sums, in order: per wave the 16 partial sums of a fade, front to back, then the wave's corrections in list order
    if (tid_end < gcount * NF) {
        const int gg = tid_end / NF, f = tid_end - gg * NF;
        float r = 0.0f;
        for (int q = 0; q < kLinWaves; ++q) {
            const float* const cells = wacc + (q * G + gg) * (kLinAccFrameBytes / 4) + f;      // quad j: 12 floats at 12 j
            float rq = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) rq += cells[j * 12];
            const unsigned key = ((unsigned)gg << 20) | ((unsigned)f << 23);
            const int nq = min(qcount[q], A.qcap);
            for (int e = 0; e < nq; ++e) {
                const u2 ent = queues[q * A.qcap + e];
                const unsigned dbits = ent.y;
                if ((ent.x & 0x07F00000u) == key) rq += __builtin_bit_cast(float, dbits);
            }
            r += rq;
        }
        r = r / Lp->blackScore;
        if (A.take_abs) r = fabsf(r);
        A.out[(long long)(F0 + gg) * A.out_frame_stride + Lp->out_off + A.fade0 + f] = r;
    }
    // a list that did not take every pair: the group's frames are left to the exact kernel (the guard's list; amt_gpu.hip analyze_run)
    if (overflow && A.force != nullptr && tid_end < gcount) A.force[F0 + tid_end] = 1;
}

// A workgroup belongs to one logo: the deinterlaced logo's rows are [1 2 1] blends of three source rows (DeintY), a field logo's are
// the source rows themselves (CopyY) -- a third of the loads and none of the blend arithmetic: two instances of the loop.
template <typename pix_t>
__device__ __forceinline__ void logo_eval_linear_split(const LinLaunch& A)
{
    if (A.logos[blockIdx.x / A.ngroups].deint) logo_eval_linear_body<pix_t, 11, true>(A);
    else logo_eval_linear_body<pix_t, 11, false>(A);
}

// Four waves per SIMD (<= 128 registers) for both sample sizes.
// (Inside the loop a spilled register would be reloaded with a wait for EVERY load in flight -- the pipeline's whole point.)
__global__ __launch_bounds__(kLinWgThreads) __attribute__((amdgpu_waves_per_eu(AMT_LIN_OCC, AMT_LIN_OCC)))
void logo_eval_linear_kernel(const LinLaunch A) { logo_eval_linear_split<uint8_t>(A); }
__global__ __launch_bounds__(kLinWgThreads) __attribute__((amdgpu_waves_per_eu(AMT_LIN_OCC16, AMT_LIN_OCC16)))
void logo_eval_linear_kernel16(const LinLaunch A) { logo_eval_linear_split<uint16_t>(A); }

hipError_t launch_logo_eval_linear(hipStream_t st, int bits, const EvalLogoDev* dlogos, const TileLogoDev* dtls, int nlogos,
                                   const float* dfades, int nfades, int fade0, const void* dY,
                                   const int* dframe_map, long long frame_stride_elems, int pitch, int nframes, int G, float* dout,
                                   int out_frame_stride, int take_abs, float bin_eps, int qlog2, int qcap, uint8_t* dforce)
{
    if (qcap < 1 || qcap > 4096) return hipErrorInvalidValue;
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
    {
        const float qscale = std::ldexp(1.0f, qlog2);
        const int qe = (int)std::ceil(bin_eps * qscale), dq = qe + 1;
        A.ydq = std::ldexp((float)dq, -(qlog2 + 3));
        A.ywin = std::ldexp((float)(dq + qe + 1), -(qlog2 + 3));
    }
    A.qcap = qcap; A.force = dforce;
    const size_t lds = (size_t)kLinWaves * kTileCap * 2 * sizeof(float) + (size_t)kLinWaves * G * kLinAccFrameBytes + (size_t)kLinWaves * qcap * 8 + (size_t)kLinWaves * sizeof(int);
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
