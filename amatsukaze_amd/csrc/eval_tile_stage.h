// eval_tile_stage.h -- device side of the tile plans (eval_tiles.hpp), shared by the tile kernels (eval_pair_kernels.hip,
// eval_linear_kernels.hip): a wave stages ITS OWN tile -- the bounding box of the 5x5 windows of its 64 mask pixels -- for one
// frame into an LDS plane nobody else touches, as {s, bg = a*s + b*maxv} pairs (LogoScan.hpp:244-251), and reads its pixels'
// windows from it.  LDS operations of one wave complete in order: no barrier between staging and evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "engine.hpp"
#include "eval_tiles.hpp"
#include "exact_math.h"
#include "eval_lds_stage.h"

namespace amt {

namespace tile {

using namespace lin;

// The 5x5 window of one pixel: 25 separate 8-byte LDS reads (merged into ds_read2_b64 they would take twice the LDS cycles and
// fall under a different bank map than the one the tile pitch was chosen for -- MI355X_MICROARCH.md, LDS table), issued in one go,
// row by row; wrow[r] = LDS byte address of the window's row r.  The compiler does not count these reads, so the waits are placed
// here: LDS operations of a wave complete in order, and window_rows_ready<N>() returns when at most N of them are outstanding.  Its
// in/out operands tie the rows it releases to the instructions that consume them.  (No scalar load is in flight at this point of
// the loops -- their consumers precede the evaluation -- so lgkmcnt counts LDS operations only.)
__device__ __forceinline__ void window_reads(const unsigned (&wrow)[5], f2 (&W)[25])
{
#pragma unroll
    for (int r = 0; r < 5; ++r)
        asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:8\n\tds_read_b64 %2, %5 offset:16\n\t"
                     "ds_read_b64 %3, %5 offset:24\n\tds_read_b64 %4, %5 offset:32"
                     : "=&v"(W[5 * r]), "=&v"(W[5 * r + 1]), "=&v"(W[5 * r + 2]), "=&v"(W[5 * r + 3]), "=&v"(W[5 * r + 4])
                     : "v"(wrow[r]) : "memory");
}
template <int OUTSTANDING, int ROW0, int NROWS>
__device__ __forceinline__ void window_rows_ready(f2 (&W)[25])
{
    static_assert(NROWS == 1 || NROWS == 2, "one or two rows per wait");
    if (NROWS == 2)
        asm volatile("s_waitcnt lgkmcnt(%10)"
                     : "+v"(W[5 * ROW0]), "+v"(W[5 * ROW0 + 1]), "+v"(W[5 * ROW0 + 2]), "+v"(W[5 * ROW0 + 3]), "+v"(W[5 * ROW0 + 4]),
                       "+v"(W[5 * ROW0 + 5]), "+v"(W[5 * ROW0 + 6]), "+v"(W[5 * ROW0 + 7]), "+v"(W[5 * ROW0 + 8]), "+v"(W[5 * ROW0 + 9])
                     : "n"(OUTSTANDING) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%5)"
                     : "+v"(W[5 * ROW0]), "+v"(W[5 * ROW0 + 1]), "+v"(W[5 * ROW0 + 2]), "+v"(W[5 * ROW0 + 3]), "+v"(W[5 * ROW0 + 4])
                     : "n"(OUTSTANDING) : "memory");
}
// window reads + {mean(s), mean(bg)} in the reference's order: the column sums ((r0+r1)+(r2+r3))+r4 (ComputeKernel.cpp:88-94) start
// as the rows arrive, then hsum256_ps' order and /25 (ComputeKernel.cpp:54-74,98)
__device__ __forceinline__ f2 window_means_as_rows_land(f2 (&W)[25]);
__device__ __forceinline__ f2 window_load_means(const unsigned (&wrow)[5], f2 (&W)[25])
{
    window_reads(wrow, W);
    return window_means_as_rows_land(W);
}
// ... the second half on its own, for callers that issue window_reads() earlier (and nothing but vector-memory requests in between)
__device__ __forceinline__ f2 window_means_as_rows_land(f2 (&W)[25])
{
    f2 c01[5], c[5];
    window_rows_ready<15, 0, 2>(W);
#pragma unroll
    for (int i = 0; i < 5; ++i) c01[i] = W[i] + W[5 + i];
    window_rows_ready<5, 2, 2>(W);
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = c01[i] + (W[10 + i] + W[15 + i]);
    window_rows_ready<0, 4, 1>(W);
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = c[i] + W[20 + i];
    return div25_pk(((c[0] + c[4]) + c[2]) + (c[1] + c[3]));
}

// The linear kernel's evaluation of a window, streamed: {mean(s), mean(bg)} in the reference's order as above and
// {corr(k, s), corr(k, bg)} = sum_i k_i w_i - mean * sum_i k_i as two FMA chains (even taps, odd taps),
// consumed as the reads land -- at most 15 of the 25 window elements are in registers at a time (two LDS round trips).
__device__ __forceinline__ void window_rows_issue2(unsigned a0, unsigned a1, f2 (&r)[10])
{
    asm volatile("ds_read_b64 %0, %10\n\tds_read_b64 %1, %10 offset:8\n\tds_read_b64 %2, %10 offset:16\n\t"
                 "ds_read_b64 %3, %10 offset:24\n\tds_read_b64 %4, %10 offset:32\n\t"
                 "ds_read_b64 %5, %11\n\tds_read_b64 %6, %11 offset:8\n\tds_read_b64 %7, %11 offset:16\n\t"
                 "ds_read_b64 %8, %11 offset:24\n\tds_read_b64 %9, %11 offset:32"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9])
                 : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void window_row_issue1(unsigned a0, f2 (&r)[5])
{
    asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:8\n\tds_read_b64 %2, %5 offset:16\n\t"
                 "ds_read_b64 %3, %5 offset:24\n\tds_read_b64 %4, %5 offset:32"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]) : "v"(a0) : "memory");
}
template <int OUTSTANDING> __device__ __forceinline__ void rows_ready2(f2 (&r)[10])
{
    asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                 "+v"(r[8]), "+v"(r[9]) : "n"(OUTSTANDING) : "memory");
}
template <int OUTSTANDING> __device__ __forceinline__ void row_ready1(f2 (&r)[5])
{
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : "n"(OUTSTANDING) : "memory");
}
// acc + {k, k} * w with k = the low (tap 2j) or the high (tap 2j + 1) half of a tap pair, the broadcast done by the instruction's op_sel.
// Written as instructions: as IR the broadcast is a shufflevector of a loop-carried value that the optimiser moves to where the pair is
// defined (50 registers of {k, k} copies instead of 13 pairs); the old remedy -- an empty asm that redefines the pairs every iteration --
// cost 26 v_mov_b64 per iteration (the register allocator kept the redefined pairs apart from the loop-carried ones).
__device__ __forceinline__ f2 pk_fma_tap(int hi, bool first, f2 kp, f2 w, f2 acc)
{
    f2 r;
    if (first) {
        if (hi) asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(kp), "v"(w));
        else asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(kp), "v"(w));
    } else {
        if (hi) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(kp), "v"(w), "v"(acc));
        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(kp), "v"(w), "v"(acc));
    }
    return r;
}
// The same evaluation with all 25 window reads issued at once (one LDS round trip; 50 registers of window instead of 30 -- the linear
// kernel has them since its taps' broadcasts moved into the multiply-adds).  Same operations in the same order: identical results.
__device__ __forceinline__ void window_eval_one_trip(const unsigned (&wrow)[5], const f2 (&Kp)[13], f2& M, f2& R)
{
    f2 W[25];
    window_reads(wrow, W);
    f2 acc0 = {0.0f, 0.0f}, acc1 = acc0, c[5];
    auto mac = [&](int e) {
        if (e & 1) acc1 = pk_fma_tap(1, e == 1, Kp[e >> 1], W[e], acc1);
        else acc0 = pk_fma_tap(0, e == 0, Kp[e >> 1], W[e], acc0);
    };
    window_rows_ready<15, 0, 2>(W);
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = W[i] + W[5 + i];
#pragma unroll
    for (int e = 0; e < 10; ++e) mac(e);
    window_rows_ready<5, 2, 2>(W);
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = c[i] + (W[10 + i] + W[15 + i]);
#pragma unroll
    for (int e = 10; e < 20; ++e) mac(e);
    window_rows_ready<0, 4, 1>(W);
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = c[i] + W[20 + i];
#pragma unroll
    for (int e = 20; e < 25; ++e) mac(e);
    M = div25_pk(((c[0] + c[4]) + c[2]) + (c[1] + c[3]));
    const f2 sum = acc0 + acc1;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(R) : "v"(Kp[12]), "v"(M), "v"(sum));
}
// ONE_TRIP: 50 registers of window at once (the 8-bit kernel has them); otherwise THREE round trips -- rows 0-1, rows 2-3, row 4 -- with
// at most 20 (the 16-bit kernel, whose raw samples in flight take twice the room).  Same operations in the same order: identical results.
template <bool ONE_TRIP>
__device__ __forceinline__ void window_eval_streamed(const unsigned (&wrow)[5], const f2 (&Kp)[13], f2& M, f2& R)
{
    if (ONE_TRIP) { window_eval_one_trip(wrow, Kp, M, R); return; }
    f2 ra[10];
    window_rows_issue2(wrow[0], wrow[1], ra);
    f2 acc0 = {0.0f, 0.0f}, acc1 = acc0, c[5];
    auto mac = [&](int e, f2 wv) {
        if (e & 1) acc1 = pk_fma_tap(1, e == 1, Kp[e >> 1], wv, acc1);
        else acc0 = pk_fma_tap(0, e == 0, Kp[e >> 1], wv, acc0);
    };
    rows_ready2<0>(ra);
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = ra[i] + ra[5 + i];
#pragma unroll
    for (int e = 0; e < 10; ++e) mac(e, ra[e]);
    window_rows_issue2(wrow[2], wrow[3], ra);    // (into the registers rows 0 and 1 have left)
    rows_ready2<0>(ra);
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = c[i] + (ra[i] + ra[5 + i]);
#pragma unroll
    for (int e = 0; e < 10; ++e) mac(10 + e, ra[e]);
    f2 r4[5];
    window_row_issue1(wrow[4], r4);
    row_ready1<0>(r4);
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = c[i] + r4[i];
#pragma unroll
    for (int e = 0; e < 5; ++e) mac(20 + e, r4[e]);
    M = div25_pk(((c[0] + c[4]) + c[2]) + (c[1] + c[3]));
    {   // R = acc0 + acc1 - {sum k, sum k} * M   (sum k = the high half of the last pair)
        const f2 sum = acc0 + acc1;
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(R) : "v"(Kp[12]), "v"(M), "v"(sum));
    }
}

// Four adjacent samples of a source row as they come out of memory, and the unit's four s values: the sample itself, or DeintY's
// (r0 + 2 r1 + r2 + 2) / 4.0f (LogoScan.hpp:763-780) -- an integer sum below 2^24 times 0.25, the reference's value bit for bit.
// bias = the blend's + 2; a row that is NOT blended (the logo's first / last row under DeintY, every row of a field logo under
// CopyY, :782-790) loads r0 = r2 = r1 and has bias 0: 4 r1 / 4 = r1 -- one code path, no branch.
template <typename pix_t> struct Quad;
template <> struct Quad<uint8_t> {
    unsigned v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff) { v = __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0); }
    // two samples per instruction: bytes 0, 2 and bytes 1, 3 spread over 16-bit lanes (sums <= 1022)
    static __device__ __forceinline__ void blend(const Quad& r0, const Quad& r1, const Quad& r2, unsigned bias, float (&s)[4])
    {
        const unsigned e0 = __builtin_amdgcn_perm(r0.v, r0.v, 0x0C020C00u), o0 = __builtin_amdgcn_perm(r0.v, r0.v, 0x0C030C01u);
        const unsigned e1 = __builtin_amdgcn_perm(r1.v, r1.v, 0x0C020C00u), o1 = __builtin_amdgcn_perm(r1.v, r1.v, 0x0C030C01u);
        const unsigned e2 = __builtin_amdgcn_perm(r2.v, r2.v, 0x0C020C00u), o2 = __builtin_amdgcn_perm(r2.v, r2.v, 0x0C030C01u);
        // (two instructions per sum -- v_lshl_add_u32, v_add3_u32 -- where the compiler's own association of
        //  ((e1 << 1) + e0) + (e2 + bias) takes three: 1 % of the scan kernel)
        unsigned te, to;
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(te) : "v"(e1), "v"(e0));
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(to) : "v"(o1), "v"(o0));
        const unsigned se = te + e2 + bias;
        const unsigned so = to + o2 + bias;
        // (ldexp, not a multiply by 0.25: the vectoriser would pair the multiplies and the pairs {s0,s1}, {s2,s3} then need four
        //  moves into the {s, bg} order of the LDS store)
        s[0] = __builtin_amdgcn_ldexpf((float)(se & 0xFFFFu), -2);
        s[1] = __builtin_amdgcn_ldexpf((float)(so & 0xFFFFu), -2);
        s[2] = __builtin_amdgcn_ldexpf((float)(se >> 16), -2);
        s[3] = __builtin_amdgcn_ldexpf((float)(so >> 16), -2);
    }
    static constexpr unsigned kBias = 0x00020002u;
    // a row that is never blended (CopyY, LogoScan.hpp:782-790): the four samples as they are
    static __device__ __forceinline__ void copy(const Quad& r1, float (&s)[4])
    {
        s[0] = (float)(r1.v & 0xFFu); s[1] = (float)((r1.v >> 8) & 0xFFu); s[2] = (float)((r1.v >> 16) & 0xFFu); s[3] = (float)(r1.v >> 24);   // v_cvt_f32_ubyte0..3
    }
};
template <> struct Quad<uint16_t> {
    u2 v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff) { v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0); }
    static __device__ __forceinline__ void blend(const Quad& r0, const Quad& r1, const Quad& r2, unsigned bias, float (&s)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned a = (r0.v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, b = (r1.v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu,
                           c = (r2.v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            s[k] = __builtin_amdgcn_ldexpf((float)(((b << 1) + a) + (c + bias)), -2);
        }
    }
    static constexpr unsigned kBias = 2u;
    static __device__ __forceinline__ void copy(const Quad& r1, float (&s)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = (float)((r1.v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
    }
};

typedef const __attribute__((address_space(4))) TileDesc* const_tile_ptr;          // constant address space: scalar loads
typedef const __attribute__((address_space(4))) int* const_int_ptr;
__device__ __forceinline__ void fetch_tile(TileDesc& D, const_tile_ptr t)
{
    D.x0 = t->x0; D.y0 = t->y0; D.nrows = t->nrows; D.ncol4 = t->ncol4; D.tp = t->tp; D.npix = t->npix; D.rcp = t->rcp;
}

// One wave's staging state.  Unit = one tile row x four columns; a tile has at most 64 * kTileUnits of them.  Lanes beyond the tile's
// last unit repeat it (same loads, same values stored to the same place).  No branch DEFINES these registers: a value that is only
// conditionally loaded gets copied at the join, and the copy waits for the load right behind its issue -- the prefetch distance
// would be gone.  Only the conversion of the units beyond the first 64 is skipped for small tiles.
// AB_LDS: the units' logo coefficients live in a second LDS plane of the wave ({a, b*maxv} at the unit's tile offset) instead of in
// 16 registers -- for kernels that are short of registers.
// SLIM: the byte offsets of the rows above / below a unit are re-derived at every request (4 more instructions) instead of kept (4
// registers), and "this row is blended" rides in the sign bit of the unit's LDS offset instead of in a register of its own.
// BLEND = false: a logo whose rows are never blended (field logos: CopyY) -- one row load per unit instead of three, no [1 2 1] sums.
template <typename pix_t, bool AB_LDS = false, bool SLIM = false, bool BLEND = true> struct TileStager {
    static constexpr int ES = (int)sizeof(pix_t);
    // (wave-uniform)
    int w, h, deint, srow0, srow_step, scol0, pitchB;
    float maxv;
    gptr_t gA, gB;
    f2* plane;
    f2* abplane;                     // AB_LDS only
    bool second_pass;                // the tile has more than 64 units
    // (per lane)
    int ulds[kTileUnits];            // pair offset in the tile plane
    int ug[kTileUnits][3];           // byte offsets in a frame of the rows above / at / below the unit
    unsigned ubias[kTileUnits];
    f4 ua[kTileUnits], ubmv[kTileUnits];                        // the unit's logo coefficients: a, b * maxv
    Quad<pix_t> raw[kTileUnits][3];

    __device__ __forceinline__ void init(const EvalLogoDev* Lp, int pitch, float maxv_, f2* plane_, f2* abplane_ = nullptr)
    {
        abplane = abplane_;
        w = Lp->w; h = Lp->h; deint = Lp->deint;
        srow0 = Lp->imgy + Lp->row0; srow_step = Lp->row_step; scol0 = Lp->imgx;
        gA = (gptr_t)Lp->a; gB = (gptr_t)Lp->b;
        pitchB = pitch * ES; maxv = maxv_; plane = plane_;
        second_pass = false;
    }
    __device__ __forceinline__ void setup_units(const TileDesc& T, int lane)
    {
        second_pass = T.nrows * T.ncol4 > 64;
#pragma unroll
        for (int k = 0; k < kTileUnits; ++k) {
            const TileUnit U = tile_unit(T, lane + 64 * k, w);
            const bool blend = BLEND && deint && U.y > 0 && U.y < h - 1;       // DeintY copies the first and the last row (LogoScan.hpp:763-780)
            ulds[k] = SLIM ? (U.lds | (blend ? (int)0x80000000 : 0)) : U.lds;
            if (!SLIM) ubias[k] = blend ? Quad<pix_t>::kBias : 0u;
            ug[k][1] = (srow0 + U.y * srow_step) * pitchB + (scol0 + U.xs) * ES;
            if (!SLIM) {
                ug[k][0] = blend ? ug[k][1] - pitchB : ug[k][1];
                ug[k][2] = blend ? ug[k][1] + pitchB : ug[k][1];
            }
            typedef f4 __attribute__((aligned(8))) f4a8;
            const f4 av = gld<f4a8>(gA, (unsigned)(U.y * w + U.xs) * 4u);
            const f4 bmv = gld<f4a8>(gB, (unsigned)(U.y * w + U.xs) * 4u) * maxv;     // rounded once, exactly as in a*s + b*maxv
            if (AB_LDS) {
                f4* d = reinterpret_cast<f4*>(abplane + U.lds);
                d[0] = f4{av[0], bmv[0], av[1], bmv[1]};
                d[1] = f4{av[2], bmv[2], av[3], bmv[3]};
            } else {
                ua[k] = av; ubmv[k] = bmv;
            }
        }
    }
    // frame = buffer descriptor of the source frame (whole plane: 32-bit byte offsets, checked on the host)
    __device__ __forceinline__ void request(const __amdgpu_buffer_rsrc_t frame)
    {
#pragma unroll
        for (int k = 0; k < kTileUnits; ++k) {
            if (!BLEND) {
                raw[k][1].load(frame, ug[k][1]);
            } else if (SLIM) {
                const int d = (ulds[k] >> 31) & pitchB;           // a blended row: the rows above and below; otherwise the row itself
                raw[k][0].load(frame, ug[k][1] - d);
                raw[k][1].load(frame, ug[k][1]);
                raw[k][2].load(frame, ug[k][1] + d);
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) raw[k][j].load(frame, ug[k][j]);
            }
        }
    }
    // raw samples -> {s, bg = a*s + b*maxv} pairs (LogoScan.hpp:247)
    __device__ __forceinline__ void convert_unit(int k)
    {
        float sv[4];
        const unsigned bias = SLIM ? ((unsigned)(ulds[k] >> 31) & Quad<pix_t>::kBias) : ubias[k];
        const int lds = SLIM ? (ulds[k] & 0x7FFFFFFF) : ulds[k];
        if (BLEND) Quad<pix_t>::blend(raw[k][0], raw[k][1], raw[k][2], bias, sv);
        else Quad<pix_t>::copy(raw[k][1], sv);
        f2* dst = plane + lds;
        if (AB_LDS) {
            const f4* c = reinterpret_cast<const f4*>(abplane + lds);
            const f4 c0 = c[0], c1 = c[1];
            reinterpret_cast<f4*>(dst)[0] = f4{sv[0], c0[0] * sv[0] + c0[1], sv[1], c0[2] * sv[1] + c0[3]};
            reinterpret_cast<f4*>(dst)[1] = f4{sv[2], c1[0] * sv[2] + c1[1], sv[3], c1[2] * sv[3] + c1[3]};
        } else {
            reinterpret_cast<f4*>(dst)[0] = f4{sv[0], ua[k][0] * sv[0] + ubmv[k][0], sv[1], ua[k][1] * sv[1] + ubmv[k][1]};
            reinterpret_cast<f4*>(dst)[1] = f4{sv[2], ua[k][2] * sv[2] + ubmv[k][2], sv[3], ua[k][3] * sv[3] + ubmv[k][3]};
        }
    }
    __device__ __forceinline__ void convert()
    {
        convert_unit(0);
        if (second_pass) {
#pragma unroll
            for (int k = 1; k < kTileUnits; ++k) convert_unit(k);
        }
    }
};

template <typename pix_t>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const void* Y, const int* frame_map, long long frame_stride, int frame)
{
    const int srcFrame = frame_map ? ((const_int_ptr)frame_map)[frame] : frame;
    const pix_t* src = reinterpret_cast<const pix_t*>(Y) + (long long)srcFrame * frame_stride;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<pix_t*>(src), 0, 0x7FFFFFFF, 0x00020000);
}

// this lane's mask pixel of a tile: its window in the tile plane, its index in the band's score row, its taps
struct TilePixel {
    f2 Kp[13];
    unsigned w0;                     // LDS byte address of the window's first element
    unsigned tp8;                    // (wave-uniform) the tile's row pitch in bytes
    int ridx;
    bool act;
    unsigned slot8;
    unsigned slotbase8;              // (wave-uniform) slot8 of lane 0: slot8 == slotbase8 + 8 * lane, for kernels short of registers
    __device__ __forceinline__ void load(const TileLogoDev* Xp, unsigned slot, const TileDesc& T, unsigned plane_base)
    {
        const gptr_t gK = (gptr_t)Xp->kp, gInfo = (gptr_t)Xp->sinfo;
        const unsigned nslots8 = (unsigned)Xp->nslots * 8u;
        const unsigned si = gld<unsigned>(gInfo, slot * 4u);
        w0 = plane_base + (si & 0xFFFu) * 8u;                     // idle lanes: the tile's first window (zero taps, never written out)
        tp8 = (unsigned)T.tp * 8u;
        ridx = (int)((si >> 12) & 0xFFFu);
        act = (si >> 31) != 0;
        slot8 = slot * 8u;
        slotbase8 = __builtin_amdgcn_readfirstlane(slot8);
#pragma unroll
        for (int j = 0; j < 13; ++j) Kp[j] = gld<f2>(gK, (unsigned)j * nslots8 + slot8);
    }
    // LDS byte addresses of the five rows of the window
    __device__ __forceinline__ void rows(unsigned (&wrow)[5]) const
    {
#pragma unroll
        for (int r = 0; r < 5; ++r) wrow[r] = w0 + (unsigned)r * tp8;
    }
};

__device__ __forceinline__ unsigned lds_address(const void* p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

} // namespace tile

} // namespace amt
