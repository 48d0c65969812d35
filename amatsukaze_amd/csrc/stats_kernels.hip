// stats_kernels.hip -- whole-frame field-difference / combing metrics (self-specified).
//
// Streaming reduction over the Y plane: every byte of every frame is read from HBM once.  A thread owns a 16-byte-wide column of a
// tile of 24 rows (8-bit samples; 16 rows at 16 bits) for a RUN of consecutive frames ((tile, column) pairs are dealt densely to the threads of the grid), so the vertical
// neighbours (rows y-1, y+1) and the previous frame's rows are all in the thread's own registers -- no LDS staging, no re-reads
// except the two halo rows per tile.  Loads are 16 B per lane, 64 lanes = 1 KiB contiguous per row.  The per-byte work is done four
// pixels at a time with v_sad_u8 / v_lerp_u8 (two per instruction for 16-bit samples).
//
// What bounds it (profiles/r04_notes.md): with the whole device the kernel moves 6.1 TB/s of actual traffic (1.15x its algorithmic
// bytes: 128-byte lines against a 1472-byte pitch, halo rows, the frame before a run) -- HBM.  Given a PART of the device
// (hipExtStreamCreateWithCUMask, to run beside the VALU-bound logo kernels) its rate scales with the CUs it owns: per CU it is bound
// by the vector ALU (~780 instructions per wave and frame in round 3's form, 4.5 cycles each) and by the bytes a CU keeps in flight.
// Hence the lean form below: raw buffer loads (one VGPR of address for the whole tile, rows outside the frame come back as zeros from
// the bounds check -- no 64-bit address arithmetic, no selects), the even-row vertical detail of the previous frame carried over
// instead of recomputed, the duplicate of the odd rows' term dropped, wave sums on DPP instead of LDS permutes.
//
// Per frame n (prev = frame n-1; rows 1..H-2 for the vertical metrics), all sums of absolute values:
//   0 DIFF_TOP   sum_{y even} |Y_n[y] - Y_prev[y]|          3 COMB       sum |Y_n[y] - avg(Y_n[y-1], Y_n[y+1])|
//   1 DIFF_BOT   sum_{y odd}  |Y_n[y] - Y_prev[y]|          4 COMB_PREV  same on the weave (even rows of n, odd rows of prev)
//   2 VERT_SAME  sum |Y_n[y-1] - Y_n[y+1]|                  5 SUM        sum Y_n
//   6 VERT_PREV  VERT_SAME of that weave                    7 reserved (0)
// avg(a,c) = (a + c) >> 1 per sample.
#include "build_knobs.h"
#include <hip/hip_runtime.h>
#include <cstdint>

namespace amt {

// (tried and closed, profiles/r04_notes.md: waves of a workgroup stacked on vertically adjacent tiles -- the halo re-reads already hit
//  the XCD's L2 thanks to the tile order below -- and a third row set with the next frame's loads issued before the evaluation)
constexpr int kStatThreads = 128;
#ifndef AMT_STATS_ROWS
#define AMT_STATS_ROWS 16
#endif
#ifndef AMT_STATS_RUN
#define AMT_STATS_RUN 32
#endif
#ifndef AMT_STATS_COLB
#define AMT_STATS_COLB 16
#endif
#ifndef AMT_STATS_LEAN
#define AMT_STATS_LEAN 1
#endif
#ifndef AMT_STATS_PINGPONG
#define AMT_STATS_PINGPONG 1
#endif
#ifndef AMT_STATS_DEAL
#define AMT_STATS_DEAL 0       /* 0: (tile, column) pairs dealt densely over the threads; 1: wave-segment dealing (see the kernel) */
#endif
#ifndef AMT_STATS_NT
#define AMT_STATS_NT 2         /* cache-policy bits of the loads of rows no other tile reads: 2 = nt (non-temporal) */
#endif
constexpr int kStatTileRows = AMT_STATS_ROWS;          // rows of a tile for 16-bit samples (two halo rows per tile are read twice)
#ifndef AMT_STATS_ROWS8
#define AMT_STATS_ROWS8 24
#endif
// 8-bit samples: 24 rows (halo 2 / 24 instead of 2 / 16 of the traffic; 244 VGPRs.  At 16 bits the same tile needs 256-264 and drops
// to one wave per SIMD: measured 2.879 -> 2.818 ms at 8 bits, 2.077 -> 2.121 at 10 -- profiles/r04_notes.md section 4)
constexpr int kStatTileRows8 = AMT_STATS_ROWS8;
constexpr int kStatTileRowsPlain = 8;          // the plain-load fallback (BUF = false, rare geometries): byte-wise tails cost registers
template <int ES, bool BUF = true> constexpr int stat_tile_rows() { return !BUF ? kStatTileRowsPlain : ES == 1 ? kStatTileRows8 : kStatTileRows; }
constexpr int kStatRun = AMT_STATS_RUN;          // frames a workgroup walks through (the frame before a run is its one re-read: 1/32)
constexpr int kStatXcds = 8;          // MI355X: 8 XCDs, workgroups are dealt to them round-robin by linear workgroup id
constexpr int kStatWords = 8;
constexpr int kStatColBytes = AMT_STATS_COLB;      // bytes of a row one lane owns: 16 (one dwordx4 load) or 8 (dwordx2: half the registers per row)
constexpr int kStatColWords = kStatColBytes / 4;

template <int ES> struct Px;
template <> struct Px<1> {
    static __device__ __forceinline__ unsigned sad(unsigned a, unsigned b, unsigned acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
    static __device__ __forceinline__ unsigned avg(unsigned a, unsigned c) { return __builtin_amdgcn_lerp(a, c, 0u); }
};
template <> struct Px<2> {
    static __device__ __forceinline__ unsigned sad(unsigned a, unsigned b, unsigned acc) { return __builtin_amdgcn_sad_u16(a, b, acc); }
    // floor((a+c)/2) in each 16-bit half without carries crossing
    static __device__ __forceinline__ unsigned avg(unsigned a, unsigned c)
    {
        return ((a >> 1) & 0x7FFF7FFFu) + ((c >> 1) & 0x7FFF7FFFu) + (a & c & 0x00010001u);
    }
};

struct alignas(kStatColBytes) Chunk { unsigned w[kStatColWords]; };

__device__ __forceinline__ Chunk chunk_zero()
{
    Chunk c;
#pragma unroll
    for (int i = 0; i < kStatColWords; ++i) c.w[i] = 0;
    return c;
}
__device__ __forceinline__ Chunk load_chunk(const uint8_t* p, int nvalid)
{
    if (nvalid >= kStatColBytes) return *reinterpret_cast<const Chunk*>(p);
    // (the ragged last column: byte by byte into a scratch array that never escapes -- a dynamically indexed member of the Chunk that
    // is returned would keep every row set out of registers: the compiler then parks them in LDS, measured 2x slower)
    uint32_t w[kStatColWords];
#pragma unroll
    for (int i = 0; i < kStatColWords; ++i) w[i] = 0;
    for (int i = 0; i < nvalid; ++i) w[i >> 2] |= (uint32_t)p[i] << ((i & 3) * 8);
    Chunk c;
#pragma unroll
    for (int i = 0; i < kStatColWords; ++i) c.w[i] = w[i];
    return c;
}

template <int ES> __device__ __forceinline__ unsigned sad16(const Chunk& a, const Chunk& b, unsigned acc)
{
#pragma unroll
    for (int i = 0; i < kStatColWords; ++i) acc = Px<ES>::sad(a.w[i], b.w[i], acc);
    return acc;
}
template <int ES> __device__ __forceinline__ Chunk avg16(const Chunk& a, const Chunk& c)
{
    Chunk r;
#pragma unroll
    for (int i = 0; i < kStatColWords; ++i) r.w[i] = Px<ES>::avg(a.w[i], c.w[i]);
    return r;
}

// sum over the 64 lanes of a wave, result in lane 63: row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast15 / row_bcast31
// across the rows -- six DPP adds on the VALU, no LDS permute and no wait
__device__ __forceinline__ unsigned wave_sum_to_lane63(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8   -> lane 15 of every row holds the row's sum
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
    return v;
}

#ifdef AMT_STATS_WAVES
#define AMT_STATS_OCC __attribute__((amdgpu_waves_per_eu(AMT_STATS_WAVES, AMT_STATS_WAVES)))
#else
#define AMT_STATS_OCC
#endif
// RAGGED: the row is not a whole number of lane columns (the last column's tail bytes are masked); the common widths -- multiples of 16
// bytes -- take the version without the masks
// BUF: rows come in through raw buffer loads whose bounds check supplies the zeros (the lean form below).  It needs every 16-byte
// column of a row to end inside the row's pitch -- a ragged row in an unpadded pitch (W = 362, pitch = 362) would have its last column
// of the bottom row straddle the end of the buffer and lose its valid bytes; such geometries take the plain loads (BUF = false).
template <int ES, bool RAGGED, bool BUF>
__global__ __launch_bounds__(kStatThreads) AMT_STATS_OCC
void frame_stats_kernel(const uint8_t* __restrict__ Y, long long frame_stride /*bytes*/, int pitch_bytes, int row_bytes, int H,
                        const uint8_t* __restrict__ prevY /* frame before the batch or null */, int nframes, int col_groups,
                        unsigned long long* __restrict__ out)
{
    constexpr int TR = stat_tile_rows<ES, BUF>();
    constexpr int R = TR + 2;
    // (tile, lane column) pairs are dealt to threads densely -- `cols` columns per tile, no idle lanes when the
    // row is not a multiple of the workgroup's span (1440 bytes = 90 columns); a wave may straddle two tiles
    const int cols = col_groups;
    // XCD-aware tile order: gridDim.x is a multiple of 8, so workgroup x of a frame run lands on XCD x % 8.  Giving XCD k the
    // CONTIGUOUS tile groups [k*per, (k+1)*per) makes vertically adjacent tiles -- which share their two halo rows --
    // neighbours on one XCD, running at the same time: the halo re-read is an L2 hit there instead of a second HBM fetch
    // (each XCD has a private L2; adjacent blockIdx.x would put every halo on a different one).
    const int per = gridDim.x / kStatXcds;
    const int wg = (blockIdx.x % kStatXcds) * per + blockIdx.x / kStatXcds;
    int tile, xb;
    if (AMT_STATS_DEAL == 1) {
        // wave-segment dealing: a row is cut into wpt = ceil(cols / 64) equal segments, one wave each; consecutive waves take the segments
        // of one tile, so (wpt = 2, two waves per workgroup) the waves that share a 128-byte line -- the segment boundary, and a row's tail
        // with the next row's head when the pitch is not a multiple of the line -- sit in ONE workgroup and ask for it at the same time
        const int wpt = (cols + 63) >> 6, cpw = (cols + wpt - 1) / wpt;
        const int wave = wg * (kStatThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        tile = wave / wpt;
        const int col = (wave - tile * wpt) * cpw + lane;
        xb = (lane < cpw && col < cols) ? col * kStatColBytes : row_bytes;          // (row_bytes: no valid bytes -> an idle lane)
    } else if (AMT_STATS_DEAL == 2) {
        // grouped dealing: every wave's span starts on a multiple of 64 columns (1 KiB) of its tile's rows, so a span boundary never cuts a
        // 128-byte line of a line-aligned row.  The full 64-column groups of a tile take a wave each; the remainders (r = cols % 64
        // columns) of P = 64 / r consecutive tiles share one wave.  1440 bytes = 90 columns: waves {tile 2b: 0-63}, {tile 2b+1: 0-63},
        // {26 + 26 remainder columns of both} -- 94 % of the lanes busy, against dense dealing's arbitrary boundaries (a line cut by a
        // boundary is fetched by both waves, and the L2 does not merge the two misses: profiles/r05_notes.md)
        const int G = cols >> 6, r = cols & 63, P = r ? 64 / r : 1, wpb = P * G + (r ? 1 : 0);      // waves per block of P tiles
        const int wave = wg * (kStatThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        const int b = wave / wpb, i = wave - b * wpb;
        int col;
        if (i < P * G) { tile = b * P + i / G; col = (i % G) * 64 + lane; }
        else { const int sub = lane / r; tile = b * P + sub; col = sub < P ? G * 64 + (lane - sub * r) : cols; }
        xb = col < cols ? col * kStatColBytes : row_bytes;
    } else {
        const int gid = wg * kStatThreads + threadIdx.x;
        tile = gid / cols;
        xb = (gid - tile * cols) * kStatColBytes;                 // byte column of this thread
    }
    const int y0 = tile * TR;
    const int nvalid = y0 < H ? min(kStatColBytes, row_bytes - xb) : 0;      // <= 0: thread has no pixels
    const int n0 = blockIdx.y * kStatRun;
    const int n1 = min(nframes, n0 + kStatRun);

    // A frame is a raw buffer of H * pitch bytes: the bounds check of the buffer load returns zeros for everything outside it -- the
    // row above the first tile (offset wraps far past the end), the rows below the frame, and ALL rows of a lane that owns no pixels
    // (its offset is parked past the end).  One VGPR holds the lane's offset; the 18 row offsets are scalar multiples of the pitch.
    const unsigned frame_bytes = (unsigned)H * (unsigned)pitch_bytes;
    const unsigned voff0 = nvalid > 0 ? (unsigned)y0 * (unsigned)pitch_bytes + (unsigned)xb : 0x80000000u;
    // mask of this lane's valid bytes (RAGGED only: the last column of a row whose width is not a multiple of the column)
    unsigned bmask[kStatColWords];
#pragma unroll
    for (int i = 0; i < kStatColWords; ++i) {
        const int k = nvalid - 4 * i;
        bmask[i] = !RAGGED || k >= 4 ? 0xFFFFFFFFu : (k <= 0 ? 0u : (0xFFFFFFFFu >> (8 * (4 - k))));
    }
    auto load_rows = [&](const uint8_t* frame, Chunk* rows) {
        if constexpr (!BUF) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 - 1 + r;
                rows[r] = (nvalid > 0 && y >= 0 && y < H) ? load_chunk(frame + (long long)y * pitch_bytes + xb, nvalid)
                                                         : chunk_zero();
            }
            return;
        }
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frame), 0, (int)frame_bytes, 0x00027000);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned off = voff0 + (unsigned)((r - 1) * pitch_bytes);      // (r = 0 of the first tile: wraps, out of range, zeros)
            // AMT_STATS_NT: rows that no other tile reads (all but this tile's first and last row and its two halo rows) are loaded
            // non-temporal, so that the rows two tiles DO share stay in the XCD's L2 until the neighbour asks for them
            constexpr int kNt = AMT_STATS_NT;
            const bool shared_row = r <= 1 || r >= R - 2;
            if (kStatColBytes == 16) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 v = (kNt && !shared_row) ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, kNt)
                                                  : __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
#pragma unroll
                for (int i = 0; i < kStatColWords; ++i) rows[r].w[i] = RAGGED ? (v[i] & bmask[i]) : v[i];
            } else {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u2 v = (kNt && !shared_row) ? __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, kNt)
                                                  : __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0);
#pragma unroll
                for (int i = 0; i < kStatColWords; ++i) rows[r].w[i] = RAGGED ? (v[i] & bmask[i]) : v[i];
            }
        }
    };

    // Even-row vertical detail of a row set: sum over the tile's even rows y (1 <= y <= H-2) of |rows[y-1] - rows[y+1]|.  The weave of
    // frame n takes its odd rows from frame n-1, so its VERT term on an even row looks at rows of frame n-1 only: that is this sum of
    // the PREVIOUS frame, which the previous iteration has already formed as part of its own VERT.
    auto vert_even = [&](const Chunk* rows) {
        unsigned a = 0;
#pragma unroll
        for (int r = 1; r <= TR; r += 2) {
            const int y = y0 - 1 + r;
            if (y >= 1 && y <= H - 2) a = sad16<ES>(rows[r - 1], rows[r + 1], a);
        }
        return a;
    };

    // one frame of this thread's tile against the frame before it; wave reduction, one atomic per word per wave.  ve_prev: vert_even
    // of the frame before; returns vert_even of this frame
    auto compute = [&](const Chunk* cur, const Chunk* prev, int n, unsigned ve_prev) {
        unsigned acc[7] = {0, 0, 0, 0, 0, 0, 0};
        unsigned ve = 0, vo = 0;
        const Chunk zero = chunk_zero();
#pragma unroll
        for (int r = 1; r <= TR; ++r) {
            const int y = y0 - 1 + r;                  // rows >= H were loaded as zeros and add nothing
            const bool odd = ((r - 1) & 1) != 0;       // tiles start on even rows: a constant once unrolled
            acc[odd ? 1 : 0] = sad16<ES>(cur[r], prev[r], acc[odd ? 1 : 0]);
            acc[5] = sad16<ES>(cur[r], zero, acc[5]);
            if (y >= 1 && y <= H - 2) {
                const Chunk mc = avg16<ES>(cur[r - 1], cur[r + 1]);
                if (odd) vo = sad16<ES>(cur[r - 1], cur[r + 1], vo); else ve = sad16<ES>(cur[r - 1], cur[r + 1], ve);
                acc[3] = sad16<ES>(cur[r], mc, acc[3]);
                if (odd) {        // weave: this row comes from prev, its neighbours from cur
                    acc[4] = sad16<ES>(prev[r], mc, acc[4]);
                } else {          // this row from cur, neighbours from prev
                    const Chunk mp = avg16<ES>(prev[r - 1], prev[r + 1]);
                    acc[4] = sad16<ES>(cur[r], mp, acc[4]);
                }
            }
        }
        acc[2] = ve + vo;                // VERT of the frame
        acc[6] = vo + ve_prev;           // VERT of the weave: odd rows look at this frame's neighbours, even rows at the previous frame's
#pragma unroll
        for (int k = 0; k < 7; ++k) {
#if AMT_STATS_LEAN
            acc[k] = wave_sum_to_lane63(acc[k]);
#else
            unsigned v = acc[k];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
            acc[k] = v;
#endif
        }
        if ((threadIdx.x & 63) == (AMT_STATS_LEAN ? 63 : 0)) {
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (acc[k]) atomicAdd(&out[(long long)n * kStatWords + k], (unsigned long long)acc[k]);
        }
        return ve;
    };
    const uint8_t* const before = n0 > 0 ? Y + (long long)(n0 - 1) * frame_stride : (prevY ? prevY : Y);
    // two row sets that swap roles every frame (the loop body holds two frames): copying cur -> prev was 4 R register moves per frame,
    // 7 % of the kernel's vector instructions
    Chunk A[R], B[R];
    load_rows(before, A);
    unsigned ve = vert_even(A);
    // (ragged rows keep the copying form: with the byte masks the doubled body needs more than 256 registers at 8 bits)
    if constexpr (AMT_STATS_PINGPONG && !RAGGED) {
        for (int n = n0; n < n1; n += 2) {
            load_rows(Y + (long long)n * frame_stride, B);
            ve = compute(B, A, n, ve);
            if (n + 1 >= n1) break;
            load_rows(Y + (long long)(n + 1) * frame_stride, A);
            ve = compute(A, B, n + 1, ve);
        }
    } else {
        for (int n = n0; n < n1; ++n) {
            load_rows(Y + (long long)n * frame_stride, B);
            ve = compute(B, A, n, ve);
#pragma unroll
            for (int r = 0; r < R; ++r) A[r] = B[r];
        }
    }
}

hipError_t launch_frame_stats(hipStream_t st, int bits, const void* dY, long long frame_stride_bytes, int pitch_elems, int W,
                              int H, const void* dprevY, int nframes, unsigned long long* dout)
{
    if (nframes <= 0) return hipSuccess;
    const int es = bits <= 8 ? 1 : 2;
    const int row_bytes = W * es;
    const int col_groups = (row_bytes + kStatColBytes - 1) / kStatColBytes;     // lane columns per row
    const bool buf = AMT_STATS_LEAN && (long long)col_groups * kStatColBytes <= (long long)pitch_elems * es;   // see the kernel's BUF
    const int tile_rows = !buf ? kStatTileRowsPlain : es == 1 ? kStatTileRows8 : kStatTileRows;
    const int tiles = (H + tile_rows - 1) / tile_rows;
    // (the lean form addresses a frame with 32-bit byte offsets below 2^31)
    if ((long long)H * pitch_elems * es >= (1LL << 31)) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(dout, 0, (size_t)nframes * kStatWords * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    const int wpt = (col_groups + 63) / 64;
    const int dG = col_groups / 64, dR = col_groups % 64, dP = dR ? 64 / dR : 1, dWpb = dP * dG + (dR ? 1 : 0);
    const int wgs = AMT_STATS_DEAL == 2 ? ((tiles + dP - 1) / dP * dWpb + kStatThreads / 64 - 1) / (kStatThreads / 64)
                  : AMT_STATS_DEAL == 1 ? (tiles * wpt + kStatThreads / 64 - 1) / (kStatThreads / 64)
                  : (tiles * col_groups + kStatThreads - 1) / kStatThreads;
    dim3 grid((unsigned)((wgs + kStatXcds - 1) / kStatXcds * kStatXcds), (unsigned)((nframes + kStatRun - 1) / kStatRun)),
        block(kStatThreads);                                      // surplus workgroups of the round-up find nvalid <= 0
    const bool ragged = row_bytes % kStatColBytes != 0;
#ifndef AMT_STATS_LDS_BYTES
#define AMT_STATS_LDS_BYTES 0      /* experiments: an LDS reservation the kernel never touches caps its workgroups per CU */
#endif
#define AMT_STATS_LAUNCH(E, RG, BF)                                                                                                          \
    hipLaunchKernelGGL((frame_stats_kernel<E, RG, BF>), grid, block, AMT_STATS_LDS_BYTES, st, (const uint8_t*)dY, frame_stride_bytes, pitch_elems * es, row_bytes, H, \
                       (const uint8_t*)dprevY, nframes, col_groups, dout)
    if (es == 1) { if (!buf) AMT_STATS_LAUNCH(1, true, false); else if (ragged) AMT_STATS_LAUNCH(1, true, true); else AMT_STATS_LAUNCH(1, false, true); }
    else { if (!buf) AMT_STATS_LAUNCH(2, true, false); else if (ragged) AMT_STATS_LAUNCH(2, true, true); else AMT_STATS_LAUNCH(2, false, true); }
#undef AMT_STATS_LAUNCH
    return hipGetLastError();
}

} // namespace amt
