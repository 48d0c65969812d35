// stats_kernels.hip -- whole-frame field-difference / combing metrics (self-specified).
//
// HBM-bound streaming reduction over the Y plane: every byte of every frame is read from HBM once.
// A thread owns a 16-byte-wide column of a 16-row tile for a RUN of consecutive frames (tiles x columns are dealt
// densely to the threads of the grid), so the vertical neighbours (rows y-1, y+1) and the previous
// frame's rows are all in the thread's own registers -- no LDS staging, no re-reads except the two halo
// rows per tile.  Loads are 16 B per lane, 64 lanes = 1 KiB contiguous per row.  The per-byte work is
// done four pixels at a time with v_sad_u8 / v_lerp_u8 (two per instruction for 16-bit samples).
//
// Per frame n (prev = frame n-1; rows 1..H-2 for the vertical metrics), all sums of absolute values:
//   0 DIFF_TOP   sum_{y even} |Y_n[y] - Y_prev[y]|          3 COMB       sum |Y_n[y] - avg(Y_n[y-1], Y_n[y+1])|
//   1 DIFF_BOT   sum_{y odd}  |Y_n[y] - Y_prev[y]|          4 COMB_PREV  same on the weave (even rows of n, odd rows of prev)
//   2 VERT_SAME  sum |Y_n[y-1] - Y_n[y+1]|                  5 SUM        sum Y_n
//   6 VERT_PREV  VERT_SAME of that weave                    7 reserved (0)
// avg(a,c) = (a + c) >> 1 per sample.
#include <hip/hip_runtime.h>
#include <cstdint>

namespace amt {

#ifndef AMT_STATS_VG
#define AMT_STATS_VG 4
#endif
// Waves of a workgroup that sit on top of one another: wave w of a workgroup owns tile (4 * supertile + w) of the SAME 16-byte columns, so
// the two halo rows a tile re-reads are rows that a sibling wave of the same workgroup -- same CU, same moment -- reads as its own: the
// second request merges with the first in the CU's vector cache / the XCD's L2 instead of going to HBM again (kStatVG = 1: round 3's
// mapping, every wave an unrelated (tile, column) range; measured traffic 1.146x the algorithmic bytes)
constexpr int kStatVG = AMT_STATS_VG;
constexpr int kStatThreads = kStatVG > 1 ? 64 * kStatVG : 128;
#ifndef AMT_STATS_ROWS
#define AMT_STATS_ROWS 16
#endif
#ifndef AMT_STATS_RUN
#define AMT_STATS_RUN 32
#endif
constexpr int kStatTileRows = AMT_STATS_ROWS;
constexpr int kStatRun = AMT_STATS_RUN;          // frames a workgroup walks through (the frame before a run is its one re-read: 1/32)
constexpr int kStatXcds = 8;          // MI355X: 8 XCDs, workgroups are dealt to them round-robin by linear workgroup id
constexpr int kStatWords = 8;

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

#ifndef AMT_STATS_COLB
#define AMT_STATS_COLB 16
#endif
#ifndef AMT_STATS_PREFETCH
#define AMT_STATS_PREFETCH 0
#endif
constexpr int kStatColBytes = AMT_STATS_COLB;      // bytes of a row one lane owns: 16 (one dwordx4 load) or 8 (dwordx2: half the registers per row)
constexpr int kStatColWords = kStatColBytes / 4;
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

#ifdef AMT_STATS_WAVES
#define AMT_STATS_OCC __attribute__((amdgpu_waves_per_eu(AMT_STATS_WAVES, AMT_STATS_WAVES)))
#else
#define AMT_STATS_OCC
#endif
template <int ES>
__global__ __launch_bounds__(kStatThreads) AMT_STATS_OCC
void frame_stats_kernel(const uint8_t* __restrict__ Y, long long frame_stride /*bytes*/, int pitch_bytes, int row_bytes, int H,
                        const uint8_t* __restrict__ prevY /* frame before the batch or null */, int nframes, int col_groups,
                        unsigned long long* __restrict__ out)
{
    constexpr int R = kStatTileRows + 2;
    // (tile, 16-byte column) pairs are dealt to threads densely -- `cols` columns per tile, no idle lanes when the
    // row is not a multiple of the workgroup's span (1440 bytes = 90 columns); a wave may straddle two tiles
    const int cols = col_groups;                                  // 16-byte columns per row
    // XCD-aware tile order: gridDim.x is a multiple of 8, so workgroup x of a frame run lands on XCD x % 8.  Giving XCD k the
    // CONTIGUOUS tile groups [k*per, (k+1)*per) makes vertically adjacent tiles -- which share their two halo rows --
    // neighbours on one XCD, running at the same time: the halo re-read is an L2 hit there instead of a second HBM fetch
    // (each XCD has a private L2; adjacent blockIdx.x would put every halo on a different one).
    const int per = gridDim.x / kStatXcds;
    const int wg = (blockIdx.x % kStatXcds) * per + blockIdx.x / kStatXcds;
    int tile, xb;
    if (kStatVG > 1) {
        // (supertile, column) pairs are dealt densely to the 64 lanes of a workgroup's waves; wave w takes tile kStatVG * supertile + w
        const int gid = wg * 64 + (threadIdx.x & 63);
        const int st = gid / cols;
        tile = st * kStatVG + (threadIdx.x >> 6);
        xb = (gid - st * cols) * kStatColBytes;
    } else {
        const int gid = wg * kStatThreads + threadIdx.x;
        tile = gid / cols;
        xb = (gid - tile * cols) * kStatColBytes;                 // byte column of this thread
    }
    const int y0 = tile * kStatTileRows;
    const int nvalid = y0 < H ? min(kStatColBytes, row_bytes - xb) : 0;      // <= 0: thread has no pixels
    const int n0 = blockIdx.y * kStatRun;
    const int n1 = min(nframes, n0 + kStatRun);

    auto load_rows = [&](const uint8_t* frame, Chunk* rows) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = y0 - 1 + r;
            rows[r] = (nvalid > 0 && y >= 0 && y < H) ? load_chunk(frame + (long long)y * pitch_bytes + xb, nvalid)
                                                     : chunk_zero();
        }
    };

    // one frame of this thread's tile against the frame before it; wave reduction, one atomic per word per wave
    auto compute = [&](const Chunk* cur, const Chunk* prev, int n) {
        unsigned acc[7] = {0, 0, 0, 0, 0, 0, 0};
        const Chunk zero = chunk_zero();
#pragma unroll
        for (int r = 1; r <= kStatTileRows; ++r) {
            const int y = y0 - 1 + r;                  // rows >= H were loaded as zeros and add nothing
            const bool odd = ((r - 1) & 1) != 0;       // tiles start on even rows: a constant once unrolled
            acc[odd ? 1 : 0] = sad16<ES>(cur[r], prev[r], acc[odd ? 1 : 0]);
            acc[5] = sad16<ES>(cur[r], zero, acc[5]);
            if (y >= 1 && y <= H - 2) {
                const Chunk mc = avg16<ES>(cur[r - 1], cur[r + 1]);
                acc[2] = sad16<ES>(cur[r - 1], cur[r + 1], acc[2]);
                acc[3] = sad16<ES>(cur[r], mc, acc[3]);
                if (odd) {        // weave: this row comes from prev, its neighbours from cur
                    acc[4] = sad16<ES>(prev[r], mc, acc[4]);
                    acc[6] = sad16<ES>(cur[r - 1], cur[r + 1], acc[6]);
                } else {          // this row from cur, neighbours from prev
                    const Chunk mp = avg16<ES>(prev[r - 1], prev[r + 1]);
                    acc[4] = sad16<ES>(cur[r], mp, acc[4]);
                    acc[6] = sad16<ES>(prev[r - 1], prev[r + 1], acc[6]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            unsigned v = acc[k];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
            acc[k] = v;
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (acc[k]) atomicAdd(&out[(long long)n * kStatWords + k], (unsigned long long)acc[k]);
        }
    };
    const uint8_t* const before = n0 > 0 ? Y + (long long)(n0 - 1) * frame_stride : (prevY ? prevY : Y);
#if AMT_STATS_PREFETCH
    // three row sets in rotation: the loads of frame n + 1 are issued BEFORE frame n is evaluated, so that every wave always has a
    // whole tile in flight -- the kernel's rate is set by the bytes a CU keeps in flight (measured: it scales with the CUs it is
    // given, ~29 GB/s per CU for the two-set form), not by HBM, as soon as it does not own the whole device
    Chunk A[R], B[R], D[R];
    load_rows(before, A);
    load_rows(Y + (long long)n0 * frame_stride, B);
    for (int n = n0;;) {
        if (n + 1 < n1) load_rows(Y + (long long)(n + 1) * frame_stride, D);
        compute(B, A, n);
        if (++n >= n1) break;
        if (n + 1 < n1) load_rows(Y + (long long)(n + 1) * frame_stride, A);
        compute(D, B, n);
        if (++n >= n1) break;
        if (n + 1 < n1) load_rows(Y + (long long)(n + 1) * frame_stride, B);
        compute(A, D, n);
        if (++n >= n1) break;
    }
#else
    Chunk prev[R], cur[R];
    load_rows(before, prev);
    for (int n = n0; n < n1; ++n) {
        load_rows(Y + (long long)n * frame_stride, cur);
        compute(cur, prev, n);
#pragma unroll
        for (int r = 0; r < R; ++r) prev[r] = cur[r];
    }
#endif
}

hipError_t launch_frame_stats(hipStream_t st, int bits, const void* dY, long long frame_stride_bytes, int pitch_elems, int W,
                              int H, const void* dprevY, int nframes, unsigned long long* dout)
{
    if (nframes <= 0) return hipSuccess;
    const int es = bits <= 8 ? 1 : 2;
    const int row_bytes = W * es;
    const int col_groups = (row_bytes + kStatColBytes - 1) / kStatColBytes;     // lane columns per row
    const int tiles = (H + kStatTileRows - 1) / kStatTileRows;
    hipError_t e = hipMemsetAsync(dout, 0, (size_t)nframes * kStatWords * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    const int wgs = kStatVG > 1 ? ((tiles + kStatVG - 1) / kStatVG * col_groups + 63) / 64 : (tiles * col_groups + kStatThreads - 1) / kStatThreads;
    dim3 grid((unsigned)((wgs + kStatXcds - 1) / kStatXcds * kStatXcds), (unsigned)((nframes + kStatRun - 1) / kStatRun)),
        block(kStatThreads);                                      // surplus workgroups of the round-up find nvalid <= 0
    if (es == 1)
        hipLaunchKernelGGL(frame_stats_kernel<1>, grid, block, 0, st, (const uint8_t*)dY, frame_stride_bytes, pitch_elems * es,
                           row_bytes, H, (const uint8_t*)dprevY, nframes, col_groups, dout);
    else
        hipLaunchKernelGGL(frame_stats_kernel<2>, grid, block, 0, st, (const uint8_t*)dY, frame_stride_bytes, pitch_elems * es,
                           row_bytes, H, (const uint8_t*)dprevY, nframes, col_groups, dout);
    return hipGetLastError();
}

} // namespace amt
