// ingest_kernels.hip -- frame assembly in HBM: AMTSource::MergeField / Copy1 / Copy2 (AMTSource.hpp:291-355).
//
// The decoder hands AMTSource one picture per field pair (`top` and `bottom` are the same AVFrame for frame-coded
// pictures, two different ones for field-coded streams); the output frame takes its even rows from `top` and its odd
// rows from `bottom` (Copy1: dst row y <- top row y, dst row y+1 <- bottom row y+1), plane by plane.  NV12 sources carry
// one interleaved UV plane that Copy2 splits into the planar U and V the rest of the pipeline expects.
// A pure HBM copy: one wave per output row, 16 bytes per lane when rows are 16-byte aligned (AVFrame lines are 32/64-byte
// aligned, AviSynth's 64), element by element otherwise.
#include "build_knobs.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>

namespace amt {

struct WeaveArgs {
    const uint8_t* srcY; const uint8_t* srcU; const uint8_t* srcV;   // decoded pictures (srcV unused for NV12)
    long long src_strideY, src_strideUV;                             // bytes between pictures
    int src_pitchY, src_pitchUV;                                     // bytes per source row
    uint8_t* dstY; uint8_t* dstU; uint8_t* dstV;
    long long dst_strideY, dst_strideUV;
    int dst_pitchY, dst_pitchUV;                                     // bytes
    int rowY, rowUV;                                                 // bytes per output row (width * es, widthUV * es)
    int H, HUV;
    int nv12, es, vec;                                               // vec: all rows 16-byte aligned
};

constexpr int kWeaveRows = 8;      // rows per workgroup (one wave per row, two rounds)

__global__ __launch_bounds__(256)
void weave_fields_kernel(WeaveArgs a, const int* __restrict__ top_index, const int* __restrict__ bottom_index)
{
    const int frame = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ti = top_index ? top_index[frame] : frame;
    const int bi = bottom_index ? bottom_index[frame] : frame;
    const int nrows = a.H + 2 * a.HUV;                                // Y rows, then U rows, then V rows
    for (int r = blockIdx.x * kWeaveRows + wave; r < min(nrows, (int)(blockIdx.x + 1) * kWeaveRows); r += 4) {
        int pl, y;
        if (r < a.H) { pl = 0; y = r; } else if (r < a.H + a.HUV) { pl = 1; y = r - a.H; } else { pl = 2; y = r - a.H - a.HUV; }
        const int pic = (y & 1) ? bi : ti;                           // even rows from top, odd rows from bottom
        if (pl == 0 || !a.nv12) {
            const uint8_t* s = (pl == 0 ? a.srcY + (long long)pic * a.src_strideY + (long long)y * a.src_pitchY
                                        : (pl == 1 ? a.srcU : a.srcV) + (long long)pic * a.src_strideUV + (long long)y * a.src_pitchUV);
            uint8_t* d = pl == 0 ? a.dstY + (long long)frame * a.dst_strideY + (long long)y * a.dst_pitchY
                                 : (pl == 1 ? a.dstU : a.dstV) + (long long)frame * a.dst_strideUV + (long long)y * a.dst_pitchUV;
            const int nb = pl == 0 ? a.rowY : a.rowUV;
            if (a.vec) {
                for (int x = lane * 16; x < nb; x += 64 * 16) {
                    if (x + 16 <= nb) *reinterpret_cast<uint4*>(d + x) = *reinterpret_cast<const uint4*>(s + x);
                    else for (int k = x; k < nb; ++k) d[k] = s[k];
                }
            } else {
                for (int x = lane; x < nb; x += 64) d[x] = s[x];
            }
        } else {
            // NV12: source chroma row = U0 V0 U1 V1 ...; plane 1 takes the even elements, plane 2 the odd ones
            const uint8_t* s = a.srcU + (long long)pic * a.src_strideUV + (long long)y * a.src_pitchUV;
            uint8_t* d = (pl == 1 ? a.dstU : a.dstV) + (long long)frame * a.dst_strideUV + (long long)y * a.dst_pitchUV;
            const int n = a.rowUV / a.es;                             // samples per output row
            if (a.es == 1) {
                for (int x = lane; x < n; x += 64) d[x] = s[2 * x + (pl - 1)];
            } else {
                const uint16_t* s16 = reinterpret_cast<const uint16_t*>(s);
                uint16_t* d16 = reinterpret_cast<uint16_t*>(d);
                for (int x = lane; x < n; x += 64) d16[x] = s16[2 * x + (pl - 1)];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Rows that arrive in pinned host memory (a slot of the staging ring, or a registered frame pool) to their pitched place in HBM:
// `nchunks` pieces of `chunk` bytes, src_stride apart in host memory, dst_stride apart on the device.  The kernel reads the
// host memory itself, over PCIe -- no copy engine: the runtime's 2-D copy took 3-5 ms per call for narrow rows on this stack
// (8192 rows of 256 bytes; profiles/r04_notes.md "Boundary"), 40x what the bytes cost.  One lane moves 16 bytes (4 or 1 when the
// geometry is not 16-byte aligned); a row's lanes are consecutive, so the PCIe reads of a wave are 1 KiB contiguous where rows allow.
// ------------------------------------------------------------------------------------------------------------------------
template <typename V>
__global__ __launch_bounds__(256)
void ingest_rows_kernel(const uint8_t* __restrict__ src, long long src_stride, uint8_t* __restrict__ dst, long long dst_stride,
                        int vecs_per_chunk, long long total_vecs)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vecs; i += (long long)gridDim.x * blockDim.x) {
        const long long c = i / vecs_per_chunk;
        const int v = (int)(i - c * vecs_per_chunk);
        reinterpret_cast<V*>(dst + c * dst_stride)[v] = reinterpret_cast<const V*>(src + c * src_stride)[v];
    }
}

hipError_t launch_ingest_rows(hipStream_t st, const void* src_host_mapped, long long src_stride, void* dst, long long dst_stride,
                              unsigned long long chunk, long long nchunks)
{
    if (!chunk || nchunks <= 0) return hipSuccess;
    const uintptr_t a = (uintptr_t)src_host_mapped | (uintptr_t)dst | (uintptr_t)src_stride | (uintptr_t)dst_stride | (uintptr_t)chunk;
    const int vb = (a % 16 == 0) ? 16 : (a % 4 == 0) ? 4 : 1;
    const long long total = (long long)(chunk / vb) * nchunks;
    // enough lanes in flight to cover the PCIe round trip (a few microseconds) at the link's rate, never more than the work
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 2048);
    if (vb == 16)
        hipLaunchKernelGGL(ingest_rows_kernel<uint4>, dim3(grid), dim3(256), 0, st, (const uint8_t*)src_host_mapped, src_stride, (uint8_t*)dst, dst_stride,
                           (int)(chunk / 16), total);
    else if (vb == 4)
        hipLaunchKernelGGL(ingest_rows_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, (const uint8_t*)src_host_mapped, src_stride, (uint8_t*)dst, dst_stride,
                           (int)(chunk / 4), total);
    else
        hipLaunchKernelGGL(ingest_rows_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, (const uint8_t*)src_host_mapped, src_stride, (uint8_t*)dst, dst_stride,
                           (int)chunk, total);
    return hipGetLastError();
}

hipError_t launch_weave_fields(hipStream_t st, const WeaveArgs& a, const int* dtop_index, const int* dbottom_index, int nframes)
{
    if (nframes <= 0) return hipSuccess;
    const int nrows = a.H + 2 * a.HUV;
    dim3 grid((unsigned)((nrows + kWeaveRows - 1) / kWeaveRows), (unsigned)nframes), block(256);
    hipLaunchKernelGGL(weave_fields_kernel, grid, block, 0, st, a, dtop_index, dbottom_index);
    return hipGetLastError();
}

} // namespace amt
