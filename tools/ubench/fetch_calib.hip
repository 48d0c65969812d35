// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE for the request shapes the logo kernels issue.
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced streaming read and "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The tile stager (eval_tile_stage.h) reads frames
// with raw_buffer_load_b32 (8-bit, 4 pixels per lane) / _b64 (16-bit): short row segments of a logo rectangle, one rectangle per frame,
// frames 2.3 MB apart.  Each kernel below touches every byte of a known set exactly once, over a footprint far beyond the 256 MiB
// Infinity Cache, so  factor = known_bytes / (FETCH_SIZE x 1024)  is the correction for that shape.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/ubench/fetch_calib      (tools/gpu_fetch_calib.sh does both and the division)
// stdout: one JSON object {kernel name: {"bytes_requested":…, "bytes_lines64":…, "bytes_lines128":…}}.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef int i2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// streaming reads: lane i of the grid reads element i, i + total, ...   (a window of <= 2 GiB per buffer descriptor)
template <int WIDTH> __global__ void stream_kernel(const char* base, size_t bytes, int* sink)
{
    const size_t total = (size_t)gridDim.x * blockDim.x;
    const size_t n = bytes / WIDTH;
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += total) {
        const size_t off = i * WIDTH;
        const size_t win = off & ~(size_t)0x3FFFFFFF;                           // 1 GiB windows
        __amdgpu_buffer_rsrc_t r = rsrc_of(base + win, 0x40000000u);
        const int vo = (int)(off - win);
        if (WIDTH == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0);
        else if (WIDTH == 8) { i2 v = __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0); acc += v.x ^ v.y; }
        else { i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

// the stager's shape: one workgroup per frame; lanes walk the rectangle's dwords row by row (row = seg_dwords consecutive dwords at
// x0_bytes, rows pitch bytes apart), exactly one request per dword
template <int WIDTH> __global__ void rect_kernel(const char* base, size_t frame_stride, int pitch, int x0_bytes, int y0, int rows, int seg_elems, int* sink)
{
    const char* fr = base + (size_t)blockIdx.x * frame_stride;
    __amdgpu_buffer_rsrc_t r = rsrc_of(fr, (unsigned)frame_stride);
    int acc = 0;
    for (int e = threadIdx.x; e < rows * seg_elems; e += blockDim.x) {
        const int y = e / seg_elems, x = e - y * seg_elems;
        const int vo = (y0 + y) * pitch + x0_bytes + x * WIDTH;
        if (WIDTH == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0);
        else { i2 v = __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0); acc += v.x ^ v.y; }
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

static void lines_of_rect(size_t nframes, size_t frame_stride, int pitch, int x0, int y0, int rows, int seg_bytes, size_t* l64, size_t* l128)
{
    // frame_stride is a multiple of 128 here, so every frame touches the same line pattern
    std::set<size_t> a, b;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < seg_bytes; ++x) {
            const size_t o = (size_t)(y0 + y) * pitch + x0 + x;
            a.insert(o / 64); b.insert(o / 128);
        }
    *l64 = a.size() * 64 * nframes; *l128 = b.size() * 128 * nframes;
}

int main()
{
    const size_t stream_bytes = (size_t)3 << 30;                               // 3 GiB: 12x the Infinity Cache
    const size_t frame_stride = 1440 * 1080 * 3 / 2 + 0;                       // 2 332 800 = 128 x 18 225: 8-bit 1440x1080 YUV420 frames
    const size_t frame_stride16 = frame_stride * 2;
    const int nframes = 1200;                                                  // 2.8 GB (8-bit) / 5.6 GB (16-bit) footprint
    char* buf = nullptr;
    int* sink = nullptr;
    CHECK(hipMalloc(&buf, frame_stride16 * nframes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, frame_stride16 * nframes));
    CHECK(hipDeviceSynchronize());
    std::printf("{\n");
    const int grid = 256 * 8, block = 256;
    stream_kernel<4><<<grid, block>>>(buf, stream_bytes, sink);
    stream_kernel<8><<<grid, block>>>(buf, stream_bytes, sink);
    stream_kernel<16><<<grid, block>>>(buf, stream_bytes, sink);
    for (int w = 4; w <= 16; w *= 2)
        std::printf(" \"stream_kernel<%d>\": {\"bytes_requested\": %zu, \"bytes_lines64\": %zu, \"bytes_lines128\": %zu},\n", w, stream_bytes, stream_bytes, stream_bytes);
    // BASELINE configs[1] logo geometry: 8-bit rectangle at x 1112, rows 52..52+92 (bounding box with the 5x5 apron), 328 bytes wide;
    // a narrow one (one wave's 64-pixel band: 72 bytes) -- the per-wave bounding boxes the stager actually requests
    struct { const char* name; int width, x0, y0, rows, seg_bytes; } rc[] = {
        {"rect_b32_wide", 4, 1112, 52, 92, 328}, {"rect_b32_wave", 4, 1112, 52, 7, 72}, {"rect_b32_wave_unaligned", 4, 1140, 52, 7, 72},
        {"rect_b64_wide", 8, 2224, 52, 92, 656}, {"rect_b64_wave", 8, 2224, 52, 7, 144},
    };
    const int nrc = (int)(sizeof rc / sizeof rc[0]);
    for (int i = 0; i < nrc; ++i) {
        const bool hi = rc[i].width == 8;
        const size_t fs = hi ? frame_stride16 : frame_stride;
        const int pitch = hi ? 2880 : 1440;
        if (hi) rect_kernel<8><<<nframes, 256>>>(buf, fs, pitch, rc[i].x0, rc[i].y0, rc[i].rows, rc[i].seg_bytes / 8, sink);
        else rect_kernel<4><<<nframes, 256>>>(buf, fs, pitch, rc[i].x0, rc[i].y0, rc[i].rows, rc[i].seg_bytes / 4, sink);
        size_t l64, l128;
        lines_of_rect((size_t)nframes, fs, pitch, rc[i].x0, rc[i].y0, rc[i].rows, rc[i].seg_bytes, &l64, &l128);
        std::printf(" \"%s\": {\"kernel\": \"rect_kernel<%d>\", \"launch_index\": %d, \"bytes_requested\": %zu, \"bytes_lines64\": %zu, \"bytes_lines128\": %zu}%s\n",
                    rc[i].name, rc[i].width, i, (size_t)nframes * rc[i].rows * rc[i].seg_bytes, l64, l128, i + 1 < nrc ? "," : "");
    }
    std::printf("}\n");
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
