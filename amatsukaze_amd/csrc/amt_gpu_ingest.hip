// amt_gpu_ingest.hip -- C ABI part 4: frame assembly (field weave, NV12 split) of decoded pictures resident in HBM.
#include "build_knobs.h"
#include "../../include/amt_gpu.h"

#include <vector>

#include "api_common.hpp"

namespace amt {
struct WeaveArgs {
    const uint8_t* srcY; const uint8_t* srcU; const uint8_t* srcV;
    long long src_strideY, src_strideUV;
    int src_pitchY, src_pitchUV;
    uint8_t* dstY; uint8_t* dstU; uint8_t* dstV;
    long long dst_strideY, dst_strideUV;
    int dst_pitchY, dst_pitchUV;
    int rowY, rowUV;
    int H, HUV;
    int nv12, es, vec;
};
hipError_t launch_weave_fields(hipStream_t st, const WeaveArgs& a, const int* dtop_index, const int* dbottom_index, int nframes);
}
using namespace amt;

extern "C" {

int amtgpu_weave_fields_batch(AmtGpuContext* c, const void* dsrcY, const void* dsrcU, const void* dsrcV, int64_t src_strideY,
                              int64_t src_strideUV, int src_pitchY, int src_pitchUV, int num_pictures, const int* top_index,
                              const int* bottom_index, int nv12, int bits, int width, int height, void* dY, void* dU, void* dV,
                              int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int nframes)
{
    return guard(c, [&] {
        if (!c) throw std::runtime_error("[AMTSource] no context");
        if (nframes <= 0) return;
        // Copy1 walks row PAIRS of every plane, the chroma planes included (AMTSource.hpp:294, 345-346)
        if (width <= 0 || height <= 0 || (width & 1) || (height & 3)) throw std::runtime_error("[AMTSource] width must be even and height a multiple of 4 (interlaced 4:2:0)");
        if (bits < 8 || bits > 16) throw std::runtime_error("[AMTSource] unsupported bit depth");
        if (!dsrcY || !dsrcU || (!nv12 && !dsrcV) || !dY || !dU || !dV) throw std::runtime_error("[AMTSource] null plane");
        const int es = bits <= 8 ? 1 : 2;
        const int wUV = width >> 1, hUV = height >> 1;
        if (src_pitchY < width || src_pitchUV < (nv12 ? width : wUV) || pitchY < width || pitchUV < wUV)
            throw std::runtime_error("[AMTSource] pitch smaller than the row");
        for (int i = 0; i < nframes; ++i) {
            const int t = top_index ? top_index[i] : i, b = bottom_index ? bottom_index[i] : i;
            if (t < 0 || t >= num_pictures || b < 0 || b >= num_pictures) throw std::runtime_error("[AMTSource] picture index out of range");
        }
        c->bind();
        WeaveArgs a;
        a.srcY = (const uint8_t*)dsrcY; a.srcU = (const uint8_t*)dsrcU; a.srcV = (const uint8_t*)dsrcV;
        a.src_strideY = src_strideY; a.src_strideUV = src_strideUV;
        a.src_pitchY = src_pitchY * es; a.src_pitchUV = src_pitchUV * es;
        a.dstY = (uint8_t*)dY; a.dstU = (uint8_t*)dU; a.dstV = (uint8_t*)dV;
        a.dst_strideY = strideY; a.dst_strideUV = strideUV;
        a.dst_pitchY = pitchY * es; a.dst_pitchUV = pitchUV * es;
        a.rowY = width * es; a.rowUV = wUV * es;
        a.H = height; a.HUV = hUV;
        a.nv12 = nv12 ? 1 : 0; a.es = es;
        auto al16 = [](const void* p, long long stride, int pitch) { return ((uintptr_t)p % 16 == 0) && stride % 16 == 0 && pitch % 16 == 0; };
        a.vec = al16(a.srcY, a.src_strideY, a.src_pitchY) && al16(a.dstY, a.dst_strideY, a.dst_pitchY) &&
                al16(a.dstU, a.dst_strideUV, a.dst_pitchUV) && al16(a.dstV, a.dst_strideUV, a.dst_pitchUV) &&
                (nv12 || (al16(a.srcU, a.src_strideUV, a.src_pitchUV) && al16(a.srcV, a.src_strideUV, a.src_pitchUV)));
        DevBuf<int> dti, dbi;
        if (top_index) dti.upload(top_index, (size_t)nframes, c->stream);
        if (bottom_index) dbi.upload(bottom_index, (size_t)nframes, c->stream);
        const int sp = c->prof_begin("weave_fields_kernel");
        AMT_HIP(launch_weave_fields(c->stream, a, top_index ? dti.get() : nullptr, bottom_index ? dbi.get() : nullptr, nframes));
        c->prof_end(sp);
        if (top_index || bottom_index) AMT_HIP(hipStreamSynchronize(c->stream));   // the index buffers die with this call
    });
}

} // extern "C"
