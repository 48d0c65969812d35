/*
 * amt_filters.hpp -- the reference's filter surface for the logo path, in C++ over the C ABI (amt_gpu.h).
 *
 * Same class names, constructor arguments, GetFrame contracts and error texts as the reference's
 *   logo::AMTAnalyzeLogo   "AMTAnalyzeLogo" "cs[maskratio]i"            (LogoScan.hpp:1106-1236, Amatsukaze.cpp:58)
 *   logo::AMTEraseLogo     "AMTEraseLogo" "ccs[logof]s[mode]i[maxfade]i" (LogoScan.hpp:1238-1519, Amatsukaze.cpp:59)
 *   logo::LogoFrame        scanFrames / selectLogo / writeResult / ...  (LogoScan.hpp:1521-1836, CMAnalyze.hpp:273-317)
 * so that FilteredSource / CMAnalyze keep calling what they call today; the per-pixel work goes to libamt_gpu.so.
 * Header-only, no HIP headers needed: frames travel through amtgpu_frames_upload / amtgpu_download.
 *
 * Host types: AviSynth's, from the real avisynth.h when AMT_FILTERS_USE_AVISYNTH_H is defined before inclusion, else
 * the stand-ins of amt_avs_min.h (namespace amtavs).
 *
 * What differs from the reference's filters, by design: AMTAnalyzeLogo evaluates a block of analysis frames per GPU
 * launch and serves GetFrame from that block (the reference computes 8 source frames per call); the frames returned
 * are byte-identical.  AMTEraseLogo mode != 0 (debug text overlay, :1402-1418) is not provided.
 */
#ifndef AMT_FILTERS_HPP
#define AMT_FILTERS_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "amt_gpu.h"
#ifdef AMT_FILTERS_USE_AVISYNTH_H
#include "avisynth.h"
#define AMT_AVS_NS
#else
#include "amt_avs_min.h"
#define AMT_AVS_NS amtavs::
#endif

namespace amtgpu {

using AMT_AVS_NS AVSValue;
using AMT_AVS_NS GenericVideoFilter;
using AMT_AVS_NS IScriptEnvironment;
using AMT_AVS_NS PClip;
using AMT_AVS_NS PVideoFrame;
using AMT_AVS_NS VideoInfo;
using AMT_AVS_NS PLANAR_U;
using AMT_AVS_NS PLANAR_V;
using AMT_AVS_NS PLANAR_Y;

inline int nblocks(int n, int block) { return (n + block - 1) / block; }      /* StreamUtils.hpp:35 */

/* one context (device, stream, pinned ring) shared by the filters of a script */
class Context {
    AmtGpuContext* g_;
public:
    explicit Context(int device = 0) : g_(amtgpu_context_create(device))
    {
        if (!g_) throw std::runtime_error("amtgpu: no HIP device (there is no CPU path)");
    }
    ~Context() { amtgpu_context_destroy(g_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    AmtGpuContext* get() const { return g_; }
    const char* error() const { return amtgpu_last_error(g_); }
};
typedef std::shared_ptr<Context> PContext;

/* device memory owned by a filter */
class DeviceBuffer {
    PContext ctx_;
    void* p_ = nullptr;
    uint64_t bytes_ = 0;
public:
    explicit DeviceBuffer(PContext c) : ctx_(std::move(c)) {}
    ~DeviceBuffer() { if (p_) amtgpu_device_free(ctx_->get(), p_); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void* reserve(uint64_t bytes)
    {
        if (bytes > bytes_) {
            if (p_) amtgpu_device_free(ctx_->get(), p_);
            p_ = amtgpu_device_alloc(ctx_->get(), bytes);
            if (!p_) throw std::runtime_error(std::string("amtgpu: device allocation failed: ") + ctx_->error());
            bytes_ = bytes;
        }
        return p_;
    }
    uint8_t* at(uint64_t off) const { return static_cast<uint8_t*>(p_) + off; }
};

/* ------------------------------------------------------------------------------------------------------------------
 * AMTAnalyzeLogo: clip of BGR32 64x5 frames, frame n = LogoAnalyzeFrame[8] {p[11],t[11],b[11]} of source frames
 * clamp(8n+i) (LogoScan.hpp:1100-1103, 1119-1161, 1195-1200).
 * ---------------------------------------------------------------------------------------------------------------- */
class AMTAnalyzeLogo : public GenericVideoFilter {
    PContext ctx_;
    AmtGpuAnalyze* an_ = nullptr;
    VideoInfo srcvi_;
    int block_;                                   /* analysis frames evaluated per GPU launch */
    DeviceBuffer dY_;
    std::mutex mu_;
    int cache_first_ = -1;
    std::vector<float> cache_;                    /* [block_][8][33] */
    int row0_ = 0, row1_ = 0;                     /* Y rows the analysis reads: the logo rectangle's */

    void fill(int first, IScriptEnvironment* env)
    {
        const int last = std::min(vi.num_frames, first + block_);
        const int nsrc = (last - first) * 8;
        const int es = srcvi_.ComponentSize();
        uint64_t plane = 0;
        int pitch = 0;
        for (int k = 0; k < nsrc; ++k) {
            const int n = std::max(0, std::min(srcvi_.num_frames - 1, first * 8 + k));
            PVideoFrame f = child->GetFrame(n, env);
            if (k == 0) {
                pitch = f->GetPitch(PLANAR_Y);
                plane = (uint64_t)pitch * srcvi_.height;
                dY_.reserve(plane * nsrc);
            } else if (f->GetPitch(PLANAR_Y) != pitch) {
                env->ThrowError("[AMTAnalyzeLogo] frames of one clip must share a pitch");
            }
            /* only the rows of the logo rectangle travel, to their place in the device frame: the analysis reads nothing else
             * (LogoScan.hpp:1132-1141) -- h * pitch bytes per frame instead of the whole plane */
            const uint64_t off = (uint64_t)row0_ * pitch;
            if (!amtgpu_frames_upload(ctx_->get(), dY_.at(plane * k + off), f->GetReadPtr(PLANAR_Y) + off, (uint64_t)(row1_ - row0_) * pitch))
                env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
        }
        if (!amtgpu_frames_upload_wait(ctx_->get())) env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
        cache_.assign((size_t)block_ * 8 * AMTGPU_ANALYZE_FLOATS, 0.0f);
        if (!amtgpu_analyze_batch_host(an_, dY_.at(0), (int64_t)plane, pitch / es, srcvi_.BitsPerComponent(), nsrc, cache_.data()))
            env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
        cache_first_ = first;
    }

public:
    AMTAnalyzeLogo(PClip clip, const std::string& logoPath, float maskratio, IScriptEnvironment* env, PContext ctx = PContext(),
                   int framesPerLaunch = 32)
        : GenericVideoFilter(clip), ctx_(ctx ? ctx : std::make_shared<Context>()), srcvi_(vi), block_(std::max(1, framesPerLaunch)),
          dY_(ctx_)
    {
        an_ = amtgpu_analyze_create(ctx_->get(), logoPath.c_str(), maskratio);
        if (!an_) env->ThrowError("Failed to read logo file (%s)", logoPath.c_str());          /* LogoScan.hpp:1174 */
        int rc[4] = {0, 0, 0, 0};
        if (!amtgpu_analyze_get_rect(an_, rc)) env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
        row0_ = std::max(0, std::min(srcvi_.height, rc[1]));
        row1_ = std::max(row0_, std::min(srcvi_.height, rc[1] + rc[3]));
        const int out_bytes = (int)sizeof(float) * AMTGPU_ANALYZE_FLOATS * 8;                  /* sizeof(LogoAnalyzeFrame) * 8 */
        vi.pixel_type = VideoInfo::CS_BGR32;
        vi.width = 64;
        vi.height = nblocks(out_bytes, vi.width * 4);
        vi.num_frames = nblocks(vi.num_frames, 8);
    }
    ~AMTAnalyzeLogo() override { if (an_) amtgpu_analyze_destroy(an_); }

    PVideoFrame GetFrame(int n, IScriptEnvironment* env) override
    {
        if (srcvi_.ComponentSize() != 1 && srcvi_.ComponentSize() != 2) env->ThrowError("[AMTAnalyzeLogo] Unsupported pixel format");
        PVideoFrame dst = env->NewVideoFrame(vi);
        std::lock_guard<std::mutex> lock(mu_);
        n = std::max(0, std::min(vi.num_frames - 1, n));
        if (cache_first_ < 0 || n < cache_first_ || n >= cache_first_ + block_) fill(n - n % block_, env);
        std::memcpy(dst->GetWritePtr(), &cache_[(size_t)(n - cache_first_) * 8 * AMTGPU_ANALYZE_FLOATS],
                    sizeof(float) * 8 * AMTGPU_ANALYZE_FLOATS);
        return dst;
    }
    int SetCacheHints(int cachehints, int) override { return cachehints == AMT_AVS_NS CACHE_GET_MTMODE ? AMT_AVS_NS MT_NICE_FILTER : 0; }
};

/* ------------------------------------------------------------------------------------------------------------------
 * AMTEraseLogo: removes the logo from frame n with the fades CalcFade derives from the analysis clip (and the
 * logoframe file, when given) -- LogoScan.hpp:1263-1341 on the host side of the library, Delogo :1248-1261 on the GPU.
 * ---------------------------------------------------------------------------------------------------------------- */
class AMTEraseLogo : public GenericVideoFilter {
    PContext ctx_;
    AmtGpuErase* er_ = nullptr;
    PClip analyzeclip_;
    int mode_, maxFadeLength_;
    DeviceBuffer dbuf_;
    std::mutex mu_;
    std::vector<float> analysis_;                 /* [num_frames][33], filled on demand from analyzeclip */
    std::vector<char> have_;                      /* per analysis frame */

    void need_analysis(int lo, int hi, IScriptEnvironment* env)
    {
        lo = std::max(0, lo);
        hi = std::min(vi.num_frames - 1, hi);
        for (int j = lo >> 3; j <= (hi >> 3); ++j) {
            if (have_[j]) continue;
            PVideoFrame f = analyzeclip_->GetFrame(j, env);
            const float* rec = reinterpret_cast<const float*>(f->GetReadPtr());
            const int cnt = std::min(8, vi.num_frames - j * 8);
            std::memcpy(&analysis_[(size_t)j * 8 * AMTGPU_ANALYZE_FLOATS], rec, sizeof(float) * AMTGPU_ANALYZE_FLOATS * cnt);
            have_[j] = 1;
        }
    }

public:
    AMTEraseLogo(PClip clip, PClip analyzeclip, const std::string& logoPath, const std::string& logofPath, int mode, int maxFadeLength,
                 IScriptEnvironment* env, PContext ctx = PContext())
        : GenericVideoFilter(clip), ctx_(ctx ? ctx : std::make_shared<Context>()), analyzeclip_(std::move(analyzeclip)), mode_(mode),
          maxFadeLength_(maxFadeLength), dbuf_(ctx_)
    {
        if (mode_ != 0) env->ThrowError("[AMTEraseLogo] mode %d (debug overlay) is not available on the GPU path", mode_);
        er_ = amtgpu_erase_create(ctx_->get(), logoPath.c_str(), logofPath.c_str(), mode_, maxFadeLength_);
        if (!er_) {
            const std::string why = ctx_->error();       /* "Failed to read logo file (..)" / "Invalid logoframe file ..." (:1174,1446,1452) */
            env->ThrowError("%s", why.c_str());
        }
        analysis_.assign((size_t)nblocks(vi.num_frames, 8) * 8 * AMTGPU_ANALYZE_FLOATS, 0.0f);
        have_.assign(nblocks(vi.num_frames, 8), 0);
    }
    ~AMTEraseLogo() override { if (er_) amtgpu_erase_destroy(er_); }

    PVideoFrame GetFrame(int n, IScriptEnvironment* env) override
    {
        const int es = vi.ComponentSize();
        if (es != 1 && es != 2) env->ThrowError("[AMTEraseLogo] Unsupported pixel format");
        PVideoFrame frame = child->GetFrame(n, env);
        env->MakeWritable(&frame);
        std::lock_guard<std::mutex> lock(mu_);
        /* CalcFade2 reads the analysis of source frames n-8 .. n+8; a logoframe transition can widen that by maxfade/2 */
        const int reach = 8 + (maxFadeLength_ >> 1) + 1;
        need_analysis(n - reach, n + reach, env);
        float fades[2];
        if (!amtgpu_erase_calc_fades(er_, analysis_.data(), vi.num_frames, n, 1, fades)) env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
        /* Delogo rewrites the logo rectangle and nothing else (LogoScan.hpp:1248-1261): only its rows cross PCIe -- w*h luma and
         * 2 * w/2*h/2 chroma samples up and back instead of two whole frames -- and a frame whose fades are both 0 is returned as
         * it came (the reference's arithmetic is the identity there) */
        int rc[5];
        if (!amtgpu_erase_get_rect(er_, rc)) env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
        const int imgx = rc[0], imgy = rc[1], w = rc[2], h = rc[3];
        if (rc[4] && fades[0] == 0.0f && fades[1] == 0.0f) return frame;
        if (imgx < 0 || imgy < 0 || imgx + w > vi.width || imgy + h > vi.height) env->ThrowError("[AMTEraseLogo] logo rectangle outside the frame");
        const int wUV = w >> 1, hUV = h >> 1, cx = imgx >> 1, cy = imgy >> 1;
        const uint64_t by = (uint64_t)w * h * es, buv = (uint64_t)wUV * hUV * es;
        dbuf_.reserve(by + 2 * buv);
        uint8_t *dY = dbuf_.at(0), *dU = dbuf_.at(by), *dV = dbuf_.at(by + buv);
        AmtGpuContext* g = ctx_->get();
        const int pY = frame->GetPitch(PLANAR_Y), pUV = frame->GetPitch(PLANAR_U);
        uint8_t* hY = frame->GetWritePtr(PLANAR_Y) + (size_t)imgy * pY + (size_t)imgx * es;
        uint8_t* hU = frame->GetWritePtr(PLANAR_U) + (size_t)cy * pUV + (size_t)cx * es;
        uint8_t* hV = frame->GetWritePtr(PLANAR_V) + (size_t)cy * pUV + (size_t)cx * es;
        if (!amtgpu_frames_upload_strided(g, dY, (int64_t)w * es, hY, pY, (uint64_t)w * es, h) ||
            !amtgpu_frames_upload_strided(g, dU, (int64_t)wUV * es, hU, pUV, (uint64_t)wUV * es, hUV) ||
            !amtgpu_frames_upload_strided(g, dV, (int64_t)wUV * es, hV, pUV, (uint64_t)wUV * es, hUV) || !amtgpu_frames_upload_wait(g))
            env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
        if (!amtgpu_erase_rect_batch(er_, dY, dU, dV, (int64_t)by, (int64_t)buv, w, wUV, vi.BitsPerComponent(), 1, fades))
            env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
        if (!amtgpu_download_strided(g, hY, pY, dY, (int64_t)w * es, (uint64_t)w * es, h) ||
            !amtgpu_download_strided(g, hU, pUV, dU, (int64_t)wUV * es, (uint64_t)wUV * es, hUV) ||
            !amtgpu_download_strided(g, hV, pUV, dV, (int64_t)wUV * es, (uint64_t)wUV * es, hUV))
            env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
        return frame;
    }
    int SetCacheHints(int cachehints, int) override { return cachehints == AMT_AVS_NS CACHE_GET_MTMODE ? AMT_AVS_NS MT_NICE_FILTER : 0; }
};

/* ------------------------------------------------------------------------------------------------------------------
 * LogoFrame: the CM all-frames scan (CMAnalyze.hpp:291-299): every logo file against every frame, then the host
 * decisions.  Same calls as logo::LogoFrame (LogoScan.hpp:1592-1835); AMTContext& becomes the GPU context.
 * ---------------------------------------------------------------------------------------------------------------- */
class LogoFrame {
    PContext ctx_;
    AmtGpuLogoFrame* lf_ = nullptr;
    int numLogos_;
    int numFrames_ = 0;
    int framesPerLaunch_;

    void check(int ok) const { if (!ok) throw std::runtime_error(ctx_->error()); }

public:
    LogoFrame(PContext ctx, const std::vector<std::string>& logofiles, float maskratio, int framesPerLaunch = 256)
        : ctx_(std::move(ctx)), numLogos_((int)logofiles.size()), framesPerLaunch_(std::max(1, framesPerLaunch))
    {
        std::vector<const char*> paths;
        for (const auto& s : logofiles) paths.push_back(s.c_str());
        lf_ = amtgpu_logoframe_create(ctx_->get(), paths.data(), numLogos_, maskratio);
        if (!lf_) throw std::runtime_error(ctx_->error());
    }
    ~LogoFrame() { if (lf_) amtgpu_logoframe_destroy(lf_); }
    LogoFrame(const LogoFrame&) = delete;
    LogoFrame& operator=(const LogoFrame&) = delete;

    void scanFrames(PClip clip, IScriptEnvironment* env)
    {
        const VideoInfo vi = clip->GetVideoInfo();
        const int es = vi.ComponentSize();
        numFrames_ = vi.num_frames;
        check(amtgpu_logoframe_begin(lf_, vi.width, vi.height, vi.BitsPerComponent(), vi.num_frames, (int)vi.fps_numerator,
                                     (int)vi.fps_denominator));
        DeviceBuffer dY(ctx_);
        int rows[2] = {0, 0};
        check(amtgpu_logoframe_get_rows(lf_, rows));
        const int r0 = std::max(0, std::min(vi.height, rows[0])), r1 = std::max(r0, std::min(vi.height, rows[1]));
        for (int n0 = 0; n0 < vi.num_frames; n0 += framesPerLaunch_) {
            const int nb = std::min(framesPerLaunch_, vi.num_frames - n0);
            uint64_t plane = 0;
            int pitch = 0;
            for (int i = 0; i < nb; ++i) {
                PVideoFrame f = clip->GetFrame(n0 + i, env);
                if (i == 0) {
                    pitch = f->GetPitch(PLANAR_Y);
                    plane = (uint64_t)pitch * vi.height;
                    dY.reserve(plane * nb);
                } else if (f->GetPitch(PLANAR_Y) != pitch) {
                    throw std::runtime_error("[LogoFrame] frames of one clip must share a pitch");
                }
                /* only the rows some logo's rectangle covers travel (to their place in the device frame) */
                const uint64_t off = (uint64_t)r0 * pitch;
                check(amtgpu_frames_upload(ctx_->get(), dY.at(plane * i + off), f->GetReadPtr(PLANAR_Y) + off, (uint64_t)(r1 - r0) * pitch));
            }
            check(amtgpu_frames_upload_wait(ctx_->get()));
            check(amtgpu_logoframe_scan_batch(lf_, dY.at(0), (int64_t)plane, pitch / es, n0, nb));
            check(amtgpu_context_synchronize(ctx_->get()));       /* the batch buffer is reused by the next round */
        }
    }
    /* num_frames * numLogos * {corr0, corr1} (EvalResult, LogoScan.hpp:1532-1535) */
    std::vector<float> evalResults() const
    {
        std::vector<float> r((size_t)numFrames_ * numLogos_ * 2);
        if (!r.empty() && !amtgpu_logoframe_get_results(lf_, r.data())) throw std::runtime_error(ctx_->error());
        return r;
    }
    void selectLogo(int numCandidates = -1) { check(amtgpu_logoframe_select_logo(lf_, numCandidates)); }
    void writeResult(const std::string& outpath, int logoIndex = -1) { check(amtgpu_logoframe_write_result(lf_, outpath.c_str(), logoIndex)); }
    int getBestLogo() const { return amtgpu_logoframe_best_logo(lf_); }
    float getLogoRatio() const { return amtgpu_logoframe_logo_ratio(lf_); }
};

} /* namespace amtgpu */
#endif /* AMT_FILTERS_HPP */
