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
 * What differs from the reference's filters, by design: both filters work a BLOCK of frames per GPU launch and serve GetFrame
 * from it -- AMTAnalyzeLogo a block of analysis frames (the reference computes 8 source frames per call), AMTEraseLogo a block
 * of child frames whose logo rectangles go up in one batch, through ONE erase launch and back in one copy (the reference
 * erases the frame it was asked for); LogoFrame::scanFrames uploads the next batch of frames while the GPU scans the current
 * one.  What they return is byte-identical to the reference's.  AMTAnalyzeLogo can also run the linear decision-guarded
 * evaluation (AMTGPU_ANALYZE_LINEAR_GUARDED: ~2x faster, every fade AMTEraseLogo derives from the clip identical, the clip's
 * floats within 1e-4) -- opt-in, since the analysis clip then is no longer the reference's bit for bit.
 * AMTEraseLogo mode != 0 (debug text overlay, :1402-1418) is not provided.
 */
#ifndef AMT_FILTERS_HPP
#define AMT_FILTERS_HPP

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
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
constexpr int kUploadGroup = 64;   /* host frames whose logo rectangles share one upload call */

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
    int row0_ = 0, row1_ = 0, col0_ = 0, col1_ = 0;   /* the Y samples the analysis reads: the logo rectangle */

    void fill(int first, IScriptEnvironment* env)
    {
        const int last = std::min(vi.num_frames, first + block_);
        const int nsrc = (last - first) * 8;
        const int es = srcvi_.ComponentSize();
        uint64_t plane = 0;
        int pitch = 0;
        std::vector<PVideoFrame> held;
        std::vector<const void*> srcs;
        int group_first = 0;
        /* only the logo rectangle travels, to its place in the device frame: the analysis reads nothing else (LogoScan.hpp:1132-1141)
         * -- w * h samples per frame instead of the whole plane; kUploadGroup frames per upload call */
        auto flush = [&]() {
            if (srcs.empty()) return;
            if (row1_ > row0_ && col1_ > col0_ &&
                !amtgpu_frames_upload_gather(ctx_->get(), dY_.at(plane * group_first) + (uint64_t)col0_ * es, pitch, srcs.data(), pitch,
                                             (uint64_t)(col1_ - col0_) * es, row1_ - row0_, (int)srcs.size()))
                env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
            held.clear();
            srcs.clear();
        };
        for (int k = 0; k < nsrc; ++k) {
            const int n = std::max(0, std::min(srcvi_.num_frames - 1, first * 8 + k));
            PVideoFrame f = child->GetFrame(n, env);
            if (k == 0) {
                pitch = f->GetPitch(PLANAR_Y);
                plane = (uint64_t)pitch * (row1_ - row0_);        /* resident part of a frame: the rectangle's rows */
                dY_.reserve(plane * nsrc + 64);
            } else if (f->GetPitch(PLANAR_Y) != pitch) {
                env->ThrowError("[AMTAnalyzeLogo] frames of one clip must share a pitch");
            }
            if (srcs.empty()) group_first = k;
            srcs.push_back(f->GetReadPtr(PLANAR_Y) + (uint64_t)row0_ * pitch + (uint64_t)col0_ * es);
            held.push_back(std::move(f));
            if ((int)srcs.size() == kUploadGroup) flush();
        }
        flush();
        if (!amtgpu_frames_upload_wait(ctx_->get())) env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
        cache_.assign((size_t)block_ * 8 * AMTGPU_ANALYZE_FLOATS, 0.0f);
        /* a device "frame" is the rectangle's rows, addressed as if the rows above were there */
        if (!amtgpu_analyze_batch_host(an_, dY_.at(0) - (uint64_t)row0_ * pitch, (int64_t)plane, pitch / es, srcvi_.BitsPerComponent(), nsrc, cache_.data()))
            env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
        cache_first_ = first;
    }

public:
    /* analysisMode: AMTGPU_ANALYZE_EXACT (the reference's records bit for bit) or AMTGPU_ANALYZE_LINEAR_GUARDED */
    AMTAnalyzeLogo(PClip clip, const std::string& logoPath, float maskratio, IScriptEnvironment* env, PContext ctx = PContext(),
                   int framesPerLaunch = 32, int analysisMode = AMTGPU_ANALYZE_EXACT)
        : GenericVideoFilter(clip), ctx_(ctx ? ctx : std::make_shared<Context>()), srcvi_(vi), block_(std::max(1, framesPerLaunch)),
          dY_(ctx_)
    {
        an_ = amtgpu_analyze_create(ctx_->get(), logoPath.c_str(), maskratio);
        if (!an_) env->ThrowError("Failed to read logo file (%s)", logoPath.c_str());          /* LogoScan.hpp:1174 */
        if (analysisMode != AMTGPU_ANALYZE_EXACT && !amtgpu_analyze_set_mode(an_, analysisMode)) env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
        int rc[4] = {0, 0, 0, 0};
        if (!amtgpu_analyze_get_rect(an_, rc)) env->ThrowError("[AMTAnalyzeLogo] %s", ctx_->error());
        row0_ = std::max(0, std::min(srcvi_.height, rc[1]));
        row1_ = std::max(row0_, std::min(srcvi_.height, rc[1] + rc[3]));
        col0_ = std::max(0, std::min(srcvi_.width, rc[0]));
        col1_ = std::max(col0_, std::min(srcvi_.width, rc[0] + rc[2]));
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
    int block_;                                   /* child frames erased per GPU launch */
    DeviceBuffer dbuf_;
    std::mutex mu_;
    std::condition_variable cv_;                  /* a block some thread is fetching is waited for, not fetched twice */
    std::set<int> inflight_;                      /* first frames of the blocks being fetched */
    std::vector<float> analysis_;                 /* [num_frames][33], filled on demand from analyzeclip */
    std::vector<char> have_;                      /* per analysis frame */
    /* the erased frames of the most recently used blocks.  The reference's filter works frame by frame and has no block to lose
     * (LogoScan.hpp:1343-1419, MT_NICE_FILTER); here Prefetch threads whose requests straddle a block boundary alternate between
     * blocks k and k + 1, and with a single cached block each such request would evict the other's block and redo its upstream
     * GetFrames and its launch.  Two entries, least recently used replaced. */
    struct CachedBlock { int first = -1; std::vector<PVideoFrame> frames; uint64_t used = 0; };
    static constexpr int kCacheBlocks = 2;
    CachedBlock cache_[kCacheBlocks];
    uint64_t tick_ = 0;
    /* (mu_ held) */
    const PVideoFrame* lookup(int n)
    {
        for (CachedBlock& b : cache_)
            if (b.first >= 0 && n >= b.first && n < b.first + (int)b.frames.size()) { b.used = ++tick_; return &b.frames[n - b.first]; }
        return nullptr;
    }

    /* what one block needs from upstream, fetched WITHOUT the filter's lock (under Prefetch the child clip's work -- AMTSource decode,
     * the analysis filter -- then runs on as many threads as the host gives this filter, as with the reference's per-frame filter) */
    struct Fetched {
        int first = 0, nb = 0;
        std::vector<PVideoFrame> frames;
        std::vector<std::pair<int, PVideoFrame>> analysis;      /* (analysis frame number, frame) */
    };

    Fetched fetch(int first, IScriptEnvironment* env)
    {
        Fetched f;
        f.first = first;
        f.nb = std::min(block_, vi.num_frames - first);
        /* CalcFade2 reads the analysis of source frames n-8 .. n+8; a logoframe transition can widen that by maxfade/2 */
        const int reach = 8 + (maxFadeLength_ >> 1) + 1;
        const int lo = std::max(0, first - reach), hi = std::min(vi.num_frames - 1, first + f.nb - 1 + reach);
        std::vector<int> missing;
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (int j = lo >> 3; j <= (hi >> 3); ++j) if (!have_[j]) missing.push_back(j);
        }
        for (int j : missing) f.analysis.emplace_back(j, analyzeclip_->GetFrame(j, env));
        f.frames.resize(f.nb);
        for (int i = 0; i < f.nb; ++i) f.frames[i] = child->GetFrame(first + i, env);
        return f;
    }

    /* frames [first, first + nb): their logo rectangles uploaded together, ONE Delogo launch (LogoScan.hpp:1248-1261, 1374-1397), one
     * copy back.  Delogo rewrites the rectangle and nothing else: w*h luma and 2 * w/2*h/2 chroma samples per frame cross PCIe each
     * way, and a frame whose fades are both 0 does not travel at all when the reference's arithmetic is the identity there
     * (amtgpu_erase_get_rect).  Called with mu_ held. */
    void process(Fetched& in, IScriptEnvironment* env)
    {
        for (auto& a : in.analysis) {
            if (have_[a.first]) continue;
            const float* rec = reinterpret_cast<const float*>(a.second->GetReadPtr());
            const int cnt = std::min(8, vi.num_frames - a.first * 8);
            std::memcpy(&analysis_[(size_t)a.first * 8 * AMTGPU_ANALYZE_FLOATS], rec, sizeof(float) * AMTGPU_ANALYZE_FLOATS * cnt);
            have_[a.first] = 1;
        }
        in.analysis.clear();
        const int first = in.first, nb = in.nb;
        const int es = vi.ComponentSize();
        std::vector<float> fades((size_t)nb * 2);
        if (!amtgpu_erase_calc_fades(er_, analysis_.data(), vi.num_frames, first, nb, fades.data())) env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
        int rc[5];
        if (!amtgpu_erase_get_rect(er_, rc)) env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
        const int imgx = rc[0], imgy = rc[1], w = rc[2], h = rc[3];
        if (imgx < 0 || imgy < 0 || imgx + w > vi.width || imgy + h > vi.height) env->ThrowError("[AMTEraseLogo] logo rectangle outside the frame");
        const int wUV = w >> 1, hUV = h >> 1, cx = imgx >> 1, cy = imgy >> 1;
        const uint64_t by = (uint64_t)w * h * es, buv = (uint64_t)wUV * hUV * es, per = by + 2 * buv;
        std::vector<PVideoFrame>& frames = in.frames;
        std::vector<int> slot(nb, -1);                /* position of the frame's rectangle in the device batch, -1: untouched */
        std::vector<float> bf;
        AmtGpuContext* g = ctx_->get();
        int m = 0;
        for (int i = 0; i < nb; ++i) {
            /* both fades 0: the reference's arithmetic is the identity -- for samples <= maxv, i.e. at 8 and 16 bits (at 10 / 12 bits
             * its clamp still rewrites out-of-range container values, LogoScan.hpp:1258): the frame is returned as it came */
            const int bpc = vi.BitsPerComponent();
            if (rc[4] && (bpc == 8 || bpc == 16) && fades[2 * i] == 0.0f && fades[2 * i + 1] == 0.0f) continue;
            env->MakeWritable(&frames[i]);
            slot[i] = m++;
            bf.push_back(fades[2 * i]); bf.push_back(fades[2 * i + 1]);
        }
        if (m > 0) {
            dbuf_.reserve(per * m);
            /* device layout: Y [m][h][w], then U [m][hUV][wUV], then V */
            uint8_t *dY = dbuf_.at(0), *dU = dbuf_.at(by * m), *dV = dbuf_.at(by * m + buv * m);
            /* frames of one clip share their pitches (AviSynth allocates them alike): the rectangles of the whole block then leave in
             * three gather uploads (Y, U, V) instead of three calls per frame */
            std::vector<const void*> sY, sU, sV;
            int pY0 = 0, pUV0 = 0;
            bool same_pitch = true;
            for (int i = 0; i < nb; ++i) {
                if (slot[i] < 0) continue;
                const int pY = frames[i]->GetPitch(PLANAR_Y), pUV = frames[i]->GetPitch(PLANAR_U);
                if (sY.empty()) { pY0 = pY; pUV0 = pUV; }
                same_pitch = same_pitch && pY == pY0 && pUV == pUV0 && frames[i]->GetPitch(PLANAR_V) == pUV0;
                sY.push_back(frames[i]->GetReadPtr(PLANAR_Y) + (size_t)imgy * pY + (size_t)imgx * es);
                sU.push_back(frames[i]->GetReadPtr(PLANAR_U) + (size_t)cy * pUV + (size_t)cx * es);
                sV.push_back(frames[i]->GetReadPtr(PLANAR_V) + (size_t)cy * pUV + (size_t)cx * es);
            }
            if (same_pitch) {
                if (!amtgpu_frames_upload_gather(g, dY, (int64_t)w * es, sY.data(), pY0, (uint64_t)w * es, h, m) ||
                    !amtgpu_frames_upload_gather(g, dU, (int64_t)wUV * es, sU.data(), pUV0, (uint64_t)wUV * es, hUV, m) ||
                    !amtgpu_frames_upload_gather(g, dV, (int64_t)wUV * es, sV.data(), pUV0, (uint64_t)wUV * es, hUV, m))
                    env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
            } else {
                for (int i = 0, k = 0; i < nb; ++i) {
                    if (slot[i] < 0) continue;
                    const int pY = frames[i]->GetPitch(PLANAR_Y), pUV = frames[i]->GetPitch(PLANAR_U);
                    if (!amtgpu_frames_upload_strided(g, dY + by * k, (int64_t)w * es, sY[k], pY, (uint64_t)w * es, h) ||
                        !amtgpu_frames_upload_strided(g, dU + buv * k, (int64_t)wUV * es, sU[k], pUV, (uint64_t)wUV * es, hUV) ||
                        !amtgpu_frames_upload_strided(g, dV + buv * k, (int64_t)wUV * es, sV[k], pUV, (uint64_t)wUV * es, hUV))
                        env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
                    ++k;
                }
            }
            if (!amtgpu_frames_upload_wait(g)) env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
            if (!amtgpu_erase_rect_batch(er_, dY, dU, dV, (int64_t)by, (int64_t)buv, w, wUV, vi.BitsPerComponent(), m, bf.data()))
                env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
            /* one copy back; the library scatters the rows into the frames while it still holds the context's lock (the landing buffer
             * belongs to the context, which other filters on other threads share) */
            std::vector<AmtGpuScatter> pieces;
            pieces.reserve((size_t)m * 3);
            for (int i = 0; i < nb; ++i) {
                if (slot[i] < 0) continue;
                const int pY = frames[i]->GetPitch(PLANAR_Y), pUV = frames[i]->GetPitch(PLANAR_U);
                uint8_t* hY = frames[i]->GetWritePtr(PLANAR_Y) + (size_t)imgy * pY + (size_t)imgx * es;
                uint8_t* hU = frames[i]->GetWritePtr(PLANAR_U) + (size_t)cy * pUV + (size_t)cx * es;
                uint8_t* hV = frames[i]->GetWritePtr(PLANAR_V) + (size_t)cy * pUV + (size_t)cx * es;
                pieces.push_back(AmtGpuScatter{hY, pY, by * slot[i], (uint64_t)w * es, h});
                pieces.push_back(AmtGpuScatter{hU, pUV, by * m + buv * slot[i], (uint64_t)wUV * es, hUV});
                pieces.push_back(AmtGpuScatter{hV, pUV, by * m + buv * m + buv * slot[i], (uint64_t)wUV * es, hUV});
            }
            if (!amtgpu_download_scatter(g, dbuf_.at(0), per * m, pieces.data(), (int)pieces.size())) env->ThrowError("[AMTEraseLogo] %s", ctx_->error());
        }
        CachedBlock* victim = &cache_[0];
        for (CachedBlock& b : cache_)
            if (b.first < 0 || (victim->first >= 0 && b.used < victim->used)) victim = &b;
        victim->frames.swap(frames);
        victim->first = first;
        victim->used = ++tick_;
    }

public:
    AMTEraseLogo(PClip clip, PClip analyzeclip, const std::string& logoPath, const std::string& logofPath, int mode, int maxFadeLength,
                 IScriptEnvironment* env, PContext ctx = PContext(), int framesPerLaunch = 32)
        : GenericVideoFilter(clip), ctx_(ctx ? ctx : std::make_shared<Context>()), analyzeclip_(std::move(analyzeclip)), mode_(mode),
          maxFadeLength_(maxFadeLength), block_(std::max(1, framesPerLaunch)), dbuf_(ctx_)
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
        n = std::max(0, std::min(vi.num_frames - 1, n));
        const int first = n - n % block_;
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
            if (const PVideoFrame* hit = lookup(n)) return *hit;
            if (!inflight_.count(first)) break;
            cv_.wait(lock);                               /* another thread is pulling this block's upstream frames: wait for it */
        }
        inflight_.insert(first);
        lock.unlock();
        struct Done {                                     /* whatever happens below, the block leaves the in-flight set and waiters wake */
            AMTEraseLogo* self; int first; std::unique_lock<std::mutex>& lock;
            ~Done() { if (!lock.owns_lock()) lock.lock(); self->inflight_.erase(first); self->cv_.notify_all(); }
        } done{this, first, lock};
        Fetched f = fetch(first, env);                    /* upstream work: not under the lock */
        lock.lock();
        if (!lookup(n)) process(f, env);
        return *lookup(n);
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
    DeviceBuffer dY_[2];                          /* the two batches in flight */
    AmtGpuMarker* done_[2] = {nullptr, nullptr};  /* "the scan of the batch in this buffer has finished": this object's own markers */

    void check(int ok) const { if (!ok) throw std::runtime_error(ctx_->error()); }
    void release()
    {
        for (auto& m : done_) { if (m) amtgpu_marker_destroy(ctx_->get(), m); m = nullptr; }
        if (lf_) amtgpu_logoframe_destroy(lf_);
        lf_ = nullptr;
    }

public:
    LogoFrame(PContext ctx, const std::vector<std::string>& logofiles, float maskratio, int framesPerLaunch = 2048)
        : ctx_(std::move(ctx)), numLogos_((int)logofiles.size()), framesPerLaunch_(std::max(1, framesPerLaunch)), dY_{DeviceBuffer(ctx_), DeviceBuffer(ctx_)}
    {
        std::vector<const char*> paths;
        for (const auto& s : logofiles) paths.push_back(s.c_str());
        lf_ = amtgpu_logoframe_create(ctx_->get(), paths.data(), numLogos_, maskratio);
        if (!lf_) throw std::runtime_error(ctx_->error());
        for (auto& m : done_) {
            m = amtgpu_marker_create(ctx_->get());
            if (!m) { const std::string why = ctx_->error(); release(); throw std::runtime_error(why); }
        }
    }
    ~LogoFrame() { release(); }
    LogoFrame(const LogoFrame&) = delete;
    LogoFrame& operator=(const LogoFrame&) = delete;

    /* Batches of framesPerLaunch frames in two device buffers: while the GPU scans batch k the host pulls the frames of batch k + 1
     * out of the clip (AMTSource::GetFrame, AMTSource.hpp:721-780) and the copy engine brings them in.  Only the rows some logo's
     * rectangle covers travel and are resident: a device "frame" is those rows, addressed as if the rest were there. */
    void scanFrames(PClip clip, IScriptEnvironment* env)
    {
        const VideoInfo vi = clip->GetVideoInfo();
        const int es = vi.ComponentSize();
        numFrames_ = vi.num_frames;
        check(amtgpu_logoframe_begin(lf_, vi.width, vi.height, vi.BitsPerComponent(), vi.num_frames, (int)vi.fps_numerator,
                                     (int)vi.fps_denominator));
        int rows[2] = {0, 0}, cols[2] = {0, 0};
        check(amtgpu_logoframe_get_rows(lf_, rows));
        check(amtgpu_logoframe_get_columns(lf_, cols));
        const int r0 = std::max(0, std::min(vi.height, rows[0])), r1 = std::max(r0, std::min(vi.height, rows[1]));
        const int c0 = std::max(0, std::min(vi.width, cols[0])), c1 = std::max(c0, std::min(vi.width, cols[1]));
        int pitch = 0;
        uint64_t part = 0;                            /* bytes of a frame that are resident: rows [r0, r1) */
        int k = 0;
        for (int n0 = 0; n0 < vi.num_frames; n0 += framesPerLaunch_, ++k) {
            const int nb = std::min(framesPerLaunch_, vi.num_frames - n0);
            DeviceBuffer& buf = dY_[k & 1];
            check(amtgpu_marker_wait_on(ctx_->get(), done_[k & 1]));  /* the scan of batch k - 2 has left this buffer */
            /* frames leave in groups: one upload call (one copy launch) per kUploadGroup frames -- per frame, the API overhead of a
             * copy is several times the time its 32 KB take */
            std::vector<PVideoFrame> held;
            std::vector<const void*> srcs;
            auto flush = [&](int first_in_batch) {
                if (srcs.empty()) return;
                if (r1 > r0 && c1 > c0)       /* the columns of those rows that some rectangle covers */
                    check(amtgpu_frames_upload_gather(ctx_->get(), buf.at(part * first_in_batch) + (uint64_t)c0 * es, pitch, srcs.data(), pitch,
                                                      (uint64_t)(c1 - c0) * es, r1 - r0, (int)srcs.size()));
                held.clear();
                srcs.clear();
            };
            int group_first = 0;
            for (int i = 0; i < nb; ++i) {
                PVideoFrame f = clip->GetFrame(n0 + i, env);
                if (pitch == 0) {
                    pitch = f->GetPitch(PLANAR_Y);
                    part = (uint64_t)(r1 - r0) * pitch;
                } else if (f->GetPitch(PLANAR_Y) != pitch) {
                    throw std::runtime_error("[LogoFrame] frames of one clip must share a pitch");
                }
                if (i == 0) buf.reserve(part * std::min(framesPerLaunch_, vi.num_frames) + 64);   /* (+ tail slack of the virtual-row addressing) */
                if (srcs.empty()) group_first = i;
                srcs.push_back(f->GetReadPtr(PLANAR_Y) + (uint64_t)r0 * pitch + (uint64_t)c0 * es);
                held.push_back(std::move(f));
                if ((int)srcs.size() == kUploadGroup) flush(group_first);
            }
            flush(group_first);
            check(amtgpu_frames_upload_wait(ctx_->get()));
            /* frame i of the batch: rows r0.. at buf + part * i, i.e. its (virtual) row 0 sits r0 * pitch bytes before that */
            check(amtgpu_logoframe_scan_batch(lf_, buf.at(0) - (uint64_t)r0 * pitch, (int64_t)part, pitch / es, n0, nb));
            check(amtgpu_marker_record_on(ctx_->get(), done_[k & 1]));
        }
        check(amtgpu_context_synchronize(ctx_->get()));
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
    /* LogoScan.hpp:1632-1643 (CMAnalyze.hpp:296 keeps the call under #if 0) */
    void dumpResult(const std::string& basepath) { check(amtgpu_logoframe_dump_result(lf_, basepath.c_str())); }
    int getBestLogo() const { return amtgpu_logoframe_best_logo(lf_); }
    float getLogoRatio() const { return amtgpu_logoframe_logo_ratio(lf_); }
};

} /* namespace amtgpu */
#endif /* AMT_FILTERS_HPP */
