/*
 * amt_avs_min.h -- the handful of AviSynth host types the filter layer in amt_filters.hpp is written against.
 *
 * On the reference's host (MSVC + AviSynthNeo, include/avisynth.h) a maintainer includes the real header instead and
 * deletes this one: the names, members and call shapes below are the ones the reference's filters use
 * (LogoScan.hpp:1106-1236, 1238-1519; CMAnalyze.hpp:273-317): PClip / IClip::GetFrame / GetVideoInfo, PVideoFrame with
 * GetReadPtr / GetWritePtr / GetPitch / GetRowSize / GetHeight per plane, IScriptEnvironment::NewVideoFrame /
 * MakeWritable / ThrowError (-> AvisynthError), GenericVideoFilter with `child` and `vi`.  Only what those filters touch
 * is here; this is a stand-in for building and testing the layer on Linux, not an AviSynth implementation.
 */
#ifndef AMT_AVS_MIN_H
#define AMT_AVS_MIN_H

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace amtavs {

class AVSValue;
enum { PLANAR_Y = 1 << 0, PLANAR_U = 1 << 1, PLANAR_V = 1 << 2 };
enum { CACHE_GET_MTMODE = 509, MT_NICE_FILTER = 1 };

struct AvisynthError {
    std::string msg;
    explicit AvisynthError(std::string m) : msg(std::move(m)) {}
};

struct VideoInfo {
    enum { CS_YV12 = 1, CS_YUV420P10 = 2, CS_YUV420P12 = 3, CS_YUV420P14 = 4, CS_YUV420P16 = 5, CS_BGR32 = 100 };
    int width = 0, height = 0;
    unsigned fps_numerator = 30000, fps_denominator = 1001;
    int num_frames = 0;
    int pixel_type = CS_YV12;

    int BitsPerComponent() const
    {
        switch (pixel_type) {
        case CS_YUV420P10: return 10;
        case CS_YUV420P12: return 12;
        case CS_YUV420P14: return 14;
        case CS_YUV420P16: return 16;
        default: return 8;
        }
    }
    int ComponentSize() const { return BitsPerComponent() > 8 ? 2 : 1; }
    bool IsPlanar() const { return pixel_type != CS_BGR32; }
};

/* Frame memory is recycled, as AviSynth recycles its VideoFrameBuffers (a released buffer of the right size serves the next
 * NewVideoFrame / MakeWritable): a host that maps and unmaps megabytes per frame measures its allocator, not its filters. */
class FrameBufferPool {
    std::mutex mu_;
    std::multimap<size_t, std::vector<uint8_t>> free_;
    size_t held_ = 0;
    static constexpr size_t kMaxHeld = (size_t)2 << 30;
public:
    static FrameBufferPool& get() { static FrameBufferPool p; return p; }
    std::vector<uint8_t> acquire(size_t n)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = free_.find(n);
            if (it != free_.end()) {
                std::vector<uint8_t> v = std::move(it->second);
                free_.erase(it);
                held_ -= n;
                if (n <= 65536) std::memset(v.data(), 0, n);      /* (small frames -- the analysis clip's -- come back clean) */
                return v;
            }
        }
        return std::vector<uint8_t>(n, 0);
    }
    void release(std::vector<uint8_t>&& v)
    {
        const size_t n = v.size();
        if (!n) return;
        std::lock_guard<std::mutex> lk(mu_);
        if (held_ + n > kMaxHeld) return;          /* (the vector frees itself) */
        held_ += n;
        free_.emplace(n, std::move(v));
    }
};

/* planes are 64-byte aligned like AviSynth's (include/avs/config.h:45 FRAME_ALIGN) */
class VideoFrame {
    std::vector<uint8_t> buf_;
    int off_[3] = {0, 0, 0}, pitch_[3] = {0, 0, 0}, row_[3] = {0, 0, 0}, rows_[3] = {0, 0, 0};
    static int idx(int plane) { return plane == PLANAR_U ? 1 : plane == PLANAR_V ? 2 : 0; }
    void place(size_t total)
    {
        buf_ = FrameBufferPool::get().acquire(total + 64);
        const int shift = (int)((64 - (reinterpret_cast<uintptr_t>(buf_.data()) & 63)) & 63);
        for (int p = 0; p < 3; ++p) off_[p] += shift;
    }
public:
    ~VideoFrame() { FrameBufferPool::get().release(std::move(buf_)); }
    VideoFrame& operator=(const VideoFrame&) = delete;
    VideoFrame(const VideoFrame& o)
    {
        size_t total = 0;
        int o0 = 0;
        for (int p = 0; p < 3; ++p) { pitch_[p] = o.pitch_[p]; row_[p] = o.row_[p]; rows_[p] = o.rows_[p]; off_[p] = o0; o0 += pitch_[p] * rows_[p]; }
        total = (size_t)o0;
        place(total);
        for (int p = 0; p < 3; ++p)
            if (rows_[p]) std::memcpy(buf_.data() + off_[p], o.buf_.data() + o.off_[p], (size_t)pitch_[p] * rows_[p]);
    }
    explicit VideoFrame(const VideoInfo& vi)
    {
        auto al = [](int v) { return (v + 63) & ~63; };
        if (vi.IsPlanar()) {
            const int es = vi.ComponentSize();
            row_[0] = vi.width * es; rows_[0] = vi.height;
            row_[1] = row_[2] = (vi.width >> 1) * es; rows_[1] = rows_[2] = vi.height >> 1;
        } else {
            row_[0] = vi.width * 4; rows_[0] = vi.height;
        }
        int o = 0;
        for (int p = 0; p < 3; ++p) { pitch_[p] = row_[p] ? al(row_[p]) : 0; off_[p] = o; o += pitch_[p] * rows_[p]; }
        place((size_t)o);
    }
    const uint8_t* GetReadPtr(int plane = PLANAR_Y) const { return buf_.data() + off_[idx(plane)]; }
    uint8_t* GetWritePtr(int plane = PLANAR_Y) { return buf_.data() + off_[idx(plane)]; }
    int GetPitch(int plane = PLANAR_Y) const { return pitch_[idx(plane)]; }
    int GetRowSize(int plane = PLANAR_Y) const { return row_[idx(plane)]; }
    int GetHeight(int plane = PLANAR_Y) const { return rows_[idx(plane)]; }
};
typedef std::shared_ptr<VideoFrame> PVideoFrame;

class IScriptEnvironment {
public:
    virtual ~IScriptEnvironment() {}
    virtual PVideoFrame NewVideoFrame(const VideoInfo& vi) { return std::make_shared<VideoFrame>(vi); }
    /* frames handed out by this stand-in are never shared with a cache: copy-on-write only when someone else holds it */
    virtual bool MakeWritable(PVideoFrame* pf)
    {
        if (pf->use_count() > 1) *pf = std::make_shared<VideoFrame>(**pf);
        return true;
    }
    /* plugin registration (avisynth.h:1389): stored by a real host, recorded by test hosts */
    typedef class AVSValue (*ApplyFunc)(class AVSValue args, void* user_data, IScriptEnvironment* env);
    virtual void AddFunction(const char* /*name*/, const char* /*params*/, ApplyFunc /*apply*/, void* /*user_data*/) {}
    [[noreturn]] virtual void ThrowError(const char* fmt, ...)
    {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        throw AvisynthError(buf);
    }
};
typedef IScriptEnvironment IScriptEnvironment2;

class IClip {
public:
    virtual ~IClip() {}
    virtual PVideoFrame GetFrame(int n, IScriptEnvironment* env) = 0;
    virtual const VideoInfo& GetVideoInfo() = 0;
    virtual int SetCacheHints(int, int) { return 0; }
};
typedef std::shared_ptr<IClip> PClip;

/* ---- script values and plugin registration: what AvisynthPluginInit3 and the filters' Create factories touch
 *      (include/avisynth.h:1166-1260 AVSValue, :1389 AddFunction, Amatsukaze.cpp:43-65) ---- */
struct AVS_Linkage;     /* the real header's function-pointer table; opaque here (the stand-in links directly) */

class AVSValue {
    enum Kind { UNDEF, CLIP, BOOL, INT, FLOAT, STRING, ARRAY } kind_ = UNDEF;
    PClip clip_;
    long long i_ = 0;
    double f_ = 0;
    std::string s_;
    std::vector<AVSValue> arr_;
public:
    AVSValue() {}
    AVSValue(IClip* c) : kind_(CLIP), clip_(c) {}            /* takes ownership, like AviSynth's ref-counted PClip(IClip*) */
    AVSValue(const PClip& c) : kind_(CLIP), clip_(c) {}
    AVSValue(bool b) : kind_(BOOL), i_(b) {}
    AVSValue(int i) : kind_(INT), i_(i) {}
    AVSValue(float f) : kind_(FLOAT), f_(f) {}
    AVSValue(double f) : kind_(FLOAT), f_(f) {}
    AVSValue(const char* s) : kind_(STRING), s_(s) {}
    AVSValue(const std::vector<AVSValue>& a) : kind_(ARRAY), arr_(a) {}
    bool Defined() const { return kind_ != UNDEF; }
    bool IsClip() const { return kind_ == CLIP; }
    bool IsInt() const { return kind_ == INT; }
    bool IsFloat() const { return kind_ == FLOAT || kind_ == INT; }
    bool IsString() const { return kind_ == STRING; }
    bool IsArray() const { return kind_ == ARRAY; }
    PClip AsClip() const { if (kind_ != CLIP) throw AvisynthError("Invalid arguments: clip expected"); return clip_; }
    int AsInt() const { if (kind_ != INT) throw AvisynthError("Invalid arguments: int expected"); return (int)i_; }
    int AsInt(int def) const { return Defined() ? AsInt() : def; }
    double AsFloat() const { if (!IsFloat()) throw AvisynthError("Invalid arguments: float expected"); return kind_ == INT ? (double)i_ : f_; }
    double AsFloat(float def) const { return Defined() ? AsFloat() : (double)def; }
    const char* AsString() const { if (kind_ != STRING) throw AvisynthError("Invalid arguments: string expected"); return s_.c_str(); }
    const char* AsString(const char* def) const { return Defined() ? AsString() : def; }
    int ArraySize() const { return kind_ == ARRAY ? (int)arr_.size() : 1; }
    const AVSValue& operator[](int i) const
    {
        if (kind_ != ARRAY) { if (i == 0) return *this; throw AvisynthError("Invalid arguments: index out of range"); }
        return arr_.at((size_t)i);
    }
};

class GenericVideoFilter : public IClip {
protected:
    PClip child;
    VideoInfo vi;
public:
    explicit GenericVideoFilter(PClip c) : child(std::move(c)), vi(child->GetVideoInfo()) {}
    PVideoFrame GetFrame(int n, IScriptEnvironment* env) override { return child->GetFrame(n, env); }
    const VideoInfo& GetVideoInfo() override { return vi; }
};

} /* namespace amtavs */
#endif /* AMT_AVS_MIN_H */
