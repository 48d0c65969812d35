/*
 * oracle/ref_shim/TranscodeSetting.hpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Stand-in for the reference's Windows-only include chain
 * (Amatsukaze/TranscodeSetting.hpp -> common.h -> <windows.h>, AviSynthNeo, FFmpeg, UtVideo)
 * so that the reference's OWN Amatsukaze/LogoScan.hpp + AMTLogo.hpp + ComputeKernel.cpp can be
 * compiled with g++ where they lie under /root/reference (see oracle/build_ref.sh).  Nothing here
 * is arithmetic: it only supplies the host types those headers name -- a mock AviSynth clip/frame
 * host, a raw-YUV "decoder" behind the FFmpeg function names, an identity "lossless codec" behind
 * the UtVideo names, File/StringBuilder/AMTContext plumbing.  The pixel math that runs is the
 * reference's.
 */
#pragma once

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iterator>
#include <map>
#include <memory>
#include <numeric>
#include <regex>
#include <string>
#include <vector>

#include "logo.h"   // the reference's include/logo.h (symlinked into the stage dir)

// ---- MSVC-isms --------------------------------------------------------------------------------
#define __declspec(x)
#define __stdcall
#define __cdecl
#define _T(x) x
typedef char tchar;
typedef std::string tstring;
inline tstring to_tstring(const char* s) { return tstring(s); }
inline tstring to_tstring(const std::string& s) { return s; }
inline std::string to_string(const tstring& s) { return s; }

template <size_t N> inline int strcpy_s(char (&dst)[N], const char* src)
{
    std::strncpy(dst, src, N - 1); dst[N - 1] = 0; return 0;
}
template <size_t N> inline int strncpy_s(char (&dst)[N], const char* src, size_t count)
{
    size_t n = std::min(count, N - 1);
    size_t i = 0;
    for (; i < n && src[i]; ++i) dst[i] = src[i];
    dst[i] = 0;
    return 0;
}
template <size_t N, typename... A> inline int sprintf_s(char (&dst)[N], const char* fmt, A... a)
{
    return std::snprintf(dst, N, fmt, a...);
}
namespace stdext {
template <typename P> struct checked_array_iterator_t { };
template <typename P> inline P checked_array_iterator(P p, size_t) { return p; }
}
// the reference spells it stdext::checked_array_iterator<float*>(k, KLEN): make that a function template call
// (template argument = pointer type, returns the raw pointer).

// ---- exceptions (CoreUtils.hpp:17-67) ---------------------------------------------------------
struct Exception {
    virtual ~Exception() { }
    virtual const char* message() const { return "No Message ..."; }
    virtual void raise() const { throw *this; }
};
#define SHIM_DEFINE_EXCEPTION(name) \
    struct name : public Exception { \
        name(const std::string& mes) : mes(mes) { } \
        virtual const char* message() const { return mes.c_str(); } \
        virtual void raise() const { throw *this; } \
    private: std::string mes; };
SHIM_DEFINE_EXCEPTION(EOFException)
SHIM_DEFINE_EXCEPTION(FormatException)
SHIM_DEFINE_EXCEPTION(InvalidOperationException)
SHIM_DEFINE_EXCEPTION(ArgumentException)
SHIM_DEFINE_EXCEPTION(IOException)
SHIM_DEFINE_EXCEPTION(RuntimeException)
SHIM_DEFINE_EXCEPTION(AviSynthException)
#undef SHIM_DEFINE_EXCEPTION

inline const char* shim_arg(const std::string& s) { return s.c_str(); }
template <typename T> inline T shim_arg(T v) { return v; }
template <typename... A> inline std::string StringFormat(const char* fmt, A... a)
{
    char buf[2048];
    std::snprintf(buf, sizeof buf, fmt, shim_arg(a)...);
    return buf;
}
inline std::string StringFormat(const char* fmt) { return fmt; }
#define THROW(exception, message) throw exception(std::string(message))
#define THROWF(exception, fmt, ...) throw exception(StringFormat(fmt, __VA_ARGS__))

struct MemoryChunk {
    MemoryChunk() : data(NULL), length(0) { }
    MemoryChunk(uint8_t* data, size_t length) : data(data), length(length) { }
    uint8_t* data;
    size_t length;
};

class StringBuilder {
    std::string s;
public:
    template <typename... A> StringBuilder& append(const char* fmt, A... a) { s += StringFormat(fmt, a...); return *this; }
    MemoryChunk getMC() { return MemoryChunk((uint8_t*)s.data(), s.size()); }
    std::string str() const { return s; }
};

// ---- File (CoreUtils.hpp:257-395), LLP64 on-disk layout for LOGO_FILE_HEADER ---------------------
class File {
    FILE* fp_;
public:
    File(const tstring& path, const tchar* mode) : fp_(std::fopen(path.c_str(), mode))
    {
        if (!fp_) THROWF(IOException, "failed to open file %s", path.c_str());
    }
    ~File() { if (fp_) std::fclose(fp_); }
    void write(MemoryChunk mc) const
    {
        if (mc.length == 0) return;
        if (std::fwrite(mc.data, mc.length, 1, fp_) != 1) THROW(IOException, "failed to write to file");
    }
    template <typename T> void writeValue(T v) const { write(MemoryChunk((uint8_t*)&v, sizeof(T))); }
    size_t read(MemoryChunk mc) const
    {
        size_t r = std::fread(mc.data, 1, mc.length, fp_);
        if (r != mc.length) THROW(IOException, "failed to read from file");
        return r;
    }
    template <typename T> T readValue() const
    {
        T v;
        read(MemoryChunk((uint8_t*)&v, sizeof(T)));
        return v;
    }
    void seek(int64_t offset, int origin) const { std::fseek(fp_, (long)offset, origin); }
    int64_t size() const
    {
        long cur = std::ftell(fp_);
        std::fseek(fp_, 0, SEEK_END);
        long sz = std::ftell(fp_);
        std::fseek(fp_, cur, SEEK_SET);
        return sz;
    }
    bool getline(std::string& line)
    {
        line.clear();
        int c;
        bool any = false;
        while ((c = std::fgetc(fp_)) != EOF) {
            any = true;
            if (c == '\n') break;
            line.push_back((char)c);
        }
        return any;
    }
    static bool exists(const tstring& path)
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (f) std::fclose(f);
        return f != nullptr;
    }
};
// `unsigned long` is 4 bytes where the reference is built (LLP64): 28 + 4 bytes on disk (include/logo.h:39-45)
template <> inline void File::writeValue<LOGO_FILE_HEADER>(LOGO_FILE_HEADER v) const
{
    write(MemoryChunk((uint8_t*)v.str, LOGO_FILE_HEADER_STR_SIZE));
    write(MemoryChunk((uint8_t*)v.logonum.c, 4));
}
template <> inline LOGO_FILE_HEADER File::readValue<LOGO_FILE_HEADER>() const
{
    LOGO_FILE_HEADER v;
    std::memset(&v, 0, sizeof v);
    read(MemoryChunk((uint8_t*)v.str, LOGO_FILE_HEADER_STR_SIZE));
    read(MemoryChunk((uint8_t*)v.logonum.c, 4));
    return v;
}

// ---- AMTContext / AMTObject (StreamUtils.hpp:343-511) -------------------------------------------
class AMTContext {
    std::string err_;
public:
    template <typename... A> void debugF(const char*, A...) const { }
    template <typename... A> void infoF(const char*, A...) const { }
    template <typename... A> void warnF(const char*, A...) const { }
    void debug(const char*) const { }
    void info(const char*) const { }
    void warn(const char*) const { }
    void setError(const Exception& e) { err_ = e.message(); }
    const std::string& getError() const { return err_; }
};
class AMTObject {
public:
    AMTObject(AMTContext& ctx) : ctx(ctx) { }
    virtual ~AMTObject() { }
    AMTContext& ctx;
};

inline static int nblocks(int n, int block) { return (n + block - 1) / block; }   // StreamUtils.hpp:35-38

template <typename T> struct shim_deleter { void operator()(T* p) const { delete p; } };
template <typename T> inline std::unique_ptr<T, shim_deleter<T>> make_unique_ptr(T* p) { return std::unique_ptr<T, shim_deleter<T>>(p); }

// YV12 crop copy (StreamUtils.hpp:934-1004 semantics: tight planar Y,U,V)
inline void CopyYV12(uint8_t* dst, const uint8_t* srcY, const uint8_t* srcU, const uint8_t* srcV,
                     int pitchY, int pitchUV, int width, int height)
{
    for (int y = 0; y < height; ++y) { std::memcpy(dst, srcY + (size_t)y * pitchY, width); dst += width; }
    for (int y = 0; y < height / 2; ++y) { std::memcpy(dst, srcU + (size_t)y * pitchUV, width / 2); dst += width / 2; }
    for (int y = 0; y < height / 2; ++y) { std::memcpy(dst, srcV + (size_t)y * pitchUV, width / 2); dst += width / 2; }
}

// ---- UtVideo stand-in: an identity "codec" (the reference only uses it as a lossless cache) -----
enum { UTVF_ULH0 = 1, UTVF_YV12 = 2, CBGROSSWIDTH_WINDOWS = 0 };
class CCodec {
    size_t frameSize_ = 0;
public:
    static CCodec* CreateInstance(int, const char*) { return new CCodec; }
    size_t EncodeGetOutputSize(int, int w, int h) { return (size_t)w * h * 3 / 2; }
    size_t EncodeGetExtraDataSize() { return 4; }
    int EncodeGetExtraData(void* p, size_t n, int, int, int) { std::memset(p, 0, n); return 0; }
    int EncodeBegin(int, int w, int h, int) { frameSize_ = (size_t)w * h * 3 / 2; return 0; }
    size_t EncodeFrame(void* out, bool* key, const void* in) { std::memcpy(out, in, frameSize_); if (key) *key = true; return frameSize_; }
    int EncodeEnd() { return 0; }
    int DecodeBegin(int, int w, int h, int, const void*, int) { frameSize_ = (size_t)w * h * 3 / 2; return 0; }
    size_t DecodeFrame(void* out, const void* in) { std::memcpy(out, in, frameSize_); return frameSize_; }
    int DecodeEnd() { return 0; }
};
typedef std::unique_ptr<CCodec, shim_deleter<CCodec>> CCodecPointer;

// lossless work file (StreamUtils.hpp:846-932): kept in process memory, keyed by path
class LosslessVideoFile {
    struct Store { int w = 0, h = 0; std::vector<uint8_t> extra; std::vector<std::vector<uint8_t>> frames; };
    static std::map<std::string, Store>& stores() { static std::map<std::string, Store> s; return s; }
    Store* st_;
public:
    LosslessVideoFile(AMTContext&, const tstring& path, const tchar* mode)
    {
        if (mode[0] == 'w') stores()[path] = Store();
        st_ = &stores()[path];
    }
    void writeHeader(int w, int h, int, const std::vector<uint8_t>& extra) { st_->w = w; st_->h = h; st_->extra = extra; }
    void readHeader() { }
    int getWidth() const { return st_->w; }
    int getHeight() const { return st_->h; }
    int getNumFrames() const { return (int)st_->frames.size(); }
    const std::vector<uint8_t>& getExtra() const { return st_->extra; }
    void writeFrame(const uint8_t* data, int len) { st_->frames.emplace_back(data, data + len); }
    int64_t readFrame(int n, uint8_t* data)
    {
        const auto& f = st_->frames.at(n);
        std::memcpy(data, f.data(), f.size());
        return (int64_t)f.size();
    }
};

// ---- FFmpeg stand-in: raw planar YUV420 8-bit clip file --------------------------------------
// file = int32 {magic 'AMTR', width, height, nframes} then frames (Y w*h, U, V tight)
enum AVCodecID { AV_CODEC_ID_RAWSHIM = 1 };
enum AVPixelFormat { AV_PIX_FMT_YUV420P = 0 };
struct AVCodec { int id; };
struct AVCodecParameters { AVCodecID codec_id; int width, height; };
struct AVStream { int index; AVCodecParameters* codecpar; };
struct AVPacket { int stream_index; int64_t pos; int frame_no; };
struct AVFrame { uint8_t* data[4]; int linesize[4]; int width, height, format; };
struct AVPixFmtDescriptor { int log2_chroma_w, log2_chroma_h; };
struct AVFormatContext {
    FILE* fp = nullptr; int w = 0, h = 0, n = 0, next = 0;
    AVStream stream; AVCodecParameters par;
};
struct AVCodecContext {
    int thread_count = 0; int w = 0, h = 0;
    AVFormatContext* fmt = nullptr; int pending = -1;
    std::vector<uint8_t> buf;
};
// the decoder reads pixels through the format context the packet came from
inline AVFormatContext*& shim_current_fmt() { static AVFormatContext* p = nullptr; return p; }

namespace av {
class InputContext {
    AVFormatContext ctx_;
public:
    InputContext(const tstring& src)
    {
        ctx_.fp = std::fopen(src.c_str(), "rb");
        if (!ctx_.fp) THROW(IOException, "avformat_open_input failed");
        int32_t hdr[4];
        if (std::fread(hdr, sizeof hdr, 1, ctx_.fp) != 1 || hdr[0] != 0x52544D41) THROW(FormatException, "bad raw clip");
        ctx_.w = hdr[1]; ctx_.h = hdr[2]; ctx_.n = hdr[3];
        ctx_.par.codec_id = AV_CODEC_ID_RAWSHIM; ctx_.par.width = ctx_.w; ctx_.par.height = ctx_.h;
        ctx_.stream.index = 0; ctx_.stream.codecpar = &ctx_.par;
        shim_current_fmt() = &ctx_;
    }
    ~InputContext() { if (ctx_.fp) std::fclose(ctx_.fp); if (shim_current_fmt() == &ctx_) shim_current_fmt() = nullptr; }
    AVFormatContext* operator()() { return &ctx_; }
};
class CodecContext {
    AVCodecContext ctx_;
public:
    CodecContext(AVCodec*) { }
    AVCodecContext* operator()() { return &ctx_; }
};
class Frame {
    AVFrame f_;
public:
    Frame() { std::memset(&f_, 0, sizeof f_); }
    AVFrame* operator()() { return &f_; }
};
inline AVStream* GetVideoStream(AVFormatContext* c, int) { return &c->stream; }
} // namespace av

inline int avformat_find_stream_info(AVFormatContext*, void*) { return 0; }
inline AVCodec* avcodec_find_decoder(AVCodecID) { static AVCodec c = {1}; return &c; }
inline int avcodec_parameters_to_context(AVCodecContext* c, const AVCodecParameters* p) { c->w = p->width; c->h = p->height; return 0; }
inline int GetProcessorCount() { return 4; }
inline int GetFFmpegThreads(int n) { return n; }
inline int avcodec_open2(AVCodecContext* c, AVCodec*, void*) { c->fmt = shim_current_fmt(); c->buf.resize((size_t)c->w * c->h * 3 / 2); return 0; }
inline int av_read_frame(AVFormatContext* c, AVPacket* pkt)
{
    if (c->next >= c->n) return -1;
    pkt->stream_index = 0;
    pkt->frame_no = c->next;
    pkt->pos = 16 + (int64_t)c->next * ((int64_t)c->w * c->h * 3 / 2);
    c->next++;
    return 0;
}
inline void av_packet_unref(AVPacket*) { }
inline int avcodec_send_packet(AVCodecContext* c, const AVPacket* pkt)
{
    c->pending = pkt ? pkt->frame_no : -1;
    return 0;
}
inline int avcodec_receive_frame(AVCodecContext* c, AVFrame* f)
{
    if (c->pending < 0) return -1;
    AVFormatContext* fc = c->fmt;
    size_t fsz = (size_t)fc->w * fc->h * 3 / 2;
    std::fseek(fc->fp, (long)(16 + (int64_t)c->pending * (int64_t)fsz), SEEK_SET);
    if (std::fread(c->buf.data(), fsz, 1, fc->fp) != 1) return -1;
    c->pending = -1;
    f->width = fc->w; f->height = fc->h; f->format = AV_PIX_FMT_YUV420P;
    f->data[0] = c->buf.data();
    f->data[1] = f->data[0] + (size_t)fc->w * fc->h;
    f->data[2] = f->data[1] + (size_t)(fc->w / 2) * (fc->h / 2);
    f->linesize[0] = fc->w; f->linesize[1] = f->linesize[2] = fc->w / 2;
    return 0;
}
inline const AVPixFmtDescriptor* av_pix_fmt_desc_get(AVPixelFormat) { static AVPixFmtDescriptor d = {1, 1}; return &d; }

// ---- AviSynth stand-in (include/avisynth.h names; planar 4:2:0 8/16-bit + BGR32 scratch frames) ----
enum { PLANAR_Y = 1 << 0, PLANAR_U = 1 << 1, PLANAR_V = 1 << 2 };
enum { CACHE_GET_MTMODE = 509 };
enum { MT_NICE_FILTER = 1, MT_MULTI_INSTANCE = 2, MT_SERIALIZED = 3 };

struct AvisynthError { const char* const msg; AvisynthError(const char* m) : msg(m) { } };

struct VideoInfo {
    enum { CS_BGR32 = 1, CS_YV12 = 2, CS_YUV420P16 = 3 };
    int width = 0, height = 0;
    unsigned fps_numerator = 30000, fps_denominator = 1001;
    int num_frames = 0;
    int pixel_type = CS_YV12;
    int bits_per_component = 8;     // only meaningful for planar types
    int BitsPerComponent() const { return pixel_type == CS_BGR32 ? 8 : bits_per_component; }
    int ComponentSize() const { return BitsPerComponent() <= 8 ? 1 : 2; }
};

class VideoFrame {
public:
    std::vector<uint8_t> buf[3];
    int pitch[3] = {0, 0, 0};
    static int pidx(int plane) { return plane == PLANAR_U ? 1 : plane == PLANAR_V ? 2 : 0; }
    const uint8_t* GetReadPtr(int plane = 0) const { return buf[pidx(plane)].data(); }
    uint8_t* GetWritePtr(int plane = 0) { return buf[pidx(plane)].data(); }
    int GetPitch(int plane = 0) const { return pitch[pidx(plane)]; }   // BYTES, like AviSynth
};
class PVideoFrame {
    std::shared_ptr<VideoFrame> p_;
public:
    PVideoFrame() { }
    PVideoFrame(std::nullptr_t) { }
    PVideoFrame(VideoFrame* f) : p_(f) { }
    VideoFrame* operator->() const { return p_.get(); }
    explicit operator bool() const { return (bool)p_; }
    long use_count() const { return p_.use_count(); }
};

inline int shim_align64(int v) { return (v + 63) & ~63; }
inline PVideoFrame shim_new_frame(const VideoInfo& vi)
{
    VideoFrame* f = new VideoFrame;
    if (vi.pixel_type == VideoInfo::CS_BGR32) {
        f->pitch[0] = shim_align64(vi.width * 4);
        f->buf[0].assign((size_t)f->pitch[0] * vi.height, 0);
    } else {
        int cs = vi.ComponentSize();
        f->pitch[0] = shim_align64(vi.width * cs);
        f->pitch[1] = f->pitch[2] = shim_align64((vi.width / 2) * cs);
        f->buf[0].assign((size_t)f->pitch[0] * vi.height, 0);
        f->buf[1].assign((size_t)f->pitch[1] * (vi.height / 2), 0);
        f->buf[2].assign((size_t)f->pitch[2] * (vi.height / 2), 0);
    }
    return PVideoFrame(f);
}

class IScriptEnvironment {
public:
    virtual ~IScriptEnvironment() { }
    void ThrowError(const char* fmt, ...)
    {
        static thread_local char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        std::vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        throw AvisynthError(buf);
    }
    PVideoFrame NewVideoFrame(const VideoInfo& vi) { return shim_new_frame(vi); }
    bool MakeWritable(PVideoFrame* pf)
    {
        if (pf->use_count() <= 1) return false;
        VideoFrame* c = new VideoFrame(*(*pf).operator->());
        *pf = PVideoFrame(c);
        return true;
    }
};
class IScriptEnvironment2 : public IScriptEnvironment { };

class IClip {
public:
    virtual ~IClip() { }
    virtual PVideoFrame GetFrame(int n, IScriptEnvironment* env) = 0;
    virtual const VideoInfo& GetVideoInfo() = 0;
    virtual int SetCacheHints(int, int) { return 0; }
};
class PClip {
    std::shared_ptr<IClip> p_;
public:
    PClip() { }
    PClip(IClip* c) : p_(c) { }
    IClip* operator->() const { return p_.get(); }
    explicit operator bool() const { return (bool)p_; }
};
class GenericVideoFilter : public IClip {
protected:
    PClip child;
    VideoInfo vi;
public:
    GenericVideoFilter(PClip c) : child(c) { vi = child->GetVideoInfo(); }
    PVideoFrame GetFrame(int n, IScriptEnvironment* env) { return child->GetFrame(n, env); }
    const VideoInfo& GetVideoInfo() { return vi; }
};

class AVSValue {
    int type_ = 0;   // 0 undefined, 1 clip, 2 string, 3 int, 4 float, 5 array
    PClip clip_; std::string s_; int i_ = 0; double f_ = 0; std::vector<AVSValue> arr_;
public:
    AVSValue() { }
    AVSValue(IClip* c) : type_(1), clip_(c) { }
    AVSValue(const PClip& c) : type_(1), clip_(c) { }
    AVSValue(const char* s) : type_(2), s_(s) { }
    AVSValue(int i) : type_(3), i_(i) { }
    AVSValue(double f) : type_(4), f_(f) { }
    AVSValue(const std::vector<AVSValue>& a) : type_(5), arr_(a) { }
    bool Defined() const { return type_ != 0; }
    PClip AsClip() const { return clip_; }
    const char* AsString() const { return s_.c_str(); }
    const char* AsString(const char* def) const { return type_ == 2 ? s_.c_str() : def; }
    int AsInt() const { return i_; }
    int AsInt(int def) const { return type_ == 3 ? i_ : def; }
    double AsFloat() const { return type_ == 3 ? i_ : f_; }
    double AsFloat(float def) const { return type_ == 3 ? (double)i_ : type_ == 4 ? f_ : (double)def; }
    const AVSValue& operator[](int i) const { return arr_.at(i); }
};
