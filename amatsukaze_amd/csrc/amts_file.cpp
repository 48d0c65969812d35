// amts_file.cpp -- reader of the stream-index file AMTSource is constructed from (amts%d.dat) and the frame-assembly plan it implies.
//
// Written by SaveAMTSource (AMTSource.hpp:835-852) with File::writeArray / writeValue (CoreUtils.hpp:275-284: int64 element count +
// raw elements; raw struct bytes), read back by LoadAMTSource (:854-871).  The structs are MSVC x64 PODs, so the layout is fixed
// here with explicit offsets instead of relying on this compiler's:
//   tchar            = wchar_t = UTF-16LE code unit (2 bytes)
//   VideoFormat      (StreamUtils.hpp:633-641)  44 bytes: enum format, 8 x int (width, height, displayWidth, displayHeight, sarWidth,
//                     sarHeight, frameRateNum, frameRateDenom), 3 x uint8 (colorPrimaries, transferCharacteristics, colorSpace),
//                     2 x bool (progressive, fixedFrameRate), 3 bytes of padding
//   AudioFormat      (:778-781)                  8 bytes: enum channels, int sampleRate
//   FilterSourceFrame (StreamReform.hpp:145-154) 48 bytes: bool halfDelay @0, int frameIndex @4, double pts @8, double frameDuration @16,
//                     int64 framePTS @24, int64 fileOffset @32, int keyFrame @40, enum cmType @44
//   FilterAudioFrame (:156-160)                  24 bytes: int frameIndex @0, int64 waveOffset @8, int waveLength @16
//   DecoderSetting   (StreamUtils.hpp:526-536)   12 bytes: 3 x enum (mpeg2, h264, hevc)
// These numbers are pinned to the reference's own definitions: oracle/ref_shim/layout_probe.cpp compiles the structs out of the reference
// headers (-fshort-wchar) and tests/test_abi_and_host.py::test_amts_layout_is_the_reference_structs checks sizeof/offsetof and reads a
// file written from those structs (tests/golden/amts_ref_layout.json, amts_ref_sample.dat) back through amtgpu_amts_load.
// The plan: AMTSource::OnFrameOutput (:482-566) matches every decoded picture to the frame list by its 33-bit PTS; a frame whose
// halfDelay is set is woven from the PREVIOUS picture's top field and this picture's bottom field (MakeFrame(prev, cur)), any other
// frame from one picture -- exactly the (top_index, bottom_index) pairs amtgpu_weave_fields_batch takes.
#include "build_knobs.h"
#include "../../include/amt_gpu.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "api_common.hpp"

struct AmtGpuAmtsFile {
    std::u16string srcpath, audiopath;
    int32_t vfmt[9] = {0};              // format, width, height, displayWidth, displayHeight, sarWidth, sarHeight, frameRateNum, frameRateDenom
    uint8_t color[3] = {0};
    uint8_t progressive = 0, fixedFrameRate = 0;
    int32_t audioChannels = 0, sampleRate = 0;
    struct Frame { uint8_t halfDelay; int32_t frameIndex; double pts, frameDuration; int64_t framePTS, fileOffset; int32_t keyFrame, cmType; };
    struct Audio { int32_t frameIndex; int64_t waveOffset; int32_t waveLength; };
    std::vector<Frame> frames;
    std::vector<Audio> audio;
    int32_t decoder[3] = {0};
};

namespace {

struct Reader {
    const std::vector<uint8_t>& b;
    size_t p = 0;
    explicit Reader(const std::vector<uint8_t>& v) : b(v) {}
    const uint8_t* take(size_t n)
    {
        if (n > b.size() - p) throw std::runtime_error("amts file truncated");
        const uint8_t* r = b.data() + p;
        p += n;
        return r;
    }
    template <typename T> T val() { T v; std::memcpy(&v, take(sizeof(T)), sizeof(T)); return v; }
    int64_t count(size_t elem)
    {
        const int64_t n = val<int64_t>();
        if (n < 0 || (uint64_t)n > (b.size() - p) / elem) throw std::runtime_error("amts file: array length out of range");
        return n;
    }
};

template <typename T> T at(const uint8_t* base, size_t off) { T v; std::memcpy(&v, base + off, sizeof(T)); return v; }

std::string utf8(const std::u16string& s) { return amt_utf8_from_utf16(reinterpret_cast<const uint16_t*>(s.data()), s.size()); }

} // namespace

extern "C" {

AmtGpuAmtsFile* amtgpu_amts_load(AmtGpuContext* c, const char* path)
{
    AmtGpuAmtsFile* out = nullptr;
    guard(c, [&] {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error(std::string("failed to open file ") + path);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        Reader r(bytes);
        std::unique_ptr<AmtGpuAmtsFile> a(new AmtGpuAmtsFile);
        for (std::u16string* s : {&a->srcpath, &a->audiopath}) {
            const int64_t n = r.count(2);
            const uint8_t* p = r.take((size_t)n * 2);
            s->resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) (*s)[(size_t)i] = (char16_t)(p[2 * i] | (p[2 * i + 1] << 8));
        }
        const uint8_t* v = r.take(44);                                 // VideoFormat
        for (int i = 0; i < 9; ++i) a->vfmt[i] = at<int32_t>(v, 4 * (size_t)i);
        a->color[0] = v[36]; a->color[1] = v[37]; a->color[2] = v[38];
        a->progressive = v[39]; a->fixedFrameRate = v[40];
        const uint8_t* af = r.take(8);                                 // AudioFormat
        a->audioChannels = at<int32_t>(af, 0); a->sampleRate = at<int32_t>(af, 4);
        const int64_t nf = r.count(48);                                // FilterSourceFrame[]
        a->frames.resize((size_t)nf);
        for (int64_t i = 0; i < nf; ++i) {
            const uint8_t* e = r.take(48);
            AmtGpuAmtsFile::Frame& F = a->frames[(size_t)i];
            F.halfDelay = e[0] != 0; F.frameIndex = at<int32_t>(e, 4); F.pts = at<double>(e, 8); F.frameDuration = at<double>(e, 16);
            F.framePTS = at<int64_t>(e, 24); F.fileOffset = at<int64_t>(e, 32); F.keyFrame = at<int32_t>(e, 40); F.cmType = at<int32_t>(e, 44);
        }
        const int64_t na = r.count(24);                                // FilterAudioFrame[]
        a->audio.resize((size_t)na);
        for (int64_t i = 0; i < na; ++i) {
            const uint8_t* e = r.take(24);
            a->audio[(size_t)i] = {at<int32_t>(e, 0), at<int64_t>(e, 8), at<int32_t>(e, 16)};
        }
        const uint8_t* d = r.take(12);                                 // DecoderSetting
        for (int i = 0; i < 3; ++i) a->decoder[i] = at<int32_t>(d, 4 * (size_t)i);
        out = a.release();
    });
    return out;
}

void amtgpu_amts_destroy(AmtGpuAmtsFile* a) { delete a; }

int amtgpu_amts_get_info(const AmtGpuAmtsFile* a, int* out19, int* num_frames, int* num_audio_frames)
{
    if (!a) return 0;
    if (out19) {
        for (int i = 0; i < 9; ++i) out19[i] = a->vfmt[i];
        out19[9] = a->color[0]; out19[10] = a->color[1]; out19[11] = a->color[2];
        out19[12] = a->progressive; out19[13] = a->fixedFrameRate;
        out19[14] = a->audioChannels; out19[15] = a->sampleRate;
        out19[16] = a->decoder[0]; out19[17] = a->decoder[1]; out19[18] = a->decoder[2];
    }
    if (num_frames) *num_frames = (int)a->frames.size();
    if (num_audio_frames) *num_audio_frames = (int)a->audio.size();
    return 1;
}

int amtgpu_amts_get_paths(const AmtGpuAmtsFile* a, char* src, int cap_src, char* audio, int cap_audio)
{
    if (!a) return 0;
    const std::string s = utf8(a->srcpath), w = utf8(a->audiopath);
    if ((src && (int)s.size() + 1 > cap_src) || (audio && (int)w.size() + 1 > cap_audio)) return 0;
    if (src) std::memcpy(src, s.c_str(), s.size() + 1);
    if (audio) std::memcpy(audio, w.c_str(), w.size() + 1);
    return 1;
}

int amtgpu_amts_get_frames(const AmtGpuAmtsFile* a, int64_t* framePTS, int64_t* fileOffset, int* keyFrame, uint8_t* halfDelay, int* cmType)
{
    if (!a) return 0;
    for (size_t i = 0; i < a->frames.size(); ++i) {
        if (framePTS) framePTS[i] = a->frames[i].framePTS;
        if (fileOffset) fileOffset[i] = a->frames[i].fileOffset;
        if (keyFrame) keyFrame[i] = a->frames[i].keyFrame;
        if (halfDelay) halfDelay[i] = a->frames[i].halfDelay;
        if (cmType) cmType[i] = a->frames[i].cmType;
    }
    return 1;
}

// AMTSource::OnFrameOutput (AMTSource.hpp:482-566) over a sequence of decoded pictures in output order
int amtgpu_amts_weave_plan(const AmtGpuAmtsFile* a, const int64_t* picture_pts, int npictures, int* top_index, int* bottom_index)
{
    if (!a || !picture_pts || !top_index || !bottom_index || npictures < 0) return 0;
    const auto& fr = a->frames;
    const int nf = (int)fr.size();
    std::fill(top_index, top_index + nf, -1);
    std::fill(bottom_index, bottom_index + nf, -1);
    if (nf == 0) return 1;
    auto lower = [&](int64_t pts) {
        return (int)(std::lower_bound(fr.begin(), fr.end(), pts, [](const AmtGpuAmtsFile::Frame& e, int64_t p) { return e.framePTS < p; }) - fr.begin());
    };
    int prev = -1;                                        // the picture before this one, -1 after a discontinuity
    for (int k = 0; k < npictures; ++k) {
        int64_t pts = picture_pts[k] & ((int64_t(1) << 33) - 1);      // only the low 33 bits are trusted (:484-486)
        int it = lower(pts);
        if (it == 0 && pts < fr[0].framePTS) {                       // too small: look one wrap later (:493-500)
            pts += int64_t(1) << 33;
            it = lower(pts);
        }
        if (it == nf || fr[(size_t)it].framePTS != pts) { prev = -1; continue; }      // after the end / unknown PTS (:502-518)
        if (fr[(size_t)it].halfDelay) {
            if (top_index[it] < 0 && prev >= 0) { top_index[it] = prev; bottom_index[it] = k; }     // MakeFrame(prev, cur) (:531-533)
            if (it + 1 < nf && fr[(size_t)it + 1].framePTS == fr[(size_t)it].framePTS && top_index[it + 1] < 0) {
                top_index[it + 1] = k; bottom_index[it + 1] = k;                                    // (:540-551)
            }
        } else if (top_index[it] < 0) {
            top_index[it] = k; bottom_index[it] = k;                                                // MakeFrame(cur, cur) (:559-561)
        }
        prev = k;
    }
    return 1;
}

} // extern "C"
