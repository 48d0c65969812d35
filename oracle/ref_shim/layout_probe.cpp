// oracle/ref_shim/layout_probe.cpp -- TEST INFRASTRUCTURE ONLY.
// Pins the byte layout amatsukaze_amd/csrc/amts_file.cpp parses to the reference's OWN type definitions: build_ref.sh cuts
//   DECODER_TYPE, DecoderSetting, CMType, VIDEO_STREAM_FORMAT, VideoFormat, AUDIO_CHANNELS, AudioFormat   (StreamUtils.hpp:520-536,
//   538-543, 570-575, 633-693, 707-781) and FilterSourceFrame, FilterAudioFrame (StreamReform.hpp:145-160)
// out of the reference headers where they lie (by name, into a temporary "ref_types.inc" in the stage directory that is deleted after the
// build -- the whole headers need Win32/FFmpeg) and this TU compiles them with -fshort-wchar.  All members are int / enum / uint8_t / bool /
// double / int64_t, for which the Itanium x86-64 ABI and MSVC x64 agree on size and alignment, so sizeof/offsetof here are the reference
// binary's.
//   layout_probe layout            -> JSON of sizeof / offsetof
//   layout_probe sample <out.dat>  -> an amts file written the way SaveAMTSource does (AMTSource.hpp:835-852: writeArray = int64 count +
//                                     raw elements, writeValue = raw struct bytes; CoreUtils.hpp:275-284) from the reference structs, with
//                                     known field values the test reads back through amtgpu_amts_load.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ref_types.inc"

static_assert(sizeof(wchar_t) == 2, "compile with -fshort-wchar: tchar is a UTF-16 code unit in the reference");

#define SZ(T) std::printf("  \"sizeof(" #T ")\": %zu,\n", sizeof(T))
#define OFF(T, m) std::printf("  \"" #T "." #m "\": %zu,\n", offsetof(T, m))

template <typename T> static void write_value(FILE* f, const T& v) { std::fwrite(&v, sizeof(T), 1, f); }
template <typename T> static void write_array(FILE* f, const T* p, int64_t n)
{
    write_value(f, n);
    std::fwrite(p, sizeof(T), (size_t)n, f);
}

int main(int argc, char** argv)
{
    if (argc >= 2 && std::strcmp(argv[1], "layout") == 0) {
        std::printf("{\n");
        SZ(wchar_t); SZ(VideoFormat); SZ(AudioFormat); SZ(FilterSourceFrame); SZ(FilterAudioFrame); SZ(DecoderSetting);
        OFF(VideoFormat, format); OFF(VideoFormat, width); OFF(VideoFormat, height); OFF(VideoFormat, displayWidth);
        OFF(VideoFormat, displayHeight); OFF(VideoFormat, sarWidth); OFF(VideoFormat, sarHeight); OFF(VideoFormat, frameRateNum);
        OFF(VideoFormat, frameRateDenom); OFF(VideoFormat, colorPrimaries); OFF(VideoFormat, transferCharacteristics);
        OFF(VideoFormat, colorSpace); OFF(VideoFormat, progressive); OFF(VideoFormat, fixedFrameRate);
        OFF(AudioFormat, channels); OFF(AudioFormat, sampleRate);
        OFF(FilterSourceFrame, halfDelay); OFF(FilterSourceFrame, frameIndex); OFF(FilterSourceFrame, pts);
        OFF(FilterSourceFrame, frameDuration); OFF(FilterSourceFrame, framePTS); OFF(FilterSourceFrame, fileOffset);
        OFF(FilterSourceFrame, keyFrame); OFF(FilterSourceFrame, cmType);
        OFF(FilterAudioFrame, frameIndex); OFF(FilterAudioFrame, waveOffset); OFF(FilterAudioFrame, waveLength);
        OFF(DecoderSetting, mpeg2); OFF(DecoderSetting, h264); OFF(DecoderSetting, hevc);
        std::printf("  \"CMTYPE_CM\": %d, \"VS_H264\": %d, \"AUDIO_32_LFE\": %d, \"DECODER_CUVID\": %d\n}\n", (int)CMTYPE_CM, (int)VS_H264,
                    (int)AUDIO_32_LFE, (int)DECODER_CUVID);
        return 0;
    }
    if (argc >= 3 && std::strcmp(argv[1], "sample") == 0) {
        FILE* f = std::fopen(argv[2], "wb");
        if (!f) return 2;
        // char16_t, not wchar_t: libc's wcslen (behind std::wstring) assumes its own 4-byte wchar_t whatever -fshort-wchar says
        static_assert(sizeof(char16_t) == sizeof(wchar_t), "tchar");
        const std::u16string src = u"D:\\rec\\\u756a\u7d44 #12.ts", wav = u"C:\\tmp\\amt0\\a0-0.wav";      // a BMP CJK pair in the name
        write_array(f, src.data(), (int64_t)src.size());
        write_array(f, wav.data(), (int64_t)wav.size());
        VideoFormat vf;
        std::memset(&vf, 0xEE, sizeof vf);                          // padding bytes are whatever the writer's stack held
        vf.format = VS_H264; vf.width = 1440; vf.height = 1080; vf.displayWidth = 1440; vf.displayHeight = 1080;
        vf.sarWidth = 4; vf.sarHeight = 3; vf.frameRateNum = 30000; vf.frameRateDenom = 1001;
        vf.colorPrimaries = 1; vf.transferCharacteristics = 6; vf.colorSpace = 9; vf.progressive = false; vf.fixedFrameRate = true;
        write_value(f, vf);
        AudioFormat af; af.channels = AUDIO_32_LFE; af.sampleRate = 48000;
        write_value(f, af);
        std::vector<FilterSourceFrame> frames(7);
        std::memset(frames.data(), 0xEE, frames.size() * sizeof(FilterSourceFrame));
        for (int i = 0; i < 7; ++i) {
            FilterSourceFrame& F = frames[(size_t)i];
            F.halfDelay = (i == 3 || i == 4);
            F.frameIndex = 100 + i;
            F.pts = 1000.5 + 3003.0 * i;
            F.frameDuration = 3003.0;
            F.framePTS = (int64_t(1) << 33) - 6006 + 3003 * (int64_t)(i - (i >= 4 ? 1 : 0));      // crosses the 33-bit wrap; 3 and 4 share a PTS
            F.fileOffset = int64_t(5000000000) + 188 * (int64_t)i;
            F.keyFrame = (i % 3 == 0) ? i : -1;
            F.cmType = (i & 1) ? CMTYPE_CM : CMTYPE_NONCM;
        }
        write_array(f, frames.data(), (int64_t)frames.size());
        std::vector<FilterAudioFrame> audio(3);
        std::memset(audio.data(), 0xEE, audio.size() * sizeof(FilterAudioFrame));
        for (int i = 0; i < 3; ++i) { audio[(size_t)i].frameIndex = i; audio[(size_t)i].waveOffset = int64_t(1) << (31 + i); audio[(size_t)i].waveLength = 4096 + i; }
        write_array(f, audio.data(), (int64_t)audio.size());
        DecoderSetting ds; ds.mpeg2 = DECODER_DEFAULT; ds.h264 = DECODER_CUVID; ds.hevc = DECODER_QSV;
        write_value(f, ds);
        std::fclose(f);
        return 0;
    }
    std::fprintf(stderr, "usage: layout_probe layout | sample <out.dat>\n");
    return 1;
}
