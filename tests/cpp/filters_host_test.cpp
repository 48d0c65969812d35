// filters_host_test.cpp -- drives include/amt_filters.hpp the way the reference's host drives its filters:
// a clip object (here: frames from a raw file) goes through LogoFrame (CMAnalyze.hpp:291-299), AMTAnalyzeLogo and
// AMTEraseLogo (FilteredSource.hpp MakeSource script: AMTEraseLogo(src, AMTAnalyzeLogo(src, logo), logo, logof)).
// Everything the filters return is dumped to files; tests/test_gpu_filters_cpp.py compares them with the CPU oracle.
//   filters_host_test <clip.raw> <logo.lgd> <logo2.lgd> <logof-in or -> <outdir> <device>
// clip.raw: int32 {W,H,bits,N,pitchY,pitchUV} then Y[N][H][pitchY], U[N][H/2][pitchUV], V[...] (elements of 1 or 2 bytes)
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "amt_filters.hpp"

using namespace amtavs;

// a host that records what a plugin registers (name, argument specification, factory), the way AviSynth's function table does
struct Registered { std::string name, params; IScriptEnvironment::ApplyFunc apply; void* user; };
class RecordingEnv : public IScriptEnvironment {
public:
    std::vector<Registered> funcs;
    void AddFunction(const char* name, const char* params, ApplyFunc apply, void* user) override { funcs.push_back({name, params, apply, user}); }
    const Registered& find(const std::string& n) const
    {
        for (const auto& f : funcs) if (f.name == n) return f;
        throw std::runtime_error("plugin did not register " + n);
    }
};
typedef const char* (*PluginInit3)(IScriptEnvironment*, const AVS_Linkage*);

static const char* load_plugin(const std::string& path, RecordingEnv& env)
{
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) throw std::runtime_error(std::string("dlopen: ") + dlerror());
    PluginInit3 init = reinterpret_cast<PluginInit3>(dlsym(h, "AvisynthPluginInit3"));
    if (!init) throw std::runtime_error("plugin exports no AvisynthPluginInit3");
    return init(&env, nullptr);
}

class RawClip : public IClip {
    VideoInfo vi_;
    int pitchY_, pitchUV_, es_;
    std::vector<uint8_t> Y_, U_, V_;
public:
    explicit RawClip(const std::string& path)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open " + path);
        int32_t h[6];
        f.read(reinterpret_cast<char*>(h), sizeof(h));
        vi_.width = h[0]; vi_.height = h[1]; vi_.num_frames = h[3];
        vi_.pixel_type = h[2] <= 8 ? VideoInfo::CS_YV12 : h[2] == 10 ? VideoInfo::CS_YUV420P10 : h[2] == 12 ? VideoInfo::CS_YUV420P12
                                                                                              : VideoInfo::CS_YUV420P16;
        pitchY_ = h[4]; pitchUV_ = h[5]; es_ = h[2] <= 8 ? 1 : 2;
        Y_.resize((size_t)h[3] * h[1] * pitchY_ * es_);
        U_.resize((size_t)h[3] * (h[1] / 2) * pitchUV_ * es_);
        V_.resize(U_.size());
        f.read(reinterpret_cast<char*>(Y_.data()), Y_.size());
        f.read(reinterpret_cast<char*>(U_.data()), U_.size());
        f.read(reinterpret_cast<char*>(V_.data()), V_.size());
        if (!f) throw std::runtime_error("short read " + path);
    }
    const VideoInfo& GetVideoInfo() override { return vi_; }
    PVideoFrame GetFrame(int n, IScriptEnvironment* env) override
    {
        n = std::max(0, std::min(vi_.num_frames - 1, n));        // AviSynth clamps frame numbers
        PVideoFrame f = env->NewVideoFrame(vi_);
        auto blit = [&](int plane, const std::vector<uint8_t>& src, int rows, int spitch) {
            const size_t frame = (size_t)rows * spitch * es_;
            for (int y = 0; y < rows; ++y)
                std::memcpy(f->GetWritePtr(plane) + (size_t)y * f->GetPitch(plane), src.data() + frame * n + (size_t)y * spitch * es_,
                            f->GetRowSize(plane));
        };
        blit(PLANAR_Y, Y_, vi_.height, pitchY_);
        blit(PLANAR_U, U_, vi_.height / 2, pitchUV_);
        blit(PLANAR_V, V_, vi_.height / 2, pitchUV_);
        return f;
    }
};

static void dump(const std::string& path, const void* p, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char*>(p), n);
}

int main(int argc, char** argv)
{
    if (argc == 3 && std::string(argv[1]) == "--registration") {
        // what the plugin registers, one "name<TAB>params" line each, then its description string (no GPU needed)
        try {
            RecordingEnv renv;
            const char* desc = load_plugin(argv[2], renv);
            for (const auto& f : renv.funcs) std::printf("%s\t%s\n", f.name.c_str(), f.params.c_str());
            std::printf("%s\n", desc);
            return 0;
        } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc < 7) { std::fprintf(stderr, "usage: %s clip.raw logo.lgd logo2.lgd logof|- outdir device\n", argv[0]); return 2; }
    const std::string clipPath = argv[1], logo = argv[2], logo2 = argv[3], logofIn = argv[4], out = argv[5];
    IScriptEnvironment env;
    try {
        auto ctx = std::make_shared<amtgpu::Context>(std::atoi(argv[6]));
        PClip src = std::make_shared<RawClip>(clipPath);
        const VideoInfo vi = src->GetVideoInfo();

        // 1. CM all-frames scan and its decisions
        amtgpu::LogoFrame lf(ctx, {logo, logo2, out + "/missing.lgd"}, 0.35f, /*framesPerLaunch*/ 7);
        lf.scanFrames(src, &env);
        const std::vector<float> ev = lf.evalResults();
        dump(out + "/eval.bin", ev.data(), ev.size() * sizeof(float));
        lf.selectLogo(2);
        lf.writeResult(out + "/logof.txt");
        {
            std::ofstream f(out + "/select.txt");
            f << lf.getBestLogo() << " " << std::hexfloat << lf.getLogoRatio() << "\n";
        }

        // 2. the analysis clip, every frame of it, in a scrambled order (block cache misses both ways)
        PClip an = std::make_shared<amtgpu::AMTAnalyzeLogo>(src, logo, 0.35f, &env, ctx, /*framesPerLaunch*/ 3);
        const VideoInfo avi = an->GetVideoInfo();
        std::vector<uint8_t> records((size_t)avi.num_frames * 1056);
        for (int pass = 0; pass < 2; ++pass)
            for (int n = pass ? avi.num_frames - 1 : 0; pass ? n >= 0 : n < avi.num_frames; pass ? --n : ++n) {
                PVideoFrame f = an->GetFrame(n, &env);
                if (pass == 0) std::memcpy(&records[(size_t)n * 1056], f->GetReadPtr(), 1056);
                else if (std::memcmp(&records[(size_t)n * 1056], f->GetReadPtr(), 1056)) throw std::runtime_error("analysis clip not reproducible");
            }
        dump(out + "/analysis.bin", records.data(), records.size());
        {
            std::ofstream f(out + "/analysis_vi.txt");
            f << avi.width << " " << avi.height << " " << avi.num_frames << " " << avi.pixel_type << "\n";
        }

        // 3. erase, frame by frame, with and without a logoframe file
        for (int variant = 0; variant < 2; ++variant) {
            const std::string lfile = variant ? (logofIn == "-" ? out + "/logof.txt" : logofIn) : "";
            PClip er = std::make_shared<amtgpu::AMTEraseLogo>(src, an, logo, lfile, 0, 16, &env, ctx);
            std::ofstream f(out + (variant ? "/erased_logof.raw" : "/erased.raw"), std::ios::binary);
            for (int n = 0; n < vi.num_frames; ++n) {
                PVideoFrame fr = er->GetFrame(n, &env);
                for (int plane : {PLANAR_Y, PLANAR_U, PLANAR_V})
                    for (int y = 0; y < fr->GetHeight(plane); ++y)
                        f.write(reinterpret_cast<const char*>(fr->GetReadPtr(plane)) + (size_t)y * fr->GetPitch(plane), fr->GetRowSize(plane));
            }
        }

        // 3b. the same graph built the way AviSynth builds it: through the factories the plugin registered with the
        //     reference's names / argument specifications (Amatsukaze.cpp:58-59), defaults left undefined
        if (argc >= 8) {
            RecordingEnv renv;
            load_plugin(argv[7], renv);
            const Registered& fa = renv.find("AMTAnalyzeLogo");
            const Registered& fe = renv.find("AMTEraseLogo");
            if (fa.params != "cs[maskratio]i" || fe.params != "ccs[logof]s[mode]i[maxfade]i") throw std::runtime_error("argument specification differs");
            AVSValue an2 = fa.apply(AVSValue(std::vector<AVSValue>{AVSValue(src), AVSValue(logo.c_str()), AVSValue()}), fa.user, &renv);
            AVSValue an3 = fa.apply(AVSValue(std::vector<AVSValue>{AVSValue(src), AVSValue(logo.c_str()), AVSValue(35)}), fa.user, &renv);
            for (int n = 0; n < avi.num_frames; ++n) {
                PVideoFrame a = an2.AsClip()->GetFrame(n, &renv), b = an3.AsClip()->GetFrame(n, &renv);
                if (std::memcmp(a->GetReadPtr(), &records[(size_t)n * 1056], 1056) || std::memcmp(b->GetReadPtr(), &records[(size_t)n * 1056], 1056))
                    throw std::runtime_error("plugin-built AMTAnalyzeLogo differs from the directly constructed one");
            }
            AVSValue er2 = fe.apply(AVSValue(std::vector<AVSValue>{AVSValue(src), an2, AVSValue(logo.c_str()), AVSValue(), AVSValue(), AVSValue()}),
                                    fe.user, &renv);
            std::ofstream f(out + "/erased_plugin.raw", std::ios::binary);
            for (int n = 0; n < vi.num_frames; ++n) {
                PVideoFrame fr = er2.AsClip()->GetFrame(n, &renv);
                for (int plane : {PLANAR_Y, PLANAR_U, PLANAR_V})
                    for (int y = 0; y < fr->GetHeight(plane); ++y)
                        f.write(reinterpret_cast<const char*>(fr->GetReadPtr(plane)) + (size_t)y * fr->GetPitch(plane), fr->GetRowSize(plane));
            }
            // a mode the GPU path does not provide must surface as a script error, as ThrowError does in the reference
            std::ofstream pf(out + "/plugin_errors.txt");
            try { fe.apply(AVSValue(std::vector<AVSValue>{AVSValue(src), an2, AVSValue(logo.c_str()), AVSValue(), AVSValue(1), AVSValue()}), fe.user, &renv); pf << "no error\n"; }
            catch (const AvisynthError& e) { pf << e.msg << "\n"; }
        }

        // 4. error texts of the constructors
        std::ofstream ef(out + "/errors.txt");
        try { amtgpu::AMTAnalyzeLogo bad(src, out + "/missing.lgd", 0.35f, &env, ctx); ef << "no error\n"; }
        catch (const AvisynthError& e) { ef << e.msg << "\n"; }
        try { amtgpu::AMTEraseLogo bad(src, an, logo, "", 1, 16, &env, ctx); ef << "no error\n"; }
        catch (const AvisynthError& e) { ef << e.msg << "\n"; }
    } catch (const AvisynthError& e) {
        std::fprintf(stderr, "AvisynthError: %s\n", e.msg.c_str());
        return 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    std::puts("ok");
    return 0;
}
