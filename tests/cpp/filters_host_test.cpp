// filters_host_test.cpp -- drives include/amt_filters.hpp the way the reference's host drives its filters:
// a clip object (here: frames from a raw file) goes through LogoFrame (CMAnalyze.hpp:291-299), AMTAnalyzeLogo and
// AMTEraseLogo (FilteredSource.hpp MakeSource script: AMTEraseLogo(src, AMTAnalyzeLogo(src, logo), logo, logof)).
// Everything the filters return is dumped to files; tests/test_gpu_filters_cpp.py compares them with the CPU oracle.
//   filters_host_test <clip.raw> <logo.lgd> <logo2.lgd> <logof-in or -> <outdir> <device>
// clip.raw: int32 {W,H,bits,N,pitchY,pitchUV} then Y[N][H][pitchY], U[N][H/2][pitchUV], V[...] (elements of 1 or 2 bytes)
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <condition_variable>
#include <mutex>
#include <thread>

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

// A clip whose frames exist once and are handed out by reference (AviSynth's cache does the same): eight base pictures, each with
// and without the logo blended in (obs = (bg - B*maxv) / A, LogoScan.hpp:320-333), the logo present on frames (n / 300) % 2 == 0 with
// linear 12-frame fades at the transitions (those frames are blended on demand).  The source costs nothing, so what --bench
// measures is the filter layer: host copies, PCIe, launches.
class SynthClip : public IClip {
    VideoInfo vi_;
    std::vector<float> a_[3], b_[3];
    int lw_ = 0, lh_ = 0, lx_ = 0, ly_ = 0;
    std::vector<PVideoFrame> plain_, logo_;
    IScriptEnvironment env_;

    static double vis(int n)
    {
        const int period = 300, fade = 12, k = n % (2 * period);
        if (k < fade) return (k + 0.5) / fade;
        if (k < period) return 1.0;
        if (k < period + fade) return 1.0 - (k - period + 0.5) / fade;
        return 0.0;
    }
    PVideoFrame blended(int base, double v)
    {
        PVideoFrame f = std::make_shared<VideoFrame>(*plain_[base]);
        if (v <= 0) return f;
        for (int p = 0; p < 3; ++p) {
            const int plane = p == 0 ? PLANAR_Y : p == 1 ? PLANAR_U : PLANAR_V;
            const int w = p ? lw_ / 2 : lw_, h = p ? lh_ / 2 : lh_, x0 = p ? lx_ / 2 : lx_, y0 = p ? ly_ / 2 : ly_;
            for (int y = 0; y < h; ++y) {
                uint8_t* row = f->GetWritePtr(plane) + (size_t)(y0 + y) * f->GetPitch(plane) + x0;
                for (int x = 0; x < w; ++x) {
                    const float A = a_[p][(size_t)y * w + x], B = b_[p][(size_t)y * w + x];
                    const float with = (row[x] - B * 255.0f) / (A != 0.0f ? A : 1.0f);
                    const float o = (float)(v * with + (1.0 - v) * row[x]);
                    row[x] = (uint8_t)std::max(0.0f, std::min(255.0f, std::floor(o + 0.5f)));
                }
            }
        }
        return f;
    }
public:
    SynthClip(int W, int H, int N, const std::string& logoPath)
    {
        vi_.width = W; vi_.height = H; vi_.num_frames = N; vi_.pixel_type = VideoInfo::CS_YV12;
        AmtGpuLogo* lg = amtgpu_logo_load(nullptr, logoPath.c_str());
        if (!lg) throw std::runtime_error("cannot read " + logoPath);
        int info[8];
        amtgpu_logo_get_info(lg, info);
        lw_ = info[0]; lh_ = info[1]; lx_ = info[6]; ly_ = info[7];
        const size_t ys = (size_t)lw_ * lh_, cs = ys / 4;
        std::vector<float> planes((ys + 2 * cs) * 2);
        amtgpu_logo_get_planes(lg, planes.data());
        amtgpu_logo_destroy(lg);
        const float* q = planes.data();
        for (int p = 0; p < 3; ++p) {
            const size_t n = p ? cs : ys;
            a_[p].assign(q, q + n); q += n;
            b_[p].assign(q, q + n); q += n;
        }
        for (int k = 0; k < 8; ++k) {
            PVideoFrame f = env_.NewVideoFrame(vi_);
            uint32_t h = 0x9E3779B9u * (k + 1);
            for (int plane : {PLANAR_Y, PLANAR_U, PLANAR_V})
                for (int y = 0; y < f->GetHeight(plane); ++y) {
                    uint8_t* row = f->GetWritePtr(plane) + (size_t)y * f->GetPitch(plane);
                    for (int x = 0; x < f->GetRowSize(plane); ++x) {
                        h = h * 1664525u + 1013904223u;
                        row[x] = (uint8_t)(plane == PLANAR_Y ? 60 + ((x + 2 * y + 37 * k) % 120) + (h >> 29) : 120 + (h >> 28));
                    }
                }
            plain_.push_back(f);
            logo_.push_back(blended(k, 1.0));
        }
    }
    const VideoInfo& GetVideoInfo() override { return vi_; }
    PVideoFrame GetFrame(int n, IScriptEnvironment*) override
    {
        n = std::max(0, std::min(vi_.num_frames - 1, n));
        const double v = vis(n);
        if (v >= 1.0) return logo_[n & 7];
        if (v <= 0.0) return plain_[n & 7];
        // A partially faded frame exists once per (base picture, fade step) and is handed out by reference like the others -- what a
        // decoder's frame pool or AviSynth's frame cache does.  Blending a FRESH 2.3 MB frame for every request (round 3's source) made
        // the measured "filter layer" rates an artefact of the allocator: each such frame is a new mmap whose pages fault in one by one
        // (slow under virtualisation) and whose munmap shoots down the address space the GPU driver tracks -- blocks then took 20-40 ms
        // in steps of 10 ms (profiles/r04_notes.md, "Boundary").  AMT_BENCH_FRESH_FADES=1 brings that source back.
        static const bool fresh = std::getenv("AMT_BENCH_FRESH_FADES") != nullptr;
        if (fresh) return blended(n & 7, v);
        const int key = (n & 7) * 1024 + (int)(v * 1000.0 + 0.5);
        std::lock_guard<std::mutex> lk(fade_mu_);
        auto it = faded_.find(key);
        if (it == faded_.end()) it = faded_.emplace(key, blended(n & 7, v)).first;
        return it->second;
    }
private:
    std::mutex fade_mu_;
    std::map<int, PVideoFrame> faded_;
};

// --bench: frames per second THROUGH include/amt_filters.hpp -- what a host that pulls frames one GetFrame at a time gets
// (LogoScan.hpp:1343-1419, 1570-1589; FilteredSource.hpp:441-475).  One JSON line.
static int run_bench(int argc, char** argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: --bench W H N logo.lgd logo2.lgd logo3.lgd [device]\n"); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
    const std::string l1 = argv[5], l2 = argv[6], l3 = argc > 7 ? argv[7] : argv[6];
    IScriptEnvironment env;
    auto ctx = std::make_shared<amtgpu::Context>(argc > 8 ? std::atoi(argv[8]) : 0);
    PClip src = std::make_shared<SynthClip>(W, H, N, l1);
    auto secs = [](auto&& fn) {
        const auto t0 = std::chrono::steady_clock::now();
        fn();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    // the source alone (frames by reference + what the filters' callers do with them: nothing)
    const double t_src = secs([&] { for (int n = 0; n < N; ++n) src->GetFrame(n, &env); });
    // LogoFrame::scanFrames, 3 logos (CMAnalyze.hpp:291-299)
    double t_scan = 1;
    if (!std::getenv("AMT_BENCH_SKIP_SCAN")) {                         // (diagnostic)
        amtgpu::LogoFrame lf(ctx, {l1, l2, l3}, 0.35f);
        lf.scanFrames(src, &env);                                     // warm-up: tables, buffers, pinned ring
        t_scan = secs([&] { lf.scanFrames(src, &env); });
    }
    double t_an[2] = {0, 0}, t_er[2] = {0, 0};
    std::map<int, int> blk_hist[2];
    uint64_t sum[2] = {0, 0};
    const bool swap_order = std::getenv("AMT_BENCH_SWAP") != nullptr;      // (diagnostic: the fast mode first)
    for (int pass = 0; pass < 2; ++pass) {
        const int mode = swap_order ? 1 - pass : pass;
        // AMTAnalyzeLogo pulled frame by frame (8 source frames per analysis frame)
        PClip an = std::make_shared<amtgpu::AMTAnalyzeLogo>(src, l1, 0.35f, &env, ctx, 32, mode ? AMTGPU_ANALYZE_LINEAR_GUARDED : AMTGPU_ANALYZE_EXACT);
        const int na = an->GetVideoInfo().num_frames;
        an->GetFrame(0, &env);
        PClip an2 = std::make_shared<amtgpu::AMTAnalyzeLogo>(src, l1, 0.35f, &env, ctx, 32, mode ? AMTGPU_ANALYZE_LINEAR_GUARDED : AMTGPU_ANALYZE_EXACT);
        if (std::getenv("AMT_BENCH_VERBOSE")) {                      // per-call times of the first blocks (diagnostic)
            PClip an4 = std::make_shared<amtgpu::AMTAnalyzeLogo>(src, l1, 0.35f, &env, ctx, 32, mode ? AMTGPU_ANALYZE_LINEAR_GUARDED : AMTGPU_ANALYZE_EXACT);
            amtgpu_profile_enable(ctx->get(), 1);
            for (int n = 0; n < std::min(na, 200); ++n) {
                const double t = secs([&] { an4->GetFrame(n, &env); });
                if (t > 2e-4) std::fprintf(stderr, "mode %d GetFrame(%d): %.3f ms\n", mode, n, t * 1e3);
            }
            char rep[4096];
            if (amtgpu_profile_report(ctx->get(), rep, sizeof rep) >= 0) std::fprintf(stderr, "kernel times (name calls total_ms):\n%s\n", rep);
            amtgpu_profile_enable(ctx->get(), 0);
        }
        // per-call latency of the calls that launch a block (the others are served from the filter's cache): histogram in whole ms
        std::map<int, int>& hist = blk_hist[mode];
        t_an[mode] = secs([&] {
            for (int n = 0; n < na; ++n) {
                const double t = secs([&] { an2->GetFrame(n, &env); });
                if (t > 2e-4) hist[(int)(t * 1e3 + 0.5)] += 1;
            }
        });
        // the MakeSource graph: AMTEraseLogo(src, AMTAnalyzeLogo(src, logo), logo) pulled in order, as the encoder does
        PClip an3 = std::make_shared<amtgpu::AMTAnalyzeLogo>(src, l1, 0.35f, &env, ctx, 32, mode ? AMTGPU_ANALYZE_LINEAR_GUARDED : AMTGPU_ANALYZE_EXACT);
        PClip er = std::make_shared<amtgpu::AMTEraseLogo>(src, an3, l1, "", 0, 16, &env, ctx);
        t_er[mode] = secs([&] {
            for (int n = 0; n < N; ++n) {
                PVideoFrame f = er->GetFrame(n, &env);
                sum[mode] += f->GetReadPtr(PLANAR_Y)[(size_t)80 * f->GetPitch(PLANAR_Y) + 1200];       // (the frames are used)
            }
        });
    }
    std::printf("{\"what\": \"frames/s through include/amt_filters.hpp (GetFrame by GetFrame, frames in host memory, PCIe inclusive)\", "
                "\"frame\": \"%dx%d 8-bit\", \"frames\": %d, \"source_alone_fps\": %.0f, \"logoframe_scan_3_logos_fps\": %.0f, "
                "\"analyze_exact_fps\": %.0f, \"analyze_linear_guarded_fps\": %.0f, \"erase_graph_exact_fps\": %.0f, "
                "\"erase_graph_linear_guarded_fps\": %.0f, \"erased_frames_identical_in_both_modes\": %s, ",
                W, H, N, N / t_src, N / t_scan, N / t_an[0], N / t_an[1], N / t_er[0], N / t_er[1], sum[0] == sum[1] ? "true" : "false");
    for (int mode = 0; mode < 2; ++mode) {
        std::printf("\"analyze_%s_block_ms_hist\": {", mode ? "linear" : "exact");
        bool firstk = true;
        for (const auto& kv : blk_hist[mode]) { std::printf("%s\"%d\": %d", firstk ? "" : ", ", kv.first, kv.second); firstk = false; }
        std::printf("}%s", mode ? "}\n" : ", ");
    }
    return 0;
}

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
    if (argc >= 2 && std::string(argv[1]) == "--bench") {
        try { return run_bench(argc, argv); }
        catch (const AvisynthError& e) { std::fprintf(stderr, "AvisynthError: %s\n", e.msg.c_str()); return 1; }
        catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
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
        lf.dumpResult(out + "/dump_");                     // LogoScan.hpp:1632-1643: dump_0, dump_1, dump_2
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
            PClip er = std::make_shared<amtgpu::AMTEraseLogo>(src, an, logo, lfile, 0, 16, &env, ctx, /*framesPerLaunch*/ 5);
            std::ofstream f(out + (variant ? "/erased_logof.raw" : "/erased.raw"), std::ios::binary);
            for (int n = 0; n < vi.num_frames; ++n) {
                PVideoFrame fr = er->GetFrame(n, &env);
                for (int plane : {PLANAR_Y, PLANAR_U, PLANAR_V})
                    for (int y = 0; y < fr->GetHeight(plane); ++y)
                        f.write(reinterpret_cast<const char*>(fr->GetReadPtr(plane)) + (size_t)y * fr->GetPitch(plane), fr->GetRowSize(plane));
            }
        }

        // 3c. AviSynth's Prefetch threads: four threads ask for frames of the SAME block at once.  The block's upstream frames must be
        //     pulled once (the other threads wait for the fetch in flight), and every thread gets the frame the serial walk got.
        {
            struct CountingClip : IClip {
                PClip child; std::atomic<int> calls{0};
                explicit CountingClip(PClip c) : child(std::move(c)) {}
                const VideoInfo& GetVideoInfo() override { return child->GetVideoInfo(); }
                PVideoFrame GetFrame(int n, IScriptEnvironment* e) override
                {
                    ++calls;
                    std::this_thread::sleep_for(std::chrono::microseconds(300));      // a decoder-bound upstream: makes the race window real
                    return child->GetFrame(n, e);
                }
            };
            auto counted = std::make_shared<CountingClip>(src);
            const int blk = 5;
            PClip er = std::make_shared<amtgpu::AMTEraseLogo>(counted, an, logo, "", 0, 16, &env, ctx, blk);
            PClip serial = std::make_shared<amtgpu::AMTEraseLogo>(src, an, logo, "", 0, 16, &env, ctx, blk);
            std::ofstream cf(out + "/concurrent.txt");
            bool same = true, once = true;
            for (int b0 = 0; b0 < vi.num_frames; b0 += blk) {
                const int nb = std::min(blk, vi.num_frames - b0), before = counted->calls.load();
                std::vector<PVideoFrame> got(4);
                std::vector<std::thread> th;
                for (int t = 0; t < 4; ++t) th.emplace_back([&, t] { IScriptEnvironment e2; got[t] = er->GetFrame(b0 + t % nb, &e2); });
                for (auto& t : th) t.join();
                once = once && counted->calls.load() - before == nb;
                for (int t = 0; t < 4; ++t) {
                    PVideoFrame want = serial->GetFrame(b0 + t % nb, &env);
                    for (int plane : {PLANAR_Y, PLANAR_U, PLANAR_V})
                        for (int y = 0; y < want->GetHeight(plane); ++y)
                            same = same && !std::memcmp(want->GetReadPtr(plane) + (size_t)y * want->GetPitch(plane),
                                                        got[t]->GetReadPtr(plane) + (size_t)y * got[t]->GetPitch(plane), want->GetRowSize(plane));
                }
            }
            cf << "upstream_frames_pulled_once " << (once ? 1 : 0) << "\nframes_equal_serial " << (same ? 1 : 0) << "\n";

            // 3d. two Prefetch threads whose requests straddle a block boundary: thread 0 walks block k, thread 1 block k + 1, turn and
            //     turn about.  Each block's upstream frames must be pulled ONCE -- a one-block cache would evict k for k + 1 and back
            //     with every request (the reference's per-frame filter has no such cliff, LogoScan.hpp:1343-1419).
            auto counted2 = std::make_shared<CountingClip>(src);
            PClip er2 = std::make_shared<amtgpu::AMTEraseLogo>(counted2, an, logo, "", 0, 16, &env, ctx, blk);
            bool once2 = true, same2 = true;
            for (int b0 = 0; b0 + blk < vi.num_frames; b0 += 2 * blk) {
                const int nb1 = std::min(blk, vi.num_frames - (b0 + blk)), before = counted2->calls.load();
                std::vector<PVideoFrame> got0(blk), got1(nb1);
                std::mutex tm; std::condition_variable tcv; int turn = 0;       // strict alternation: the worst case for one cached block
                auto walk = [&](int me, int base, int cnt, std::vector<PVideoFrame>& gotv) {
                    IScriptEnvironment e2;
                    for (int i = 0; i < blk; ++i) {
                        { std::unique_lock<std::mutex> lk(tm); tcv.wait(lk, [&] { return turn == me; }); }
                        if (i < cnt) gotv[i] = er2->GetFrame(base + i, &e2);
                        { std::lock_guard<std::mutex> lk(tm); turn = 1 - me; }
                        tcv.notify_all();
                    }
                };
                std::thread t0(walk, 0, b0, blk, std::ref(got0)), t1(walk, 1, b0 + blk, nb1, std::ref(got1));
                t0.join(); t1.join();
                once2 = once2 && counted2->calls.load() - before == blk + nb1;
                for (int i = 0; i < blk + nb1; ++i) {
                    PVideoFrame want = serial->GetFrame(b0 + i, &env), have = i < blk ? got0[i] : got1[i - blk];
                    for (int plane : {PLANAR_Y, PLANAR_U, PLANAR_V})
                        for (int y = 0; y < want->GetHeight(plane); ++y)
                            same2 = same2 && !std::memcmp(want->GetReadPtr(plane) + (size_t)y * want->GetPitch(plane),
                                                          have->GetReadPtr(plane) + (size_t)y * have->GetPitch(plane), want->GetRowSize(plane));
                }
            }
            cf << "alternating_blocks_pulled_once " << (once2 ? 1 : 0) << "\nalternating_frames_equal_serial " << (same2 ? 1 : 0) << "\n";
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
