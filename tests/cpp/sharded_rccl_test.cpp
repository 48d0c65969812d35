// sharded_rccl_test.cpp -- the frame-sharded drivers of the C ABI over RCCL, the way a C++ host with one rank per GPU runs
// them (include/amt_rccl_collectives.hpp): LogoFrame::scanFrames sharded + all-gather of the records (LogoScan.hpp:1577-1584) and
// ScanLogo sharded -- quota of the first numMaxFrames valid frames in stream order (:885), three exact int64 all-reduces
// (:917-1036) -- and the CM / KFM frame metrics sharded with their one-frame halo + all-gather of the 64-byte records + replicated
// cadence / scene-change decisions (amtgpu_framestats_sharded), against the same calls on one GPU.  One thread per rank, communicators from ncclCommInitAll; the world is every
// visible GPU (1 on a single-GPU box: the collectives then run over a one-rank communicator -- RCCL is still what executes them).
//   sharded_rccl_test <clip.raw> <logo.lgd> <logo2.lgd> <outdir> imgx imgy w h numMaxFrames [max_ranks]
// clip.raw: int32 {W,H,bits(8),N,pitchY,pitchUV} then Y[N][H][pitchY], U[N][H/2][pitchUV], V[...]
// outdir gets single.lgd, sharded.lgd, eval_single.bin, eval_rank<r>.bin; prints "ok world=<n> ..." when everything matches.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "amt_rccl_collectives.hpp"

#define HIPCHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) throw std::runtime_error(std::string(#e) + ": " + hipGetErrorString(e_)); } while (0)

struct Clip {
    int W, H, N, pY, pUV;
    std::vector<uint8_t> Y, U, V;
};
static Clip read_clip(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    int32_t h[6];
    f.read(reinterpret_cast<char*>(h), sizeof h);
    if (h[2] != 8) throw std::runtime_error("8-bit clips only (ScanLogo is 8-bit, LogoScan.hpp:813)");
    Clip c{h[0], h[1], h[3], h[4], h[5], {}, {}, {}};
    c.Y.resize((size_t)c.N * c.H * c.pY);
    c.U.resize((size_t)c.N * (c.H / 2) * c.pUV);
    c.V.resize(c.U.size());
    f.read(reinterpret_cast<char*>(c.Y.data()), c.Y.size());
    f.read(reinterpret_cast<char*>(c.U.data()), c.U.size());
    f.read(reinterpret_cast<char*>(c.V.data()), c.V.size());
    if (!f) throw std::runtime_error("short read " + path);
    return c;
}
static std::string slurp(const std::string& p)
{
    std::ifstream f(p, std::ios::binary);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
// frames [a, b) of the clip on the current device, and the frame before them (the frame metrics' one-frame halo; null for a = 0)
struct DevShard {
    void *Y = nullptr, *U = nullptr, *V = nullptr, *prevY = nullptr;
    int n = 0;
    DevShard(const Clip& c, int a, int b) : n(b - a)
    {
        const size_t fy = (size_t)c.H * c.pY, fc = (size_t)(c.H / 2) * c.pUV;
        if (a > 0) {
            HIPCHK(hipMalloc(&prevY, fy));
            HIPCHK(hipMemcpy(prevY, c.Y.data() + fy * (a - 1), fy, hipMemcpyHostToDevice));
        }
        HIPCHK(hipMalloc(&Y, std::max<size_t>(1, fy * n)));
        HIPCHK(hipMalloc(&U, std::max<size_t>(1, fc * n)));
        HIPCHK(hipMalloc(&V, std::max<size_t>(1, fc * n)));
        if (n > 0) {
            HIPCHK(hipMemcpy(Y, c.Y.data() + fy * a, fy * n, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(U, c.U.data() + fc * a, fc * n, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(V, c.V.data() + fc * a, fc * n, hipMemcpyHostToDevice));
        }
    }
    ~DevShard() { (void)hipFree(Y); (void)hipFree(U); (void)hipFree(V); (void)hipFree(prevY); }
};
static void shard_range(int n, int rank, int world, int& a, int& b) { a = (int)((long long)n * rank / world); b = (int)((long long)n * (rank + 1) / world); }

int main(int argc, char** argv)
{
    if (argc < 10) { std::fprintf(stderr, "usage: %s clip.raw logo.lgd logo2.lgd outdir imgx imgy w h numMaxFrames [max_ranks]\n", argv[0]); return 2; }
    const std::string out = argv[4];
    const int imgx = std::atoi(argv[5]), imgy = std::atoi(argv[6]), w = std::atoi(argv[7]), h = std::atoi(argv[8]), cap = std::atoi(argv[9]);
    try {
        const Clip clip = read_clip(argv[1]);
        int ndev = 0;
        HIPCHK(hipGetDeviceCount(&ndev));
        if (ndev < 1) throw std::runtime_error("no HIP device");
        const int asked = argc > 10 ? std::atoi(argv[10]) : ndev;
        const int world = std::max(1, std::min(ndev, asked));
        if (asked > ndev)        // RCCL refuses two ranks of one communicator on the same device ("Duplicate GPU detected"): the multi-rank
            std::fprintf(stderr, "note: %d ranks asked for, %d device(s) visible: running %d rank(s) (RCCL does not place several ranks of a communicator on one "
                                 "device; the multi-rank control flow on one device is covered over gloo by tests/test_gpu_sharded.py)\n", asked, ndev, world);
        const char* paths[2] = {argv[2], argv[3]};

        // ---- one GPU: the answers the sharded runs must reproduce ----
        std::vector<float> eval_single((size_t)clip.N * 2 * 2);
        std::vector<uint64_t> metrics_single((size_t)clip.N * AMTGPU_FS_WORDS);
        {
            HIPCHK(hipSetDevice(0));
            AmtGpuContext* ctx = amtgpu_context_create(0);
            if (!ctx) throw std::runtime_error("amtgpu_context_create(0)");
            DevShard all(clip, 0, clip.N);
            if (!amtgpu_scanlogo(ctx, all.Y, all.U, all.V, (int64_t)clip.H * clip.pY, (int64_t)(clip.H / 2) * clip.pUV, clip.pY, clip.pUV, clip.W, clip.H,
                                 clip.N, 1041, (out + "/single.lgd").c_str(), imgx, imgy, w, h, 12, cap, nullptr))
                throw std::runtime_error(std::string("single-GPU ScanLogo: ") + amtgpu_last_error(ctx));
            AmtGpuLogoFrame* lf = amtgpu_logoframe_create(ctx, paths, 2, 0.35f);
            if (!lf) throw std::runtime_error(std::string("logoframe_create: ") + amtgpu_last_error(ctx));
            if (!amtgpu_logoframe_begin(lf, clip.W, clip.H, 8, clip.N, 30000, 1001) ||
                !amtgpu_logoframe_scan_batch(lf, all.Y, (int64_t)clip.H * clip.pY, clip.pY, 0, clip.N) || !amtgpu_logoframe_get_results(lf, eval_single.data()))
                throw std::runtime_error(std::string("single-GPU scan: ") + amtgpu_last_error(ctx));
            amtgpu_logoframe_destroy(lf);
            AmtGpuFrameStats* fs = amtgpu_framestats_create(ctx, clip.W, clip.H, 8);
            void* dm = nullptr;
            HIPCHK(hipMalloc(&dm, metrics_single.size() * 8));
            if (!fs || !amtgpu_framestats_batch(fs, all.Y, (int64_t)clip.H * clip.pY, clip.pY, nullptr, clip.N, (uint64_t*)dm) || !amtgpu_context_synchronize(ctx))
                throw std::runtime_error(std::string("single-GPU frame metrics: ") + amtgpu_last_error(ctx));
            HIPCHK(hipMemcpy(metrics_single.data(), dm, metrics_single.size() * 8, hipMemcpyDeviceToHost));
            (void)hipFree(dm);
            amtgpu_framestats_destroy(fs);
            amtgpu_context_destroy(ctx);
            std::ofstream(out + "/eval_single.bin", std::ios::binary).write(reinterpret_cast<const char*>(eval_single.data()), eval_single.size() * 4);
        }

        // ---- one rank per GPU over RCCL ----
        std::vector<ncclComm_t> comms(world);
        std::vector<int> devs(world);
        for (int r = 0; r < world; ++r) devs[r] = r;
        if (ncclCommInitAll(comms.data(), world, devs.data()) != ncclSuccess) throw std::runtime_error("ncclCommInitAll");
        int nccl_version = 0;
        ncclGetVersion(&nccl_version);
        std::vector<std::string> errors(world);
        std::vector<std::vector<float>> evals(world, std::vector<float>((size_t)clip.N * 2 * 2));
        std::vector<std::vector<uint64_t>> metrics(world, std::vector<uint64_t>((size_t)clip.N * AMTGPU_FS_WORDS));
        std::vector<std::thread> th;
        for (int r = 0; r < world; ++r)
            th.emplace_back([&, r] {
                try {
                    HIPCHK(hipSetDevice(r));
                    int nranks = 0, myrank = -1;
                    ncclCommCount(comms[r], &nranks);
                    ncclCommUserRank(comms[r], &myrank);
                    if (nranks != world || myrank != r) throw std::runtime_error("communicator does not describe this rank");
                    AmtGpuContext* ctx = amtgpu_context_create(r);
                    if (!ctx) throw std::runtime_error("amtgpu_context_create");
                    amtgpu::RcclCollectives coll(comms[r], r, world, r);
                    int a, b;
                    shard_range(clip.N, r, world, a, b);
                    DevShard mine(clip, a, b);
                    // the all-frames scan: own frames, then every rank holds the whole clip's records
                    AmtGpuLogoFrame* lf = amtgpu_logoframe_create(ctx, paths, 2, 0.35f);
                    if (!lf) throw std::runtime_error(std::string("logoframe_create: ") + amtgpu_last_error(ctx));
                    if (!amtgpu_logoframe_begin(lf, clip.W, clip.H, 8, clip.N, 30000, 1001) ||
                        !amtgpu_logoframe_scan_batch(lf, mine.Y, (int64_t)clip.H * clip.pY, clip.pY, a, b - a) ||
                        !amtgpu_logoframe_allgather_results(lf, coll.get(), a, b - a) || !amtgpu_logoframe_get_results(lf, evals[r].data()))
                        throw std::runtime_error(std::string("sharded scan: ") + amtgpu_last_error(ctx) + " / " + coll.last_error());
                    amtgpu_logoframe_destroy(lf);
                    // CM / KFM frame metrics: own range with the frame before it as halo, records all-gathered
                    AmtGpuFrameStats* fs = amtgpu_framestats_create(ctx, clip.W, clip.H, 8);
                    if (!fs || !amtgpu_framestats_sharded(fs, coll.get(), mine.Y, (int64_t)clip.H * clip.pY, clip.pY, mine.prevY, a, b - a, clip.N, metrics[r].data()))
                        throw std::runtime_error(std::string("sharded frame metrics: ") + amtgpu_last_error(ctx) + " / " + coll.last_error());
                    amtgpu_framestats_destroy(fs);
                    // logo generation
                    const std::string dst = out + "/sharded.lgd";
                    if (!amtgpu_scanlogo_sharded(ctx, coll.get(), mine.Y, mine.U, mine.V, (int64_t)clip.H * clip.pY, (int64_t)(clip.H / 2) * clip.pUV, clip.pY,
                                                 clip.pUV, clip.W, clip.H, b - a, 1041, r == 0 ? dst.c_str() : nullptr, imgx, imgy, w, h, 12, cap, nullptr))
                        throw std::runtime_error(std::string("sharded ScanLogo: ") + amtgpu_last_error(ctx) + " / " + coll.last_error());
                    amtgpu_context_destroy(ctx);
                } catch (const std::exception& e) { errors[r] = e.what(); }
            });
        for (auto& t : th) t.join();
        for (auto c : comms) ncclCommDestroy(c);
        for (int r = 0; r < world; ++r)
            if (!errors[r].empty()) throw std::runtime_error("rank " + std::to_string(r) + ": " + errors[r]);
        bool ok = slurp(out + "/single.lgd") == slurp(out + "/sharded.lgd") && !slurp(out + "/single.lgd").empty();
        if (!ok) std::fprintf(stderr, "sharded .lgd differs from the single-GPU one\n");
        for (int r = 0; r < world; ++r) {
            std::ofstream(out + "/eval_rank" + std::to_string(r) + ".bin", std::ios::binary).write(reinterpret_cast<const char*>(evals[r].data()), evals[r].size() * 4);
            if (std::memcmp(evals[r].data(), eval_single.data(), eval_single.size() * 4)) { std::fprintf(stderr, "rank %d: gathered records differ\n", r); ok = false; }
        }
        // the replicated decisions: every rank's gathered records are the single-GPU ones, so are the cadence and the scene changes
        auto decide = [&](const std::vector<uint64_t>& m, std::vector<uint8_t>& cad, std::vector<int>& sc) {
            cad.assign((size_t)clip.N * 2, 0);
            sc.assign((size_t)std::max(1, clip.N), 0);
            int nsc = 0;
            if (!amtgpu_kfm_cadence(m.data(), clip.N, clip.W, clip.H, cad.data(), cad.data() + clip.N) ||
                !amtgpu_cm_scene_changes(m.data(), clip.N, clip.W, clip.H, sc.data(), clip.N, &nsc))
                throw std::runtime_error("decisions");
            sc.resize(nsc);
        };
        std::vector<uint8_t> cad0, cadr;
        std::vector<int> sc0, scr;
        decide(metrics_single, cad0, sc0);
        for (int r = 0; r < world; ++r) {
            if (metrics[r] != metrics_single) { std::fprintf(stderr, "rank %d: gathered frame metrics differ\n", r); ok = false; }
            decide(metrics[r], cadr, scr);
            if (cadr != cad0 || scr != sc0) { std::fprintf(stderr, "rank %d: replicated decisions differ\n", r); ok = false; }
        }
        if (!ok) return 1;
        std::printf("ok world=%d devices=%d rccl=%d\n", world, ndev, nccl_version);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
