// amt_gpu_stats.hip -- C ABI part 3: self-specified whole-frame metrics and their host decisions.
#include "build_knobs.h"
#include "../../include/amt_gpu.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "api_common.hpp"
#include "stats_decisions.hpp"

namespace amt {
hipError_t launch_frame_stats(hipStream_t st, int bits, const void* dY, long long frame_stride_bytes, int pitch_elems, int W,
                              int H, const void* dprevY, int nframes, unsigned long long* dout);
}
using namespace amt;

struct AmtGpuFrameStats {
    AmtGpuContext* ctx;
    int width, height, bits;
    DevBuf<unsigned long long> dShard;      // amtgpu_framestats_sharded: the local records before the exchange
};

namespace {
// the exchange of amtgpu_framestats_allgather.  A rank whose own part failed (local_error) still enters every collective with
// neutral data -- the others would block in it for ever -- and all ranks throw together once the status is known.
void gather_frame_metrics(const AmtGpuCollectives* coll, const uint64_t* local, int first, int nlocal, int num_frames,
                          uint64_t* out, std::string local_error)
{
    const size_t rec = AMTGPU_FS_WORDS;
    const bool sharded = coll && coll->world > 1;
    if (local_error.empty()) {
        if (num_frames < 0 || first < 0 || nlocal < 0 || (long long)first + nlocal > num_frames) local_error = "frame range outside the clip";
        else if (!out || (nlocal > 0 && !local)) local_error = "null metrics pointer";
    }
    if (!sharded) {
        if (!local_error.empty()) throw std::runtime_error(local_error);
        if (first != 0 || nlocal != num_frames) throw std::runtime_error("a single rank must hold the whole clip");
        if (nlocal) std::memcpy(out, local, (size_t)nlocal * rec * sizeof(uint64_t));
        return;
    }
    if (!coll->allgather || coll->rank < 0 || coll->rank >= coll->world) throw std::runtime_error("AmtGpuCollectives incomplete");
    const bool ok = local_error.empty();
    const int64_t mine[3] = {ok ? first : 0, ok ? nlocal : 0, ok ? 1 : 0};
    std::vector<int64_t> ranges((size_t)coll->world * 3);
    if (!coll->allgather(coll->user, mine, ranges.data(), sizeof mine)) throw std::runtime_error("allgather failed");
    bool all_ok = true;
    int64_t nmax = 0, next = 0;
    bool tiles = true;                          // ranks hold contiguous ranges in rank order that tile [0, num_frames)
    for (int r = 0; r < coll->world; ++r) {
        const int64_t f = ranges[3 * r], n = ranges[3 * r + 1];
        all_ok = all_ok && ranges[3 * r + 2] == 1;
        tiles = tiles && f == next && n >= 0;
        next = f + n;
        nmax = std::max(nmax, n);
    }
    tiles = tiles && next == num_frames;
    if (!ok) throw std::runtime_error(local_error);
    if (!all_ok) throw std::runtime_error("another rank failed before the exchange of the frame metrics");
    if (!tiles) throw std::runtime_error("the ranks' frame ranges do not tile the clip in rank order");
    if (nmax == 0) return;
    std::vector<uint64_t> send((size_t)nmax * rec, 0), recv((size_t)nmax * rec * coll->world);
    if (nlocal) std::memcpy(send.data(), local, (size_t)nlocal * rec * sizeof(uint64_t));
    if (!coll->allgather(coll->user, send.data(), recv.data(), (int64_t)(send.size() * sizeof(uint64_t)))) throw std::runtime_error("allgather failed");
    for (int r = 0; r < coll->world; ++r) {
        const int64_t f = ranges[3 * r], n = ranges[3 * r + 1];
        if (n) std::memcpy(out + (size_t)f * rec, recv.data() + (size_t)r * nmax * rec, (size_t)n * rec * sizeof(uint64_t));
    }
}
} // namespace

extern "C" {

AmtGpuFrameStats* amtgpu_framestats_create(AmtGpuContext* c, int width, int height, int bits)
{
    AmtGpuFrameStats* fs = nullptr;
    guard(c, [&] {
        if (width <= 0 || height < 4 || bits < 8 || bits > 15) throw std::runtime_error("[FrameStats] unsupported frame format");
        fs = new AmtGpuFrameStats{c, width, height, bits};
    });
    return fs;
}
void amtgpu_framestats_destroy(AmtGpuFrameStats* fs) { delete fs; }

int amtgpu_framestats_batch(AmtGpuFrameStats* fs, const void* dY, int64_t frame_stride, int pitch, const void* dprevY, int nframes,
                            uint64_t* dout)
{
    return guard(fs->ctx, [&] {
        fs->ctx->bind();
        const int sp = fs->ctx->prof_begin("frame_stats_kernel");
        AMT_HIP(launch_frame_stats(fs->ctx->stream, fs->bits, dY, frame_stride, pitch, fs->width, fs->height, dprevY, nframes,
                                   (unsigned long long*)dout));
        fs->ctx->prof_end(sp);
    });
}

int amtgpu_framestats_allgather(AmtGpuFrameStats* fs, const AmtGpuCollectives* coll, const uint64_t* local_metrics, int first, int nlocal,
                                int num_frames, uint64_t* metrics_out)
{
    return guard(fs->ctx, [&] { gather_frame_metrics(coll, local_metrics, first, nlocal, num_frames, metrics_out, std::string()); });
}

int amtgpu_framestats_sharded(AmtGpuFrameStats* fs, const AmtGpuCollectives* coll, const void* dY, int64_t frame_stride, int pitch,
                              const void* dprevY, int first, int nlocal, int num_frames, uint64_t* metrics_out)
{
    return guard(fs->ctx, [&] {
        std::string err;
        std::vector<uint64_t> local;
        try {
            if (first < 0 || nlocal < 0 || (long long)first + nlocal > num_frames) throw std::runtime_error("frame range outside the clip");
            // the first frame of a shard is compared with the frame before it: only the shard that starts the clip has none
            if (first > 0 && nlocal > 0 && !dprevY) throw std::runtime_error("[FrameStats] a shard that does not start the clip needs the frame before it (dprevY)");
            if (nlocal > 0) {
                fs->ctx->bind();
                const size_t n = (size_t)nlocal * AMTGPU_FS_WORDS;
                if (fs->dShard.size() < n) fs->dShard.alloc(n);
                const int sp = fs->ctx->prof_begin("frame_stats_kernel");
                AMT_HIP(launch_frame_stats(fs->ctx->stream, fs->bits, dY, frame_stride, pitch, fs->width, fs->height, first > 0 ? dprevY : nullptr,
                                           nlocal, fs->dShard.get()));
                fs->ctx->prof_end(sp);
                local.resize(n);
                download_via_pinned(fs->ctx, local.data(), fs->dShard.get(), n * sizeof(uint64_t));
            }
        } catch (const std::exception& e) { err = e.what(); }
        gather_frame_metrics(coll, local.data(), first, nlocal, num_frames, metrics_out, err);
    });
}

int amtgpu_cm_scene_changes(const uint64_t* metrics, int nframes, int width, int height, int* sc_out, int cap, int* nsc)
{
    try {
        const std::vector<int> sc = scene_changes(metrics, nframes, width, height);
        if (nsc) *nsc = (int)sc.size();
        for (int i = 0; i < (int)sc.size() && i < cap; ++i) sc_out[i] = sc[i];
        return (int)sc.size() <= cap ? 1 : 0;
    } catch (...) { return 0; }
}

int amtgpu_kfm_cadence(const uint64_t* metrics, int nframes, int width, int height, uint8_t* cadence_out, uint8_t* phase_out)
{
    try { classify_cadence(metrics, nframes, width, height, cadence_out, phase_out); return 1; }
    catch (...) { return 0; }
}

int amtgpu_kfm_write_durations(const uint8_t* cadence, const uint8_t* phase, int nframes, const char* path, int* nout)
{
    try {
        const std::vector<int> d = cadence_durations(cadence, phase, nframes);
        FILE* fp = std::fopen(path, "w");
        if (!fp) return 0;
        for (int v : d) std::fprintf(fp, "%d\n", v);
        std::fclose(fp);
        if (nout) *nout = (int)d.size();
        return 1;
    } catch (...) { return 0; }
}

// KFM timecode contract (FilteredSource.hpp:163-188): one integer per output frame = its start in ms (atoi'ed), comment lines
// start with '#'; a "# total: <seconds>" line ends the parse and gives the duration.  Output frame k starts at the sum of
// the 60p ticks of frames 0..k-1; a tick is fps_den / (2 * fps_num) seconds (1001/60000 for 29.97 fps sources).
int amtgpu_kfm_write_timecode(const uint8_t* cadence, const uint8_t* phase, int nframes, int fps_num, int fps_den, const char* path,
                              int* nout)
{
    try {
        if (fps_num <= 0 || fps_den <= 0) return 0;
        const std::vector<int> d = cadence_durations(cadence, phase, nframes);
        FILE* fp = std::fopen(path, "w");
        if (!fp) return 0;
        std::fprintf(fp, "# timecode format v2\n");
        const double tick_ms = 1000.0 * fps_den / (2.0 * fps_num);
        long long ticks = 0;
        for (int v : d) {
            std::fprintf(fp, "%lld\n", (long long)(ticks * tick_ms + 0.5));
            ticks += v;
        }
        std::fprintf(fp, "# total: %.6f\n", ticks * tick_ms / 1000.0);
        std::fclose(fp);
        if (nout) *nout = (int)d.size();
        return 1;
    } catch (...) { return 0; }
}

// chapter_exe's output as CMAnalyze::readSceneChanges parses it (CMAnalyze.hpp:411-439): everything up to a line that
// starts with "----" is header; then "SCPos: <frame>" lines (and "mute<k>: <a> - <b>" lines, which this build never
// writes: audio silence detection is out of scope).  The raw file also goes to join_logo_scp (:346-347).
int amtgpu_cm_write_chapter_exe(const int* scene_changes, int nsc, int nframes, const char* path)
{
    FILE* fp = std::fopen(path, "w");
    if (!fp) return 0;
    std::fprintf(fp, "amtgpu scene changes (self-specified field-difference detector), %d frames\n", nframes);
    std::fprintf(fp, "----------------------------------------\n");
    for (int i = 0; i < nsc; ++i) std::fprintf(fp, "\tSCPos: %d %d\n", scene_changes[i], scene_changes[i]);
    std::fclose(fp);
    return 1;
}

} // extern "C"
