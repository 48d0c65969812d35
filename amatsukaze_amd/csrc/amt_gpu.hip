// amt_gpu.hip -- implementation of the C ABI declared in include/amt_gpu.h (part 1: context, ingest,
// logo model, LogoFrame, AMTAnalyzeLogo).  Parts 2/3 live in amt_gpu_erase_scan.hip / amt_gpu_stats.hip.
#include "build_knobs.h"
#include "../../include/amt_gpu.h"

#include <link.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "api_common.hpp"
#include "host_parallel.hpp"

using namespace amt;
#ifdef AMT_TRACE_CALLS
namespace amt { void trace_stamp(AmtGpuContext* c, hipStream_t st, int slot); }
#endif

extern "C" {

int amtgpu_abi_version(void) { return AMTGPU_ABI_VERSION; }
void amtgpu_host_set_parallelism(int max_threads, int min_frames_per_thread)
{
    amt::HostParallelism::max_threads().store(std::max(0, max_threads));
    amt::HostParallelism::min_frames().store(std::max(0, min_frames_per_thread));
}

int amtgpu_hip_runtimes_loaded(char* paths, int cap)
{
    struct Acc { std::vector<std::string> v; } acc;
    dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* user) {
        const char* name = info->dlpi_name ? info->dlpi_name : "";
        const char* base = std::strrchr(name, '/');
        base = base ? base + 1 : name;
        if (std::strncmp(base, "libamdhip64.so", 14) == 0) {
            auto& v = static_cast<Acc*>(user)->v;
            if (std::find(v.begin(), v.end(), std::string(name)) == v.end()) v.emplace_back(name);
        }
        return 0;
    }, &acc);
    if (paths && cap > 0) {
        std::string all;
        for (const auto& p : acc.v) { all += p; all += '\n'; }
        const size_t n = std::min(all.size(), (size_t)cap - 1);
        std::memcpy(paths, all.data(), n);
        paths[n] = 0;
    }
    return (int)acc.v.size();
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
AmtGpuContext* amtgpu_context_create(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr;
    AmtGpuContext* c = new AmtGpuContext;
    c->device = device;
    try {
        c->bind();
        AMT_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        AMT_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        AMT_HIP(hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
        for (auto& e : c->slot_free) AMT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        amt::upload_pool_default(c);
        c->stream = c->own_stream;
#ifdef AMT_TRACE_CALLS
        if (std::getenv("AMT_SAME_STREAM")) {                 // instrumented builds: uploads on the compute stream (no cross-stream wait)
            (void)hipStreamDestroy(c->copy_stream);
            c->copy_stream = c->own_stream;
        }
#endif
    } catch (const std::exception&) {
        amtgpu_context_destroy(c);
        return nullptr;
    }
    return c;
}

void amtgpu_context_destroy(AmtGpuContext* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    amt::context_stop_threads(c);                      // keep-alive heartbeat and staging workers
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);    // no copy may still read a slot or a registered range below
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (auto& r : c->registered) (void)hipHostUnregister((void*)r.first);
    for (auto& sp : c->prof_spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_down) (void)hipHostFree(c->pinned_down);
    for (auto& e : c->markers) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->slot_free) if (e) (void)hipEventDestroy(e);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->copy_stream && c->copy_stream != c->own_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* amtgpu_last_error(const AmtGpuContext* c) { return c ? c->err.c_str() : "no context"; }

int amtgpu_context_set_stream(AmtGpuContext* c, void* s)
{
    if (!c) return 0;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (s == AMTGPU_STREAM_LEGACY_DEFAULT) c->stream = nullptr;          // HIP's legacy default ("null") stream
    else c->stream = s ? (hipStream_t)s : c->own_stream;
    return 1;
}
// (the legacy default stream is reported by its sentinel, so that set_stream(get_stream()) is an identity)
void* amtgpu_context_get_stream(AmtGpuContext* c) { return !c ? nullptr : (c->stream ? (void*)c->stream : AMTGPU_STREAM_LEGACY_DEFAULT); }
int amtgpu_context_synchronize(AmtGpuContext* c)
{
    return guard(c, [&] { c->bind(); AMT_HIP(hipStreamSynchronize(c->stream)); });
}

void* amtgpu_stream_create_cu_range(AmtGpuContext* c, int first_cu, int num_cus)
{
    hipStream_t st = nullptr;
    guard(c, [&] {
        if (!c) throw std::runtime_error("no context");
        c->bind();
        hipDeviceProp_t prop;
        AMT_HIP(hipGetDeviceProperties(&prop, c->device));
        const int ncu = prop.multiProcessorCount;
        if (first_cu < 0 || num_cus <= 0 || first_cu + num_cus > ncu) throw std::runtime_error("compute-unit range outside the device");
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int i = first_cu; i < first_cu + num_cus; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
        AMT_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    });
    return (void*)st;
}
void amtgpu_stream_destroy(AmtGpuContext* c, void* s)
{
    if (c && s) { (void)hipSetDevice(c->device); (void)hipStreamDestroy((hipStream_t)s); }
}
int amtgpu_device_cu_count(AmtGpuContext* c)
{
    int n = 0;
    guard(c, [&] {
        if (!c) throw std::runtime_error("no context");
        hipDeviceProp_t prop;
        AMT_HIP(hipGetDeviceProperties(&prop, c->device));
        n = prop.multiProcessorCount;
    });
    return n;
}

// per-kernel timing (HIP events on the launch stream) for bench.py's roofline figures
int amtgpu_profile_enable(AmtGpuContext* c, int on)
{
    return guard(c, [&] {
        c->bind();
        c->prof_resolve();
        c->profiling = on != 0;
        if (on) { std::fill(c->prof_ms.begin(), c->prof_ms.end(), 0.0); std::fill(c->prof_calls.begin(), c->prof_calls.end(), 0LL); }
    });
}
// text: one line per kernel "name calls total_ms\n"; returns bytes written or -1
int amtgpu_profile_report(AmtGpuContext* c, char* out, int cap)
{
    int n = -1;
    guard(c, [&] {
        c->bind();
        c->prof_resolve();
        std::string s;
        char line[160];
        for (size_t i = 0; i < c->prof_names.size(); ++i) {
            std::snprintf(line, sizeof line, "%s %lld %.6f\n", c->prof_names[i].c_str(), c->prof_calls[i], c->prof_ms[i]);
            s += line;
        }
        if ((int)s.size() + 1 > cap) throw std::runtime_error("profile buffer too small");
        std::memcpy(out, s.c_str(), s.size() + 1);
        n = (int)s.size();
    });
    return n;
}

// ---------------------------------------------------------------------------------------------
// logo model
// ---------------------------------------------------------------------------------------------
AmtGpuLogo* amtgpu_logo_load(AmtGpuContext* c, const char* path)
{
    AmtGpuLogo* l = nullptr;
    guard(c, [&] { l = new AmtGpuLogo{load_lgd(path)}; });
    return l;
}
AmtGpuLogo* amtgpu_logo_loadW(AmtGpuContext* c, const uint16_t* path) { return amtgpu_logo_load(c, amt_utf8_from_utf16z(path).c_str()); }
int amtgpu_logo_get_header(const AmtGpuLogo* l, char* name, int name_cap, int* serviceId)
{
    if (!l) return 0;
    if (name) {
        if ((int)l->planes.name.size() + 1 > name_cap) return 0;
        std::memcpy(name, l->planes.name.c_str(), l->planes.name.size() + 1);
    }
    if (serviceId) *serviceId = l->planes.serviceId;
    return 1;
}
int amtgpu_logo_set_header(AmtGpuLogo* l, const char* name, int serviceId)
{
    if (!l) return 0;
    if (name) l->planes.name = std::string(name).substr(0, 254);          // char name[255] (AMTLogo.hpp:26)
    l->planes.serviceId = serviceId;
    return 1;
}

AmtGpuLogo* amtgpu_logo_from_planes(AmtGpuContext* c, int w, int h, int logUVx, int logUVy, int imgw, int imgh,
                                    int imgx, int imgy, const float* planes)
{
    AmtGpuLogo* l = nullptr;
    guard(c, [&] {
        if (w <= 0 || h <= 0 || (w & 1) || (h & 1)) throw std::runtime_error("logo size must be positive and even");
        std::unique_ptr<AmtGpuLogo> n(new AmtGpuLogo);
        LogoPlanes& P = n->planes;
        P.w = w; P.h = h; P.logUVx = logUVx; P.logUVy = logUVy; P.imgw = imgw; P.imgh = imgh; P.imgx = imgx; P.imgy = imgy;
        P.allocate();
        if (planes) std::memcpy(P.data.data(), planes, P.data.size() * sizeof(float));
        l = n.release();
    });
    return l;
}

int amtgpu_logo_save(AmtGpuContext* c, const AmtGpuLogo* l, const char* path, const char* name, int serviceId)
{
    return guard(c, [&] { save_lgd(l->planes, path, name ? name : "", serviceId); });
}
int amtgpu_logo_saveW(AmtGpuContext* c, const AmtGpuLogo* l, const uint16_t* path, const char* name, int serviceId)
{
    return amtgpu_logo_save(c, l, amt_utf8_from_utf16z(path).c_str(), name, serviceId);
}
void amtgpu_logo_destroy(AmtGpuLogo* l) { delete l; }

int amtgpu_logo_get_info(const AmtGpuLogo* l, int* o)
{
    if (!l || !o) return 0;
    const LogoPlanes& P = l->planes;
    o[0] = P.w; o[1] = P.h; o[2] = P.logUVx; o[3] = P.logUVy; o[4] = P.imgw; o[5] = P.imgh; o[6] = P.imgx; o[7] = P.imgy;
    return 1;
}
int amtgpu_logo_get_planes(const AmtGpuLogo* l, float* out)
{
    if (!l || !out) return 0;
    std::memcpy(out, l->planes.data.data(), l->planes.data.size() * sizeof(float));
    return 1;
}

int amtgpu_logo_mask_tables(AmtGpuContext* c, const AmtGpuLogo* l, int kind, float maskratio, int* maskpixels, int* count,
                            float* blackScore, uint8_t* mask, float* kernels, float* scales)
{
    return guard(c, [&] {
        if (kind < 0 || kind > 2) throw std::runtime_error("kind must be 0 (deint), 1 (top) or 2 (bottom)");
        LogoPlanes E = kind == 0 ? deinterlaced_logo(l->planes) : field_logo(l->planes, kind == 2);
        MaskTables T = build_mask_tables(E, maskratio);
        if (maskpixels) *maskpixels = T.maskpixels;
        if (count) *count = T.count;
        if (blackScore) *blackScore = T.blackScore;
        if (mask) std::memcpy(mask, T.mask.data(), T.mask.size());
        if (kernels) std::memcpy(kernels, T.kernels.data(), T.kernels.size() * sizeof(float));
        if (scales) std::memcpy(scales, T.scales.data(), T.scales.size() * sizeof(float));
    });
}

// ---------------------------------------------------------------------------------------------
// LogoFrame
// ---------------------------------------------------------------------------------------------
struct AmtGpuLogoFrame {
    AmtGpuContext* ctx;
    float maskratio;
    std::vector<std::unique_ptr<LogoPlanes>> logos;    // null = unreadable file
    // clip
    int width = 0, height = 0, bits = 8, numFrames = 0, fpsNum = 30000, fpsDen = 1001;
    std::unique_ptr<EvalEngine> engine;
    std::vector<int> slotOfEngineLogo;                  // engine logo -> logo index
    DevBuf<float> dResults;                             // numFrames * nlogos * 2
    std::vector<float> results;                         // host copy
    bool hostValid = false;
    LogoSelection sel;
    bool selected = false;
};

static AmtGpuLogoFrame* logoframe_new(AmtGpuContext* c, std::vector<std::unique_ptr<LogoPlanes>> logos, float maskratio)
{
    AmtGpuLogoFrame* lf = new AmtGpuLogoFrame;
    lf->ctx = c;
    lf->maskratio = maskratio;
    lf->logos = std::move(logos);
    return lf;
}

AmtGpuLogoFrame* amtgpu_logoframe_create(AmtGpuContext* c, const char* const* paths, int nlogos, float maskratio)
{
    AmtGpuLogoFrame* lf = nullptr;
    guard(c, [&] {
        std::vector<std::unique_ptr<LogoPlanes>> v(nlogos);
        for (int i = 0; i < nlogos; ++i) {
            try { v[i].reset(new LogoPlanes(load_lgd(paths[i]))); }
            catch (const std::exception&) { /* read errors are ignored, the slot scores {0,-1} */ }
        }
        lf = logoframe_new(c, std::move(v), maskratio);
    });
    return lf;
}

AmtGpuLogoFrame* amtgpu_logoframe_create_from_logos(AmtGpuContext* c, const AmtGpuLogo* const* logos, int nlogos, float maskratio)
{
    AmtGpuLogoFrame* lf = nullptr;
    guard(c, [&] {
        std::vector<std::unique_ptr<LogoPlanes>> v(nlogos);
        for (int i = 0; i < nlogos; ++i) if (logos[i]) v[i].reset(new LogoPlanes(logos[i]->planes));
        lf = logoframe_new(c, std::move(v), maskratio);
    });
    return lf;
}

void amtgpu_logoframe_destroy(AmtGpuLogoFrame* lf) { delete lf; }

int amtgpu_logoframe_begin(AmtGpuLogoFrame* lf, int width, int height, int bits, int num_frames, int fps_num, int fps_den)
{
    return guard(lf->ctx, [&] {
        if (num_frames < 0 || bits < 8 || bits > 16) throw std::runtime_error("[LogoFrame] Unsupported pixel format");
        lf->width = width; lf->height = height; lf->bits = bits; lf->numFrames = num_frames;
        lf->fpsNum = fps_num; lf->fpsDen = fps_den;
        const int nl = (int)lf->logos.size();
        std::vector<EvalLogoSpec> specs;
        lf->slotOfEngineLogo.clear();
        for (int i = 0; i < nl; ++i) {
            const LogoPlanes* P = lf->logos[i].get();
            if (!P || P->imgw != width || P->imgh != height) continue;      // scores {0,-1}
            if (P->imgx < 0 || P->imgy < 0 || P->imgx + P->w > width || P->imgy + P->h > height)
                throw std::runtime_error("logo rectangle outside the frame");
            EvalLogoSpec S;
            S.planes = deinterlaced_logo(*P);
            S.tables = build_mask_tables(S.planes, lf->maskratio);
            S.imgx = P->imgx; S.imgy = P->imgy; S.row0 = 0; S.row_step = 1; S.deint = 1;
            S.out_off = i * 2;
            specs.push_back(std::move(S));
            lf->slotOfEngineLogo.push_back(i);
        }
        lf->engine.reset(specs.empty() ? nullptr : new EvalEngine(lf->ctx, std::move(specs), {0.0f, 1.0f}, false, nl * 2,
                                                                        "logo_eval_fused_kernel.scan"));
        // invalid / mismatching logos keep {corr0, corr1} = {0, -1}
        lf->results.assign((size_t)num_frames * nl * 2, 0.0f);
        for (size_t i = 0; i < (size_t)num_frames * nl; ++i) lf->results[i * 2 + 1] = -1.0f;
        lf->dResults.alloc(std::max<size_t>(1, lf->results.size()));
        lf->ctx->bind();
        if (!lf->results.empty())
            AMT_HIP(hipMemcpyAsync(lf->dResults.get(), lf->results.data(), lf->results.size() * sizeof(float), hipMemcpyHostToDevice, lf->ctx->stream));
        AMT_HIP(hipStreamSynchronize(lf->ctx->stream));
        lf->hostValid = true;
        lf->selected = false;
    });
}

int amtgpu_logoframe_scan_batch(AmtGpuLogoFrame* lf, const void* dY, int64_t frame_stride, int pitch, int first, int nframes)
{
    return guard(lf->ctx, [&] {
        if (first < 0 || nframes < 0 || first + nframes > lf->numFrames) throw std::runtime_error("frame range outside the clip");
        if (!lf->engine || nframes == 0) return;
        const int nl = (int)lf->logos.size();
        lf->engine->run(dY, frame_stride, pitch, lf->bits, nframes, lf->dResults.get() + (size_t)first * nl * 2);
        lf->hostValid = false;
        lf->selected = false;
    });
}

static void logoframe_sync_results(AmtGpuLogoFrame* lf)
{
    if (lf->hostValid) return;
    lf->ctx->bind();
    download_via_pinned(lf->ctx, lf->results.data(), lf->dResults.get(), lf->results.size() * sizeof(float));
    lf->hostValid = true;
}

int amtgpu_logoframe_get_rows(const AmtGpuLogoFrame* lf, int* out2)
{
    if (!lf || !out2) return 0;
    int lo = 0x7FFFFFFF, hi = 0;
    for (const auto& l : lf->logos)
        if (l) { lo = std::min(lo, l->imgy); hi = std::max(hi, l->imgy + l->h); }
    if (lo >= hi) { lo = 0; hi = 0; }
    out2[0] = lo; out2[1] = hi;
    return 1;
}

int amtgpu_logoframe_get_columns(const AmtGpuLogoFrame* lf, int* out2)
{
    if (!lf || !out2) return 0;
    int lo = 0x7FFFFFFF, hi = 0;
    for (const auto& l : lf->logos)
        if (l) { lo = std::min(lo, l->imgx); hi = std::max(hi, l->imgx + l->w); }
    if (lo >= hi) { lo = 0; hi = 0; }
    out2[0] = lo; out2[1] = hi;
    return 1;
}

int amtgpu_logoframe_get_results(AmtGpuLogoFrame* lf, float* out)
{
    return guard(lf->ctx, [&] {
        logoframe_sync_results(lf);
        std::memcpy(out, lf->results.data(), lf->results.size() * sizeof(float));
    });
}

int amtgpu_logoframe_set_results(AmtGpuLogoFrame* lf, int first, int nframes, const float* evals)
{
    return guard(lf->ctx, [&] {
        if (first < 0 || nframes < 0 || first + nframes > lf->numFrames) throw std::runtime_error("frame range outside the clip");
        logoframe_sync_results(lf);
        const size_t nl = lf->logos.size();
        std::memcpy(lf->results.data() + (size_t)first * nl * 2, evals, (size_t)nframes * nl * 2 * sizeof(float));
        lf->ctx->bind();
        if (nframes)
            AMT_HIP(hipMemcpyAsync(lf->dResults.get() + (size_t)first * nl * 2, evals, (size_t)nframes * nl * 2 * sizeof(float),
                                   hipMemcpyHostToDevice, lf->ctx->stream));
        AMT_HIP(hipStreamSynchronize(lf->ctx->stream));
        lf->selected = false;
    });
}

int amtgpu_logoframe_allgather_results(AmtGpuLogoFrame* lf, const AmtGpuCollectives* coll, int first, int nlocal)
{
    return guard(lf->ctx, [&] {
        if (!coll || coll->world <= 1) {
            if (first < 0 || nlocal < 0 || first + nlocal > lf->numFrames) throw std::runtime_error("frame range outside the clip");
            return;
        }
        if (!coll->allgather) throw std::runtime_error("AmtGpuCollectives incomplete");
        // A rank whose own work failed must still enter every collective (the others would block in it for ever): what is wrong
        // here travels as a status word next to the rank's range, and every rank throws after the exchange.
        std::string local_error;
        try {
            if (first < 0 || nlocal < 0 || first + nlocal > lf->numFrames) throw std::runtime_error("frame range outside the clip");
            logoframe_sync_results(lf);
        } catch (const std::exception& e) { local_error = e.what(); }
        const size_t rec = lf->logos.size() * 2;                               // floats per frame
        // ragged shards: gather {first, nlocal, ok}, then records padded to the largest shard
        const int64_t mine[3] = {local_error.empty() ? first : 0, local_error.empty() ? nlocal : 0, local_error.empty() ? 1 : 0};
        std::vector<int64_t> ranges((size_t)coll->world * 3);
        if (!coll->allgather(coll->user, mine, ranges.data(), sizeof mine)) throw std::runtime_error("allgather failed");
        int64_t nmax = 0;
        bool all_ok = true, ranges_ok = true;
        for (int r = 0; r < coll->world; ++r) {
            const int64_t f = ranges[3 * r], n = ranges[3 * r + 1];
            all_ok = all_ok && ranges[3 * r + 2] == 1;
            ranges_ok = ranges_ok && f >= 0 && n >= 0 && f + n <= lf->numFrames;
            nmax = std::max(nmax, n);
        }
        if (!local_error.empty()) throw std::runtime_error(local_error);
        if (!all_ok) throw std::runtime_error("another rank failed before the exchange of the scan records");
        if (!ranges_ok) throw std::runtime_error("a rank reported a frame range outside the clip");
        if (nmax == 0 || rec == 0) return;
        std::vector<float> send((size_t)nmax * rec, 0.0f), recv((size_t)nmax * rec * coll->world);
        std::memcpy(send.data(), lf->results.data() + (size_t)first * rec, (size_t)nlocal * rec * sizeof(float));
        if (!coll->allgather(coll->user, send.data(), recv.data(), (int64_t)(send.size() * sizeof(float)))) throw std::runtime_error("allgather failed");
        for (int r = 0; r < coll->world; ++r) {
            const int64_t f = ranges[3 * r], n = ranges[3 * r + 1];
            std::memcpy(lf->results.data() + (size_t)f * rec, recv.data() + (size_t)r * nmax * rec, (size_t)n * rec * sizeof(float));
        }
        lf->ctx->bind();
        AMT_HIP(hipMemcpyAsync(lf->dResults.get(), lf->results.data(), lf->results.size() * sizeof(float), hipMemcpyHostToDevice, lf->ctx->stream));
        AMT_HIP(hipStreamSynchronize(lf->ctx->stream));
        lf->selected = false;
    });
}

int amtgpu_logoframe_decide_host(const float* evals, int num_frames, int num_logos, int num_candidates, int logo_index,
                                 int fps_num, int fps_den, int* best_logo, float* logo_ratio, char* text, int cap, int* text_len)
{
    try {
        if (!evals || num_frames < 0 || num_logos <= 0 || fps_num <= 0 || fps_den <= 0 || logo_index >= num_logos) return 0;
        if (num_candidates > num_logos) return 0;               // the records hold num_logos pairs per frame, no more
        const LogoSelection sel = select_logo(evals, num_frames, num_logos, num_candidates);
        if (best_logo) *best_logo = sel.bestLogo;
        if (logo_ratio) *logo_ratio = sel.logoRatio;
        const int li = logo_index < 0 ? sel.bestLogo : logo_index;
        const std::string t = li < 0 ? std::string() : logoframe_text(evals, num_frames, num_logos, li, fps_num, fps_den);
        if (text_len) *text_len = (int)t.size();
        if (!text && cap == 0) return 1;                        // a length query
        if (!text || cap < (int)t.size()) return 0;             // buffer too small: a failure like any other (the library's int 1/0
                                                                // convention, `if (!call) fail;` keeps working) -- *text_len > cap tells it apart
        std::memcpy(text, t.data(), t.size());
        return 1;
    } catch (...) { return 0; }
}

int amtgpu_logoframe_select_logo(AmtGpuLogoFrame* lf, int ncand)
{
    return guard(lf->ctx, [&] {
        logoframe_sync_results(lf);
        if (ncand > (int)lf->logos.size()) throw std::runtime_error("num_candidates exceeds the number of logos");
        lf->sel = select_logo(lf->results.data(), lf->numFrames, (int)lf->logos.size(), ncand);
        lf->selected = true;
    });
}

int amtgpu_logoframe_write_result(AmtGpuLogoFrame* lf, const char* outpath, int logo_index)
{
    return guard(lf->ctx, [&] {
        logoframe_sync_results(lf);
        if (logo_index < 0) {
            if (!lf->selected) { lf->sel = select_logo(lf->results.data(), lf->numFrames, (int)lf->logos.size(), -1); lf->selected = true; }
            logo_index = lf->sel.bestLogo;
        }
        if (logo_index < 0 || logo_index >= (int)lf->logos.size()) throw std::runtime_error("logo index out of range");
        const std::string text = logoframe_text(lf->results.data(), lf->numFrames, (int)lf->logos.size(), logo_index, lf->fpsNum, lf->fpsDen);
        std::ofstream f(outpath, std::ios::binary);
        if (!f) throw std::runtime_error(std::string("failed to open file ") + outpath);
        f.write(text.data(), (std::streamsize)text.size());
    });
}

int amtgpu_logoframe_dump_result(AmtGpuLogoFrame* lf, const char* basepath)
{
    return guard(lf->ctx, [&] {
        if (!basepath) throw std::runtime_error("no base path");
        logoframe_sync_results(lf);
        const int nl = (int)lf->logos.size();
        std::string sb;
        char line[128];                                          // two "%f" of -FLT_MAX take 47 characters each: 96 in all with ",\n"
        for (int i = 0; i < nl; ++i) {
            sb.clear();
            for (int n = 0; n < lf->numFrames; ++n) {
                const float* r = lf->results.data() + ((size_t)n * nl + i) * 2;
                const int len = std::snprintf(line, sizeof line, "%f,%f\n", r[0], r[1]);
                sb.append(line, (size_t)std::min<int>(std::max(len, 0), (int)sizeof line - 1));
            }
            const std::string path = std::string(basepath) + std::to_string(i);
            std::ofstream f(path, std::ios::binary);
            if (!f) throw std::runtime_error("failed to open file " + path);
            f.write(sb.data(), (std::streamsize)sb.size());
        }
    });
}

int amtgpu_logoframe_best_logo(const AmtGpuLogoFrame* lf) { return lf->sel.bestLogo; }
float amtgpu_logoframe_logo_ratio(const AmtGpuLogoFrame* lf) { return lf->sel.logoRatio; }

// ---------------------------------------------------------------------------------------------
// AMTAnalyzeLogo
// ---------------------------------------------------------------------------------------------
struct AmtGpuAnalyze {
    AmtGpuContext* ctx;
    LogoPlanes logo;
    std::unique_ptr<EvalEngine> engine;
    DevBuf<float> dTmp;
    int mode = AMTGPU_ANALYZE_EXACT;
    DevBuf<int> dList, dCount;          // decision guard of the linear mode: frames to re-evaluate exactly
    DevBuf<uint8_t> dForce;             // ... and the frames that carry samples above maxv (9..15-bit clips in 16-bit containers)
};

// one batch in the selected mode.  Linear mode: all fades from one window evaluation of s and of bg, then the decision guard --
// frames whose argmin over the fades of p, t or b is not safe against the evaluation's error bound are listed on the device and
// re-evaluated by the exact kernel in the same stream (no host round trip)
static void analyze_run(AmtGpuAnalyze* an, const void* dY, int64_t frame_stride, int pitch, int bits, int nframes, float* dout)
{
    if (an->mode == AMTGPU_ANALYZE_EXACT) {
        an->engine->run(dY, frame_stride, pitch, bits, nframes, dout);
        return;
    }
    if (nframes <= 0) return;
    an->ctx->bind();
    if (an->dList.size() < (size_t)nframes) an->dList.alloc(nframes);
    if (an->dCount.size() < 1) an->dCount.alloc(1);
    // frames the linear evaluation must not be trusted on, one byte each: container values above maxv (9..15-bit clips: outside the error
    // bound's assumption), and -- set by the kernel itself -- the frames of a workgroup whose list of bin checks overflowed
    const bool guarded = an->mode != AMTGPU_ANALYZE_LINEAR_UNGUARDED;
    uint8_t* force = nullptr;
    if (guarded) {
        if (an->dForce.size() < (size_t)nframes) an->dForce.alloc(nframes);
        force = an->dForce.get();
        if (bits > 8 && bits < 16) {
            const int spf = an->ctx->prof_begin("rect_range_flag_kernel");
            AMT_HIP(launch_rect_range_flag(an->ctx->stream, dY, frame_stride / 2, pitch, an->logo.imgx, an->logo.imgy, an->logo.w, an->logo.h, bits, nframes, force));
            an->ctx->prof_end(spf);
        } else {
            AMT_HIP(hipMemsetAsync(force, 0, (size_t)nframes, an->ctx->stream));
        }
    }
    an->engine->run_linear(dY, frame_stride, pitch, bits, nframes, dout, nullptr, force);
    if (!guarded) return;
    float eps[3];
    for (int k = 0; k < 3; ++k) eps[k] = 2.0f * an->engine->linear_error_bound(k, bits);
    const int sp = an->ctx->prof_begin("analysis_mark_kernel");
    AMT_HIP(launch_analysis_mark(an->ctx->stream, dout, AMTGPU_ANALYZE_FLOATS, nframes, 3, AMTGPU_NUM_FADE, eps, an->dList.get(), an->dCount.get(), force));
    an->ctx->prof_end(sp);
    an->engine->run_listed(dY, frame_stride, pitch, bits, nframes, an->dList.get(), an->dCount.get(), dout);
}

static AmtGpuAnalyze* analyze_new(AmtGpuContext* c, LogoPlanes logo, float maskratio)
{
    std::unique_ptr<AmtGpuAnalyze> an(new AmtGpuAnalyze);
    an->ctx = c;
    an->logo = std::move(logo);
    const LogoPlanes& P = an->logo;
    if (P.h < 12 || P.w < 6) throw std::runtime_error("logo too small");
    std::vector<EvalLogoSpec> specs(3);
    for (int kind = 0; kind < 3; ++kind) {
        EvalLogoSpec& S = specs[kind];
        S.planes = kind == 0 ? deinterlaced_logo(P) : field_logo(P, kind == 2);
        S.tables = build_mask_tables(S.planes, maskratio);
        S.imgx = P.imgx; S.imgy = P.imgy;
        S.deint = kind == 0;
        S.row0 = kind == 2 ? 1 : 0;
        S.row_step = kind == 0 ? 1 : 2;
        S.out_off = kind * AMTGPU_NUM_FADE;
    }
    std::vector<float> fades(AMTGPU_NUM_FADE);
    for (int f = 0; f < AMTGPU_NUM_FADE; ++f) fades[f] = (float)f / 10.0f;
    an->engine.reset(new EvalEngine(c, std::move(specs), fades, true, AMTGPU_ANALYZE_FLOATS, "logo_eval_fused_kernel.analysis"));
    return an.release();
}

AmtGpuAnalyze* amtgpu_analyze_create(AmtGpuContext* c, const char* logopath, float maskratio)
{
    AmtGpuAnalyze* an = nullptr;
    guard(c, [&] {
        LogoPlanes P;
        try { P = load_lgd(logopath); }
        catch (const std::exception&) { throw std::runtime_error(std::string("Failed to read logo file (") + logopath + ")"); }
        an = analyze_new(c, std::move(P), maskratio);
    });
    return an;
}

AmtGpuAnalyze* amtgpu_analyze_create_from_logo(AmtGpuContext* c, const AmtGpuLogo* logo, float maskratio)
{
    AmtGpuAnalyze* an = nullptr;
    guard(c, [&] { an = analyze_new(c, logo->planes, maskratio); });
    return an;
}

void amtgpu_analyze_destroy(AmtGpuAnalyze* an) { delete an; }

int amtgpu_analyze_batch(AmtGpuAnalyze* an, const void* dY, int64_t frame_stride, int pitch, int bits, int nframes, float* dout)
{
    return guard(an->ctx, [&] {
        if (bits < 8 || bits > 16) throw std::runtime_error("[AMTAnalyzeLogo] Unsupported pixel format");
        analyze_run(an, dY, frame_stride, pitch, bits, nframes, dout);
    });
}

int amtgpu_analyze_set_mode(AmtGpuAnalyze* an, int mode)
{
    return guard(an->ctx, [&] {
        if (mode != AMTGPU_ANALYZE_EXACT && mode != AMTGPU_ANALYZE_LINEAR_GUARDED && mode != AMTGPU_ANALYZE_LINEAR_UNGUARDED)
            throw std::runtime_error("unknown analysis mode");
        if (mode != AMTGPU_ANALYZE_EXACT) (void)an->engine->linear_error_bound(0, 8);   // builds the tables; throws for logos the kernel does not take (wider than 256)
        an->mode = mode;
    });
}

int amtgpu_analyze_set_fixup_queue(AmtGpuAnalyze* an, int entries)
{
    return guard(an->ctx, [&] {
        if (entries < 16 || entries > 640) throw std::runtime_error("fix-up queue: 16 .. 640 entries per wave");
        an->engine->set_linear_queue(entries);
    });
}

int amtgpu_analyze_last_refined(AmtGpuAnalyze* an)
{
    int n = -1;
    guard(an->ctx, [&] {
        if (an->mode != AMTGPU_ANALYZE_LINEAR_GUARDED || an->dCount.size() < 1) { n = 0; return; }
        an->ctx->bind();
        download_via_pinned(an->ctx, &n, an->dCount.get(), sizeof(int));
    });
    return n;
}

int amtgpu_analyze_get_rect(const AmtGpuAnalyze* an, int* out4)
{
    if (!an || !out4) return 0;
    out4[0] = an->logo.imgx; out4[1] = an->logo.imgy; out4[2] = an->logo.w; out4[3] = an->logo.h;
    return 1;
}

float amtgpu_analyze_error_bound(AmtGpuAnalyze* an, int group, int bits)
{
    float e = -1.0f;
    guard(an->ctx, [&] {
        if (group < 0 || group > 2 || bits < 8 || bits > 16) throw std::runtime_error("bad group / bits");
        e = an->mode == AMTGPU_ANALYZE_EXACT ? 0.0f : an->engine->linear_error_bound(group, bits);
    });
    return e;
}

int amtgpu_analyze_batch_host(AmtGpuAnalyze* an, const void* dY, int64_t frame_stride, int pitch, int bits, int nframes, float* hout)
{
    return guard(an->ctx, [&] {
        if (bits < 8 || bits > 16) throw std::runtime_error("[AMTAnalyzeLogo] Unsupported pixel format");
        const size_t n = (size_t)nframes * AMTGPU_ANALYZE_FLOATS;
        if (an->dTmp.size() < n) an->dTmp.alloc(n);
#ifdef AMT_TRACE_CALLS
        static const bool no_kernel = std::getenv("AMT_TRACE_NO_ANALYSIS_KERNEL") != nullptr;
        if (!no_kernel)
#endif
        { AMT_TRACE_SCOPE("analyze_batch_host.launches"); analyze_run(an, dY, frame_stride, pitch, bits, nframes, an->dTmp.get()); }
        an->ctx->bind();
#ifdef AMT_TRACE_CALLS
        AmtGpuContext* c = an->ctx;
        if (!c->tr_kernels) { AMT_HIP(hipEventCreate(&c->tr_kernels)); AMT_HIP(hipEventCreate(&c->tr_landed)); }
        AMT_HIP(hipEventRecord(c->tr_kernels, c->stream));
        const double h_launched = AmtTrace::now();
        if (c->tr_block_open) amt::trace_stamp(c, c->stream, 3);
#endif
        {
            AMT_TRACE_SCOPE("analyze_batch_host.download_via_pinned");
            download_via_pinned(an->ctx, hout, an->dTmp.get(), n * sizeof(float));
        }
#ifdef AMT_TRACE_CALLS
        AMT_HIP(hipEventRecord(c->tr_landed, c->stream));
        AMT_HIP(hipEventSynchronize(c->tr_landed));
        if (c->tr_block_open && c->tr_first_copy && c->tr_copies_done && c->tr_released) {
            float a = 0, b = 0, k = 0;
            (void)hipEventElapsedTime(&a, c->tr_first_copy, c->tr_copies_done);
            (void)hipEventElapsedTime(&b, c->tr_copies_done, c->tr_released);
            (void)hipEventElapsedTime(&k, c->tr_released, c->tr_kernels);
            AmtTrace& t = AmtTrace::get();
            std::lock_guard<std::mutex> lk(t.m);
            const double now = AmtTrace::now();
            t.recs.push_back({"gpu.first_copy_to_copies_done", now, a * 1e3});
            t.recs.push_back({"gpu.copies_done_to_wait_released", now, b * 1e3});
            t.recs.push_back({"gpu.wait_released_to_kernels_done", now, k * 1e3});
            // the block on ONE clock (us after the host began the block): device stamps mapped onto the host's clock
            auto g = [&](int i) { return (double)c->tr_stamps[i] / 100.0 + c->tr_offset_us - c->tr_host_block_begin; };
            t.recs.push_back({"blk.1_gpu_copy_stream_reached_block", now, g(0)});
            t.recs.push_back({"blk.2_gpu_copies_done", now, g(1)});
            t.recs.push_back({"blk.3_gpu_compute_wait_released", now, g(2)});
            t.recs.push_back({"blk.4_host_all_enqueued", now, h_launched - c->tr_host_block_begin});
            t.recs.push_back({"blk.5_gpu_kernels_done", now, g(3)});
            t.recs.push_back({"blk.6_host_woke_up", now, now - c->tr_host_block_begin});
        }
        c->tr_block_open = false;
#endif
    });
}

} // extern "C"
