// amt_gpu_erase_scan.hip -- C ABI part 2: AMTEraseLogo, LogoScan, ScanLogo.
#include "build_knobs.h"
#include "../../include/amt_gpu.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

#include "api_common.hpp"
#include "logo_fit.hpp"

namespace amt {
struct EraseGeom {
    int w, h, wUV, hUV;
    int imgx, imgy, cx, cy;
    int uvparity;
};
hipError_t launch_delogo(hipStream_t st, int bits, const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV, long long strideY,
                         long long strideUV, int pitchY, int pitchUV, const float* dplanes, EraseGeom g, int nframes, const float2* dfades,
                         int zero_identity);
hipError_t launch_ingest_rows(hipStream_t st, const void* src, long long src_stride, void* dst, long long dst_stride, unsigned long long chunk,
                              long long nchunks);
hipError_t launch_calc_fades(hipStream_t st, const float* danalysis, int analysis_first, int analysis_count, int num_frames, int first,
                             int nframes, const uint8_t* dstate, int half, float2* dout);
hipError_t launch_scan_border(hipStream_t st, int bits, const void* dY, const void* dU, const void* dV, long long strideY,
                              long long strideUV, int pitchY, int pitchUV, int imgx, int imgy, int cx, int cy, int w, int h,
                              int wUV, int hUV, int thy, int nframes, int4* dout);
hipError_t launch_scan_accumulate(hipStream_t st, int bits, const void* dY, const void* dU, const void* dV, long long strideY,
                                  long long strideUV, int pitchY, int pitchUV, int imgx, int imgy, int cx, int cy, int w, int h,
                                  int wUV, int hUV, const int4* daccepted, int naccepted, unsigned long long* dacc);
}

using namespace amt;

// ---------------------------------------------------------------------------------------------
// AMTEraseLogo
// ---------------------------------------------------------------------------------------------
struct AmtGpuErase {
    AmtGpuContext* ctx;
    LogoPlanes logo;
    std::vector<int> frameState;    // empty = no logoframe file (always analyse)
    bool haveLogof = false;
    std::string logofText;
    int mode = 0, maxFade = 16;
    bool zeroIdentity = false;      // every a*s + b*maxv of this logo is finite: Delogo with fade 0 returns the frame unchanged
    DevBuf<float> dPlanes;
    // the caller's fades go through two pinned slots (each with its own device copy and event), so that a batch returns without
    // waiting for the stream: a slot is rewritten only after the upload that read it has completed
    float2* hFades[2] = {nullptr, nullptr};
    size_t fadesCap[2] = {0, 0};
    DevBuf<float2> dFades[2];
    hipEvent_t fadesUploaded[2] = {nullptr, nullptr};
    int fadeSlot = 0;
    DevBuf<uint8_t> dState;         // frameState on the device (amtgpu_erase_calc_fades_device), for dStateFrames frames
    int dStateFrames = -1;
    ~AmtGpuErase()
    {
        for (int i = 0; i < 2; ++i) {
            if (hFades[i]) (void)hipHostFree(hFades[i]);
            if (fadesUploaded[i]) (void)hipEventDestroy(fadesUploaded[i]);
        }
    }
};

static AmtGpuErase* erase_new(AmtGpuContext* c, LogoPlanes logo, const std::string& logofText, bool haveLogof, int mode, int maxfade)
{
    std::unique_ptr<AmtGpuErase> er(new AmtGpuErase);
    er->ctx = c;
    er->logo = std::move(logo);
    er->mode = mode;
    er->maxFade = maxfade;
    er->haveLogof = haveLogof;
    er->logofText = logofText;
    if (haveLogof) (void)parse_logoframe(logofText, 0);     // report a malformed file at construction, like the filter does
    // fade 0: dst = (pixel)min(max(0*bg + 1*s + 0.5, 0), maxv) == s as long as bg = a*s + b*maxv cannot overflow to inf / NaN
    er->zeroIdentity = true;
    for (float v : er->logo.data) er->zeroIdentity = er->zeroIdentity && std::fabs(v) < 1e30f;
    c->bind();
    er->dPlanes.upload(er->logo.data, c->stream);
    return er.release();
}

extern "C" {

AmtGpuErase* amtgpu_erase_create(AmtGpuContext* c, const char* logopath, const char* logofpath, int mode, int maxfade)
{
    AmtGpuErase* er = nullptr;
    guard(c, [&] {
        LogoPlanes P;
        try { P = load_lgd(logopath); }
        catch (const std::exception&) { throw std::runtime_error(std::string("Failed to read logo file (") + logopath + ")"); }
        std::string text;
        const bool have = logofpath && logofpath[0];
        if (have) {
            std::ifstream f(logofpath, std::ios::binary);
            if (!f) throw std::runtime_error(std::string("Failed to read dat file (") + logofpath + ")");
            std::stringstream ss;
            ss << f.rdbuf();
            text = ss.str();
        }
        er = erase_new(c, std::move(P), text, have, mode, maxfade);
    });
    return er;
}

AmtGpuErase* amtgpu_erase_create_from_logo(AmtGpuContext* c, const AmtGpuLogo* logo, const char* logof_text, int mode, int maxfade)
{
    AmtGpuErase* er = nullptr;
    guard(c, [&] {
        const bool have = logof_text && logof_text[0];
        er = erase_new(c, logo->planes, have ? logof_text : "", have, mode, maxfade);
    });
    return er;
}

void amtgpu_erase_destroy(AmtGpuErase* er) { delete er; }

int amtgpu_erase_calc_fades(AmtGpuErase* er, const float* analysis, int num_frames, int first, int nframes, float* fades_out)
{
    return guard(er->ctx, [&] {
        if (first < 0 || nframes < 0 || first + nframes > num_frames) throw std::runtime_error("frame range outside the clip");
        if (er->haveLogof && (int)er->frameState.size() != num_frames) er->frameState = parse_logoframe(er->logofText, num_frames);
        static const std::vector<int> none;
        for (int i = 0; i < nframes; ++i) {
            const FadePair fp = fade_for_frame(er->haveLogof ? er->frameState : none, er->maxFade, analysis, num_frames, first + i);
            fades_out[2 * i] = fp.top;
            fades_out[2 * i + 1] = fp.bottom;
        }
    });
}

// sY / sU / sV: the planes Delogo READS (null = dY / dU / dV: in place)
static void erase_launch(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV, int pitchY, int pitchUV,
                         int bits, int nframes, const float* fades, bool rect_only, const float* d_fades = nullptr,
                         const void* sY = nullptr, const void* sU = nullptr, const void* sV = nullptr)
{
    if (!sY) { sY = dY; sU = dU; sV = dV; }
    if (bits < 8 || bits > 16) throw std::runtime_error("[AMTEraseLogo] Unsupported pixel format");
    if (er->mode != 0) throw std::runtime_error("[AMTEraseLogo] only mode 0 is supported (debug overlay modes are out of scope)");
    if (nframes <= 0) return;
    const LogoPlanes& P = er->logo;
    const int es = bits <= 8 ? 1 : 2;
    if (strideY % es || strideUV % es) throw std::runtime_error("frame stride not a multiple of the sample size");
    if (rect_only && (pitchY < P.w || pitchUV < P.wUV())) throw std::runtime_error("[AMTEraseLogo] rectangle pitch smaller than the logo width");
    er->ctx->bind();
    const float2* dfades = reinterpret_cast<const float2*>(d_fades);
    if (!dfades) {
        if (!fades) throw std::runtime_error("[AMTEraseLogo] null fades");
        const int slot = er->fadeSlot;
        er->fadeSlot ^= 1;
        if (!er->fadesUploaded[slot]) AMT_HIP(hipEventCreateWithFlags(&er->fadesUploaded[slot], hipEventDisableTiming));
        else AMT_HIP(hipEventSynchronize(er->fadesUploaded[slot]));          // the upload two batches ago
        if (er->fadesCap[slot] < (size_t)nframes) {
            if (er->hFades[slot]) AMT_HIP(hipHostFree(er->hFades[slot]));
            er->hFades[slot] = nullptr;
            AMT_HIP(hipHostMalloc((void**)&er->hFades[slot], (size_t)nframes * sizeof(float2), hipHostMallocDefault));
            er->fadesCap[slot] = (size_t)nframes;
            er->dFades[slot].alloc(nframes);                                   // hipFree waits for the kernels that read the old copy
        }
        std::memcpy(er->hFades[slot], fades, (size_t)nframes * sizeof(float2));
        AMT_HIP(hipMemcpyAsync(er->dFades[slot].get(), er->hFades[slot], (size_t)nframes * sizeof(float2), hipMemcpyHostToDevice, er->ctx->stream));
        AMT_HIP(hipEventRecord(er->fadesUploaded[slot], er->ctx->stream));
        dfades = er->dFades[slot].get();
    }
    EraseGeom g;
    g.w = P.w; g.h = P.h; g.wUV = P.wUV(); g.hUV = P.hUV();
    // rectangle-only planes start at the logo's top-left sample; the chroma row parity is a property of the logo's position in
    // the frame (LogoScan.hpp:1374-1397), not of the buffer
    g.imgx = rect_only ? 0 : P.imgx; g.imgy = rect_only ? 0 : P.imgy;
    g.cx = rect_only ? 0 : P.imgx >> P.logUVx; g.cy = rect_only ? 0 : P.imgy >> P.logUVy;
    g.uvparity = (P.imgy / 2) % 2;
    const int sp = er->ctx->prof_begin("delogo_kernel");
    // fade 0 returns a sample unchanged only while the sample is <= maxv: min(tmp + 0.5, maxv) (LogoScan.hpp:1258) clamps a 10- or
    // 12-bit clip's out-of-range container values.  At 8 and 16 bits every container value is in range: only there are fade-0
    // frames skipped.
    const bool skip_fade0 = er->zeroIdentity && (bits == 8 || bits == 16);
    AMT_HIP(launch_delogo(er->ctx->stream, bits, sY, sU, sV, dY, dU, dV, strideY / es, strideUV / es, pitchY, pitchUV, er->dPlanes.get(), g,
                          nframes, dfades, skip_fade0 ? 1 : 0));
    er->ctx->prof_end(sp);
}

int amtgpu_erase_batch(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV, int pitchY,
                       int pitchUV, int bits, int nframes, const float* fades)
{
    return guard(er->ctx, [&] { erase_launch(er, dY, dU, dV, strideY, strideUV, pitchY, pitchUV, bits, nframes, fades, false); });
}

int amtgpu_erase_rect_batch(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV, int pitchY,
                            int pitchUV, int bits, int nframes, const float* fades)
{
    return guard(er->ctx, [&] { erase_launch(er, dY, dU, dV, strideY, strideUV, pitchY, pitchUV, bits, nframes, fades, true); });
}

int amtgpu_erase_batch_dfades(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV, int pitchY,
                              int pitchUV, int bits, int nframes, const float* d_fades)
{
    return guard(er->ctx, [&] {
        if (!d_fades && nframes > 0) throw std::runtime_error("[AMTEraseLogo] null device fades");
        erase_launch(er, dY, dU, dV, strideY, strideUV, pitchY, pitchUV, bits, nframes, nullptr, false, d_fades);
    });
}

int amtgpu_erase_rect_batch_dfades(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV, int pitchY,
                                   int pitchUV, int bits, int nframes, const float* d_fades)
{
    return guard(er->ctx, [&] {
        if (!d_fades && nframes > 0) throw std::runtime_error("[AMTEraseLogo] null device fades");
        erase_launch(er, dY, dU, dV, strideY, strideUV, pitchY, pitchUV, bits, nframes, nullptr, true, d_fades);
    });
}

int amtgpu_erase_batch_dfades_to(AmtGpuErase* er, const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV, int64_t strideY,
                                 int64_t strideUV, int pitchY, int pitchUV, int bits, int nframes, const float* d_fades)
{
    return guard(er->ctx, [&] {
        if (!d_fades && nframes > 0) throw std::runtime_error("[AMTEraseLogo] null device fades");
        if (nframes > 0 && (!sY || !sU || !sV || !dY || !dU || !dV)) throw std::runtime_error("[AMTEraseLogo] null plane");
        erase_launch(er, dY, dU, dV, strideY, strideUV, pitchY, pitchUV, bits, nframes, nullptr, false, d_fades, sY, sU, sV);
    });
}

// CalcFade / CalcFade2 on the device (erase_scan_kernels.hip calc_fades_kernel); the host routine above stays the checker's
// counterpart and what the per-frame filter layer uses
int amtgpu_erase_calc_fades_device(AmtGpuErase* er, const float* d_analysis, int analysis_first, int analysis_count, int num_frames,
                                   int first, int nframes, float* d_fades_out)
{
    return guard(er->ctx, [&] {
        if (first < 0 || nframes < 0 || first + nframes > num_frames) throw std::runtime_error("frame range outside the clip");
        if (nframes == 0) return;
        if (!d_analysis || !d_fades_out) throw std::runtime_error("[AMTEraseLogo] null device pointer");
        // every record the decision of [first, first + nframes) can read: n - 8 .. n + 8, clamped the reference's way (which keeps
        // reads at the clip's ends inside the first / last eight frames)
        const int need0 = std::max(0, first - 8), need1 = std::min(num_frames, first + nframes + 8);
        if (analysis_first < 0 || analysis_count <= 0 || analysis_first > need0 || analysis_first + analysis_count < need1)
            throw std::runtime_error("[AMTEraseLogo] the analysis records do not cover frames first-8 .. first+nframes+8 (CalcFade2's window)");
        er->ctx->bind();
        const uint8_t* dstate = nullptr;
        if (er->haveLogof) {
            if ((int)er->frameState.size() != num_frames) er->frameState = parse_logoframe(er->logofText, num_frames);
            if (er->dStateFrames != num_frames) {
                std::vector<uint8_t> st(num_frames);
                for (int i = 0; i < num_frames; ++i) st[i] = (uint8_t)er->frameState[i];
                er->dState.upload(st, er->ctx->stream);          // (synchronous: once per clip)
                er->dStateFrames = num_frames;
            }
            dstate = er->dState.get();
        }
        const int sp = er->ctx->prof_begin("calc_fades_kernel");
        AMT_HIP(launch_calc_fades(er->ctx->stream, d_analysis, analysis_first, analysis_count, num_frames, first, nframes, dstate,
                                  er->maxFade >> 1, reinterpret_cast<float2*>(d_fades_out)));
        er->ctx->prof_end(sp);
    });
}

int amtgpu_erase_get_rect(const AmtGpuErase* er, int* out5)
{
    if (!er || !out5) return 0;
    const LogoPlanes& P = er->logo;
    out5[0] = P.imgx; out5[1] = P.imgy; out5[2] = P.w; out5[3] = P.h; out5[4] = er->zeroIdentity ? 1 : 0;
    return 1;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// LogoScan
// ---------------------------------------------------------------------------------------------
struct AmtGpuLogoScan {
    AmtGpuContext* ctx;
    ScanSums sums;
    int thy = 0;
    DevBuf<unsigned long long> dAcc;
    DevBuf<int4> dVerdict, dAccepted;
    std::vector<int4> lastVerdicts;      // verdicts of the most recent add_batch (frame-local)
    std::vector<int4> accHost;           // accepted list of the most recent add_batch: source of an async upload, so it
    hipEvent_t accUploaded = nullptr;    //   lives here and is rewritten only after this event
    ~AmtGpuLogoScan() { if (accUploaded) (void)hipEventDestroy(accUploaded); }
    bool accDirty = false;               // device accumulators newer than sums.px
};

static void logoscan_pull(AmtGpuLogoScan* s)
{
    if (!s->accDirty) return;
    s->ctx->bind();
    download_via_pinned(s->ctx, s->sums.px.data(), s->dAcc.get(), s->sums.px.size() * sizeof(int64_t));
    s->accDirty = false;
}

// border verdicts for a batch, then accumulate the accepted subset; returns accepted count
static int logoscan_add(AmtGpuLogoScan* s, const void* dY, const void* dU, const void* dV, int64_t strideY, int64_t strideUV,
                        int pitchY, int pitchUV, int bits, int imgx, int imgy, int nframes, int max_valid, const uint8_t* use_mask,
                        uint8_t* valid_out, const int* frame_ids /* optional: batch slot -> frame index in dY */,
                        const int4* known_verdicts /* optional: skip the border kernel */)
{
    if (bits < 8 || bits > 12) throw std::runtime_error("[LogoScan] 8..12 bit only");
    const int es = bits <= 8 ? 1 : 2;
    if (strideY % es || strideUV % es) throw std::runtime_error("frame stride not a multiple of the sample size");
    ScanSums& S = s->sums;
    const int wUV = S.w >> S.logUVx, hUV = S.h >> S.logUVy;
    const int cx = imgx >> S.logUVx, cy = imgy >> S.logUVy;
    s->ctx->bind();
    std::vector<int4>& v = s->lastVerdicts;
    if (known_verdicts) {
        v.assign(known_verdicts, known_verdicts + nframes);
    } else {
        if (s->dVerdict.size() < (size_t)nframes) s->dVerdict.alloc(nframes);
        const int spb = s->ctx->prof_begin("scan_border_kernel");
        AMT_HIP(launch_scan_border(s->ctx->stream, bits, dY, dU, dV, strideY / es, strideUV / es, pitchY, pitchUV, imgx, imgy, cx, cy,
                                   S.w, S.h, wUV, hUV, s->thy, nframes, s->dVerdict.get()));
        s->ctx->prof_end(spb);
        v.resize(nframes);
        download_via_pinned(s->ctx, v.data(), s->dVerdict.get(), (size_t)nframes * sizeof(int4));
    }
    std::vector<int4>& acc = s->accHost;
    if (s->accUploaded) AMT_HIP(hipEventSynchronize(s->accUploaded));
    else AMT_HIP(hipEventCreateWithFlags(&s->accUploaded, hipEventDisableTiming));
    acc.clear();
    for (int i = 0; i < nframes; ++i) {
        if (valid_out) valid_out[i] = 0;
        if ((int)acc.size() >= max_valid) continue;          // stream order: later frames are not even looked at
        if (use_mask && !use_mask[i]) continue;
        if (!v[i].x) continue;
        if (valid_out) valid_out[i] = 1;
        acc.push_back(make_int4(frame_ids ? frame_ids[i] : i, v[i].y, v[i].z, v[i].w));
        S.plane[0] += v[i].y; S.plane[1] += (int64_t)v[i].y * v[i].y;
        S.plane[2] += v[i].z; S.plane[3] += (int64_t)v[i].z * v[i].z;
        S.plane[4] += v[i].w; S.plane[5] += (int64_t)v[i].w * v[i].w;
    }
    if (!acc.empty()) {
        if (s->dAccepted.size() < acc.size()) s->dAccepted.alloc(acc.size());
        AMT_HIP(hipMemcpyAsync(s->dAccepted.get(), acc.data(), acc.size() * sizeof(int4), hipMemcpyHostToDevice, s->ctx->stream));
        AMT_HIP(hipEventRecord(s->accUploaded, s->ctx->stream));
        const int spa = s->ctx->prof_begin("scan_accumulate_kernel");
        AMT_HIP(launch_scan_accumulate(s->ctx->stream, bits, dY, dU, dV, strideY / es, strideUV / es, pitchY, pitchUV, imgx, imgy, cx, cy,
                                       S.w, S.h, wUV, hUV, s->dAccepted.get(), (int)acc.size(), s->dAcc.get()));
        s->ctx->prof_end(spa);
        s->accDirty = true;
        S.nframes += (int)acc.size();
    }
    return (int)acc.size();
}

static AmtGpuLogoScan* logoscan_new(AmtGpuContext* c, int w, int h, int logUVx, int logUVy, int thy)
{
    if (w <= 0 || h <= 0 || (w & 1) || (h & 1) || logUVx != 1 || logUVy != 1)
        throw std::runtime_error("[LogoScan] rectangle must be even-sized 4:2:0");
    std::unique_ptr<AmtGpuLogoScan> s(new AmtGpuLogoScan);
    s->ctx = c;
    s->thy = thy;
    s->sums.w = w; s->sums.h = h; s->sums.logUVx = logUVx; s->sums.logUVy = logUVy;
    s->sums.px.assign(s->sums.npixels() * 3, 0);
    c->bind();
    s->dAcc.alloc(s->sums.px.size());
    AMT_HIP(hipMemsetAsync(s->dAcc.get(), 0, s->sums.px.size() * sizeof(int64_t), c->stream));
    AMT_HIP(hipStreamSynchronize(c->stream));
    return s.release();
}

extern "C" {

AmtGpuLogoScan* amtgpu_logoscan_create(AmtGpuContext* c, int w, int h, int logUVx, int logUVy, int thy)
{
    AmtGpuLogoScan* s = nullptr;
    guard(c, [&] { s = logoscan_new(c, w, h, logUVx, logUVy, thy); });
    return s;
}
void amtgpu_logoscan_destroy(AmtGpuLogoScan* s) { delete s; }

int amtgpu_logoscan_add_batch(AmtGpuLogoScan* s, const void* dY, const void* dU, const void* dV, int64_t strideY, int64_t strideUV,
                              int pitchY, int pitchUV, int bits, int imgx, int imgy, int nframes, int max_valid,
                              const uint8_t* use_mask, uint8_t* valid_out, int* naccepted)
{
    return guard(s->ctx, [&] {
        const int n = logoscan_add(s, dY, dU, dV, strideY, strideUV, pitchY, pitchUV, bits, imgx, imgy, nframes, max_valid, use_mask,
                                   valid_out, nullptr, nullptr);
        if (naccepted) *naccepted = n;
    });
}

int amtgpu_logoscan_nframes(const AmtGpuLogoScan* s) { return s->sums.nframes; }

int amtgpu_logoscan_get_sums(AmtGpuLogoScan* s, int64_t* sums, int64_t* plane_sums)
{
    return guard(s->ctx, [&] {
        logoscan_pull(s);
        if (sums) std::memcpy(sums, s->sums.px.data(), s->sums.px.size() * sizeof(int64_t));
        if (plane_sums) std::memcpy(plane_sums, s->sums.plane, sizeof s->sums.plane);
    });
}

int amtgpu_logoscan_set_sums(AmtGpuLogoScan* s, const int64_t* sums, const int64_t* plane_sums, int nframes)
{
    return guard(s->ctx, [&] {
        std::memcpy(s->sums.px.data(), sums, s->sums.px.size() * sizeof(int64_t));
        std::memcpy(s->sums.plane, plane_sums, sizeof s->sums.plane);
        s->sums.nframes = nframes;
        s->ctx->bind();
        AMT_HIP(hipMemcpyAsync(s->dAcc.get(), s->sums.px.data(), s->sums.px.size() * sizeof(int64_t), hipMemcpyHostToDevice, s->ctx->stream));
        AMT_HIP(hipStreamSynchronize(s->ctx->stream));
        s->accDirty = false;
    });
}

AmtGpuLogo* amtgpu_logoscan_get_logo(AmtGpuLogoScan* s, int maxv, int clean, int imgw, int imgh, int imgx, int imgy)
{
    AmtGpuLogo* l = nullptr;
    guard(s->ctx, [&] {
        logoscan_pull(s);
        std::unique_ptr<AmtGpuLogo> n(new AmtGpuLogo);
        if (!fit_logo(s->sums, maxv, clean != 0, n->planes)) throw std::runtime_error("Insufficient logo frames");
        n->planes.imgw = imgw; n->planes.imgh = imgh; n->planes.imgx = imgx; n->planes.imgy = imgy;
        l = n.release();
    });
    return l;
}

// LogoAnalyzer::ScanLogo (LogoScan.hpp:1058-1079): initial logo from every flat-bordered frame (stop at
// numMaxFrames), then twice: evaluate 20 fades per kept frame, re-accumulate only frames whose best fade
// index is > 8, regress again with clean-up; save.
//
// coll != nullptr: this rank holds one contiguous shard of the stream.  What is global in the reference's three
// sequential rounds is (a) which valid frames fall inside the numMaxFrames quota ("first N in stream order", :885) and
// (b) the accumulators each regression reads -- both integers, so an all-gather of valid counts and an all-reduce of
// int64 sums reproduce the single-GPU result exactly; the regression then runs redundantly on every rank.
} // extern "C" (helpers)

namespace {

// A sharded run must never leave ranks behind in a collective: a rank whose own work threw (a HIP error, a bad argument, a
// cancelled callback) keeps entering every exchange with neutral data, its status rides along, and ALL ranks throw right after
// the exchange in which the status becomes known.
struct ShardGuard {
    const AmtGpuCollectives* coll = nullptr;     // nullptr / world 1: plain exceptions
    std::string error;                           // this rank's failure, if any
    int64_t cancel = 0;
    bool sharded() const { return coll && coll->world > 1; }
    template <typename F> void attempt(F&& fn)
    {
        if (!sharded()) { fn(); return; }
        if (!error.empty()) return;
        try { fn(); } catch (const std::exception& e) { error = e.what(); } catch (...) { error = "unknown error"; }
    }
    int64_t status() const { return (cancel ? 1 : 0) + (error.empty() ? 0 : (int64_t)1 << 32); }
    // `summed` = sum over ranks of status(): everyone leaves together
    void agree(int64_t summed) const
    {
        if (!error.empty()) throw std::runtime_error(error);
        if (summed >> 32) throw std::runtime_error("another rank failed; the sharded run was abandoned on every rank");
        if (summed & 0xFFFFFFFF) throw std::runtime_error("Cancel requested");
    }
};

// sums of all ranks -> every rank (px sums, plane sums, frame count, and the ranks' status riding along); npx = 3 * samples of the
// scan rectangle (known even when this rank has no scan object to contribute)
void reduce_scan(AmtGpuLogoScan* s, ShardGuard& sg, size_t npx)
{
    if (!sg.sharded()) return;
    std::vector<int64_t> buf(npx + 8, 0);
    sg.attempt([&] {
        if (!s) throw std::runtime_error("no scan to reduce");
        logoscan_pull(s);
        if (s->sums.px.size() != npx) throw std::runtime_error("scan size mismatch");
        std::copy(s->sums.px.begin(), s->sums.px.end(), buf.begin());
        for (int k = 0; k < 6; ++k) buf[npx + k] = s->sums.plane[k];
        buf[npx + 6] = s->sums.nframes;
    });
    if (!sg.error.empty()) std::fill(buf.begin(), buf.end(), 0);
    buf[npx + 7] = sg.status();
    if (!sg.coll->allreduce_sum_i64(sg.coll->user, buf.data(), (int64_t)buf.size())) throw std::runtime_error("allreduce_sum_i64 failed");
    sg.agree(buf[npx + 7]);
    if (!amtgpu_logoscan_set_sums(s, buf.data(), buf.data() + npx, (int)buf[npx + 6])) throw std::runtime_error(s->ctx->err);
}

// ReMakeLogo twice (LogoScan.hpp:923-1036, 1065-1071) over a device-resident clip: `kept` lists the frames round 0 accepted (indices into
// the clip), whose rectangle sits at (cx, cy); the finished logo's header gets (himgw, himgh, himgx, himgy).
template <typename Progress>
std::unique_ptr<AmtGpuLogo> remake_rounds(AmtGpuContext* c, ShardGuard& sg, AmtGpuLogoScan* scan0, const void* dY, const void* dU,
                                          const void* dV, int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int cx, int cy, int w, int h,
                                          int thy, const std::vector<int>& kept, const std::vector<int4>& keptVerdict, int himgw, int himgh,
                                          int himgx, int himgy, Progress&& progress)
{
    const size_t npx = (size_t)3 * ((size_t)w * h + 2 * (size_t)(w / 2) * (h / 2));
    const int bits = 8;
    const int numFrames = (int)kept.size();
    std::unique_ptr<AmtGpuLogo> logo(amtgpu_logoscan_get_logo(scan0, 255, 0, himgw, himgh, himgx, himgy));
    if (!logo) throw std::runtime_error(c->err);
    DevBuf<int> dMap;
    if (numFrames) dMap.upload(kept, c->stream);
    std::vector<float> fades(20);
    for (int fi = 0; fi < 20; ++fi) fades[fi] = 0.1f * fi;
    DevBuf<float> dEval((size_t)std::max(1, numFrames) * 20);
    std::vector<float> hEval((size_t)numFrames * 20);
    for (int round = 0; round < 2; ++round) {
        std::unique_ptr<AmtGpuLogoScan> rescan;
        // (the regression that produced `logo` ran on the same reduced sums on every rank: all ranks are here, or none)
        sg.attempt([&] {
            EvalLogoSpec S;
            S.planes = deinterlaced_logo(logo->planes);
            S.tables = build_mask_tables(S.planes, 0.1f);
            S.imgx = cx; S.imgy = cy; S.row0 = 0; S.row_step = 1; S.deint = 1; S.out_off = 0;
            std::vector<EvalLogoSpec> specs;
            specs.push_back(std::move(S));
            EvalEngine eng(c, std::move(specs), fades, true, 20, "logo_eval_fused_kernel.remake");
            std::vector<uint8_t> use(numFrames, 0);
            if (numFrames) {
                eng.run(dY, strideY, pitchY, bits, numFrames, dEval.get(), dMap.get());
                download_via_pinned(c, hEval.data(), dEval.get(), hEval.size() * sizeof(float));
            }
            for (int i = 0; i < numFrames; ++i) {
                float best = FLT_MAX;
                int bestIdx = 0;
                for (int fi = 0; fi < 20; ++fi)
                    if (hEval[(size_t)i * 20 + fi] < best) { best = hEval[(size_t)i * 20 + fi]; bestIdx = fi; }
                use[i] = bestIdx > 8;                       // logo clearly present in this frame
            }
            progress(50.0f + 25.0f * round + 12.5f, numFrames, numFrames, numFrames);
            rescan.reset(logoscan_new(c, w, h, 1, 1, thy));
            if (numFrames)
                logoscan_add(rescan.get(), dY, dU, dV, strideY, strideUV, pitchY, pitchUV, bits, cx, cy, numFrames, numFrames, use.data(), nullptr,
                             kept.data(), keptVerdict.data());
        });
        reduce_scan(rescan.get(), sg, npx);
        logo.reset(amtgpu_logoscan_get_logo(rescan.get(), 255, 1, himgw, himgh, himgx, himgy));
        if (!logo) throw std::runtime_error(c->err);
    }
    return logo;
}

int scanlogo_impl(AmtGpuContext* c, const AmtGpuCollectives* coll, const void* dY, const void* dU, const void* dV, int64_t strideY,
                  int64_t strideUV, int pitchY, int pitchUV, int imgw, int imgh, int nframes, int serviceid, const char* dstpath,
                  int imgx, int imgy, int w, int h, int thy, int numMaxFrames, AMTGPU_LOGO_ANALYZE_CB cb)
{
    return guard(c, [&] {
        const bool sharded = coll && coll->world > 1;
        if (sharded && (!coll->allgather || !coll->allreduce_sum_i64 || coll->rank < 0 || coll->rank >= coll->world))
            throw std::runtime_error("AmtGpuCollectives incomplete");
        ShardGuard sg;
        sg.coll = sharded ? coll : nullptr;
        auto progress = [&](float p, int nread, int total, int ngather) {
            if (cb && !cb(p, nread, total, ngather)) {
                if (!sharded) throw std::runtime_error("Cancel requested");
                sg.cancel = 1;                                  // the other ranks learn about it with the next exchange
            }
        };
        const int bits = 8;                                   // the reference's scan path is 8-bit only (:813)
        const size_t npx = (size_t)3 * ((size_t)std::max(0, w) * std::max(0, h) + 2 * (size_t)(std::max(0, w) / 2) * (std::max(0, h) / 2));
        std::unique_ptr<AmtGpuLogoScan> scan;
        std::vector<int> kept;              // frame index (within this rank's frames) of every kept frame
        std::vector<int4> keptVerdict;      // its {1,bgY,bgU,bgV}
        const int chunk = 4096;
        auto at = [&](const void* base, int64_t stride, int f0) { return (const void*)((const uint8_t*)base + (int64_t)f0 * stride); };
        if (!sharded) {
            if (imgx < 0 || imgy < 0 || imgx + w > imgw || imgy + h > imgh) throw std::runtime_error("scan rectangle outside the frame");
            scan.reset(logoscan_new(c, w, h, 1, 1, thy));
            // round 0: every frame in stream order until numMaxFrames are kept
            for (int f0 = 0; f0 < nframes && (int)kept.size() < numMaxFrames; f0 += chunk) {
                const int n = std::min(chunk, nframes - f0);
                std::vector<uint8_t> valid(n);
                logoscan_add(scan.get(), at(dY, strideY, f0), at(dU, strideUV, f0), at(dV, strideUV, f0), strideY, strideUV, pitchY, pitchUV,
                             bits, imgx, imgy, n, numMaxFrames - (int)kept.size(), nullptr, valid.data(), nullptr, nullptr);
                for (int i = 0; i < n; ++i)
                    if (valid[i]) { kept.push_back(f0 + i); keptVerdict.push_back(scan->lastVerdicts[i]); }
                progress(50.0f * (f0 + n) / std::max(1, nframes), f0 + n, 0, (int)kept.size());
            }
        } else {
            // round 0, sharded: border verdicts of every local frame (nothing accepted yet: max_valid = 0) ...
            sg.attempt([&] {
                if (imgx < 0 || imgy < 0 || imgx + w > imgw || imgy + h > imgh) throw std::runtime_error("scan rectangle outside the frame");
                scan.reset(logoscan_new(c, w, h, 1, 1, thy));
                for (int f0 = 0; f0 < nframes; f0 += chunk) {
                    const int n = std::min(chunk, nframes - f0);
                    logoscan_add(scan.get(), at(dY, strideY, f0), at(dU, strideUV, f0), at(dV, strideUV, f0), strideY, strideUV, pitchY, pitchUV,
                                 bits, imgx, imgy, n, 0, nullptr, nullptr, nullptr, nullptr);
                    for (int i = 0; i < n; ++i)
                        if (scan->lastVerdicts[i].x) { kept.push_back(f0 + i); keptVerdict.push_back(scan->lastVerdicts[i]); }
                    progress(50.0f * (f0 + n) / std::max(1, nframes), f0 + n, 0, (int)kept.size());
                }
            });
            // ... this rank's share of "the first numMaxFrames valid frames of the stream" (and how every rank is doing) ...
            std::vector<int64_t> counts((size_t)coll->world * 2, 0);
            const int64_t mine[2] = {sg.error.empty() ? (int64_t)kept.size() : 0, sg.status()};
            if (!coll->allgather(coll->user, mine, counts.data(), sizeof mine)) throw std::runtime_error("allgather failed");
            int64_t before = 0, summed = 0;
            for (int r = 0; r < coll->world; ++r) summed += counts[2 * r + 1];
            for (int r = 0; r < coll->rank; ++r) before += counts[2 * r];
            sg.agree(summed);
            const int quota = (int)std::max<int64_t>(0, std::min<int64_t>(mine[0], (int64_t)numMaxFrames - before));
            kept.resize(quota);
            keptVerdict.resize(quota);
            // ... accumulated locally, summed over ranks
            sg.attempt([&] {
                if (quota)
                    logoscan_add(scan.get(), dY, dU, dV, strideY, strideUV, pitchY, pitchUV, bits, imgx, imgy, quota, quota, nullptr, nullptr,
                                 kept.data(), keptVerdict.data());
            });
            reduce_scan(scan.get(), sg, npx);
        }
        const int numFrames = (int)kept.size();
        std::unique_ptr<AmtGpuLogo> logo = remake_rounds(c, sg, scan.get(), dY, dU, dV, strideY, strideUV, pitchY, pitchUV, imgx, imgy, w, h,
                                                         thy, kept, keptVerdict, imgw, imgh, imgx, imgy, progress);
        progress(1, numFrames, numFrames, numFrames);
        if (dstpath && (!sharded || coll->rank == 0)) save_lgd(logo->planes, dstpath, "No Name", serviceid);
    });
}

} // namespace

extern "C" {

int amtgpu_scanlogo(AmtGpuContext* c, const void* dY, const void* dU, const void* dV, int64_t strideY, int64_t strideUV,
                    int pitchY, int pitchUV, int imgw, int imgh, int nframes, int serviceid, const char* dstpath, int imgx,
                    int imgy, int w, int h, int thy, int numMaxFrames, AMTGPU_LOGO_ANALYZE_CB cb)
{
    return scanlogo_impl(c, nullptr, dY, dU, dV, strideY, strideUV, pitchY, pitchUV, imgw, imgh, nframes, serviceid, dstpath, imgx, imgy,
                         w, h, thy, numMaxFrames, cb);
}

// The reference's exported ScanLogo, argument for argument (LogoScan.hpp:1083-1098; C# P/Invoke AmatsukazeNatives.cs:391-393), over a
// raw 8-bit 4:2:0 clip file instead of a transport stream (decode is out of scope): int32 {'AMTR', width, height, nframes} followed by
// tight Y, U, V planes per frame.  Frames are streamed through the pinned ring in chunks; only the rectangles of accepted frames
// stay in HBM for the two ReMakeLogo rounds (the reference keeps them in `workfile` through a lossless codec, :840-912 -- here the
// argument is accepted and the file left untouched).
int amtgpu_scanlogo_file(AmtGpuContext* c, const char* srcpath, int serviceid, const char* workfile, const char* dstpath, int imgx, int imgy,
                         int w, int h, int thy, int numMaxFrames, AMTGPU_LOGO_ANALYZE_CB cb)
{
    (void)workfile;
    return guard(c, [&] {
        auto progress = [&](float p, int nread, int total, int ngather) {
            if (cb && !cb(p, nread, total, ngather)) throw std::runtime_error("Cancel requested");
        };
        std::ifstream f(srcpath, std::ios::binary);
        if (!f) throw std::runtime_error(std::string("failed to open file ") + srcpath);
        int32_t hdr[4];
        f.read(reinterpret_cast<char*>(hdr), sizeof hdr);
        if (!f || hdr[0] != 0x52544D41 || hdr[1] <= 0 || hdr[2] <= 0 || hdr[3] < 0 || (hdr[1] & 1) || (hdr[2] & 1))
            throw std::runtime_error("not a raw AMTR clip (int32 'AMTR', width, height, frames; 8-bit 4:2:0 planes)");
        const int W = hdr[1], H = hdr[2], N = hdr[3];
        if (imgx < 0 || imgy < 0 || imgx + w > W || imgy + h > H) throw std::runtime_error("scan rectangle outside the frame");
        if (numMaxFrames < 0) numMaxFrames = 0;
        const size_t ysz = (size_t)W * H, csz = (size_t)(W / 2) * (H / 2), fsz = ysz + 2 * csz;
        const int chunk = (int)std::max<size_t>(1, std::min<size_t>(1024, (256u << 20) / fsz));
        const int wUV = w / 2, hUV = h / 2;
        c->bind();
        std::unique_ptr<AmtGpuLogoScan> scan(logoscan_new(c, w, h, 1, 1, thy));
        DevBuf<uint8_t> dChunk(fsz * chunk);
        // rectangles of the accepted frames, tight: Y [n][h][w], U / V [n][h/2][w/2]
        const int cap = std::min(numMaxFrames, N);
        DevBuf<uint8_t> cropY((size_t)std::max(1, cap) * w * h), cropU((size_t)std::max(1, cap) * wUV * hUV), cropV((size_t)std::max(1, cap) * wUV * hUV);
        std::vector<uint8_t> host(fsz * chunk), planar(fsz * chunk);
        std::vector<int4> keptVerdict;
        int nkept = 0;
        for (int f0 = 0; f0 < N && nkept < numMaxFrames; f0 += chunk) {
            const int n = std::min(chunk, N - f0);
            f.read(reinterpret_cast<char*>(host.data()), (std::streamsize)(fsz * n));
            if (!f) throw std::runtime_error("raw clip truncated");
            // file order is frame-interleaved (Y,U,V per frame); the device batch is plane-major: Y[n], U[n], V[n]
            for (int i = 0; i < n; ++i) {
                std::memcpy(planar.data() + ysz * i, host.data() + fsz * i, ysz);
                std::memcpy(planar.data() + ysz * n + csz * i, host.data() + fsz * i + ysz, csz);
                std::memcpy(planar.data() + ysz * n + csz * n + csz * i, host.data() + fsz * i + ysz + csz, csz);
            }
            if (!amtgpu_frames_upload(c, dChunk.get(), planar.data(), fsz * n) || !amtgpu_frames_upload_wait(c)) throw std::runtime_error(c->err);
            const uint8_t *dY = dChunk.get(), *dU = dY + ysz * n, *dV = dU + csz * n;
            std::vector<uint8_t> valid(n);
            logoscan_add(scan.get(), dY, dU, dV, (int64_t)ysz, (int64_t)csz, W, W / 2, 8, imgx, imgy, n, numMaxFrames - nkept, nullptr, valid.data(),
                         nullptr, nullptr);
            for (int i = 0; i < n; ++i) {
                if (!valid[i]) continue;
                // (the library's own row kernel, not the runtime's 2-D copy: amt_gpu_upload.hip)
                AMT_HIP(launch_ingest_rows(c->stream, dY + ysz * i + (size_t)imgy * W + imgx, W, cropY.get() + (size_t)nkept * w * h, w, (unsigned long long)w, h));
                AMT_HIP(launch_ingest_rows(c->stream, dU + csz * i + (size_t)(imgy / 2) * (W / 2) + imgx / 2, W / 2, cropU.get() + (size_t)nkept * wUV * hUV, wUV,
                                           (unsigned long long)wUV, hUV));
                AMT_HIP(launch_ingest_rows(c->stream, dV + csz * i + (size_t)(imgy / 2) * (W / 2) + imgx / 2, W / 2, cropV.get() + (size_t)nkept * wUV * hUV, wUV,
                                           (unsigned long long)wUV, hUV));
                keptVerdict.push_back(scan->lastVerdicts[i]);
                ++nkept;
            }
            AMT_HIP(hipStreamSynchronize(c->stream));          // the chunk buffer is refilled by the next upload
            progress(50.0f * (f0 + n) / std::max(1, N), f0 + n, 0, nkept);
        }
        std::vector<int> kept(nkept);
        for (int i = 0; i < nkept; ++i) kept[i] = i;
        ShardGuard unsharded;
        std::unique_ptr<AmtGpuLogo> logo = remake_rounds(c, unsharded, scan.get(), cropY.get(), cropU.get(), cropV.get(), (int64_t)w * h,
                                                         (int64_t)wUV * hUV, w, wUV, 0, 0, w, h, thy, kept, keptVerdict, W, H, imgx, imgy, progress);
        progress(1, nkept, nkept, nkept);
        save_lgd(logo->planes, dstpath, "No Name", serviceid);
    });
}

int amtgpu_scanlogo_fileW(AmtGpuContext* c, const uint16_t* srcpath, int serviceid, const uint16_t* workfile, const uint16_t* dstpath, int imgx,
                          int imgy, int w, int h, int thy, int numMaxFrames, AMTGPU_LOGO_ANALYZE_CB cb)
{
    const std::string src = amt_utf8_from_utf16z(srcpath), work = amt_utf8_from_utf16z(workfile), dst = amt_utf8_from_utf16z(dstpath);
    return amtgpu_scanlogo_file(c, src.c_str(), serviceid, work.c_str(), dst.c_str(), imgx, imgy, w, h, thy, numMaxFrames, cb);
}

int amtgpu_scanlogo_sharded(AmtGpuContext* c, const AmtGpuCollectives* coll, const void* dY, const void* dU, const void* dV,
                            int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int imgw, int imgh, int nframes_local,
                            int serviceid, const char* dstpath, int imgx, int imgy, int w, int h, int thy, int numMaxFrames,
                            AMTGPU_LOGO_ANALYZE_CB cb)
{
    return scanlogo_impl(c, coll, dY, dU, dV, strideY, strideUV, pitchY, pitchUV, imgw, imgh, nframes_local, serviceid, dstpath, imgx,
                         imgy, w, h, thy, numMaxFrames, cb);
}

} // extern "C"
