/*
 * oracle/ref_shim/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C ABI around the REAL reference classes (Amatsukaze/LogoScan.hpp, AMTLogo.hpp, ComputeKernel.cpp,
 * compiled where they lie through the shim headers in this directory; see oracle/build_ref.sh).
 * Used by tests/test_oracle_vs_ref.py to pin oracle/amt_oracle.cpp, and by tools/make_golden.py to
 * produce tests/golden/.  Contains no pixel arithmetic of its own: it builds mock clips over caller
 * memory, instantiates the reference's LogoDataParam / LogoScan / LogoFrame / AMTAnalyzeLogo /
 * AMTEraseLogo / ScanLogo and copies their outputs out.
 */
#include "TranscodeSetting.hpp"     // shim (all std headers come in here, before the access hack)

#define private public
#define protected public
#define class struct              // default member access of the reference classes -> public
#include "LogoScan.hpp"             // the reference's, via the stage-dir symlink
#undef class
#undef private
#undef protected

namespace {

AMTContext g_ctx;
std::string g_err;

// clip over caller memory: planar 4:2:0, 8 or 16 bit; pitches in elements
class MemClip : public IClip {
    VideoInfo vi_;
    const uint8_t *Y_, *U_, *V_;
    int64_t strideY_, strideUV_;
    int pitchY_, pitchUV_;
public:
    MemClip(int w, int h, int bits, int nframes, int fps_num, int fps_den, const void* Y, const void* U,
            const void* V, int64_t strideY, int64_t strideUV, int pitchY, int pitchUV)
        : Y_((const uint8_t*)Y), U_((const uint8_t*)U), V_((const uint8_t*)V),
          strideY_(strideY), strideUV_(strideUV), pitchY_(pitchY), pitchUV_(pitchUV)
    {
        vi_.width = w; vi_.height = h; vi_.num_frames = nframes;
        vi_.fps_numerator = fps_num; vi_.fps_denominator = fps_den;
        vi_.bits_per_component = bits;
        vi_.pixel_type = bits <= 8 ? VideoInfo::CS_YV12 : VideoInfo::CS_YUV420P16;
    }
    const VideoInfo& GetVideoInfo() { return vi_; }
    PVideoFrame GetFrame(int n, IScriptEnvironment* env)
    {
        n = std::max(0, std::min(vi_.num_frames - 1, n));
        PVideoFrame f = env->NewVideoFrame(vi_);
        int cs = vi_.ComponentSize();
        for (int y = 0; y < vi_.height; ++y)
            std::memcpy(f->GetWritePtr(PLANAR_Y) + (size_t)y * f->GetPitch(PLANAR_Y),
                        Y_ + n * strideY_ + (size_t)y * pitchY_ * cs, (size_t)vi_.width * cs);
        if (U_ && V_)
            for (int y = 0; y < vi_.height / 2; ++y) {
                std::memcpy(f->GetWritePtr(PLANAR_U) + (size_t)y * f->GetPitch(PLANAR_U),
                            U_ + n * strideUV_ + (size_t)y * pitchUV_ * cs, (size_t)(vi_.width / 2) * cs);
                std::memcpy(f->GetWritePtr(PLANAR_V) + (size_t)y * f->GetPitch(PLANAR_V),
                            V_ + n * strideUV_ + (size_t)y * pitchUV_ * cs, (size_t)(vi_.width / 2) * cs);
            }
        return f;
    }
};

// what AviSynth puts in front of every filter instance: its cache clamps the frame number
class ClampCache : public IClip {
    PClip c_;
public:
    ClampCache(PClip c) : c_(c) { }
    const VideoInfo& GetVideoInfo() { return c_->GetVideoInfo(); }
    PVideoFrame GetFrame(int n, IScriptEnvironment* env)
    {
        const VideoInfo& vi = c_->GetVideoInfo();
        return c_->GetFrame(std::max(0, std::min(vi.num_frames - 1, n)), env);
    }
};

struct RefLogo {
    logo::LogoHeader header;
    std::unique_ptr<logo::LogoDataParam> p;
};

template <typename F> int guarded(F f)
{
    try { f(); return 1; }
    catch (const Exception& e) { g_err = e.message(); }
    catch (const AvisynthError& e) { g_err = e.msg; }
    catch (const std::exception& e) { g_err = e.what(); }
    return 0;
}

} // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }
int ref_is_avx() { return IsAVXAvailable() ? 1 : 0; }
float ref_corr5x5_scalar(const float* k, const float* Y, int x, int y, int w, float* pavg) { return CalcCorrelation5x5(k, Y, x, y, w, pavg); }
float ref_corr5x5_avx(const float* k, const float* Y, int x, int y, int w, float* pavg) { return CalcCorrelation5x5_AVX(k, Y, x, y, w, pavg); }

void* ref_logo_load(const char* path)
{
    RefLogo* r = new RefLogo;
    if (!guarded([&] { r->p.reset(new logo::LogoDataParam(logo::LogoData::Load(path, &r->header), &r->header)); })) { delete r; return nullptr; }
    return r;
}
int ref_logo_save(void* h, const char* path)
{
    RefLogo* r = (RefLogo*)h;
    return guarded([&] { r->p->Save(path, &r->header); });
}
void ref_logo_free(void* h) { delete (RefLogo*)h; }
void* ref_logo_deint(void* h)      // as AMTAnalyzeLogo ctor / LogoFrame ctor do (LogoScan.hpp:1177-1179, 1605-1606)
{
    RefLogo* s = (RefLogo*)h;
    RefLogo* r = new RefLogo;
    r->header = s->header;
    r->p.reset(new logo::LogoDataParam(logo::LogoData(s->header.w, s->header.h, s->header.logUVx, s->header.logUVy), &s->header));
    logo::DeintLogo(*r->p, *s->p, s->header.w, s->header.h);
    return r;
}
void* ref_logo_field(void* h, int bottom)
{
    RefLogo* s = (RefLogo*)h;
    RefLogo* r = new RefLogo;
    r->header = s->header;
    r->header.h /= 2; r->header.imgh /= 2; r->header.imgy /= 2;
    r->p = s->p->MakeFieldLogo(bottom != 0);
    return r;
}
void ref_logo_create_mask(void* h, float maskratio) { ((RefLogo*)h)->p->CreateLogoMask(maskratio); }
void ref_logo_info(void* h, int* o)
{
    logo::LogoDataParam& p = *((RefLogo*)h)->p;
    o[0] = p.w; o[1] = p.h; o[2] = p.logUVx; o[3] = p.logUVy; o[4] = p.imgw; o[5] = p.imgh; o[6] = p.imgx; o[7] = p.imgy;
    o[8] = p.mask ? p.maskpixels : 0; o[9] = 0;
}
const float* ref_logo_data(void* h) { return ((RefLogo*)h)->p->data.get(); }
const uint8_t* ref_logo_mask(void* h) { return ((RefLogo*)h)->p->GetMask(); }
const float* ref_logo_kernels(void* h) { return ((RefLogo*)h)->p->GetKernels(); }
const float* ref_logo_scales(void* h) { return (const float*)((RefLogo*)h)->p->scales.get(); }
float ref_logo_black_score(void* h) { return ((RefLogo*)h)->p->blackScore; }
float ref_evaluate_logo(void* h, const float* src, float maxv, float fade, float* work, int stride)
{
    return ((RefLogo*)h)->p->EvaluateLogo(src, maxv, fade, work, stride);
}
void ref_deint_y_u8(float* d, const uint8_t* s, int p, int w, int hh) { logo::DeintY(d, s, p, w, hh); }
void ref_deint_y_u16(float* d, const uint16_t* s, int p, int w, int hh) { logo::DeintY(d, s, p, w, hh); }
void ref_copy_y_u8(float* d, const uint8_t* s, int p, int w, int hh) { logo::CopyY(d, s, p, w, hh); }

// LogoFrame: ctor + scanFrames + selectLogo(ncand) + writeResult(tmp, logoIndex) (CMAnalyze.hpp:291-299)
int ref_logoframe(const char* const* paths, int nlogos, float maskratio, const void* Y, int64_t frame_stride,
                  int pitch, int bits, int w, int h, int nframes, int fps_num, int fps_den, float* evals_out,
                  int ncand, int* best, float* ratio, int logoIndex, const char* tmp_path, char* text, int cap)
{
    return guarded([&] {
        std::vector<tstring> files(paths, paths + nlogos);
        logo::LogoFrame lf(g_ctx, files, maskratio);
        PClip clip(new MemClip(w, h, bits, nframes, fps_num, fps_den, Y, nullptr, nullptr, frame_stride, 0, pitch, 0));
        IScriptEnvironment2 env;
        lf.scanFrames(clip, &env);
        for (int i = 0; i < nframes * nlogos; ++i) { evals_out[i * 2] = lf.evalResults[i].corr0; evals_out[i * 2 + 1] = lf.evalResults[i].corr1; }
        lf.selectLogo(ncand);
        *best = lf.getBestLogo();
        *ratio = lf.getLogoRatio();
        lf.writeResult(tmp_path, logoIndex);
        std::ifstream in(tmp_path, std::ios::binary);
        std::string s((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if ((int)s.size() + 1 > cap) THROW(RuntimeException, "text buffer too small");
        std::memcpy(text, s.c_str(), s.size() + 1);
    });
}

// AMTAnalyzeLogo over a memory clip; out = nframes*33 (source frame order)
int ref_analyze(const char* logopath, float maskratio, const void* Y, const void* U, const void* V,
                int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int bits, int w, int h, int nframes, float* out)
{
    return guarded([&] {
        IScriptEnvironment2 env;
        PClip src(new MemClip(w, h, bits, nframes, 30000, 1001, Y, U, V, strideY, strideUV, pitchY, pitchUV));
        PClip an(new logo::AMTAnalyzeLogo(src, logopath, maskratio, &env));
        int na = an->GetVideoInfo().num_frames;
        for (int q = 0; q < na; ++q) {
            PVideoFrame f = an->GetFrame(q, &env);
            const logo::LogoAnalyzeFrame* p = (const logo::LogoAnalyzeFrame*)f->GetReadPtr();
            for (int i = 0; i < 8; ++i) {
                int n = q * 8 + i;
                if (n < nframes) std::memcpy(out + (size_t)n * 33, &p[i], sizeof(float) * 33);
            }
        }
    });
}

// AMTEraseLogo(AMTAnalyzeLogo(logo), logo, logof, maxfade=) (FilteredSource.hpp:456-457) over a memory clip;
// planes are rewritten in place; fades_out = nframes*2 (CalcFade results)
int ref_erase(const char* logopath, const char* logofpath, int maxfade, float maskratio, void* Y, void* U, void* V,
              int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int bits, int w, int h, int nframes, float* fades_out)
{
    return guarded([&] {
        IScriptEnvironment2 env;
        PClip src(new MemClip(w, h, bits, nframes, 30000, 1001, Y, U, V, strideY, strideUV, pitchY, pitchUV));
        PClip an(new ClampCache(PClip(new logo::AMTAnalyzeLogo(src, logopath, maskratio, &env))));
        logo::AMTEraseLogo* er = new logo::AMTEraseLogo(src, an, logopath, logofpath, 0, maxfade, &env);
        PClip erc(er);
        int cs = bits <= 8 ? 1 : 2;
        std::vector<std::vector<uint8_t>> outY(nframes), outU(nframes), outV(nframes);
        for (int n = 0; n < nframes; ++n) {
            if (fades_out) {
                float t, b;
                er->CalcFade(n, t, b, &env);
                fades_out[n * 2] = t; fades_out[n * 2 + 1] = b;
            }
            PVideoFrame f = erc->GetFrame(n, &env);
            outY[n].assign(f->GetReadPtr(PLANAR_Y), f->GetReadPtr(PLANAR_Y) + (size_t)f->GetPitch(PLANAR_Y) * h);
            outU[n].assign(f->GetReadPtr(PLANAR_U), f->GetReadPtr(PLANAR_U) + (size_t)f->GetPitch(PLANAR_U) * (h / 2));
            outV[n].assign(f->GetReadPtr(PLANAR_V), f->GetReadPtr(PLANAR_V) + (size_t)f->GetPitch(PLANAR_V) * (h / 2));
        }
        VideoInfo vi = src->GetVideoInfo();
        int py = shim_align64(w * cs), puv = shim_align64((w / 2) * cs);
        for (int n = 0; n < nframes; ++n) {
            for (int y = 0; y < h; ++y)
                std::memcpy((uint8_t*)Y + n * strideY + (size_t)y * pitchY * cs, &outY[n][(size_t)y * py], (size_t)w * cs);
            for (int y = 0; y < h / 2; ++y) {
                std::memcpy((uint8_t*)U + n * strideUV + (size_t)y * pitchUV * cs, &outU[n][(size_t)y * puv], (size_t)(w / 2) * cs);
                std::memcpy((uint8_t*)V + n * strideUV + (size_t)y * pitchUV * cs, &outV[n][(size_t)y * puv], (size_t)(w / 2) * cs);
            }
        }
        (void)vi;
    });
}

// LogoScan accumulate / regress
void* ref_scan_create(int w, int h, int lx, int ly, int thy) { return new logo::LogoScan(w, h, lx, ly, thy); }
void ref_scan_free(void* s) { delete (logo::LogoScan*)s; }
int ref_scan_add_frame_u8(void* s, const uint8_t* Y, const uint8_t* U, const uint8_t* V, int pitchY, int pitchUV)
{
    return ((logo::LogoScan*)s)->AddFrame(Y, U, V, pitchY, pitchUV) ? 1 : 0;
}
int ref_scan_nframes(void* s) { return ((logo::LogoScan*)s)->nframes; }
void ref_scan_sums(void* sp, double* out)
{
    logo::LogoScan* s = (logo::LogoScan*)sp;
    int ny = s->scanw * s->scanh, nc = ny >> (s->logUVx + s->logUVy);
    auto dump = [&](const logo::LogoColor* v, int n) {
        for (int i = 0; i < n; ++i) { *out++ = v[i].sumF; *out++ = v[i].sumB; *out++ = v[i].sumF2; *out++ = v[i].sumB2; *out++ = v[i].sumFB; }
    };
    dump(s->logoY.get(), ny); dump(s->logoU.get(), nc); dump(s->logoV.get(), nc);
}
// Normalize(maxv) + GetLogo(clean); the sums are restored afterwards so the call can be repeated
void* ref_scan_get_logo(void* sp, int maxv, int clean, int imgw, int imgh, int imgx, int imgy)
{
    logo::LogoScan* s = (logo::LogoScan*)sp;
    int ny = s->scanw * s->scanh, nc = ny >> (s->logUVx + s->logUVy);
    std::vector<logo::LogoColor> sy(s->logoY.get(), s->logoY.get() + ny), su(s->logoU.get(), s->logoU.get() + nc), sv(s->logoV.get(), s->logoV.get() + nc);
    s->Normalize(maxv);
    std::unique_ptr<logo::LogoData> d = s->GetLogo(clean != 0);
    std::copy(sy.begin(), sy.end(), s->logoY.get());
    std::copy(su.begin(), su.end(), s->logoU.get());
    std::copy(sv.begin(), sv.end(), s->logoV.get());
    if (!d) return nullptr;
    RefLogo* r = new RefLogo;
    r->header = logo::LogoHeader(s->scanw, s->scanh, s->logUVx, s->logUVy, imgw, imgh, imgx, imgy, "No Name");
    r->p.reset(new logo::LogoDataParam(std::move(*d), &r->header));
    return r;
}

// the reference's own exported C entry point (LogoScan.hpp:1083-1098) over a raw clip file (see shim av layer)
int ref_scanlogo(const char* srcpath, int serviceid, const char* workfile, const char* dstpath,
                 int imgx, int imgy, int w, int h, int thy, int numMaxFrames)
{
    int r = logo::ScanLogo(&g_ctx, srcpath, serviceid, workfile, dstpath, imgx, imgy, w, h, thy, numMaxFrames,
                           [](float, int, int, int) { return true; });
    if (!r) g_err = g_ctx.getError();
    return r;
}

} // extern "C"
