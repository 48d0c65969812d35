// logo_model.cpp -- see logo_model.hpp.  Host code; fp32 order-sensitive parts go through exact_math.h.
#include "build_knobs.h"
#include "logo_model.hpp"
#include "exact_math.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace amt {

// ------------------------------------------------------------------------------------------------
// .lgd on-disk layout (AMTLogo.hpp:169-196,239-279; include/logo.h:31-78).  Fixed-width types: the
// reference is an LLP64 build, its `unsigned long logonum` is 4 bytes.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kFileHeaderBytes = 32;   // char[28] magic string + uint32 big-endian logo count
constexpr int kBaseHeaderBytes = 48;   // char name[32]; int16 x,y,h,w,fi,fo,st,ed
constexpr int kBasePixelBytes = 12;    // int16 dp_y,y,dp_cb,cb,dp_cr,cr
constexpr int kExtHeaderBytes = 540;   // int32[10]; char name[255]; pad; int32 serviceId; int32 reserved[60]
const char kMagic[] = "<logo data file ver0.1>";

struct Reader {
    FILE* fp;
    explicit Reader(const std::string& p) : fp(std::fopen(p.c_str(), "rb"))
    {
        if (!fp) throw std::runtime_error("failed to open file " + p);
    }
    ~Reader() { std::fclose(fp); }
    void get(void* dst, size_t n)
    {
        if (n && std::fread(dst, 1, n, fp) != n) throw std::runtime_error("failed to read from file");
    }
    void skip(long n) { if (std::fseek(fp, n, SEEK_CUR) != 0) throw std::runtime_error("failed to seek"); }
};

int32_t rd32(const uint8_t* p) { int32_t v; std::memcpy(&v, p, 4); return v; }
int16_t rd16(const uint8_t* p) { int16_t v; std::memcpy(&v, p, 2); return v; }
void wr32(uint8_t* p, int32_t v) { std::memcpy(p, &v, 4); }
void wr16(uint8_t* p, int16_t v) { std::memcpy(p, &v, 2); }

// AviUtl YC48 conversion of one (A,B) pair -> (colour, opacity) int16 pair (AMTLogo.hpp:58-167).
// luma: YV12->YC48 maps via ((v*255*1197)>>6)-299, chroma via (((v*255-128)*4681+164)>>8);
// the probe points 0 and 2048 of the YC48 axis are first mapped to YV12 [0,1] values.
struct Yc48Map {
    float yv12_at0, yv12_at2048;
    float (*to_yc48)(float);
};
float yc48_luma(float y) { return float(((int(y * 255) * 1197) >> 6) - 299); }
float yc48_chroma(float u) { return float(((int(u * 255) - 128) * 4681 + 164) >> 8); }
float yv12_luma(int v) { return float((((v * 219 + 383) >> 12) + 16) / 255.0f); }
float yv12_chroma(int v) { return float(((((v + 2048) * 7 + 66) >> 7) + 16) / 255.0f); }

void base_pixel_pair(float A, float B, bool luma, int16_t& colour, int16_t& opacity)
{
    const float x0 = luma ? yv12_luma(0) : yv12_chroma(0);
    const float x1 = luma ? yv12_luma(2048) : yv12_chroma(2048);
    float y0 = (x0 - B) / A;
    float y1 = (x1 - B) / A;
    y0 = luma ? yc48_luma(y0) : yc48_chroma(y0);
    y1 = luma ? yc48_luma(y1) : yc48_chroma(y1);
    const float B48 = y0;
    const float A48 = (y1 - y0) / 2048.0f;
    colour = opacity = 0;
    if (A48 == 1) return;
    float t = B48 / (1 - A48) + 0.5f;
    if (!(std::abs(t) < 0x7FFF)) return;
    const int16_t col = (int16_t)t;
    t = (1 - A48) * 1000 + 0.5f;
    if (std::abs(t) > 0x3FFF || int16_t(t) == 0) return;
    colour = col;
    opacity = (int16_t)t;
}

} // namespace

LogoPlanes load_lgd(const std::string& path)
{
    Reader r(path);
    uint8_t fh[kFileHeaderBytes], bh[kBaseHeaderBytes], eh[kExtHeaderBytes];
    r.get(fh, sizeof fh);
    r.get(bh, sizeof bh);
    const int bhH = rd16(bh + 36), bhW = rd16(bh + 38);
    r.skip((long)bhH * bhW * kBasePixelBytes);          // base section is only for AviUtl
    r.get(eh, sizeof eh);
    LogoPlanes L;
    L.w = rd32(eh + 8); L.h = rd32(eh + 12); L.logUVx = rd32(eh + 16); L.logUVy = rd32(eh + 20);
    L.imgw = rd32(eh + 24); L.imgh = rd32(eh + 28); L.imgx = rd32(eh + 32); L.imgy = rd32(eh + 36);
    if (L.w <= 0 || L.h <= 0 || L.w > 16384 || L.h > 16384 || L.logUVx < 0 || L.logUVx > 2 || L.logUVy < 0 || L.logUVy > 2)
        throw std::runtime_error("bad logo header in " + path);
    L.name.assign((const char*)eh + 40, strnlen((const char*)eh + 40, 255));
    L.serviceId = rd32(eh + 296);
    L.allocate();
    r.get(L.data.data(), L.data.size() * sizeof(float));
    return L;
}

void save_lgd(const LogoPlanes& L, const std::string& path, const std::string& name, int serviceId)
{
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("failed to open file " + path);
    std::vector<uint8_t> buf(kFileHeaderBytes + kBaseHeaderBytes + (size_t)L.w * L.h * kBasePixelBytes + kExtHeaderBytes, 0);
    uint8_t* p = buf.data();
    std::memcpy(p, kMagic, sizeof kMagic - 1);
    p[31] = 1;                                           // logonum = 1, big endian
    p += kFileHeaderBytes;
    std::strncpy((char*)p, name.c_str(), 31);
    wr16(p + 32, (int16_t)L.imgx); wr16(p + 34, (int16_t)L.imgy);
    wr16(p + 36, (int16_t)L.h); wr16(p + 38, (int16_t)L.w);
    p += kBaseHeaderBytes;
    const int wUV = L.wUV();
    for (int y = 0; y < L.h; ++y)
        for (int x = 0; x < L.w; ++x, p += kBasePixelBytes) {
            const int o = x + y * L.w, oc = (x >> L.logUVx) + (y >> L.logUVy) * wUV;
            int16_t col, dp;
            base_pixel_pair(L.A(0)[o], L.B(0)[o], true, col, dp);   wr16(p + 0, dp); wr16(p + 2, col);
            base_pixel_pair(L.A(1)[oc], L.B(1)[oc], false, col, dp); wr16(p + 4, dp); wr16(p + 6, col);
            base_pixel_pair(L.A(2)[oc], L.B(2)[oc], false, col, dp); wr16(p + 8, dp); wr16(p + 10, col);
        }
    wr32(p + 0, 0x12345); wr32(p + 4, 1);
    wr32(p + 8, L.w); wr32(p + 12, L.h); wr32(p + 16, L.logUVx); wr32(p + 20, L.logUVy);
    wr32(p + 24, L.imgw); wr32(p + 28, L.imgh); wr32(p + 32, L.imgx); wr32(p + 36, L.imgy);
    std::strncpy((char*)p + 40, name.c_str(), 254);      // char name[255]; (LogoHeader's ctor itself keeps 31 chars, AMTLogo.hpp:45)
    wr32(p + 296, serviceId);
    bool ok = std::fwrite(buf.data(), 1, buf.size(), fp) == buf.size() &&
              std::fwrite(L.data.data(), sizeof(float), L.data.size(), fp) == L.data.size();
    std::fclose(fp);
    if (!ok) throw std::runtime_error("failed to write to file " + path);
}

LogoPlanes deinterlaced_logo(const LogoPlanes& S)
{
    LogoPlanes D = S;
    D.allocate();                                        // chroma stays unset like the reference's (never read)
    const int w = S.w, h = S.h;
    for (int pl = 0; pl < 2; ++pl) {                     // A then B of the Y plane
        const float* s = pl ? S.B(0) : S.A(0);
        float* d = pl ? D.B(0) : D.A(0);
        std::memcpy(d, s, sizeof(float) * w);
        std::memcpy(d + (size_t)(h - 1) * w, s + (size_t)(h - 1) * w, sizeof(float) * w);
        for (int y = 1; y < h - 1; ++y) {
            const float *up = s + (size_t)(y - 1) * w, *mid = up + w, *dn = mid + w;
            float* o = d + (size_t)y * w;
            for (int x = 0; x < w; ++x) o[x] = (up[x] + 2 * mid[x] + dn[x]) / 4.0f;
        }
    }
    return D;
}

LogoPlanes field_logo(const LogoPlanes& S, bool bottom)
{
    LogoPlanes F;
    F.w = S.w; F.h = S.h / 2; F.logUVx = S.logUVx; F.logUVy = S.logUVy;
    F.imgw = S.imgw; F.imgh = S.imgh / 2; F.imgx = S.imgx; F.imgy = S.imgy / 2;
    F.name = S.name; F.serviceId = S.serviceId;
    F.allocate();
    const int w = S.w;
    for (int y = 0; y < F.h; ++y) {
        std::memcpy(F.A(0) + (size_t)y * w, S.A(0) + (size_t)((bottom ? 1 : 0) + 2 * y) * w, sizeof(float) * w);
        std::memcpy(F.B(0) + (size_t)y * w, S.B(0) + (size_t)((bottom ? 1 : 0) + 2 * y) * w, sizeof(float) * w);
    }
    const int cw = F.wUV(), ch = F.hUV();
    const int first = (bottom ? 1 : 0) ^ (F.imgy % 2);   // chroma row parity follows the field's own imgy
    for (int pl = 1; pl <= 2; ++pl)
        for (int y = 0; y < ch; ++y) {
            std::memcpy(F.A(pl) + (size_t)y * cw, S.A(pl) + (size_t)(first + 2 * y) * cw, sizeof(float) * cw);
            std::memcpy(F.B(pl) + (size_t)y * cw, S.B(pl) + (size_t)(first + 2 * y) * cw, sizeof(float) * cw);
        }
    return F;
}

// ------------------------------------------------------------------------------------------------
// evaluation tables (LogoDataParam::CreateLogoMask, LogoScan.hpp:112-229)
// ------------------------------------------------------------------------------------------------
namespace {

inline void gather_window(const float* plane, int w, int x, int y, float v[5][5])
{
    for (int r = 0; r < 5; ++r)
        for (int c = 0; c < 5; ++c) v[r][c] = plane[(x - 2 + c) + (size_t)(y - 2 + r) * w];
}

// 25-tap patch with its (sequentially summed) mean removed -- the order std::accumulate uses
inline void centred_patch(const float* plane, int w, int x, int y, float k[25])
{
    float s = 0.0f;
    for (int r = 0; r < 5; ++r)
        for (int c = 0; c < 5; ++c) {
            float t = plane[(x - 2 + c) + (size_t)(y - 2 + r) * w];
            k[r * 5 + c] = t;
            s += t;
        }
    const float m = s / 25;
    for (int i = 0; i < 25; ++i) k[i] = k[i] - m;
}

} // namespace

float correlation_score_host(const MaskTables& t, const float* work)
{
    float total = 0;
    float v[5][5];
    for (int n = 0; n < t.count; ++n) {
        const int x = t.pos[n] & 0xFFFF, y = t.pos[n] >> 16;
        gather_window(work, t.w, x, y, v);
        float mean;
        const float c = corr5x5(&t.kernels[(size_t)n * 25], v, &mean);
        const float* sl = &t.scales[((size_t)n * 32 + score_bin(mean)) * 2];
        total += score_term(c, sl[0], sl[1]);
    }
    return total;
}

MaskTables build_mask_tables(const LogoPlanes& L, float maskratio)
{
    // maskratio arrives from the user's script (AMTAnalyzeLogo's [maskratio], CMAnalyze's setting): a negative or NaN value
    // would turn into a negative pixel count below
    if (!(maskratio >= 0.0f) || !std::isfinite(maskratio)) throw std::runtime_error("maskratio must be a finite value >= 0");
    MaskTables T;
    const int w = T.w = L.w, h = T.h = L.h;
    const int npx = w * h;
    const float* a = L.A(0);
    const float* b = L.B(0);

    // the logo composited over 32 flat backgrounds 0,8,...,248 (always on a 0..255 scale)
    std::vector<float> flat((size_t)npx * 32);
    for (int i = 0; i < npx; ++i) {
        const float lift = b[i] * 255;
        for (int c = 0; c < 32; ++c) {
            const float base = (float)(c << 3);
            flat[(size_t)c * npx + i] = (a[i] > 0) ? (base - lift) / a[i] : base;
        }
    }
    auto level = [&](int c) { return flat.data() + (size_t)c * npx; };

    // feature strength: variance of the mean-removed window on the mid-grey composite
    std::vector<std::pair<float, int>> strength(npx);
    for (int i = 0; i < npx; ++i) strength[i] = {0.0f, i};
    {
        float k[25];
        const float* mid = level(16);
        for (int y = 2; y < h - 2; ++y)
            for (int x = 2; x < w - 2; ++x) {
                centred_patch(mid, w, x, y, k);
                float s = 0.0f;
                for (int i = 0; i < 25; ++i) s = s + k[i] * k[i];
                strength[x + y * w].first = s;
            }
    }
    T.maskpixels = std::min(npx, (int)(npx * maskratio));
    // strongest first; equal strengths rank the larger index first (descending pair order)
    auto stronger = [](const std::pair<float, int>& p, const std::pair<float, int>& q) { return q < p; };
    if (T.maskpixels < npx) std::nth_element(strength.begin(), strength.begin() + T.maskpixels, strength.end(), stronger);
    T.mask.assign(npx, 0);
    for (int i = 0; i < T.maskpixels; ++i) T.mask[strength[i].second] = 1;

    for (int y = 2; y < h - 2; ++y)
        for (int x = 2; x < w - 2; ++x)
            if (T.mask[x + y * w]) T.pos.push_back(((uint32_t)y << 16) | (uint32_t)x);
    T.count = (int)T.pos.size();
    T.kernels.resize((size_t)T.count * 25);
    T.scales.assign((size_t)T.count * 64, 0.0f);
    T.resp.assign((size_t)T.count * 32, 0.0f);

    // expected response of every kernel on every flat level; running mean in visiting order
    float total = 0.0f;
    float v[5][5];
    for (int n = 0; n < T.count; ++n) {
        const int x = T.pos[n] & 0xFFFF, y = T.pos[n] >> 16;
        float* k = &T.kernels[(size_t)n * 25];
        centred_patch(level(0), w, x, y, k);
        for (int c = 0; c < 32; ++c) {
            gather_window(level(c), w, x, y, v);
            float mean;
            const float signedResp = corr5x5(k, v, &mean);
            const float r = std::abs(signedResp);
            T.resp[(size_t)n * 32 + c] = signedResp;
            T.scales[((size_t)n * 32 + c) * 2] = r;
            total += r;
        }
    }
    const float meanResp = total / (T.maskpixels * 32);
    const float floorResp = meanResp * 0.2f;
    T.floorResp = floorResp;
    for (size_t i = 0; i < (size_t)T.count * 32; ++i) {
        const float r = T.scales[i * 2];
        T.scales[i * 2] = (r > 0) ? (1.0f / r) : 0.0f;
        T.scales[i * 2 + 1] = std::min(1.0f, r / floorResp);
    }
    T.blackScore = correlation_score_host(T, level(2));     // flat 16 = broadcast black
    return T;
}

} // namespace amt
