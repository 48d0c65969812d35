/*
 * amt_oracle.cpp -- CPU oracle: restatement of the reference's logo hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see amt_oracle.h).  Every function cites the reference
 * file:line it follows (paths relative to /root/reference).  Arithmetic types and
 * evaluation order are kept exactly as the reference writes them: IEEE fp32, no FMA
 * contraction (build with -ffp-contract=off), double only where the reference uses
 * double.  Build: see oracle/Makefile (g++ -O2 -mavx -ffp-contract=off -fno-fast-math).
 *
 * Reference quirks mirrored on purpose (SURVEY.md section 8a "Q" notes):
 *   - CreateLogoMask always composites with maxv 255 and sorts ties by larger index first
 *   - CorrelationScore ignores its maxv argument (bin = clamp((int)avg,0,255)>>3)
 *   - maxfilter() never changes dist, so GetLogo(clean) thresholds the raw dist
 *   - CalcFade2 samples n+2i (clamp(n+i)+i), wrapping through the analyze clip's 8-slot frames
 *   - .lgd file header uses a 4-byte logonum (LLP64 unsigned long)
 */
#include "amt_oracle.h"

#include <immintrin.h>
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <numeric>
#include <regex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

enum { KLEN = 25, CSHIFT = 3, CLEN = 256 >> CSHIFT };   // LogoScan.hpp:63-68

typedef float (*corr_fn)(const float*, const float*, int, int, int, float*);

// LogoScan.hpp:24-41 -- row-major sequential sums
float corr_scalar(const float* k, const float* Y, int x, int y, int w, float* pavg)
{
    float mean = 0.0f;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx)
            mean += Y[(x + dx) + (y + dy) * w];
    mean /= 25;
    float acc = 0.0f;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx)
            acc += k[(dx + 2) + (dy + 2) * 5] * (Y[(x + dx) + (y + dy) * w] - mean);
    if (pavg) *pavg = mean;
    return acc;
}

// ComputeKernel.cpp:54-74 -- ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7)); lanes 5..7 are zero here
inline float hsum8(__m256 v)
{
    __m128 hi = _mm256_extractf128_ps(v, 1);
    __m128 lo = _mm256_castps256_ps128(v);
    __m128 q = _mm_add_ps(lo, hi);
    __m128 d = _mm_add_ps(q, _mm_movehl_ps(q, q));
    __m128 s = _mm_add_ss(d, _mm_shuffle_ps(d, d, 0x1));
    float r;
    _mm_store_ss(&r, s);
    return r;
}

// ComputeKernel.cpp:77-121 -- reads 3 floats past each row and past k+24
float corr_avx(const float* k, const float* Y, int x, int y, int w, float* pavg)
{
    const __m256 keep5 = _mm256_castsi256_ps(_mm256_set_epi32(0, 0, 0, -1, -1, -1, -1, -1));
    const float* p = Y + (x - 2) + w * (y - 2);
    __m256 r0 = _mm256_loadu_ps(p);
    __m256 r1 = _mm256_loadu_ps(p + w);
    __m256 r2 = _mm256_loadu_ps(p + 2 * w);
    __m256 r3 = _mm256_loadu_ps(p + 3 * w);
    __m256 r4 = _mm256_loadu_ps(p + 4 * w);
    __m256 col = _mm256_add_ps(_mm256_add_ps(_mm256_add_ps(r0, r1), _mm256_add_ps(r2, r3)), r4);
    float mean = hsum8(_mm256_and_ps(col, keep5));
    mean /= 25;
    __m256 vm = _mm256_broadcast_ss(&mean);
    __m256 t0 = _mm256_mul_ps(_mm256_loadu_ps(k + 0), _mm256_sub_ps(r0, vm));
    __m256 t1 = _mm256_mul_ps(_mm256_loadu_ps(k + 5), _mm256_sub_ps(r1, vm));
    __m256 t2 = _mm256_mul_ps(_mm256_loadu_ps(k + 10), _mm256_sub_ps(r2, vm));
    __m256 t3 = _mm256_mul_ps(_mm256_loadu_ps(k + 15), _mm256_sub_ps(r3, vm));
    __m256 t4 = _mm256_mul_ps(_mm256_loadu_ps(k + 20), _mm256_sub_ps(r4, vm));
    __m256 pr = _mm256_add_ps(_mm256_add_ps(_mm256_add_ps(t0, t1), _mm256_add_ps(t2, t3)), t4);
    float acc = hsum8(_mm256_and_ps(pr, keep5));
    if (pavg) *pavg = mean;
    return acc;
}

struct ScaleLimit { float scale, scale2; };   // LogoScan.hpp:72-75

} // namespace

// LogoDataParam : LogoData (+ the LogoHeader fields that travel with it)
struct OrcLogo {
    int w = 0, h = 0, logUVx = 0, logUVy = 0;
    int imgw = 0, imgh = 0, imgx = 0, imgy = 0;
    std::vector<float> data;            // aY,bY,aU,bU,aV,bV (AMTLogo.hpp:204-212)
    float *aY = nullptr, *bY = nullptr, *aU = nullptr, *bU = nullptr, *aV = nullptr, *bV = nullptr;
    std::vector<uint8_t> mask;
    std::vector<float> kernels;         // maskpixels*25 + 8
    std::vector<ScaleLimit> scales;     // maskpixels*32
    int maskpixels = 0;
    int count = 0;                      // mask pixels actually visited (interior)
    float blackScore = 0;
    corr_fn corr = corr_avx;

    void alloc(int w_, int h_, int lx, int ly)
    {
        w = w_; h = h_; logUVx = lx; logUVy = ly;
        int wUV = w >> lx, hUV = h >> ly;
        data.assign((size_t)(w * h + wUV * hUV * 2) * 2, 0.0f);
        aY = data.data();
        bY = aY + w * h;
        aU = bY + w * h;
        bU = aU + wUV * hUV;
        aV = bU + wUV * hUV;
        bV = aV + wUV * hUV;
    }
};

namespace {

// LogoScan.hpp:320-333
void add_logo(const OrcLogo& L, float* Y, int maxv)
{
    for (int y = 0; y < L.h; ++y)
        for (int x = 0; x < L.w; ++x) {
            float a = L.aY[x + y * L.w];
            float b = L.bY[x + y * L.w];
            if (a > 0) Y[x + y * L.w] = (Y[x + y * L.w] - b * maxv) / a;
        }
}

// LogoScan.hpp:135-147
void make_kernel(float* k, const float* Y, int x, int y, int w)
{
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx)
            k[(dx + 2) + (dy + 2) * 5] = Y[(x + dx) + (y + dy) * w];
    float mean = std::accumulate(k, k + KLEN, 0.0f) / KLEN;
    for (int i = 0; i < KLEN; ++i) k[i] = k[i] - mean;
}

// LogoScan.hpp:288-318
float correlation_score(const OrcLogo& L, const float* work)
{
    int n = 0;
    float total = 0;
    for (int y = 2; y < L.h - 2; ++y)
        for (int x = 2; x < L.w - 2; ++x) {
            if (!L.mask[x + y * L.w]) continue;
            const float* k = &L.kernels[(size_t)n * KLEN];
            float mean;
            float s = L.corr(k, work, x, y, L.w, &mean);
            ScaleLimit sl = L.scales[(size_t)n * CLEN + (std::max(0, std::min(255, (int)mean)) >> CSHIFT)];
            float normalized = std::max(-1.0f, std::min(1.0f, s * sl.scale));
            float score = normalized * sl.scale2;
            total += score;
            ++n;
        }
    return total;
}

// LogoScan.hpp:112-229
void create_logo_mask(OrcLogo& L, float maskratio)
{
    const float corrLowerLimit = 0.2f;
    const int w = L.w, h = L.h;
    const int YSize = w * h;
    std::vector<float> memWork((size_t)YSize * CLEN + 8);
    for (int c = 0; c < CLEN; ++c) {
        float* slice = &memWork[(size_t)c * YSize];
        std::fill_n(slice, YSize, (float)(c << CSHIFT));
        add_logo(L, slice, 255);
    }
    std::vector<std::pair<float, int>> variance(YSize);
    for (int y = 2; y < h - 2; ++y)
        for (int x = 2; x < w - 2; ++x) {
            const float* slice = &memWork[(size_t)(CLEN >> 1) * YSize];
            float k[KLEN];
            make_kernel(k, slice, x, y, w);
            variance[x + y * w].first =
                std::accumulate(k, k + KLEN, 0.0f, [](float s, float v) { return s + v * v; });
        }
    for (int i = 0; i < YSize; ++i) variance[i].second = i;
    std::sort(variance.begin(), variance.end(), std::greater<std::pair<float, int>>());
    L.mask.assign(YSize, 0);
    L.maskpixels = std::min(YSize, (int)(YSize * maskratio));
    for (int i = 0; i < L.maskpixels; ++i) L.mask[variance[i].second] = 1;

    L.kernels.assign((size_t)L.maskpixels * KLEN + 8, 0.0f);
    L.scales.assign((size_t)L.maskpixels * CLEN, ScaleLimit{0.0f, 0.0f});  // reference leaves the unvisited tail uninitialised
    int n = 0;
    float avgCorr = 0.0f;
    for (int y = 2; y < h - 2; ++y)
        for (int x = 2; x < w - 2; ++x) {
            if (!L.mask[x + y * w]) continue;
            float* k = &L.kernels[(size_t)n * KLEN];
            ScaleLimit* s = &L.scales[(size_t)n * CLEN];
            make_kernel(k, memWork.data(), x, y, w);
            for (int i = 0; i < CLEN; ++i) {
                const float* slice = &memWork[(size_t)i * YSize];
                avgCorr += s[i].scale = std::abs(L.corr(k, slice, x, y, w, nullptr));
            }
            ++n;
        }
    L.count = n;
    avgCorr /= L.maskpixels * CLEN;
    float limitCorr = avgCorr * corrLowerLimit;
    for (int i = 0; i < n * CLEN; ++i) {       // tail beyond n*CLEN is never read back (count stops at n)
        float c = L.scales[i].scale;
        L.scales[i].scale = (c > 0) ? (1.0f / c) : 0.0f;
        L.scales[i].scale2 = std::min(1.0f, c / limitCorr);
    }
    const float* slice = &memWork[(size_t)(16 >> CSHIFT) * YSize];
    L.blackScore = correlation_score(L, slice);
}

// LogoScan.hpp:231-255
float evaluate_logo(const OrcLogo& L, const float* src, float maxv, float fade, float* work, int stride)
{
    if (stride == -1) stride = L.w;
    for (int y = 0; y < L.h; ++y)
        for (int x = 0; x < L.w; ++x) {
            float srcv = src[x + y * stride];
            float a = L.aY[x + y * L.w];
            float b = L.bY[x + y * L.w];
            float bg = a * srcv + b * maxv;
            float dstv = fade * bg + (1 - fade) * srcv;
            work[x + y * L.w] = dstv;
        }
    return correlation_score(L, work) / L.blackScore;
}

// LogoScan.hpp:763-780
template <typename T> void deint_y(float* dst, const T* src, int pitch, int w, int h)
{
    for (int x = 0; x < w; ++x) {
        dst[x] = src[x];
        dst[x + (h - 1) * w] = src[x + (h - 1) * pitch];
    }
    for (int y = 1; y < h - 1; ++y)
        for (int x = 0; x < w; ++x) {
            int a = src[x + (y - 1) * pitch], b = src[x + y * pitch], c = src[x + (y + 1) * pitch];
            dst[x + y * w] = (a + 2 * b + c + 2) / 4.0f;
        }
}

// LogoScan.hpp:782-790
template <typename T> void copy_y(float* dst, const T* src, int pitch, int w, int h)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) dst[x + y * w] = src[x + y * pitch];
}

// LogoScan.hpp:1248-1261
template <typename T>
void delogo(T* dst, int w, int h, int logopitch, int imgpitch, float maxv, const float* A, const float* B, float fade)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float srcv = dst[x + y * imgpitch];
            float a = A[x + y * logopitch];
            float b = B[x + y * logopitch];
            float bg = a * srcv + b * maxv;
            float tmp = fade * bg + (1 - fade) * srcv;
            dst[x + y * imgpitch] = (T)std::min(std::max(tmp + 0.5f, 0.0f), maxv);
        }
}

// LogoScan.hpp:1343-1400 (mode 0)
template <typename T>
void erase_frame(const OrcLogo& L, T* Y, T* U, T* V, int pitchY, int pitchUV, float maxv, float fadeT, float fadeB)
{
    int off = L.imgx + L.imgy * pitchY;
    int offUV = (L.imgx >> L.logUVx) + (L.imgy >> L.logUVy) * pitchUV;
    int wUV = L.w >> L.logUVx, hUV = L.h >> L.logUVy;
    if (fadeT == fadeB) {
        delogo(Y + off, L.w, L.h, L.w, pitchY, maxv, L.aY, L.bY, fadeT);
        delogo(U + offUV, wUV, hUV, wUV, pitchUV, maxv, L.aU, L.bU, fadeT);
        delogo(V + offUV, wUV, hUV, wUV, pitchUV, maxv, L.aV, L.bV, fadeT);
    } else {
        delogo(Y + off, L.w, L.h / 2, L.w * 2, pitchY * 2, maxv, L.aY, L.bY, fadeT);
        delogo(Y + off + pitchY, L.w, L.h / 2, L.w * 2, pitchY * 2, maxv, L.aY + L.w, L.bY + L.w, fadeB);
        int uvparity = ((L.imgy / 2) % 2);
        int tuvoff = uvparity * pitchUV, buvoff = !uvparity * pitchUV;
        int tuvoffl = uvparity * wUV, buvoffl = !uvparity * wUV;
        delogo(U + offUV + tuvoff, wUV, hUV / 2, wUV * 2, pitchUV * 2, maxv, L.aU + tuvoffl, L.bU + tuvoffl, fadeT);
        delogo(V + offUV + tuvoff, wUV, hUV / 2, wUV * 2, pitchUV * 2, maxv, L.aV + tuvoffl, L.bV + tuvoffl, fadeT);
        delogo(U + offUV + buvoff, wUV, hUV / 2, wUV * 2, pitchUV * 2, maxv, L.aU + buvoffl, L.bU + buvoffl, fadeB);
        delogo(V + offUV + buvoff, wUV, hUV / 2, wUV * 2, pitchUV * 2, maxv, L.aV + buvoffl, L.bV + buvoffl, fadeB);
    }
}

// LogoScan.hpp:1543-1568 (element pitch; see amt_oracle.h)
template <typename T>
void logoframe_scan(OrcLogo* const* logos, int nlogos, const uint8_t* base, int64_t frame_stride, int pitch,
                    float maxv, int vi_w, int vi_h, int nframes, float* out)
{
    int maxY = 0;
    for (int i = 0; i < nlogos; ++i)
        if (logos[i]) maxY = std::max(maxY, logos[i]->w * logos[i]->h);
    std::vector<float> memDeint(maxY + 8), memWork(maxY + 8);
    for (int n = 0; n < nframes; ++n) {
        const T* srcY = reinterpret_cast<const T*>(base + n * frame_stride);
        for (int i = 0; i < nlogos; ++i) {
            float* r = out + ((size_t)n * nlogos + i) * 2;
            const OrcLogo* L = logos[i];
            if (L == nullptr || L->imgw != vi_w || L->imgh != vi_h) { r[0] = 0; r[1] = -1; continue; }
            int off = L->imgx + L->imgy * pitch;
            deint_y(memDeint.data(), srcY + off, pitch, L->w, L->h);
            r[0] = evaluate_logo(*L, memDeint.data(), maxv, 0, memWork.data(), -1);
            r[1] = evaluate_logo(*L, memDeint.data(), maxv, 1, memWork.data(), -1);
        }
    }
}

// LogoScan.hpp:1119-1161, per source frame
template <typename T>
void analyze_frames(const OrcLogo& D, const OrcLogo& FT, const OrcLogo& FB, const uint8_t* base,
                    int64_t frame_stride, int pitch, float maxv, int nframes, float* out)
{
    const int w = D.w, h = D.h;
    size_t YSize = (size_t)w * h;
    std::vector<float> memCopy(YSize + 8), memDeint(YSize + 8), memWork(YSize + 8);
    for (int n = 0; n < nframes; ++n) {
        const T* srcY = reinterpret_cast<const T*>(base + n * frame_stride);
        int off = D.imgx + D.imgy * pitch;
        copy_y(memCopy.data(), srcY + off, pitch, w, h);
        deint_y(memDeint.data(), srcY + off, pitch, w, h);
        float* o = out + (size_t)n * 33;
        for (int f = 0; f <= 10; ++f) {
            o[f] = std::abs(evaluate_logo(D, memDeint.data(), maxv, (float)f / 10.0f, memWork.data(), -1));
            o[11 + f] = std::abs(evaluate_logo(FT, memCopy.data(), maxv, (float)f / 10.0f, memWork.data(), w * 2));
            o[22 + f] = std::abs(evaluate_logo(FB, memCopy.data() + w, maxv, (float)f / 10.0f, memWork.data(), w * 2));
        }
    }
}

// LogoScan.hpp:336-342
void approxim_line(int n, double sx, double sy, double sx2, double sxy, double& a, double& b)
{
    double t = (double)n * sx2 - sx * sx;
    a = ((double)n * sxy - sx * sy) / t;
    b = (sx2 * sy - sx * sxy) / t;
}

// LogoScan.hpp:344-396
struct LogoColor {
    double sumF = 0, sumB = 0, sumF2 = 0, sumB2 = 0, sumFB = 0;
    void Add(int f, int b) { sumF += f; sumB += b; sumF2 += f * f; sumB2 += b * b; sumFB += f * b; }
    void Normalize(int maxv)
    {
        sumF /= (double)maxv; sumB /= (double)maxv;
        sumF2 /= (double)maxv * maxv; sumB2 /= (double)maxv * maxv; sumFB /= (double)maxv * maxv;
    }
    bool GetAB(float& A, float& B, int n) const
    {
        double A1, A2, B1, B2;
        approxim_line(n, sumF, sumB, sumF2, sumFB, A1, B1);
        approxim_line(n, sumB, sumF, sumB2, sumFB, A2, B2);
        A = (float)((A1 + (1 / A2)) / 2);
        B = (float)((B1 + (-B2 / A2)) / 2);
        if (std::isnan(A) || std::isnan(B) || std::isinf(A) || std::isinf(B) || A == 0) return false;
        return true;
    }
};

float calc_dist(float a, float b) { return (1.0f / 3.0f) * (a - 1) * (a - 1) + (a - 1) * b + b * b; }  // :430-432

// reverse find: largest i in (lo, from] with pred(i-1) -> returns i (the .base() of the reverse iterator), lo if none
template <typename P> int rfind_base(int from, int lo, P pred)
{
    int i = from;
    while (i > lo && !pred(i - 1)) --i;
    return i;
}

} // namespace

// LogoScan (LogoScan.hpp:398-660)
struct OrcScan {
    int scanw, scanh, logUVx, logUVy, thy;
    int nframes = 0;
    std::vector<LogoColor> logoY, logoU, logoV;
    std::vector<short> tmpY, tmpU, tmpV;

    // :414-428
    static int med_average(const std::vector<short>& s)
    {
        double t = 0;
        int nn = 0;
        int n = (int)s.size();
        for (int i = n / 4; i < n - (n / 4); i++, nn++) t += s[i];
        t = (t + nn / 2) / nn;
        return (int)t;
    }

    // :594-659 and :568-592
    template <typename T> bool AddFrame(const T* Y, const T* U, const T* V, int pitchY, int pitchUV)
    {
        int uvw = scanw >> logUVx, uvh = scanh >> logUVy;
        tmpY.clear(); tmpU.clear(); tmpV.clear();
        for (int x = 0; x < scanw; ++x) { tmpY.push_back(Y[x]); tmpY.push_back(Y[x + (scanh - 1) * pitchY]); }
        for (int y = 1; y < scanh - 1; ++y) { tmpY.push_back(Y[y * pitchY]); tmpY.push_back(Y[scanw - 1 + y * pitchY]); }
        for (int x = 0; x < uvw; ++x) {
            tmpU.push_back(U[x]); tmpU.push_back(U[x + (uvh - 1) * pitchUV]);
            tmpV.push_back(V[x]); tmpV.push_back(V[x + (uvh - 1) * pitchUV]);
        }
        for (int y = 1; y < uvh - 1; ++y) {
            tmpU.push_back(U[y * pitchUV]); tmpU.push_back(U[uvw - 1 + y * pitchUV]);
            tmpV.push_back(V[y * pitchUV]); tmpV.push_back(V[uvw - 1 + y * pitchUV]);
        }
        std::sort(tmpY.begin(), tmpY.end());
        if (abs(tmpY.front() - tmpY.back()) > thy) return false;
        std::sort(tmpU.begin(), tmpU.end());
        if (abs(tmpU.front() - tmpU.back()) > thy) return false;
        std::sort(tmpV.begin(), tmpV.end());
        if (abs(tmpV.front() - tmpV.back()) > thy) return false;
        int bgY = med_average(tmpY), bgU = med_average(tmpU), bgV = med_average(tmpV);
        for (int y = 0; y < scanh; ++y)
            for (int x = 0; x < scanw; ++x) logoY[x + y * scanw].Add(Y[x + y * pitchY], bgY);
        for (int y = 0; y < uvh; ++y)
            for (int x = 0; x < uvw; ++x) {
                logoU[x + y * uvw].Add(U[x + y * pitchUV], bgU);
                logoV[x + y * uvw].Add(V[x + y * pitchUV], bgV);
            }
        ++nframes;
        return true;
    }
};

namespace {

// Normalize(maxv) :471-488 then GetLogo(clean) :490-566, on a copy of the sums
OrcLogo* scan_get_logo(const OrcScan& S, int maxv, bool clean, int imgw, int imgh, int imgx, int imgy)
{
    int uvw = S.scanw >> S.logUVx, uvh = S.scanh >> S.logUVy;
    std::vector<LogoColor> cy = S.logoY, cu = S.logoU, cv = S.logoV;
    for (auto& c : cy) c.Normalize(maxv);
    for (int i = 0; i < uvw * uvh; ++i) { cu[i].Normalize(maxv); cv[i].Normalize(maxv); }
    std::unique_ptr<OrcLogo> L(new OrcLogo);
    L->alloc(S.scanw, S.scanh, S.logUVx, S.logUVy);
    L->imgw = imgw; L->imgh = imgh; L->imgx = imgx; L->imgy = imgy;
    for (int i = 0; i < S.scanw * S.scanh; ++i)
        if (!cy[i].GetAB(L->aY[i], L->bY[i], S.nframes)) return nullptr;
    for (int i = 0; i < uvw * uvh; ++i) {
        if (!cu[i].GetAB(L->aU[i], L->bU[i], S.nframes)) return nullptr;
        if (!cv[i].GetAB(L->aV[i], L->bV[i], S.nframes)) return nullptr;
    }
    if (clean) {
        // dist*1000 for every pixel first (:527-540); the three maxfilter() calls (:543-546) only
        // write their scratch buffer, never dist -> no-op; then threshold (:549-562)
        std::vector<float> dist((size_t)S.scanw * S.scanh);
        for (int y = 0; y < S.scanh; ++y)
            for (int x = 0; x < S.scanw; ++x) {
                int off = x + y * S.scanw;
                int offUV = (x >> S.logUVx) + (y >> S.logUVy) * uvw;
                dist[off] = calc_dist(L->aY[off], L->bY[off]) + calc_dist(L->aU[offUV], L->bU[offUV]) +
                            calc_dist(L->aV[offUV], L->bV[offUV]);
                dist[off] *= 1000;
            }
        for (int y = 0; y < S.scanh; ++y)
            for (int x = 0; x < S.scanw; ++x) {
                int off = x + y * S.scanw;
                int offUV = (x >> S.logUVx) + (y >> S.logUVy) * uvw;
                if (dist[off] < 0.3f) {
                    L->aY[off] = 1; L->bY[off] = 0;
                    L->aU[offUV] = 1; L->bU[offUV] = 0;
                    L->aV[offUV] = 1; L->bV[offUV] = 0;
                }
            }
    }
    return L.release();
}

// ---- AMTLogo.hpp:58-167 (AviUtl base section) ----
void ToYC48Y(float& y) { y = float(((int(y * 255) * 1197) >> 6) - 299); }
void ToYC48C(float& u) { u = float(((int(u * 255) - 128) * 4681 + 164) >> 8); }
void ToYV12Y(float& y) { y = float(((((int)y * 219 + 383) >> 12) + 16) / 255.0f); }
void ToYV12C(float& u) { u = float((((((int)u + 2048) * 7 + 66) >> 7) + 16) / 255.0f); }
void ToYC48ABY(float& A, float& B)
{
    float x0 = 0, x1 = 2048;
    ToYV12Y(x0); ToYV12Y(x1);
    float y0 = (x0 - B) / A, y1 = (x1 - B) / A;
    ToYC48Y(y0); ToYC48Y(y1);
    B = y0;
    A = (y1 - y0) / 2048.0f;
}
void ToYC48ABC(float& A, float& B)
{
    float x0 = 0, x1 = 2048;
    ToYV12C(x0); ToYV12C(x1);
    float y0 = (x0 - B) / A, y1 = (x1 - B) / A;
    ToYC48C(y0); ToYC48C(y1);
    B = y0;
    A = (y1 - y0) / 2048.0f;
}
void to_lgp_pair(float A, float B, bool luma, short& col, short& dp)
{
    if (luma) ToYC48ABY(A, B); else ToYC48ABC(A, B);
    if (A == 1) { col = dp = 0; return; }
    float temp = B / (1 - A) + 0.5f;
    if (std::abs(temp) < 0x7FFF) {
        col = (short)temp;
        temp = (1 - A) * 1000 + 0.5f;
        if (std::abs(temp) > 0x3FFF || short(temp) == 0) col = dp = 0;
        else dp = (short)temp;
    } else col = dp = 0;
}

#pragma pack(push, 1)
struct LgdFileHeader { char str[28]; uint8_t logonum[4]; };              // include/logo.h:39-45 (LLP64)
struct LgdLogoHeader { char name[32]; int16_t x, y, h, w, fi, fo, st, ed; }; // include/logo.h:59-65
struct LgdPixel { int16_t dp_y, y, dp_cb, cb, dp_cr, cr; };              // include/logo.h:71-78
struct LgdExtHeader {                                                     // AMTLogo.hpp:19-28
    int32_t magic, version, w, h, logUVx, logUVy, imgw, imgh, imgx, imgy;
    char name[255]; char pad_; int32_t serviceId; int32_t reserved[60];
};
#pragma pack(pop)
static_assert(sizeof(LgdFileHeader) == 32 && sizeof(LgdLogoHeader) == 48 && sizeof(LgdPixel) == 12 &&
              sizeof(LgdExtHeader) == 540, "lgd layout");

} // namespace

extern "C" {

float orc_corr5x5_scalar(const float* k, const float* Y, int x, int y, int w, float* pavg) { return corr_scalar(k, Y, x, y, w, pavg); }
float orc_corr5x5_avx(const float* k, const float* Y, int x, int y, int w, float* pavg) { return corr_avx(k, Y, x, y, w, pavg); }

OrcLogo* orc_logo_create(int w, int h, int lx, int ly, int imgw, int imgh, int imgx, int imgy, const float* data)
{
    OrcLogo* L = new OrcLogo;
    L->alloc(w, h, lx, ly);
    L->imgw = imgw; L->imgh = imgh; L->imgx = imgx; L->imgy = imgy;
    if (data) std::memcpy(L->data.data(), data, L->data.size() * sizeof(float));
    return L;
}

OrcLogo* orc_logo_load(const char* path)
{
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return nullptr;
    LgdFileHeader fh; LgdLogoHeader lh; LgdExtHeader eh;
    OrcLogo* L = nullptr;
    if (std::fread(&fh, sizeof fh, 1, fp) == 1 && std::fread(&lh, sizeof lh, 1, fp) == 1 &&
        std::fseek(fp, (long)lh.h * lh.w * (long)sizeof(LgdPixel), SEEK_CUR) == 0 &&
        std::fread(&eh, sizeof eh, 1, fp) == 1) {
        L = orc_logo_create(eh.w, eh.h, eh.logUVx, eh.logUVy, eh.imgw, eh.imgh, eh.imgx, eh.imgy, nullptr);
        if (std::fread(L->data.data(), sizeof(float), L->data.size(), fp) != L->data.size()) { delete L; L = nullptr; }
    }
    std::fclose(fp);
    return L;
}

int orc_logo_save(const OrcLogo* L, const char* path, const char* name, int serviceId)
{
    FILE* fp = std::fopen(path, "wb");
    if (!fp) return 0;
    int wUV = L->w >> L->logUVx;
    std::vector<LgdPixel> base((size_t)L->w * L->h);
    for (int y = 0; y < L->h; ++y)
        for (int x = 0; x < L->w; ++x) {
            int off = x + y * L->w, offUV = (x >> L->logUVx) + (y >> L->logUVy) * wUV;
            LgdPixel& p = base[off];
            to_lgp_pair(L->aY[off], L->bY[off], true, p.y, p.dp_y);
            to_lgp_pair(L->aU[offUV], L->bU[offUV], false, p.cb, p.dp_cb);
            to_lgp_pair(L->aV[offUV], L->bV[offUV], false, p.cr, p.dp_cr);
        }
    LgdFileHeader fh; std::memset(&fh, 0, sizeof fh);
    std::strcpy(fh.str, "<logo data file ver0.1>");
    fh.logonum[3] = 1;                                       // SWAP_ENDIAN(1), AMTLogo.hpp:173
    LgdLogoHeader lh; std::memset(&lh, 0, sizeof lh);
    std::strncpy(lh.name, name, sizeof(lh.name) - 1);
    lh.x = (int16_t)L->imgx; lh.y = (int16_t)L->imgy; lh.w = (int16_t)L->w; lh.h = (int16_t)L->h;
    LgdExtHeader eh; std::memset(&eh, 0, sizeof eh);
    eh.magic = 0x12345; eh.version = 1; eh.w = L->w; eh.h = L->h; eh.logUVx = L->logUVx; eh.logUVy = L->logUVy;
    eh.imgw = L->imgw; eh.imgh = L->imgh; eh.imgx = L->imgx; eh.imgy = L->imgy;
    std::strncpy(eh.name, name, 31);                         // sizeof(std::string)-1 on MSVC x64 release, AMTLogo.hpp:45
    eh.serviceId = serviceId;
    bool ok = std::fwrite(&fh, sizeof fh, 1, fp) == 1 && std::fwrite(&lh, sizeof lh, 1, fp) == 1 &&
              std::fwrite(base.data(), sizeof(LgdPixel), base.size(), fp) == base.size() &&
              std::fwrite(&eh, sizeof eh, 1, fp) == 1 &&
              std::fwrite(L->data.data(), sizeof(float), L->data.size(), fp) == L->data.size();
    std::fclose(fp);
    return ok ? 1 : 0;
}

void orc_logo_free(OrcLogo* l) { delete l; }

OrcLogo* orc_logo_deint(const OrcLogo* S)
{
    OrcLogo* D = orc_logo_create(S->w, S->h, S->logUVx, S->logUVy, S->imgw, S->imgh, S->imgx, S->imgy, nullptr);
    const int w = S->w, h = S->h;
    auto merge = [](float a, float b, float c) { return (a + 2 * b + c) / 4.0f; };
    for (int x = 0; x < w; ++x) {
        D->aY[x] = S->aY[x]; D->bY[x] = S->bY[x];
        D->aY[x + (h - 1) * w] = S->aY[x + (h - 1) * w];
        D->bY[x + (h - 1) * w] = S->bY[x + (h - 1) * w];
    }
    for (int y = 1; y < h - 1; ++y)
        for (int x = 0; x < w; ++x) {
            D->aY[x + y * w] = merge(S->aY[x + (y - 1) * w], S->aY[x + y * w], S->aY[x + (y + 1) * w]);
            D->bY[x + y * w] = merge(S->bY[x + (y - 1) * w], S->bY[x + y * w], S->bY[x + (y + 1) * w]);
        }
    return D;
}

OrcLogo* orc_logo_field(const OrcLogo* S, int bottom)
{
    OrcLogo* F = orc_logo_create(S->w, S->h / 2, S->logUVx, S->logUVy, S->imgw, S->imgh / 2, S->imgx, S->imgy / 2, nullptr);
    const int w = S->w;
    for (int y = 0; y < F->h; ++y)
        for (int x = 0; x < F->w; ++x) {
            F->aY[x + y * w] = S->aY[x + (bottom + y * 2) * w];
            F->bY[x + y * w] = S->bY[x + (bottom + y * 2) * w];
        }
    int UVoffset = ((int)(bottom != 0) ^ (F->imgy % 2));
    int wUV = F->w >> S->logUVx, hUV = F->h >> S->logUVy;
    for (int y = 0; y < hUV; ++y)
        for (int x = 0; x < wUV; ++x) {
            F->aU[x + y * wUV] = S->aU[x + (UVoffset + y * 2) * wUV];
            F->bU[x + y * wUV] = S->bU[x + (UVoffset + y * 2) * wUV];
            F->aV[x + y * wUV] = S->aV[x + (UVoffset + y * 2) * wUV];
            F->bV[x + y * wUV] = S->bV[x + (UVoffset + y * 2) * wUV];
        }
    return F;
}

void orc_logo_info(const OrcLogo* l, int* o)
{
    o[0] = l->w; o[1] = l->h; o[2] = l->logUVx; o[3] = l->logUVy; o[4] = l->imgw; o[5] = l->imgh;
    o[6] = l->imgx; o[7] = l->imgy; o[8] = l->maskpixels; o[9] = l->count;
}
const float* orc_logo_data(const OrcLogo* l) { return l->data.data(); }
void orc_logo_create_mask(OrcLogo* l, float maskratio, int use_avx)
{
    l->corr = use_avx ? corr_avx : corr_scalar;
    create_logo_mask(*l, maskratio);
}
const uint8_t* orc_logo_mask(const OrcLogo* l) { return l->mask.data(); }
const float* orc_logo_kernels(const OrcLogo* l) { return l->kernels.data(); }
const float* orc_logo_scales(const OrcLogo* l) { return reinterpret_cast<const float*>(l->scales.data()); }
float orc_logo_black_score(const OrcLogo* l) { return l->blackScore; }
float orc_evaluate_logo(const OrcLogo* l, const float* src, float maxv, float fade, float* work, int stride)
{
    return evaluate_logo(*l, src, maxv, fade, work, stride);
}

void orc_deint_y_u8(float* d, const uint8_t* s, int p, int w, int h) { deint_y(d, s, p, w, h); }
void orc_deint_y_u16(float* d, const uint16_t* s, int p, int w, int h) { deint_y(d, s, p, w, h); }
void orc_copy_y_u8(float* d, const uint8_t* s, int p, int w, int h) { copy_y(d, s, p, w, h); }
void orc_copy_y_u16(float* d, const uint16_t* s, int p, int w, int h) { copy_y(d, s, p, w, h); }

void orc_logoframe_scan(OrcLogo* const* logos, int nlogos, const void* planeY, int64_t frame_stride, int pitch,
                        int bits, int vi_w, int vi_h, int nframes, float* out)
{
    float maxv = (float)((1 << bits) - 1);                  // LogoScan.hpp:1575
    const uint8_t* base = static_cast<const uint8_t*>(planeY);
    if (bits <= 8) logoframe_scan<uint8_t>(logos, nlogos, base, frame_stride, pitch, maxv, vi_w, vi_h, nframes, out);
    else logoframe_scan<uint16_t>(logos, nlogos, base, frame_stride, pitch, maxv, vi_w, vi_h, nframes, out);
}

// LogoScan.hpp:1647-1682
void orc_logoframe_select(const float* evals, int numFrames, int numLogos, int numCandidates, int* bestLogo, float* logoRatio)
{
    const float THRESH = 0.2f;
    if (numCandidates < 0) numCandidates = numLogos;
    struct Summary { float cost; int numFrames; };
    std::vector<Summary> sum(numCandidates, Summary{0.0f, 0});
    for (int n = 0; n < numFrames; ++n)
        for (int i = 0; i < numCandidates; ++i) {
            float c0 = evals[((size_t)n * numLogos + i) * 2], c1 = evals[((size_t)n * numLogos + i) * 2 + 1];
            if (c0 > THRESH && std::abs(c1) < THRESH) { sum[i].numFrames++; sum[i].cost += std::abs(c1); }
        }
    std::vector<float> score(numCandidates);
    for (int i = 0; i < numCandidates; ++i) {
        auto& s = sum[i];
        score[i] = (s.numFrames == 0) ? INFINITY : (s.cost / s.numFrames) * (numFrames / (float)s.numFrames);
    }
    *bestLogo = (int)(std::min_element(score.begin(), score.end()) - score.begin());
    *logoRatio = (float)sum[*bestLogo].numFrames / numFrames;
}

// LogoScan.hpp:1686-1827
int orc_logoframe_write_result(const float* evals, int numFrames, int numLogos, int logoIndex,
                               int fps_num, int fps_den, char* out, int cap)
{
    const float THRESH = 0.2f;
    const float threshL = 0.5f;
    const float avgDur = 1.0f, medianDur = 0.5f;
    int framesPerSec = (int)std::round((float)fps_num / fps_den);   // :1586
    int halfAvgFrames = int(framesPerSec * avgDur / 2 + 0.5f);
    int aveFrames = halfAvgFrames * 2 + 1;
    int halfMedianFrames = int(framesPerSec * medianDur / 2 + 0.5f);
    int medianFrames = halfMedianFrames * 2 + 1;
    int winFrames = std::max(aveFrames, medianFrames);
    int halfWinFrames = winFrames / 2;
    std::vector<float> raw_(numFrames + winFrames);
    float* raw = raw_.data() + halfWinFrames;
    for (int n = 0; n < numFrames; ++n) {
        float c0 = evals[((size_t)n * numLogos + logoIndex) * 2], c1 = evals[((size_t)n * numLogos + logoIndex) * 2 + 1];
        raw[n] = std::max(0.0f, c0) + std::min(0.0f, c1);
    }
    std::fill(raw_.data(), raw, raw[0]);
    std::fill(raw + numFrames, raw_.data() + raw_.size(), raw[numFrames - 1]);

    std::vector<int> result(numFrames);
    std::vector<float> score(numFrames);
    std::vector<float> medianBuf(medianFrames);
    for (int i = 0; i < numFrames; ++i) {
        float beforeMax = *std::max_element(raw + i - halfAvgFrames, raw + i);
        float afterMax = *std::max_element(raw + i + 1, raw + i + 1 + halfAvgFrames);
        float minMax = std::min(beforeMax, afterMax);
        int minMaxResult = (std::abs(minMax) < threshL) ? 1 : (minMax < 0.0f) ? 0 : 2;
        float avg = std::accumulate(raw + i - halfAvgFrames, raw + i + halfAvgFrames + 1, 0.0f) / aveFrames;
        int avgResult = (std::abs(avg) < THRESH) ? 1 : (avg < 0.0f) ? 0 : 2;
        result[i] = (minMaxResult != avgResult) ? 1 : minMaxResult;
        std::copy(raw + i - halfMedianFrames, raw + i + halfMedianFrames + 1, medianBuf.begin());
        // NaN evidence (corr0 = +inf, corr1 = -inf) makes the reference's plain std::sort undefined behaviour (:1746); the checker
        // fixes the order -- NaN after every number -- which is the reference's own result whenever the window holds no NaN
        std::sort(medianBuf.begin(), medianBuf.end(), [](float a, float b) { return a < b || (b != b && a == a); });
        score[i] = medianBuf[halfMedianFrames];
    }
    const int N = numFrames;
    auto find_from = [&](int from, auto pred) { int i = from; while (i < N && !pred(i)) ++i; return i; };
    // unknown runs take their neighbours' value when both sides agree (:1754-1765)
    for (int it = 0; it != N;) {
        int first1 = find_from(it, [&](int i) { return result[i] == 1; });
        it = find_from(first1, [&](int i) { return result[i] != 1; });
        int prev = (first1 == 0) ? 0 : result[first1 - 1];
        int next = (it == N) ? 0 : result[it];
        if (prev == next) for (int i = first1; i < it; ++i) result[i] = prev;
    }
    std::string sb;
    char line[128];
    for (int it = 0; it != N;) {
        int sEnd_ = find_from(it, [&](int i) { return result[i] == 2; });
        int eEnd_ = find_from(sEnd_, [&](int i) { return result[i] == 0; });
        int sEnd = sEnd_, eEnd = eEnd_;
        if (sEnd != N) {
            if (score[sEnd] >= THRESH) sEnd = rfind_base(sEnd, 0, [&](int i) { return score[i] < THRESH; });
            else sEnd = find_from(sEnd, [&](int i) { return score[i] >= THRESH; });
        }
        if (eEnd != N) {
            if (score[eEnd] <= -THRESH) eEnd = rfind_base(eEnd, sEnd, [&](int i) { return score[i] > -THRESH; });
            else eEnd = find_from(eEnd, [&](int i) { return score[i] <= -THRESH; });
        }
        int sStart = rfind_base(sEnd, it, [&](int i) { return score[i] <= -THRESH; });
        int eStart = rfind_base(eEnd, sEnd, [&](int i) { return score[i] >= THRESH; });
        int sBest = sStart; while (sBest < sEnd && !(score[sBest] > 0)) ++sBest;
        int eBest = rfind_base(eEnd, eStart, [&](int i) { return score[i] > 0; });
        if (sEnd != eEnd) {
            std::snprintf(line, sizeof line, "%6d S 0 ALL %6d %6d\n", sBest, sStart, sEnd); sb += line;
            std::snprintf(line, sizeof line, "%6d E 0 ALL %6d %6d\n", eBest - 1, eStart - 1, eEnd - 1); sb += line;
        }
        it = eEnd_;
    }
    if ((int)sb.size() + 1 > cap) return -1;
    std::memcpy(out, sb.c_str(), sb.size() + 1);
    return (int)sb.size();
}

void orc_analyze_frames(const OrcLogo* D, const OrcLogo* FT, const OrcLogo* FB, const void* planeY,
                        int64_t frame_stride, int pitch, int bits, int nframes, float* out)
{
    float maxv = (float)((1 << bits) - 1);                  // LogoScan.hpp:1130
    const uint8_t* base = static_cast<const uint8_t*>(planeY);
    if (bits <= 8) analyze_frames<uint8_t>(*D, *FT, *FB, base, frame_stride, pitch, maxv, nframes, out);
    else analyze_frames<uint16_t>(*D, *FT, *FB, base, frame_stride, pitch, maxv, nframes, out);
}

void orc_delogo_u8(uint8_t* d, int w, int h, int lp, int ip, float maxv, const float* A, const float* B, float fade) { delogo(d, w, h, lp, ip, maxv, A, B, fade); }
void orc_delogo_u16(uint16_t* d, int w, int h, int lp, int ip, float maxv, const float* A, const float* B, float fade) { delogo(d, w, h, lp, ip, maxv, A, B, fade); }

// LogoScan.hpp:1263-1315.  analyzeclip->GetFrame(k>>3) is assumed to clamp its frame number into
// the analyze clip (AviSynth's cache does); slot k&7 of analysis frame q describes source
// frame clamp(q*8+slot) (:1133).
void orc_calc_fade2(const float* analysis, int num_frames, int n, float* fadeT, float* fadeB)
{
    enum { DIST = 4 };
    const float* fr[DIST * 2 + 1];
    int nanalyze = (num_frames + 7) / 8;                    // :1200
    for (int i = -DIST; i <= DIST; ++i) {
        int nsrc = std::max(0, std::min(num_frames - 1, n + i));
        int analyze_n = (nsrc + i) >> 3;
        int idx = (nsrc + i) & 7;
        analyze_n = std::max(0, std::min(nanalyze - 1, analyze_n));
        int src = std::max(0, std::min(num_frames - 1, analyze_n * 8 + idx));
        fr[i + DIST] = analysis + (size_t)src * 33;
    }
    int minfades[DIST * 2 + 1];
    for (int i = 0; i < DIST * 2 + 1; ++i) minfades[i] = (int)(std::min_element(fr[i], fr[i] + 11) - fr[i]);
    int minT = (int)(std::min_element(fr[DIST] + 11, fr[DIST] + 22) - (fr[DIST] + 11));
    int minB = (int)(std::min_element(fr[DIST] + 22, fr[DIST] + 33) - (fr[DIST] + 22));
    float before_fades = 0, after_fades = 0;
    for (int i = 1; i <= 4; ++i) { before_fades += minfades[DIST - i]; after_fades += minfades[DIST + i]; }
    before_fades /= 4 * 10;
    after_fades /= 4 * 10;
    if ((before_fades < 0.3 && after_fades > 0.7) || (before_fades > 0.7 && after_fades < 0.3)) {
        *fadeT = minT / 10.0f;
        *fadeB = minB / 10.0f;
    } else {
        *fadeT = *fadeB = (minfades[DIST] / 10.0f);
    }
}

// LogoScan.hpp:1317-1341
void orc_calc_fade(const int* frameResult, int has_result, int maxFadeLength, const float* analysis,
                   int num_frames, int n, float* fadeT, float* fadeB)
{
    if (!has_result) { orc_calc_fade2(analysis, num_frames, n, fadeT, fadeB); return; }
    int halfWidth = (maxFadeLength >> 1);
    std::vector<int> fr(halfWidth * 2 + 1);
    for (int i = -halfWidth; i <= halfWidth; ++i) {
        int nsrc = std::max(0, std::min(num_frames - 1, n + i));
        fr[i + halfWidth] = frameResult[nsrc];
    }
    if (std::all_of(fr.begin(), fr.end(), [&](int p) { return p == fr[0]; }))
        *fadeT = *fadeB = ((fr[halfWidth] == 2) ? 1.0f : 0.0f);
    else
        orc_calc_fade2(analysis, num_frames, n, fadeT, fadeB);
}

// LogoScan.hpp:1421-1461
int orc_read_logoframe(const char* text, int num_frames, int* frameResult)
{
    struct Elem { bool isStart; int best, start, end; };
    std::vector<Elem> el;
    std::regex re("^\\s*(\\d+)\\s+(\\S)\\s+(\\d+)\\s+(\\S+)\\s+(\\d+)\\s+(\\d+)");
    std::string all(text);
    size_t pos = 0;
    while (pos < all.size()) {
        size_t nl = all.find('\n', pos);
        std::string line = all.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = (nl == std::string::npos) ? all.size() : nl + 1;
        std::smatch m;
        if (std::regex_search(line, m, re))
            el.push_back(Elem{std::tolower(m[2].str()[0]) == 's', std::stoi(m[1].str()), std::stoi(m[5].str()), std::stoi(m[6].str())});
    }
    std::fill(frameResult, frameResult + num_frames, 0);
    if (el.size() % 2) return -1;
    for (size_t i = 0; i < el.size(); i += 2) {
        if (el[i].isStart == false || el[i + 1].isStart) return -1;
        std::fill(frameResult + std::min(num_frames, el[i].start), frameResult + std::min(num_frames, el[i].end + 1), 1);
        std::fill(frameResult + std::min(num_frames, el[i].end), frameResult + std::min(num_frames, el[i + 1].start + 1), 2);
        std::fill(frameResult + std::min(num_frames, el[i + 1].start + 1), frameResult + std::min(num_frames, el[i + 1].end + 1), 1);
    }
    return 0;
}

void orc_erase_frame(const OrcLogo* logo, void* Y, void* U, void* V, int pitchY, int pitchUV, int bits, float fadeT, float fadeB)
{
    float maxv = (float)((1 << bits) - 1);                  // LogoScan.hpp:1349
    if (bits <= 8) erase_frame(*logo, (uint8_t*)Y, (uint8_t*)U, (uint8_t*)V, pitchY, pitchUV, maxv, fadeT, fadeB);
    else erase_frame(*logo, (uint16_t*)Y, (uint16_t*)U, (uint16_t*)V, pitchY, pitchUV, maxv, fadeT, fadeB);
}

OrcScan* orc_scan_create(int w, int h, int lx, int ly, int thy)
{
    OrcScan* s = new OrcScan;
    s->scanw = w; s->scanh = h; s->logUVx = lx; s->logUVy = ly; s->thy = thy;
    s->logoY.resize((size_t)w * h);
    s->logoU.resize((size_t)(w * h) >> (lx + ly));
    s->logoV.resize((size_t)(w * h) >> (lx + ly));
    return s;
}
void orc_scan_free(OrcScan* s) { delete s; }
int orc_scan_add_frame_u8(OrcScan* s, const uint8_t* Y, const uint8_t* U, const uint8_t* V, int pitchY, int pitchUV)
{
    return s->AddFrame(Y, U, V, pitchY, pitchUV) ? 1 : 0;
}
int orc_scan_nframes(const OrcScan* s) { return s->nframes; }
void orc_scan_sums(const OrcScan* s, double* out)
{
    auto dump = [&](const std::vector<LogoColor>& v) {
        for (const auto& c : v) { *out++ = c.sumF; *out++ = c.sumB; *out++ = c.sumF2; *out++ = c.sumB2; *out++ = c.sumFB; }
    };
    dump(s->logoY); dump(s->logoU); dump(s->logoV);
}
OrcLogo* orc_scan_get_logo(OrcScan* s, int maxv, int clean, int imgw, int imgh, int imgx, int imgy)
{
    return scan_get_logo(*s, maxv, clean != 0, imgw, imgh, imgx, imgy);
}

// LogoScan.hpp:794-1080 without the codec/file plumbing: valid crops are kept raw in memory
// The two ReMakeLogo rounds evaluate every kept frame at 20 fades independently of every other frame (:957-984): with threads > 1 the
// frames are dealt over host threads, each with its own memDeint / memWork (evaluate_logo keeps no state).  Everything that is ordered in
// the reference stays ordered here: the quota of the first numMaxFrames valid frames in stream order (:885), both AddFrame accumulations.
static OrcLogo* scanlogo_impl(const uint8_t* Y, const uint8_t* U, const uint8_t* V, int64_t strideY, int64_t strideUV,
                              int pitchY, int pitchUV, int imgw, int imgh, int nframes_total, int scanx, int scany,
                              int scanw, int scanh, int thy, int numMaxFrames, int use_avx, int* num_valid_out, int* minfades_out,
                              int threads, int* frames_read_out)
{
    const int logUVx = 1, logUVy = 1;                       // YUV420 (:866-867)
    const int uvw = scanw >> logUVx, uvh = scanh >> logUVy;
    const size_t ysz = (size_t)scanw * scanh, csz = (size_t)uvw * uvh, fsz = ysz + 2 * csz;
    std::vector<uint8_t> crops;
    int numFrames = 0, nread = 0;
    std::unique_ptr<OrcScan> scan(orc_scan_create(scanw, scanh, logUVx, logUVy, thy));
    // MakeInitialLogo :917-921 / onFrame :881-914
    for (int n = 0; n < nframes_total; ++n) {
        if (numFrames >= numMaxFrames) break;
        ++nread;
        int offY = scanx + scany * pitchY;
        int offUV = (scanx >> logUVx) + (scany >> logUVy) * pitchUV;
        const uint8_t* sy = Y + n * strideY + offY;
        const uint8_t* su = U + n * strideUV + offUV;
        const uint8_t* sv = V + n * strideUV + offUV;
        if (scan->AddFrame(sy, su, sv, pitchY, pitchUV)) {
            ++numFrames;
            size_t base = crops.size();
            crops.resize(base + fsz);
            uint8_t* d = &crops[base];
            for (int y = 0; y < scanh; ++y) std::memcpy(d + (size_t)y * scanw, sy + (size_t)y * pitchY, scanw);
            for (int y = 0; y < uvh; ++y) {
                std::memcpy(d + ysz + (size_t)y * uvw, su + (size_t)y * pitchUV, uvw);
                std::memcpy(d + ysz + csz + (size_t)y * uvw, sv + (size_t)y * pitchUV, uvw);
            }
        }
    }
    if (num_valid_out) *num_valid_out = numFrames;
    if (frames_read_out) *frames_read_out = nread;
    std::unique_ptr<OrcLogo> logodata(scan_get_logo(*scan, 255, false, imgw, imgh, scanx, scany));  // :845-846
    if (!logodata) return nullptr;
    std::vector<int> minFades(numFrames);
    const int T = std::max(1, std::min(threads, numFrames));
    for (int round = 0; round < 2; ++round) {               // ReMakeLogo x2 (:1067-1069), body :923-1036
        std::unique_ptr<OrcLogo> deint(orc_logo_deint(logodata.get()));
        deint->imgw = scanw; deint->imgh = scanh;
        orc_logo_create_mask(deint.get(), 0.1f, use_avx);
        const int numFade = 20;
        auto eval_frames = [&](int first, int step) {
            std::vector<float> memDeint(ysz + 8), memWork(ysz + 8);
            for (int i = first; i < numFrames; i += step) {
                deint_y(memDeint.data(), &crops[(size_t)i * fsz], scanw, scanw, scanh);
                float minResult = FLT_MAX;
                int minFadeIndex = 0;
                for (int fi = 0; fi < numFade; ++fi) {
                    float fade = 0.1f * fi;
                    float result = std::abs(evaluate_logo(*deint, memDeint.data(), 255.0f, fade, memWork.data(), -1));
                    if (result < minResult) { minResult = result; minFadeIndex = fi; }
                }
                minFades[i] = minFadeIndex;
            }
        };
        if (T == 1) {
            eval_frames(0, 1);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(eval_frames, t, T);
            for (auto& x : th) x.join();
        }
        std::unique_ptr<OrcScan> rescan(orc_scan_create(scanw, scanh, logUVx, logUVy, thy));
        for (int i = 0; i < numFrames; ++i)
            if (minFades[i] > 8) {
                const uint8_t* p = &crops[(size_t)i * fsz];
                rescan->AddFrame(p, p + ysz, p + ysz + csz, scanw, uvw);
            }
        logodata.reset(scan_get_logo(*rescan, 255, true, imgw, imgh, scanx, scany));
        if (!logodata) return nullptr;
    }
    if (minfades_out) std::memcpy(minfades_out, minFades.data(), sizeof(int) * numFrames);
    return logodata.release();
}

OrcLogo* orc_scanlogo(const uint8_t* Y, const uint8_t* U, const uint8_t* V, int64_t strideY, int64_t strideUV,
                      int pitchY, int pitchUV, int imgw, int imgh, int nframes_total, int scanx, int scany,
                      int scanw, int scanh, int thy, int numMaxFrames, int use_avx, int* num_valid_out, int* minfades_out)
{
    return scanlogo_impl(Y, U, V, strideY, strideUV, pitchY, pitchUV, imgw, imgh, nframes_total, scanx, scany, scanw, scanh, thy,
                         numMaxFrames, use_avx, num_valid_out, minfades_out, 1, nullptr);
}

OrcLogo* orc_scanlogo_mt(const uint8_t* Y, const uint8_t* U, const uint8_t* V, int64_t strideY, int64_t strideUV,
                         int pitchY, int pitchUV, int imgw, int imgh, int nframes_total, int scanx, int scany,
                         int scanw, int scanh, int thy, int numMaxFrames, int use_avx, int* num_valid_out, int* minfades_out,
                         int threads, int* frames_read_out)
{
    return scanlogo_impl(Y, U, V, strideY, strideUV, pitchY, pitchUV, imgw, imgh, nframes_total, scanx, scany, scanw, scanh, thy,
                         numMaxFrames, use_avx, num_valid_out, minfades_out, threads, frames_read_out);
}


// self-specified metrics (parity unpinned): see amt_oracle.h
extern "C++" {
template <typename T>
void frame_metrics_t(const uint8_t* base, int64_t frame_stride, int pitch, int W, int H, int nframes,
                            const uint8_t* prev_first, uint64_t* out)
{
    for (int n = 0; n < nframes; ++n) {
        const T* cur = reinterpret_cast<const T*>(base + n * frame_stride);
        const T* prev = n > 0 ? reinterpret_cast<const T*>(base + (n - 1) * frame_stride)
                              : (prev_first ? reinterpret_cast<const T*>(prev_first) : cur);
        uint64_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int y = 0; y < H; ++y) {
            const T* c = cur + (size_t)y * pitch;
            const T* p = prev + (size_t)y * pitch;
            uint64_t d = 0, sum = 0;
            for (int x = 0; x < W; ++x) { d += (uint64_t)std::abs((int)c[x] - (int)p[x]); sum += c[x]; }
            m[y & 1] += d;
            m[5] += sum;
            if (y >= 1 && y <= H - 2) {
                const T *ca = c - pitch, *cc = c + pitch, *pa = p - pitch, *pc = p + pitch;
                uint64_t vert = 0, comb = 0, combp = 0, vertp = 0;
                const bool odd = y & 1;
                for (int x = 0; x < W; ++x) {
                    const int a = ca[x], b = c[x], e = cc[x];
                    vert += (uint64_t)std::abs(a - e);
                    comb += (uint64_t)std::abs(b - ((a + e) >> 1));
                    // weave: even rows from cur, odd rows from prev
                    const int wa = odd ? a : (int)pa[x], wb = odd ? (int)p[x] : b, wc = odd ? e : (int)pc[x];
                    combp += (uint64_t)std::abs(wb - ((wa + wc) >> 1));
                    vertp += (uint64_t)std::abs(wa - wc);
                }
                m[2] += vert; m[3] += comb; m[4] += combp; m[6] += vertp;
            }
        }
        std::memcpy(out + (size_t)n * 8, m, sizeof m);
    }
}
} // extern "C++"

extern "C++" {
// ---- AMTSource::MergeField / Copy1 / Copy2 (AMTSource.hpp:291-355): the frame AMTSource hands to AviSynth takes its
// even rows from the `top` picture and its odd rows from the `bottom` picture (Copy1 :291-302: dst row y <- top row y,
// dst row y+1 <- bottom row y+1), chroma planes likewise (:345-347), NV12 chroma de-interleaved (Copy2 :304-322, srcV =
// srcU + 1 :330).  Restatement only: AMTSource.hpp needs FFmpeg/AviSynth and is not part of oracle/_ref.  Pitches are in
// ELEMENTS here; the reference passes byte pitches into T* arithmetic, which is only right for T = uint8_t.
template <typename T>
static void copy1_t(T* dst, const T* top, const T* bottom, int w, int h, int dpitch, int tpitch, int bpitch)
{
    for (int y = 0; y < h; y += 2) {
        memcpy(dst + (size_t)dpitch * (y + 0), top + (size_t)tpitch * (y + 0), sizeof(T) * w);
        memcpy(dst + (size_t)dpitch * (y + 1), bottom + (size_t)bpitch * (y + 1), sizeof(T) * w);
    }
}
template <typename T>
static void copy2_t(T* dstU, T* dstV, const T* top, const T* bottom, int w, int h, int dpitch, int tpitch, int bpitch)
{
    for (int y = 0; y < h; y += 2) {
        const T* src0 = top + (size_t)tpitch * (y + 0);
        const T* src1 = bottom + (size_t)bpitch * (y + 1);
        for (int x = 0; x < w; ++x) {
            dstU[(size_t)dpitch * (y + 0) + x] = src0[x * 2 + 0];
            dstV[(size_t)dpitch * (y + 0) + x] = src0[x * 2 + 1];
            dstU[(size_t)dpitch * (y + 1) + x] = src1[x * 2 + 0];
            dstV[(size_t)dpitch * (y + 1) + x] = src1[x * 2 + 1];
        }
    }
}
template <typename T>
static void merge_field_t(const T* tY, const T* tU, const T* tV, const T* bY, const T* bU, const T* bV, int spitchY, int spitchUV,
                          int nv12, int W, int H, T* dY, T* dU, T* dV, int pitchY, int pitchUV)
{
    copy1_t<T>(dY, tY, bY, W, H, pitchY, spitchY, spitchY);
    const int wUV = W >> 1, hUV = H >> 1;                 // 4:2:0: log2_chroma_w = log2_chroma_h = 1
    if (!nv12) {
        copy1_t<T>(dU, tU, bU, wUV, hUV, pitchUV, spitchUV, spitchUV);
        copy1_t<T>(dV, tV, bV, wUV, hUV, pitchUV, spitchUV, spitchUV);
    } else {
        copy2_t<T>(dU, dV, tU, bU, wUV, hUV, pitchUV, spitchUV, spitchUV);
    }
}
} // extern "C++"

void orc_merge_field(const void* tY, const void* tU, const void* tV, const void* bY, const void* bU, const void* bV, int spitchY,
                     int spitchUV, int nv12, int bits, int W, int H, void* dY, void* dU, void* dV, int pitchY, int pitchUV)
{
    if (bits <= 8)
        merge_field_t<uint8_t>((const uint8_t*)tY, (const uint8_t*)tU, (const uint8_t*)tV, (const uint8_t*)bY, (const uint8_t*)bU,
                               (const uint8_t*)bV, spitchY, spitchUV, nv12, W, H, (uint8_t*)dY, (uint8_t*)dU, (uint8_t*)dV, pitchY, pitchUV);
    else
        merge_field_t<uint16_t>((const uint16_t*)tY, (const uint16_t*)tU, (const uint16_t*)tV, (const uint16_t*)bY, (const uint16_t*)bU,
                                (const uint16_t*)bV, spitchY, spitchUV, nv12, W, H, (uint16_t*)dY, (uint16_t*)dU, (uint16_t*)dV, pitchY, pitchUV);
}

void orc_frame_metrics(const void* Y, int64_t frame_stride, int pitch, int bits, int W, int H, int nframes,
                       const void* prev_first, uint64_t* out)
{
    if (bits <= 8) frame_metrics_t<uint8_t>((const uint8_t*)Y, frame_stride, pitch, W, H, nframes, (const uint8_t*)prev_first, out);
    else frame_metrics_t<uint16_t>((const uint8_t*)Y, frame_stride, pitch, W, H, nframes, (const uint8_t*)prev_first, out);
}

} // extern "C"
