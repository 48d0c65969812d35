// logo_fit.cpp -- see logo_fit.hpp
#include "build_knobs.h"
#include "logo_fit.hpp"

#include <cmath>

namespace amt {

namespace {

// least-squares line y = a*x + b through n points given their sums
inline void line_fit(int n, double sx, double sy, double sxx, double sxy, double& a, double& b)
{
    const double det = (double)n * sxx - sx * sx;
    a = ((double)n * sxy - sx * sy) / det;
    b = (sxx * sy - sx * sxy) / det;
}

// regress background on foreground and foreground on background, average the two lines
inline bool fit_pixel(double sF, double sB, double sF2, double sB2, double sFB, int n, float& A, float& B)
{
    double a1, b1, a2, b2;
    line_fit(n, sF, sB, sF2, sFB, a1, b1);
    line_fit(n, sB, sF, sB2, sFB, a2, b2);
    A = (float)((a1 + (1 / a2)) / 2);
    B = (float)((b1 + (-b2 / a2)) / 2);
    return !(std::isnan(A) || std::isnan(B) || std::isinf(A) || std::isinf(B) || A == 0);
}

inline float identity_distance(float a, float b) { return (1.0f / 3.0f) * (a - 1) * (a - 1) + (a - 1) * b + b * b; }

} // namespace

bool fit_logo(const ScanSums& s, int maxv, bool clean, LogoPlanes& out)
{
    out = LogoPlanes();
    out.w = s.w; out.h = s.h; out.logUVx = s.logUVx; out.logUVy = s.logUVy;
    out.allocate();
    const size_t ysz = out.ysize(), csz = out.csize();
    const double m1 = (double)maxv, m2 = (double)maxv * maxv;     // samples are normalised to 0..1 first
    for (int pl = 0; pl < 3; ++pl) {
        const size_t n = pl == 0 ? ysz : csz;
        const size_t base = pl == 0 ? 0 : ysz + (size_t)(pl - 1) * csz;
        const double sB = (double)s.plane[pl * 2] / m1, sB2 = (double)s.plane[pl * 2 + 1] / m2;
        float* A = out.A(pl);
        float* B = out.B(pl);
        for (size_t i = 0; i < n; ++i) {
            const int64_t* p = &s.px[(base + i) * 3];
            if (!fit_pixel((double)p[0] / m1, sB, (double)p[1] / m2, sB2, (double)p[2] / m2, s.nframes, A[i], B[i])) return false;
        }
    }
    if (clean) {
        // pixels whose (A,B) is indistinguishable from "no logo" become exactly A=1,B=0, luma and the
        // chroma sample under it together.  Distances are taken before any pixel is rewritten.  (The
        // reference's three maxfilter passes never write back into the distance map.)
        const int cw = out.wUV();
        std::vector<float> dist(ysz);
        for (int y = 0; y < s.h; ++y)
            for (int x = 0; x < s.w; ++x) {
                const size_t o = x + (size_t)y * s.w, oc = (x >> s.logUVx) + (size_t)(y >> s.logUVy) * cw;
                float d = identity_distance(out.A(0)[o], out.B(0)[o]) + identity_distance(out.A(1)[oc], out.B(1)[oc]) +
                          identity_distance(out.A(2)[oc], out.B(2)[oc]);
                d *= 1000;
                dist[o] = d;
            }
        for (int y = 0; y < s.h; ++y)
            for (int x = 0; x < s.w; ++x) {
                const size_t o = x + (size_t)y * s.w, oc = (x >> s.logUVx) + (size_t)(y >> s.logUVy) * cw;
                if (dist[o] < 0.3f) {
                    out.A(0)[o] = 1; out.B(0)[o] = 0;
                    out.A(1)[oc] = 1; out.B(1)[oc] = 0;
                    out.A(2)[oc] = 1; out.B(2)[oc] = 0;
                }
            }
    }
    return true;
}

} // namespace amt
