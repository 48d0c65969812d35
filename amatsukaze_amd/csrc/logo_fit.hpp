// logo_fit.hpp -- per-pixel linear regression foreground vs background -> logo planes.
// Replaces LogoColor::Normalize/GetAB (LogoScan.hpp:366-395), approxim_line (:336-342) and
// LogoScan::GetLogo (:490-566).  Host, double precision, once per scan round.
#pragma once

#include <cstdint>
#include <vector>

#include "logo_model.hpp"

namespace amt {

struct ScanSums {
    int w = 0, h = 0, logUVx = 1, logUVy = 1;
    int nframes = 0;
    std::vector<int64_t> px;        // 3 per pixel {sumF, sumF2, sumFB}; Y pixels, then U, then V
    int64_t plane[6] = {0, 0, 0, 0, 0, 0};   // {sumB, sumB2} for Y, U, V
    size_t npixels() const { return (size_t)w * h + 2 * (size_t)(w >> logUVx) * (h >> logUVy); }
};

// false when any pixel's regression degenerates (NaN / Inf / zero slope): "Insufficient logo frames"
bool fit_logo(const ScanSums& s, int maxv, bool clean, LogoPlanes& out);

} // namespace amt
