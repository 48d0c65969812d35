// logo_model.hpp -- host-side logo model: planes, .lgd file format, evaluation tables.
//
// Replaces logo::LogoData / LogoHeader (AMTLogo.hpp:19-280), DeintLogo (LogoScan.hpp:734-761),
// LogoDataParam::MakeFieldLogo (:257-283) and ::CreateLogoMask (:112-229).  Setup-time only (once
// per logo per filter instance), so it stays on the host; the tables are uploaded for the kernels.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace amt {

struct LogoPlanes {
    int w = 0, h = 0, logUVx = 1, logUVy = 1;
    int imgw = 0, imgh = 0, imgx = 0, imgy = 0;
    std::string name;
    int serviceId = 0;
    std::vector<float> data;     // aY,bY,aU,bU,aV,bV  (AMTLogo.hpp:204-212)

    int wUV() const { return w >> logUVx; }
    int hUV() const { return h >> logUVy; }
    size_t ysize() const { return (size_t)w * h; }
    size_t csize() const { return (size_t)wUV() * hUV(); }
    size_t total() const { return (ysize() + 2 * csize()) * 2; }
    void allocate() { data.assign(total(), 0.0f); }
    float* A(int plane) { return data.data() + (plane == 0 ? 0 : 2 * ysize() + (plane - 1) * 2 * csize()); }
    float* B(int plane) { return A(plane) + (plane == 0 ? ysize() : csize()); }
    const float* A(int plane) const { return const_cast<LogoPlanes*>(this)->A(plane); }
    const float* B(int plane) const { return const_cast<LogoPlanes*>(this)->B(plane); }
};

// .lgd reader / writer (AviUtl-compatible base section + extended section); throw std::runtime_error
LogoPlanes load_lgd(const std::string& path);
void save_lgd(const LogoPlanes& logo, const std::string& path, const std::string& name, int serviceId);

LogoPlanes deinterlaced_logo(const LogoPlanes& src);          // vertical [1 2 1]/4 on the Y planes
LogoPlanes field_logo(const LogoPlanes& src, bool bottom);    // every other row; chroma parity by imgy

// what CorrelationScore needs, in raster order of the visited (interior) mask pixels
struct MaskTables {
    int w = 0, h = 0;
    int maskpixels = 0;              // min(w*h, int(w*h*maskratio))
    int count = 0;                   // mask pixels with 2 <= x < w-2, 2 <= y < h-2
    std::vector<uint8_t> mask;       // w*h
    std::vector<uint32_t> pos;       // count: (y << 16) | x
    std::vector<float> kernels;      // count*25 (row-major 5x5, mean removed)
    std::vector<float> scales;       // count*32*{scale, scale2}
    std::vector<float> resp;         // count*32: the SIGNED response of the pixel's kernel on flat level c (scale = 1 / |resp|); the
                                     // composite of a flat level is affine in the level, so resp is P + Q c up to rounding -- what the
                                     // linear analysis kernel evaluates instead of gathering scales (eval_engine.hip ensure_linear)
    float floorResp = 0;             // limitCorr (LogoScan.hpp:203): scale2 = min(1, |resp| / floorResp)
    float blackScore = 0;
};
MaskTables build_mask_tables(const LogoPlanes& evalLogo, float maskratio);

// CPU evaluation of one source plane with the tables -- used at setup for blackScore only
float correlation_score_host(const MaskTables& t, const float* work);

} // namespace amt
