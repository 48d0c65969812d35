// eval_plan.h -- device-side description of a logo evaluation (shared by host code and kernels).
#pragma once
#include <cstdint>

namespace amt {

constexpr int kTablePad = 256;      // mask-pixel tables are padded to a multiple of this
constexpr int kNumBins = 32;        // 256 >> 3 background levels (LogoScan.hpp:63-68)
// kernel variants: (mask pixels per thread PXT, threads per workgroup NT) -> staged rectangle pixels per thread
// STG (LDS plane floats <= NT*STG); 0 = variant not built
inline constexpr int eval_stage_per_thread(int pxt, int nt)
{
    return (pxt == 1 && nt == 256) ? 8 : (pxt == 1 && nt == 512) ? 8 : (pxt == 1 && nt == 1024) ? 4
         : (pxt == 2 && nt == 256) ? 12 : (pxt == 2 && nt == 512) ? 8 : (pxt == 4 && nt == 256) ? 16 : 0;
}

// one evaluation logo (a LogoDataParam after CreateLogoMask) resident in HBM.  Mask-pixel tables are in raster
// order of the visited pixels (index m).
struct EvalLogoDev {
    const float* a;          // [h*w]   A plane of the evaluation logo (deint or field)
    const float* b;          // [h*w]
    const uint32_t* pos;     // [count_pad]  (y << 16) | x
    const uint32_t* slots;   // [nslots]  run slots: (n << 28) | m0 -- n <= PXT horizontally adjacent mask pixels
                             // m0..m0+n-1 (adjacent in x, hence consecutive in raster order) evaluated by ONE thread
                             // from one shared 5 x (4+PXT) register window
    const float* kern;       // [25][count_pad]   tap-major so lanes read consecutive floats
    const float2* scales;    // [32][count_pad]   bin-major {scale, scale2}
    int w, h;                // evaluation-logo size (field logos: h/2)
    int count, count_pad;
    int imgx, imgy;          // rectangle origin in the full frame (full-frame rows)
    int row0, row_step;      // source row of logo row y = imgy + row0 + y*row_step
    int deint;               // 1: source is the [1 2 1] vertical blend of rows y-1,y,y+1 (DeintY)
    int score_off;           // float offset of this logo's block in a frame's score scratch
    float blackScore;
    int out_off;             // float offset of this logo's results within a frame's output record
    int lp;                  // LDS row pitch in floats: ((w+31)&~31)+8 -> consecutive rows sit 8 banks apart
    uint32_t lp_magic;       // ceil(2^32 / lp): i / lp == __umulhi(i, lp_magic) for i*lp < 2^32
};

// a band = up to NT consecutive run slots (one per thread) and the logo rows their windows touch
struct EvalBand {
    int logo;
    int s0, nslots;
    int y0, nrows;           // staged logo rows [y0, y0+nrows)
};

} // namespace amt
