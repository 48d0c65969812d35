// eval_plan.h -- device-side description of a logo evaluation (shared by host code and kernels).
#pragma once
#include <cstdint>

namespace amt {

constexpr int kTablePad = 256;      // mask-pixel tables are padded to a multiple of this
constexpr int kNumBins = 32;        // 256 >> 3 background levels (LogoScan.hpp:63-68)
constexpr int kEvalThreads = 256;   // threads per workgroup = run slots per band
constexpr int kEvalStage = 12;      // staged rectangle pixels per thread: an LDS plane holds <= 256*12 floats
constexpr int kEvalBandPixels = 2 * kEvalThreads;   // a run slot holds up to two mask pixels
constexpr int kEvalScorePad = 36;   // score-row pitch = 512 + 36 floats: the 16 summing lanes of a ds_read_b128 group land
                                    // on distinct banks, and the read-ahead of the sum stays inside the row
constexpr int kEvalMaxFades = 24;   // fades per launch (one LDS score row and one summing lane per fade)

// one evaluation logo (a LogoDataParam after CreateLogoMask) resident in HBM.  Mask pixels are numbered in raster
// order of the visited pixels (index m); horizontally adjacent mask pixels are grouped into run slots of up to two.
struct EvalLogoDev {
    const float* a;          // [h*w]   A plane of the evaluation logo (deint or field)
    const float* b;          // [h*w]
    const float2* scales;    // [32][count_pad]   bin-major {scale, scale2}
    const float2* kslot;     // [25][nslots_pad]  kernel taps of a slot's two pixels as packed pairs: entry c*5+r, c<4:
                             // {k_px0[r*5+c+1], k_px1[r*5+c]} (both look at window column c+1); c==4: {k_px0[r*5], k_px1[r*5+4]}
    const uint2* slot2;      // [nslots_pad]  {(n << 28) | m0, LDS float offset of the window's top-left element}
    int w, h;                // evaluation-logo size (field logos: h/2)
    int count, count_pad;
    int nslots_pad;
    int band0, nbands;       // this logo's bands in the band table
    int imgx, imgy;          // rectangle origin in the full frame (full-frame rows)
    int row0, row_step;      // source row of logo row y = imgy + row0 + y*row_step
    int deint;               // 1: source is the [1 2 1] vertical blend of rows y-1,y,y+1 (DeintY)
    float blackScore;
    int out_off;             // float offset of this logo's results within a frame's output record
    int lp;                  // LDS row pitch in floats: ((w+31)&~31)+8
};

constexpr int kLinMaxFades = 12;    // fades the linear kernel's source is written for (11 for AMTAnalyzeLogo, the instantiated case)
// Most frames a workgroup of the linear kernel takes; how many it does take follows from the LDS its list of bin checks leaves
// (EvalEngine::run_linear: FOUR workgroups of four waves share a CU's 160 KB).
#ifndef AMT_LIN_G
#define AMT_LIN_G 7
#endif
constexpr int kLinMaxFrames = AMT_LIN_G;
#ifndef AMT_LIN_G16
#define AMT_LIN_G16 7
#endif
constexpr int kLinMaxFrames16 = AMT_LIN_G16; // ... of its 16-bit instantiation: the same budget (four workgroups per CU, <= 128 registers)
#ifndef AMT_LIN_WGS_MIN16
#define AMT_LIN_WGS_MIN16 2048      /* (round 5: 4 112-frame launches at 10 bits 6 frames 1.542 ms, 5: 1.568, 4: 1.615; 2 500-frame launches 6: 1.022, 4: 0.972 -- enough workgroups for ~3 rounds first) */
#endif

// a band = up to kEvalThreads consecutive run slots (one per thread) and the logo rows their windows touch
struct EvalBand {
    int logo;
    int s0, nslots;
    int y0, nrows;           // staged logo rows [y0, y0+nrows)
    int m0, npix;            // the band's mask pixels [m0, m0+npix) (raster order)
    int x0, bw;              // staged logo columns [x0, x0+bw): the whole width while five rows of it fit an LDS plane, else the
                             // columns the band's windows touch (x0 a multiple of 4) -- logos of any width
    int lp;                  // LDS row pitch of this band in floats: ((bw+31)&~31)+8
};

} // namespace amt
