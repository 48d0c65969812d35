/*
 * amt_oracle.h -- C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the reference's
 * logo hot path (Amatsukaze/LogoScan.hpp, Amatsukaze/ComputeKernel.cpp,
 * Amatsukaze/AMTLogo.hpp, include/logo.h).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (amatsukaze_amd/) never
 * links, imports or calls anything declared here.
 *
 * Parity status: PINNED against the real reference sources compiled through
 * oracle/ref_shim (oracle/_ref/libamt_ref.so, see oracle/build_ref.sh and
 * tests/test_oracle_vs_ref.py) for rows a1-a14 of SURVEY.md section 8.  The
 * self-specified CM / KFM passes (orc_cm_*, orc_kfm_*) have no in-tree reference
 * arithmetic: "parity unpinned" for those.
 */
#ifndef AMT_ORACLE_H
#define AMT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcLogo OrcLogo;   /* LogoDataParam (LogoScan.hpp:61-334) + LogoHeader */
typedef struct OrcScan OrcScan;   /* LogoScan     (LogoScan.hpp:398-660) */

/* ---- 5x5 correlation (LogoScan.hpp:24-41, ComputeKernel.cpp:54-121) ---- */
float orc_corr5x5_scalar(const float* k, const float* Y, int x, int y, int w, float* pavg);
float orc_corr5x5_avx(const float* k, const float* Y, int x, int y, int w, float* pavg);

/* ---- logo model (AMTLogo.hpp:49-280) ---- */
/* data = aY,bY,aU,bU,aV,bV back to back (AMTLogo.hpp:204-212) */
OrcLogo* orc_logo_create(int w, int h, int logUVx, int logUVy,
                         int imgw, int imgh, int imgx, int imgy, const float* data);
OrcLogo* orc_logo_load(const char* path);                       /* AMTLogo.hpp:257-279 */
int      orc_logo_save(const OrcLogo* l, const char* path, const char* name, int serviceId); /* :239-255 */
void     orc_logo_free(OrcLogo* l);
OrcLogo* orc_logo_deint(const OrcLogo* src);                    /* DeintLogo LogoScan.hpp:734-761 */
OrcLogo* orc_logo_field(const OrcLogo* src, int bottom);        /* MakeFieldLogo :257-283 */
void     orc_logo_info(const OrcLogo* l, int* out10);           /* w,h,logUVx,logUVy,imgw,imgh,imgx,imgy,maskpixels,count */
const float* orc_logo_data(const OrcLogo* l);
/* use_avx: 1 = ComputeKernel.cpp order (what any AVX x86 runs), 0 = scalar order */
void     orc_logo_create_mask(OrcLogo* l, float maskratio, int use_avx);  /* :112-229 */
const uint8_t* orc_logo_mask(const OrcLogo* l);
const float*   orc_logo_kernels(const OrcLogo* l);
const float*   orc_logo_scales(const OrcLogo* l);               /* pairs {scale,scale2} */
float    orc_logo_black_score(const OrcLogo* l);
float    orc_evaluate_logo(const OrcLogo* l, const float* src, float maxv, float fade,
                           float* work, int stride);            /* :231-255 */

/* ---- frame helpers (LogoScan.hpp:763-790) ---- */
void orc_deint_y_u8(float* dst, const uint8_t* src, int pitch, int w, int h);
void orc_deint_y_u16(float* dst, const uint16_t* src, int pitch, int w, int h);
void orc_copy_y_u8(float* dst, const uint8_t* src, int pitch, int w, int h);
void orc_copy_y_u16(float* dst, const uint16_t* src, int pitch, int w, int h);

/* ---- LogoFrame (LogoScan.hpp:1521-1836) ----
 * planeY: frame n's Y plane starts at planeY + n*frame_stride (bytes); pitch in
 * ELEMENTS (the reference's byte-pitch quirk for 16 bit, :1547, is NOT mirrored; see
 * DESIGN.md).  logos[] are deint logos with masks built.  out = nframes*nlogos*2. */
void orc_logoframe_scan(OrcLogo* const* logos, int nlogos, const void* planeY,
                        int64_t frame_stride, int pitch, int bits, int vi_w, int vi_h,
                        int nframes, float* out);
/* selectLogo :1647-1682 */
void orc_logoframe_select(const float* evals, int nframes, int nlogos, int ncand,
                          int* bestLogo, float* logoRatio);
/* writeResult :1686-1827; returns bytes written (text), -1 if cap too small */
int  orc_logoframe_write_result(const float* evals, int nframes, int nlogos, int logoIndex,
                                int fps_num, int fps_den, char* out, int cap);

/* ---- AMTAnalyzeLogo (LogoScan.hpp:1100-1161) ---- out: nframes*33 floats (p,t,b) per
 * SOURCE frame (the reference packs 8 per output frame; indices n*8+i clamped) */
void orc_analyze_frames(const OrcLogo* deint, const OrcLogo* fieldT, const OrcLogo* fieldB,
                        const void* planeY, int64_t frame_stride, int pitch, int bits,
                        int nframes, float* out);

/* ---- AMTEraseLogo (LogoScan.hpp:1238-1519) ---- */
void orc_delogo_u8(uint8_t* dst, int w, int h, int logopitch, int imgpitch, float maxv,
                   const float* A, const float* B, float fade);     /* :1248-1261 */
void orc_delogo_u16(uint16_t* dst, int w, int h, int logopitch, int imgpitch, float maxv,
                    const float* A, const float* B, float fade);
/* analysis = per-source-frame 33 floats, laid out as the analyze clip is (frame k>>3, slot k&7);
 * n_analyze_src = number of source frames the analysis covers (clamped like the clip). */
void orc_calc_fade2(const float* analysis, int num_frames, int n, float* fadeT, float* fadeB); /* :1263-1315 */
void orc_calc_fade(const int* frameResult, int has_result, int maxFadeLength,
                   const float* analysis, int num_frames, int n, float* fadeT, float* fadeB); /* :1317-1341 */
/* ReadLogoFrameFile :1421-1461: returns 0 ok, -1 bad order */
int  orc_read_logoframe(const char* text, int num_frames, int* frameResult);
/* GetFrameT mode 0 :1343-1400 on one frame (in place) */
void orc_erase_frame(const OrcLogo* logo, void* Y, void* U, void* V, int pitchY, int pitchUV,
                     int bits, float fadeT, float fadeB);

/* ---- LogoScan (LogoScan.hpp:336-660) ---- */
OrcScan* orc_scan_create(int w, int h, int logUVx, int logUVy, int thy);
void     orc_scan_free(OrcScan* s);
int      orc_scan_add_frame_u8(OrcScan* s, const uint8_t* Y, const uint8_t* U, const uint8_t* V,
                               int pitchY, int pitchUV);           /* AddFrame :594-659 */
int      orc_scan_nframes(const OrcScan* s);
void     orc_scan_sums(const OrcScan* s, double* out);             /* 5 doubles per pixel: F,B,F2,B2,FB */
/* Normalize(maxv)+GetLogo(clean) :471-566; NULL if regression fails */
OrcLogo* orc_scan_get_logo(OrcScan* s, int maxv, int clean, int imgw, int imgh, int imgx, int imgy);

/* ---- LogoAnalyzer::ScanLogo (LogoScan.hpp:794-1080) on in-memory 8-bit frames ----
 * full frames: Y/U/V planes at base + n*stride.  minfades_out (optional) receives the
 * last ReMakeLogo round's argmin fades (numFrames ints).  Returns logo or NULL. */
OrcLogo* orc_scanlogo(const uint8_t* Y, const uint8_t* U, const uint8_t* V,
                      int64_t strideY, int64_t strideUV, int pitchY, int pitchUV,
                      int imgw, int imgh, int nframes_total,
                      int imgx, int imgy, int w, int h, int thy, int numMaxFrames,
                      int use_avx, int* num_valid_out, int* minfades_out);
/* the same, the ReMakeLogo rounds' per-frame evaluations (:957-984, independent per frame) dealt over `threads` host threads; the
 * stream-order quota (:885) and both accumulations stay sequential.  frames_read_out (optional): frames consumed before the quota
 * closed the stream (readCount, :883).  Byte-identical to orc_scanlogo for any thread count (tests/test_oracle_vs_ref.py). */
OrcLogo* orc_scanlogo_mt(const uint8_t* Y, const uint8_t* U, const uint8_t* V,
                         int64_t strideY, int64_t strideUV, int pitchY, int pitchUV,
                         int imgw, int imgh, int nframes_total,
                         int imgx, int imgy, int w, int h, int thy, int numMaxFrames,
                         int use_avx, int* num_valid_out, int* minfades_out, int threads, int* frames_read_out);

/* ---- SELF-SPECIFIED whole-frame metrics -- PARITY UNPINNED: the reference has no
 * in-tree arithmetic for them.  C restatement of oracle/frame_stats_oracle.py, used as
 * the CPU baseline leg for these passes.  out = nframes*8 uint64, layout AMTGPU_FS_*; prev_first may be NULL. */
/* AMTSource::MergeField (AMTSource.hpp:291-355): weave one frame from a top and a bottom picture; pitches in elements */
void orc_merge_field(const void* tY, const void* tU, const void* tV, const void* bY, const void* bU, const void* bV, int spitchY,
                     int spitchUV, int nv12, int bits, int W, int H, void* dY, void* dU, void* dV, int pitchY, int pitchUV);
void orc_frame_metrics(const void* Y, int64_t frame_stride, int pitch, int bits, int W, int H, int nframes,
                       const void* prev_first, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
