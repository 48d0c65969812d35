/*
 * amt_gpu.h -- C ABI of the MI355X-native logo / CM / KFM analysis hot path.
 *
 * Drop-in boundary for the per-frame pixel analysis that nekopanda/Amatsukaze runs as scalar C++
 * inside Amatsukaze.dll.  Every entry point names the reference interface it replaces (paths relative
 * to the reference tree).  Conventions follow the reference's own exported C API
 * (LogoScan.hpp:1083-1098 ScanLogo, StreamUtils.hpp:1037-1039 AMTContext_*, LogoGUISupport.hpp:254-275):
 * opaque handles, int return 1 = ok / 0 = failure with the message kept on the context
 * (amtgpu_last_error), no exceptions across the boundary, handles freed by *_destroy.
 *
 * Frames are 4:2:0 planar, 8-bit (uint8) or 9..16-bit (uint16 containers), the layout AMTSource hands to
 * AviSynth (AMTSource.hpp:428-442).  A "batch" is `nframes` frames whose planes sit at
 * base + n*frame_stride (bytes); `pitch` is in ELEMENTS.  Pointers named d* are DEVICE pointers (HBM,
 * hipMalloc'ed by the caller or by amtgpu_frames_upload); everything else is host memory.
 *
 * All kernels are launched on the context's HIP stream (amtgpu_context_set_stream); calls that return
 * results to host memory synchronise that stream, calls documented "async" do not.
 * Calls on one context are serialised by a lock inside it (filters sharing a context may be driven from several
 * AviSynth threads); use one context per thread for concurrency.
 */
#ifndef AMT_GPU_H
#define AMT_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMTGPU_ABI_VERSION 5      /* 2: amtgpu_framestats_create lost two unused parameters; *W entry points, logo header access, markers
                                   * 3: additions only -- device-side CalcFade (amtgpu_erase_calc_fades_device, *_dfades), sharded frame
                                   *    metrics (amtgpu_framestats_allgather / _sharded), registered host frames (amtgpu_frames_register),
                                   *    amtgpu_download_scatter, owned markers
                                   * 4: amtgpu_logoframe_decide_host returns -1 (was 0) when the text buffer is too small and refuses
                                   *    num_candidates > num_logos; additions: amtgpu_logoframe_dump_result, amtgpu_host_set_parallelism
                                   * 5: amtgpu_logoframe_decide_host is back to 1 / 0 like every other entry point (a too-small text buffer
                                   *    is 0 with *text_len > cap: `if (!call) fail;` written against ABI <= 3 is right again); additions:
                                   *    amtgpu_analyze_set_fixup_queue, amtgpu_erase_batch_dfades_to */
#define AMTGPU_NUM_FADE 11            /* LogoAnalyzeFrame p/t/b[11]  (LogoScan.hpp:1100-1103) */
#define AMTGPU_ANALYZE_FLOATS 33      /* floats per source frame in an analysis record */

typedef struct AmtGpuContext AmtGpuContext;       /* AMTContext            (StreamUtils.hpp:343-511) */
typedef struct AmtGpuLogo AmtGpuLogo;             /* logo::LogoData+LogoHeader (AMTLogo.hpp:19-280) */
typedef struct AmtGpuLogoFrame AmtGpuLogoFrame;   /* logo::LogoFrame       (LogoScan.hpp:1521-1836) */
typedef struct AmtGpuAnalyze AmtGpuAnalyze;       /* logo::AMTAnalyzeLogo  (LogoScan.hpp:1106-1236) */
typedef struct AmtGpuErase AmtGpuErase;           /* logo::AMTEraseLogo    (LogoScan.hpp:1238-1519) */
typedef struct AmtGpuLogoScan AmtGpuLogoScan;     /* logo::LogoScan        (LogoScan.hpp:398-660) */
typedef struct AmtGpuFrameStats AmtGpuFrameStats; /* self-specified CM / KFM whole-frame metrics */

/* progress callback of ScanLogo (LogoScan.hpp:792): return 0 to cancel */
typedef int (*AMTGPU_LOGO_ANALYZE_CB)(float progress, int nread, int total, int ngather);

int amtgpu_abi_version(void);
/* Host threads of the O(frames) decision routines (selectLogo / writeResult text, scene changes, cadence): they cut the clip into
 * contiguous frame ranges -- the window filters are local -- and leave only the two small state machines sequential.  max_threads /
 * min_frames_per_thread <= 0 restore the defaults (half the host's cores up to 32; 32 768 frames).  Process-wide.  Results never depend
 * on either value.  A host with its own thread pool (AviSynth MT) passes 1 to keep the library on the calling thread. */
void amtgpu_host_set_parallelism(int max_threads, int min_frames_per_thread);
/* How many copies of the HIP runtime (libamdhip64) are mapped into this process, and where from (newline-separated paths in `paths`,
 * truncated to cap; may be NULL).  More than one -- e.g. this library bound to /opt/rocm's copy while another component brought its
 * own -- means device pointers and streams of one are unknown to the other: copies fail with "invalid argument", kernels fault.
 * A host that mixes ROCm users checks this once after loading everything (the Python mirror does, and loads torch first so that
 * both bind to the same copy). */
int amtgpu_hip_runtimes_loaded(char* paths, int cap);

/* ---- context: replaces AMTContext_Create / ATMContext_Delete / AMTContext_GetError
 *      (StreamUtils.hpp:1037-1039) ---- */
AmtGpuContext* amtgpu_context_create(int device);       /* NULL if no HIP device / bad index */
void           amtgpu_context_destroy(AmtGpuContext* ctx);
const char*    amtgpu_last_error(const AmtGpuContext* ctx);
/* use an existing hipStream_t (e.g. the caller's compute stream); NULL = the context's own stream, which is created
 * hipStreamNonBlocking: it is NOT ordered against the legacy default ("null") stream.  A caller whose other GPU work runs
 * on the null stream (handle 0, e.g. PyTorch without an explicit stream) passes AMTGPU_STREAM_LEGACY_DEFAULT
 * (== hipStreamLegacy) so that the "async" entry points below are stream-ordered with that work.  On any other stream
 * pair the caller orders the two with events or amtgpu_context_synchronize before consuming device outputs. */
#define AMTGPU_STREAM_LEGACY_DEFAULT ((void*)1)
int            amtgpu_context_set_stream(AmtGpuContext* ctx, void* hip_stream);
void*          amtgpu_context_get_stream(AmtGpuContext* ctx);
int            amtgpu_context_synchronize(AmtGpuContext* ctx);
/* Device partitions.  The whole-frame metrics are bandwidth work, the logo kernels arithmetic: run BESIDE each other they finish sooner
 * than one after the other -- but the hardware only co-schedules them when the device is partitioned (the logo kernels otherwise
 * take every CU's registers and LDS).  amtgpu_stream_create_cu_range returns a hipStream_t (hipStreamNonBlocking) whose kernels run on
 * `num_cus` compute units starting at `first_cu` in the driver's mask order, in which consecutive units go round the XCDs: a contiguous
 * range is spread evenly over all of them and their L2s.  Measured on MI355X (profiles/r04_notes.md): the effective granularity is 32
 * units (4 per XCD); the frame metrics on units [0, 64) beside the analysis + scan on [64, 256) take 8.2 ms per 10 000 frames instead
 * of 9.2 ms one after the other.  Use: one context per partition (amtgpu_context_set_stream), ordered against each other by the
 * caller's events.  NULL on failure (message on the context).  amtgpu_device_cu_count: compute units of the context's device. */
void*          amtgpu_stream_create_cu_range(AmtGpuContext* ctx, int first_cu, int num_cus);
void           amtgpu_stream_destroy(AmtGpuContext* ctx, void* hip_stream);
int            amtgpu_device_cu_count(AmtGpuContext* ctx);
/* per-kernel timing with HIP events on the launch stream (no counterpart in the reference, which only logs
 * phase wall times, CMAnalyze.hpp:37-39).  enable(1) resets the totals; report writes
 * "kernel_name calls total_ms\n" lines and returns the byte count (-1 on error). */
int            amtgpu_profile_enable(AmtGpuContext* ctx, int on);
int            amtgpu_profile_report(AmtGpuContext* ctx, char* out, int cap);

/* ---- frame ingest: pinned staging + hipMemcpyAsync on a side stream, double buffered against the
 *      compute stream (the step AMTSource::GetFrame feeds, AMTSource.hpp:721-780).  Allocates the device
 *      batch; free with amtgpu_device_free. ---- */
void* amtgpu_device_alloc(AmtGpuContext* ctx, uint64_t bytes);
void  amtgpu_device_free(AmtGpuContext* ctx, void* dptr);
int   amtgpu_frames_upload(AmtGpuContext* ctx, void* ddst, const void* hsrc, uint64_t bytes);   /* async on side stream */
/* the same for nchunks pieces of chunk_bytes that sit dst_stride apart on the device and src_stride apart on the host -- e.g. only
 * the logo rectangle's rows of every frame of a batch (h*pitch bytes instead of a whole frame: the logo passes read nothing
 * else).  async on the side stream */
int   amtgpu_frames_upload_strided(AmtGpuContext* ctx, void* ddst, int64_t dst_stride, const void* hsrc, int64_t src_stride,
                                   uint64_t chunk_bytes, int nchunks);
/* the same for `nsrc` separately allocated host frames at once: piece j of source i lands at ddst + (i * chunks_per_src + j) *
 * dst_stride -- one call and one copy launch for a whole group of PVideoFrames' logo rectangles (per-frame calls cost more in
 * API overhead than in bytes).  async on the side stream */
int   amtgpu_frames_upload_gather(AmtGpuContext* ctx, void* ddst, int64_t dst_stride, const void* const* hsrc, int64_t src_stride,
                                  uint64_t chunk_bytes, int chunks_per_src, int nsrc);
int   amtgpu_frames_upload_wait(AmtGpuContext* ctx);   /* make the compute stream wait for pending uploads */
/* Uploads from pageable host memory are staged through a ring of four pinned 32 MiB slots; the staging copy of a large upload is
 * shared out over `nthreads` threads (the caller's included; default min(4, cores / 4)) because one core's memcpy is below what
 * PCIe Gen5 x16 carries.  1 = the calling thread alone. */
int   amtgpu_context_set_upload_threads(AmtGpuContext* ctx, int nthreads);
/* Page-lock a host range in place (hipHostRegister) -- a decoder's frame pool, the buffers AMTSource keeps its frames in
 * (AMTSource.hpp:428-442): amtgpu_frames_upload / _upload_strided whose source lies inside a registered range skip the staging ring
 * and go out as one DMA copy straight from the caller's memory, which must stay untouched until amtgpu_frames_upload_wait's
 * consumer has run (or amtgpu_context_synchronize).  Registration costs about as much as touching every page once: register pools,
 * not frames.  unregister waits for the side stream first. */
int   amtgpu_frames_register(AmtGpuContext* ctx, void* hptr, uint64_t bytes);
int   amtgpu_frames_unregister(AmtGpuContext* ctx, void* hptr);
int   amtgpu_download(AmtGpuContext* ctx, void* hdst, const void* dsrc, uint64_t bytes);        /* synchronous */
/* nchunks pieces of chunk_bytes, src_stride apart on the device, dst_stride apart on the host (an erased rectangle back into
 * the rows of a host frame).  synchronous */
int   amtgpu_download_strided(AmtGpuContext* ctx, void* hdst, int64_t dst_stride, const void* dsrc, int64_t src_stride,
                              uint64_t chunk_bytes, int nchunks);
/* `bytes` from the device into a pinned landing buffer of the context: one asynchronous copy and one wait; *hptr is valid until
 * the next call.  For callers that scatter the bytes into several host frames themselves (a block of erased rectangles back into
 * the frames AMTEraseLogo::GetFrame serves, LogoScan.hpp:1343-1400). */
int   amtgpu_download_pinned(AmtGpuContext* ctx, const void* dsrc, uint64_t bytes, const void** hptr);
/* The same single copy, with the scatter done by the library while the context is still locked -- the form to use when several host
 * threads share one context (several filters of one script under Prefetch): piece i is nchunks runs of chunk_bytes that sit back to
 * back at dsrc + src_offset and go to hdst, dst_stride apart (an erased rectangle's rows back into the rows of a host frame). */
typedef struct AmtGpuScatter {
    void*    hdst;
    int64_t  dst_stride;
    uint64_t src_offset;
    uint64_t chunk_bytes;
    int      nchunks;
} AmtGpuScatter;
int   amtgpu_download_scatter(AmtGpuContext* ctx, const void* dsrc, uint64_t bytes, const AmtGpuScatter* pieces, int npieces);
/* Markers on the compute stream, ids 0..15: record(id) behind a batch's launches, wait(id) on the host before that batch's device
 * buffer is written again.  What a double-buffered caller (LogoFrame::scanFrames, LogoScan.hpp:1570-1589, over AMTSource::GetFrame)
 * needs instead of amtgpu_context_synchronize, which would also wait for the batch in flight.  wait on a never recorded id returns
 * at once. */
int   amtgpu_marker_record(AmtGpuContext* ctx, int id);
int   amtgpu_marker_wait(AmtGpuContext* ctx, int id);
/* The same pair on a marker the caller owns: users of a shared context (two LogoFrame scans, a filter next to user code) cannot
 * re-record each other's markers, whatever ids they would have picked. */
typedef struct AmtGpuMarker AmtGpuMarker;
AmtGpuMarker* amtgpu_marker_create(AmtGpuContext* ctx);
void  amtgpu_marker_destroy(AmtGpuContext* ctx, AmtGpuMarker* m);
int   amtgpu_marker_record_on(AmtGpuContext* ctx, AmtGpuMarker* m);
int   amtgpu_marker_wait_on(AmtGpuContext* ctx, AmtGpuMarker* m);     /* never recorded: returns at once */


/* ---- frame assembly: replaces AMTSource::MakeFrame -> MergeField / Copy1 / Copy2 (AMTSource.hpp:291-366) on decoded
 *      pictures already in HBM (uploaded with amtgpu_frames_upload): output frame i takes its even rows from picture
 *      top_index[i] and its odd rows from picture bottom_index[i] (the same picture for frame-coded streams, two for
 *      field-coded ones), plane by plane; nv12 != 0: the source chroma is ONE interleaved UV plane (dsrcU; dsrcV ignored)
 *      that is split into planar U and V (Copy2).  Pitches in ELEMENTS, strides in bytes; top_index / bottom_index are
 *      host arrays of nframes entries or NULL (= i).  The reference multiplies BYTE pitches into uint16_t pointers for
 *      > 8-bit pictures (:303-306 with T = uint16_t, :345-350); this uses element pitches, as every consumer of the frame
 *      does.  async unless index arrays are given ---- */
int   amtgpu_weave_fields_batch(AmtGpuContext* ctx, const void* dsrcY, const void* dsrcU, const void* dsrcV,
                                int64_t src_strideY, int64_t src_strideUV, int src_pitchY, int src_pitchUV, int num_pictures,
                                const int* top_index, const int* bottom_index, int nv12, int bits, int width, int height,
                                void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV, int pitchY, int pitchUV,
                                int nframes);

/* ---- the stream-index file AMTSource is built from: replaces LoadAMTSource's reader (AMTSource.hpp:854-871; writer :835-852,
 *      File::writeArray framing CoreUtils.hpp:275-284, FilterSourceFrame StreamReform.hpp:145-154) and the frame-assembly rule of
 *      AMTSource::OnFrameOutput (:482-566).  MSVC x64 POD layouts with 2-byte wchar_t, parsed with fixed offsets. ---- */
typedef struct AmtGpuAmtsFile AmtGpuAmtsFile;
AmtGpuAmtsFile* amtgpu_amts_load(AmtGpuContext* ctx, const char* path);      /* ctx may be NULL (no message kept then) */
void amtgpu_amts_destroy(AmtGpuAmtsFile* a);
/* out19 = VideoFormat {format, width, height, displayWidth, displayHeight, sarWidth, sarHeight, frameRateNum, frameRateDenom,
 * colorPrimaries, transferCharacteristics, colorSpace, progressive, fixedFrameRate}, AudioFormat {channels, sampleRate},
 * DecoderSetting {mpeg2, h264, hevc} */
int  amtgpu_amts_get_info(const AmtGpuAmtsFile* a, int* out19, int* num_frames, int* num_audio_frames);
/* source TS path and audio wave path, UTF-16 converted to UTF-8; 0 if a buffer is too small */
int  amtgpu_amts_get_paths(const AmtGpuAmtsFile* a, char* srcpath, int cap_src, char* audiopath, int cap_audio);
/* per-frame columns of the FilterSourceFrame list (num_frames entries each; any pointer may be NULL) */
int  amtgpu_amts_get_frames(const AmtGpuAmtsFile* a, int64_t* framePTS, int64_t* fileOffset, int* keyFrame, uint8_t* halfDelay, int* cmType);
/* Which decoded pictures make which frame: picture_pts = PTS of the decoded pictures in output order.  For frame i the top field
 * comes from picture top_index[i] and the bottom field from bottom_index[i] (both -1: the frame cannot be made from this sequence,
 * e.g. a half-delayed frame right after a discontinuity) -- the arrays amtgpu_weave_fields_batch takes. */
int  amtgpu_amts_weave_plan(const AmtGpuAmtsFile* a, const int64_t* picture_pts, int npictures, int* top_index, int* bottom_index);

/* ---- logo model: replaces LogoData::Load / Save (AMTLogo.hpp:239-279), LogoFile_* getters
 *      (LogoGUISupport.hpp:254-275) ---- */
AmtGpuLogo* amtgpu_logo_load(AmtGpuContext* ctx, const char* path);
/* paths as the reference's exports take them: NUL-terminated UTF-16 (const tchar* = wchar_t* on Windows, LogoScan.hpp:1083-1086;
 * C# CharSet.Unicode, AmatsukazeNatives.cs:391-393).  Converted to UTF-8 for the file system here (Linux). */
AmtGpuLogo* amtgpu_logo_loadW(AmtGpuContext* ctx, const uint16_t* path);
/* planes = aY,bY,aU,bU,aV,bV back to back (AMTLogo.hpp:204-212) */
AmtGpuLogo* amtgpu_logo_from_planes(AmtGpuContext* ctx, int w, int h, int logUVx, int logUVy,
                                    int imgw, int imgh, int imgx, int imgy, const float* planes);
int  amtgpu_logo_save(AmtGpuContext* ctx, const AmtGpuLogo* logo, const char* path, const char* name, int serviceId);
int  amtgpu_logo_saveW(AmtGpuContext* ctx, const AmtGpuLogo* logo, const uint16_t* path, const char* name, int serviceId);
/* LogoFile_GetName / GetServiceId / SetName / SetServiceId (LogoGUISupport.hpp:254-275) of the extended header (AMTLogo.hpp:19-47):
 * name is UTF-8 bytes as stored (at most 254 + NUL); either out pointer of get may be NULL */
int  amtgpu_logo_get_header(const AmtGpuLogo* logo, char* name, int name_cap, int* serviceId);
int  amtgpu_logo_set_header(AmtGpuLogo* logo, const char* name, int serviceId);
void amtgpu_logo_destroy(AmtGpuLogo* logo);
/* out[8] = w,h,logUVx,logUVy,imgw,imgh,imgx,imgy */
int  amtgpu_logo_get_info(const AmtGpuLogo* logo, int* out8);
int  amtgpu_logo_get_planes(const AmtGpuLogo* logo, float* out);
/* evaluation tables of LogoDataParam::CreateLogoMask (LogoScan.hpp:112-229) for inspection / tests:
 * kind 0 = DeintLogo'ed logo, 1 = top-field logo, 2 = bottom-field logo (MakeFieldLogo :257-283).
 * Any out pointer may be NULL.  mask: w*h bytes; kernels: count*25; scales: count*32*{scale,scale2}. */
int  amtgpu_logo_mask_tables(AmtGpuContext* ctx, const AmtGpuLogo* logo, int kind, float maskratio,
                             int* maskpixels, int* count, float* blackScore,
                             uint8_t* mask, float* kernels, float* scales);

/* ---- CM all-frames logo scan: replaces logo::LogoFrame (ctor :1592-1616, scanFrames :1618-1630,
 *      selectLogo :1647-1682, writeResult :1686-1827, getBestLogo/getLogoRatio :1829-1835) as driven by
 *      CMAnalyze::logoFrame (CMAnalyze.hpp:273-317).  Unreadable logo files are ignored like the
 *      reference does (:1612-1614) and score {0,-1}. ---- */
AmtGpuLogoFrame* amtgpu_logoframe_create(AmtGpuContext* ctx, const char* const* logopaths, int nlogos, float maskratio);
AmtGpuLogoFrame* amtgpu_logoframe_create_from_logos(AmtGpuContext* ctx, const AmtGpuLogo* const* logos, int nlogos, float maskratio);
void amtgpu_logoframe_destroy(AmtGpuLogoFrame* lf);
/* declare the clip (VideoInfo): resets results to num_frames entries */
int  amtgpu_logoframe_begin(AmtGpuLogoFrame* lf, int width, int height, int bits, int num_frames, int fps_num, int fps_den);
/* scan frames [first, first+nframes) of the clip from a device batch (Y plane only).  async */
int  amtgpu_logoframe_scan_batch(AmtGpuLogoFrame* lf, const void* dY, int64_t frame_stride, int pitch, int first, int nframes);
/* results: num_frames*nlogos*{corr0,corr1} (EvalResult, LogoScan.hpp:1532-1535) */
/* out2 = {first row, one past the last row} of the Y plane that the scan of these logos reads (the union of their rectangles) */
int  amtgpu_logoframe_get_rows(const AmtGpuLogoFrame* lf, int* out2);
/* ... and {first column, one past the last column}: a caller that ships frames over PCIe needs to bring no other samples */
int  amtgpu_logoframe_get_columns(const AmtGpuLogoFrame* lf, int* out2);
int  amtgpu_logoframe_get_results(AmtGpuLogoFrame* lf, float* out);
/* sharded scans: install results computed elsewhere (other ranks) for frames [first, first+nframes) */
int  amtgpu_logoframe_set_results(AmtGpuLogoFrame* lf, int first, int nframes, const float* evals);
int  amtgpu_logoframe_select_logo(AmtGpuLogoFrame* lf, int num_candidates);        /* -1 = all */
int  amtgpu_logoframe_write_result(AmtGpuLogoFrame* lf, const char* outpath, int logo_index); /* -1 = best */
/* LogoFrame::dumpResult (LogoScan.hpp:1632-1643): one text file per logo, "<basepath><logo index>", a line "%f,%f\n" {corr0, corr1} per frame */
int  amtgpu_logoframe_dump_result(AmtGpuLogoFrame* lf, const char* basepath);
int  amtgpu_logoframe_best_logo(const AmtGpuLogoFrame* lf);
/* The same two decisions -- LogoFrame::selectLogo and the text LogoFrame::writeResult writes (LogoScan.hpp:1647-1827) -- from scan
 * records alone, on the host, no device and no LogoFrame object: what a rank (or a tool) that only holds the gathered
 * records [num_frames][num_logos]{corr0, corr1} needs.  logo_index -1 = the selected logo.  text may be NULL (cap 0) to ask for
 * the length.  Returns 1 = done, 0 = failed: bad arguments (NULL records, logo_index or num_candidates > num_logos, fps <= 0) or cap
 * too small -- then *text_len (set whenever the arguments are good) is > cap and says how much it takes.  O(1) work per frame.  Evidence that is NaN (corr0 = +inf with corr1 = -inf) sorts after
 * every number in the median window -- the reference's std::sort over it is undefined. */
int  amtgpu_logoframe_decide_host(const float* evals, int num_frames, int num_logos, int num_candidates, int logo_index,
                                  int fps_num, int fps_den, int* best_logo, float* logo_ratio, char* text, int cap, int* text_len);
float amtgpu_logoframe_logo_ratio(const AmtGpuLogoFrame* lf);

/* ---- encode-time analysis: replaces logo::AMTAnalyzeLogo ("AMTAnalyzeLogo" "cs[maskratio]i",
 *      Amatsukaze.cpp:58; ctor :1164-1201, GetFrameT :1119-1161).  One record of 33 floats
 *      {p[11],t[11],b[11]} per SOURCE frame; the AviSynth shim packs 8 per BGR32 frame (:1195-1200). ---- */
AmtGpuAnalyze* amtgpu_analyze_create(AmtGpuContext* ctx, const char* logopath, float maskratio);
AmtGpuAnalyze* amtgpu_analyze_create_from_logo(AmtGpuContext* ctx, const AmtGpuLogo* logo, float maskratio);
void amtgpu_analyze_destroy(AmtGpuAnalyze* an);
/* dout: device buffer of nframes*33 floats.  async */
int  amtgpu_analyze_batch(AmtGpuAnalyze* an, const void* dY, int64_t frame_stride, int pitch, int bits, int nframes, float* dout);
/* convenience: same, results copied to host (synchronises) */
/* out4 = {imgx, imgy, w, h}: the rectangle the analysis reads (LogoScan.hpp:1132-1141) -- a host that uploads frames may ship only
 * the rows [imgy, imgy+h) of every Y plane, at their place in the frame: nothing else is read */
int  amtgpu_analyze_get_rect(const AmtGpuAnalyze* an, int* out4);
int  amtgpu_analyze_batch_host(AmtGpuAnalyze* an, const void* dY, int64_t frame_stride, int pitch, int bits, int nframes, float* hout);

/* Evaluation mode of the 33 scores per frame.
 *   AMTGPU_ANALYZE_EXACT (default): every fade evaluated in the reference's fp32 order -- records are bit-identical to
 *     AMTAnalyzeLogo::GetFrameT's.
 *   AMTGPU_ANALYZE_LINEAR_GUARDED: CalcCorrelation5x5 is linear in the window (ComputeKernel.cpp:77-121), so the 11 blends are
 *     formed from ONE evaluation of the source window and ONE of the background-estimate window per mask pixel (~4x less
 *     arithmetic).  Scores differ from the reference's by rounding only (|diff| <= amtgpu_analyze_error_bound, typically 1e-6;
 *     the north star allows 1e-4), and the integer decisions taken from them are guarded: bins are selected from the exactly
 *     evaluated mean whenever the interpolated one is near a bin edge, and every frame whose argmin over the fades of p, t or b
 *     (all CalcFade2 ever reads, LogoScan.hpp:1288-1314) is not separated by more than twice the error bound is re-evaluated by
 *     the exact kernel before the batch is handed out -- amtgpu_erase_calc_fades returns identical fades in both modes. */
#define AMTGPU_ANALYZE_EXACT 0
#define AMTGPU_ANALYZE_LINEAR_GUARDED 1
#define AMTGPU_ANALYZE_LINEAR_UNGUARDED 2   /* the linear evaluation alone, without the argmin guard: for accuracy tests and profiling */
int   amtgpu_analyze_set_mode(AmtGpuAnalyze* an, int mode);
/* frames of the most recent batch that the guard re-evaluated exactly (synchronises); 0 in exact mode, -1 on error */
int   amtgpu_analyze_last_refined(AmtGpuAnalyze* an);
/* Linear modes: (pixel, frame, fade) pairs whose window mean lies within the evaluation's error bound of a bin edge (LogoScan.hpp:304 is
 * discontinuous there) are listed per wave and settled exactly when the workgroup has finished; a workgroup whose list overflows leaves
 * its frames to the exact kernel (they are counted by amtgpu_analyze_last_refined).  entries = pairs a wave can list, 16 .. 640,
 * default 256; fewer frames share a workgroup as the list grows (LDS).  A tuning knob: results do not depend on it. */
int   amtgpu_analyze_set_fixup_queue(AmtGpuAnalyze* an, int entries);
/* the linear mode's bound on |score - reference score| for group 0 = p, 1 = t, 2 = b at the given bit depth; 0 in exact mode */
float amtgpu_analyze_error_bound(AmtGpuAnalyze* an, int group, int bits);

/* ---- encode-time erase: replaces logo::AMTEraseLogo ("AMTEraseLogo" "ccs[logof]s[mode]i[maxfade]i",
 *      Amatsukaze.cpp:59; ctor :1464-1481, ReadLogoFrameFile :1421-1461, CalcFade :1317-1341,
 *      CalcFade2 :1263-1315, Delogo :1248-1261, GetFrameT mode 0 :1343-1400).  logofpath may be "" ---- */
AmtGpuErase* amtgpu_erase_create(AmtGpuContext* ctx, const char* logopath, const char* logofpath, int mode, int maxfade);
AmtGpuErase* amtgpu_erase_create_from_logo(AmtGpuContext* ctx, const AmtGpuLogo* logo, const char* logof_text, int mode, int maxfade);
void amtgpu_erase_destroy(AmtGpuErase* er);
/* fades for frames [first, first+nframes) of a num_frames clip from the per-source-frame analysis of the
 * WHOLE clip (host, num_frames*33 floats).  out = nframes*{fadeT,fadeB}.  Host-only (tiny). */
int  amtgpu_erase_calc_fades(AmtGpuErase* er, const float* analysis, int num_frames, int first, int nframes, float* fades_out);
/* in-place erase of a device batch with the given per-frame fades (host array nframes*2).  async */
int  amtgpu_erase_batch(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV,
                        int pitchY, int pitchUV, int bits, int nframes, const float* fades);
/* the same on planes that hold ONLY the logo rectangle (w x h luma, w/2 x h/2 chroma samples per frame, first sample = the
 * rectangle's top-left): what a per-frame host filter ships instead of whole frames -- Delogo touches nothing else
 * (LogoScan.hpp:1248-1261, 1374-1397).  async */
int  amtgpu_erase_rect_batch(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV,
                             int pitchY, int pitchUV, int bits, int nframes, const float* fades);
/* CalcFade / CalcFade2 (LogoScan.hpp:1263-1341) on the DEVICE: the same decision as amtgpu_erase_calc_fades, frame by frame in one small
 * kernel, from analysis records that are still in HBM (the output of amtgpu_analyze_batch) -- no host round trip between analysis
 * and erase, the whole analyse -> decide -> erase chain is stream-ordered.  d_analysis holds the records of source frames
 * [analysis_first, analysis_first + analysis_count) of a num_frames clip, 33 floats each; it must cover every record the fades of
 * [first, first + nframes) read: frames max(0, first - 8) .. min(num_frames, first + nframes + 8) - 1 (CalcFade2 samples n - 8 .. n + 8;
 * near the clip ends the reference's clamped indices stay inside that range).  d_fades_out: nframes * {fadeT, fadeB} floats on the
 * device, bit-identical to the host routine's.  async */
int  amtgpu_erase_calc_fades_device(AmtGpuErase* er, const float* d_analysis, int analysis_first, int analysis_count, int num_frames,
                                    int first, int nframes, float* d_fades_out);
/* amtgpu_erase_batch / amtgpu_erase_rect_batch with the fades taken from DEVICE memory (nframes * 2 floats, e.g. the output of
 * amtgpu_erase_calc_fades_device).  async */
int  amtgpu_erase_batch_dfades(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV,
                               int pitchY, int pitchUV, int bits, int nframes, const float* d_fades);
int  amtgpu_erase_rect_batch_dfades(AmtGpuErase* er, void* dY, void* dU, void* dV, int64_t strideY, int64_t strideUV,
                                    int pitchY, int pitchUV, int bits, int nframes, const float* d_fades);
/* amtgpu_erase_batch_dfades from a SOURCE batch into a DESTINATION batch of the same geometry (strides, pitches, bit depth) that already
 * holds a copy of the source frames -- the writable copy AMTEraseLogo::GetFrameT takes before it calls Delogo (env->MakeWritable,
 * LogoScan.hpp:1346-1347): Delogo reads the source's rectangle and writes the destination's.  The source stays intact for its other
 * consumers (the frame metrics, the scan, the next pass over the same frames); what Delogo leaves alone -- every sample outside the
 * rectangle, frames whose fades are {0, 0} (see amtgpu_erase_get_rect), an odd last chroma row in field mode -- is NOT written, it is
 * the copy's.  sY == dY (all three) is amtgpu_erase_batch_dfades.  async (ABI 5) */
int  amtgpu_erase_batch_dfades_to(AmtGpuErase* er, const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV,
                                  int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int bits, int nframes, const float* d_fades);
/* out5 = {imgx, imgy, w, h, fade0_is_identity}: the rectangle Delogo rewrites; the last word is 1 when a frame whose two fades
 * are 0 comes back unchanged (every a*s + b*maxv of this logo is finite), i.e. the host may skip the call for such frames --
 * of an 8- or 16-bit clip: at 10 / 12 bits Delogo's min(tmp + 0.5, maxv) (LogoScan.hpp:1258) still clamps container values
 * above maxv, so those frames must go through (the library itself skips fade-0 frames only at 8 and 16 bits) */
int  amtgpu_erase_get_rect(const AmtGpuErase* er, int* out5);

/* ---- logo generation: replaces logo::LogoScan (AddFrame :594-659, AddScanFrame :568-592,
 *      Normalize :471-488, GetLogo :490-566) and LogoAnalyzer / the exported ScanLogo (:794-1098) ---- */
AmtGpuLogoScan* amtgpu_logoscan_create(AmtGpuContext* ctx, int w, int h, int logUVx, int logUVy, int thy);
void amtgpu_logoscan_destroy(AmtGpuLogoScan* s);
/* planes of full frames; the scan rectangle sits at (imgx,imgy).  valid_out (host, nframes bytes, may be
 * NULL) receives AddFrame's verdict per frame.  At most `max_valid` further frames are accepted, in
 * stream order (numMaxFrames, :885).  Returns 1/0; *naccepted = frames accepted by this call.
 * use_mask (host, nframes bytes, may be NULL): only frames with use_mask[i]!=0 are offered (ReMakeLogo :1018) */
int  amtgpu_logoscan_add_batch(AmtGpuLogoScan* s, const void* dY, const void* dU, const void* dV,
                               int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int bits,
                               int imgx, int imgy, int nframes, int max_valid, const uint8_t* use_mask,
                               uint8_t* valid_out, int* naccepted);
int  amtgpu_logoscan_nframes(const AmtGpuLogoScan* s);
/* exact integer sums per pixel {sumF, sumF2, sumFB} (Y then U then V) + per plane {sumB, sumB2}: for the
 * sharded all-reduce.  sums: 3*(w*h+2*wUV*hUV) int64; plane_sums: 6 int64. */
int  amtgpu_logoscan_get_sums(AmtGpuLogoScan* s, int64_t* sums, int64_t* plane_sums);
int  amtgpu_logoscan_set_sums(AmtGpuLogoScan* s, const int64_t* sums, const int64_t* plane_sums, int nframes);
/* Normalize(maxv) + GetLogo(clean); NULL (+ error) when the regression fails ("Insufficient logo frames") */
AmtGpuLogo* amtgpu_logoscan_get_logo(AmtGpuLogoScan* s, int maxv, int clean, int imgw, int imgh, int imgx, int imgy);
/* Same signature family as the reference's ScanLogo (LogoScan.hpp:1083-1098), over a device-resident
 * 8-bit clip instead of a TS path: 1 ok / 0 fail. */
int  amtgpu_scanlogo(AmtGpuContext* ctx, const void* dY, const void* dU, const void* dV,
                     int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int imgw, int imgh,
                     int nframes, int serviceid, const char* dstpath, int imgx, int imgy, int w, int h,
                     int thy, int numMaxFrames, AMTGPU_LOGO_ANALYZE_CB cb);

/* The reference's exported ScanLogo with its own argument list (LogoScan.hpp:1083-1098; AmatsukazeNatives.cs:391-393):
 * (ctx, srcpath, serviceid, workfile, dstpath, imgx, imgy, w, h, thy, numMaxFrames, cb) -> 1 ok / 0 fail + amtgpu_last_error.
 * srcpath is a raw 8-bit 4:2:0 clip (int32 'AMTR', width, height, frames, then tight Y,U,V per frame) instead of a transport stream
 * (demux / decode are out of scope); frames stream through the pinned ring, accepted rectangles stay in HBM; workfile is unused. */
int  amtgpu_scanlogo_file(AmtGpuContext* ctx, const char* srcpath, int serviceid, const char* workfile, const char* dstpath,
                          int imgx, int imgy, int w, int h, int thy, int numMaxFrames, AMTGPU_LOGO_ANALYZE_CB cb);
/* the same with the reference's string type: NUL-terminated UTF-16 paths (P/Invoke CharSet.Unicode keeps working) */
int  amtgpu_scanlogo_fileW(AmtGpuContext* ctx, const uint16_t* srcpath, int serviceid, const uint16_t* workfile, const uint16_t* dstpath,
                           int imgx, int imgy, int w, int h, int thy, int numMaxFrames, AMTGPU_LOGO_ANALYZE_CB cb);

/* ---- frame-sharded runs (one process per GPU).  The library does no communication itself: the host
 *      supplies two collectives over HOST memory -- RCCL in a C++ host (include/amt_rccl_collectives.hpp wraps an ncclComm_t),
 *      torch.distributed in the Python mirror (amatsukaze_amd/sharding.py).  Ranks hold contiguous frame ranges in stream
 *      order (rank 0 the first frames).  Both callbacks return 1 on success, 0 on failure. ---- */
typedef struct AmtGpuCollectives {
    int rank, world;
    /* every rank contributes `bytes` bytes; recv receives world*bytes in rank order */
    int (*allgather)(void* user, const void* send, void* recv, int64_t bytes);
    /* in-place sum over all ranks of `count` int64 */
    int (*allreduce_sum_i64)(void* user, int64_t* buf, int64_t count);
    void* user;
} AmtGpuCollectives;

/* LogoFrame::scanFrames sharded (LogoScan.hpp:1577-1584, frames independent): after this rank has scanned its frames
 * [first, first+nlocal) with amtgpu_logoframe_scan_batch, exchange the {corr0,corr1} records so that EVERY rank holds the
 * whole clip's results (8 B per frame per logo) and select_logo / write_result give the single-GPU answer anywhere. */
int  amtgpu_logoframe_allgather_results(AmtGpuLogoFrame* lf, const AmtGpuCollectives* coll, int first, int nlocal);

/* ScanLogo sharded (LogoAnalyzer, LogoScan.hpp:917-1079): dY/dU/dV hold this rank's `nframes_local` frames.  The three
 * globally sequential rounds are kept: round 0 accepts the first numMaxFrames valid frames IN STREAM ORDER (:885; an
 * all-gather of per-rank valid counts gives every rank its share of the quota), each round's exact int64 accumulators are
 * all-reduced, and every rank solves the same regression -> the .lgd is byte-identical to the single-GPU one at any world
 * size.  Rank 0 writes dstpath (other ranks may pass NULL).  A cancel from any rank's callback stops all ranks. */
int  amtgpu_scanlogo_sharded(AmtGpuContext* ctx, const AmtGpuCollectives* coll, const void* dY, const void* dU, const void* dV,
                             int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int imgw, int imgh,
                             int nframes_local, int serviceid, const char* dstpath, int imgx, int imgy, int w, int h,
                             int thy, int numMaxFrames, AMTGPU_LOGO_ANALYZE_CB cb);

/* ---- self-specified whole-frame passes (NO in-tree reference arithmetic: SURVEY.md section 0;
 *      "parity unpinned").  Stand in for what chapter_exe (CMAnalyze.hpp:319-337) and KFMDeint's
 *      analysis passes (Misc.cs:1300-1324, FilteredSource.hpp:232-238) compute.  Integer metrics,
 *      specified in DESIGN.md section 6; per frame 8 x uint32/uint64 words, see AMTGPU_FS_*. ---- */
#define AMTGPU_FS_WORDS 8
/* word indices of one frame's record (uint64 each); all sums of absolute sample differences, prev = n-1,
 * vertical metrics over rows 1..H-2, avg(a,c) = (a+c)>>1 */
#define AMTGPU_FS_DIFF_TOP   0   /* sum over even rows |Y_n - Y_prev| */
#define AMTGPU_FS_DIFF_BOT   1   /* sum over odd rows  |Y_n - Y_prev| */
#define AMTGPU_FS_VERT       2   /* sum |Y_n[y-1] - Y_n[y+1]|  (detail inside one field) */
#define AMTGPU_FS_COMB       3   /* sum |Y_n[y] - avg(Y_n[y-1], Y_n[y+1])|  (combing energy of the frame) */
#define AMTGPU_FS_COMB_PREV  4   /* COMB of the weave (even rows of n, odd rows of n-1) */
#define AMTGPU_FS_SUM        5   /* sum of luma */
#define AMTGPU_FS_VERT_PREV  6   /* VERT of that weave */
#define AMTGPU_FS_RESERVED   7
AmtGpuFrameStats* amtgpu_framestats_create(AmtGpuContext* ctx, int width, int height, int bits);
void amtgpu_framestats_destroy(AmtGpuFrameStats* fs);
/* dprevY: Y plane of the frame before the batch (device), or NULL -> frame 0 compares with itself.
 * dout: nframes*AMTGPU_FS_WORDS uint64 (device).  async */
int  amtgpu_framestats_batch(AmtGpuFrameStats* fs, const void* dY, int64_t frame_stride, int pitch,
                             const void* dprevY, int nframes, uint64_t* dout);
/* Frame-sharded runs of the whole-frame passes (fourth row): every rank computes the metrics of its own
 * contiguous frame range [first, first + nlocal) with amtgpu_framestats_batch, passing the frame before its range as dprevY (the
 * one-frame halo: a rank decodes / generates frame first - 1 itself, nothing is exchanged for it; rank 0 passes NULL).  What IS
 * exchanged is the 64-byte record per frame: amtgpu_framestats_allgather takes this rank's records (HOST, nlocal * 8 uint64) and
 * leaves the records of the WHOLE clip (num_frames * 8 uint64, host) in metrics_out on every rank -- ragged shards padded to the
 * largest, ranges checked to tile [0, num_frames) exactly.  The cadence / scene-change decisions (amtgpu_kfm_cadence,
 * amtgpu_cm_scene_changes) then run replicated on every rank from identical input: sequential over the clip, integer only, tiny.
 * A rank-local failure travels as a status word through the exchange; all ranks return 0 together. */
int  amtgpu_framestats_allgather(AmtGpuFrameStats* fs, const AmtGpuCollectives* coll, const uint64_t* local_metrics, int first,
                                 int nlocal, int num_frames, uint64_t* metrics_out);
/* the same for a shard that is resident in HBM as ONE batch: metrics kernel over dY (nlocal frames, dprevY = frame first - 1 or
 * NULL on the rank that holds frame 0), then the exchange.  Synchronises. */
int  amtgpu_framestats_sharded(AmtGpuFrameStats* fs, const AmtGpuCollectives* coll, const void* dY, int64_t frame_stride, int pitch,
                               const void* dprevY, int first, int nlocal, int num_frames, uint64_t* metrics_out);
/* host decisions from the metrics of a whole clip (nframes*8 uint64, host): scene-change list in
 * chapter_exe's "SCPos:" sense and per-frame cadence class 0=30i/60p 1=24p(3:2) 2=30p + 3:2 phase.
 * sc_out: up to cap frame numbers, returns count via *nsc.  cadence_out: nframes bytes, phase_out: nframes bytes */
int  amtgpu_cm_scene_changes(const uint64_t* metrics, int nframes, int width, int height, int* sc_out, int cap, int* nsc);
int  amtgpu_kfm_cadence(const uint64_t* metrics, int nframes, int width, int height, uint8_t* cadence_out, uint8_t* phase_out);
/* KFM output-file contract (AMTDecimate, FilteredSource.hpp:645-654): one integer per output frame = how many
 * frames of the 60p clip it spans; sum == 2*nframes.  *nout = output frames written. */
int  amtgpu_kfm_write_durations(const uint8_t* cadence, const uint8_t* phase, int nframes, const char* path, int* nout);
/* KFM timecode contract (readTimecodeFile / readTimecode, FilteredSource.hpp:163-212): "# timecode format v2", one integer
 * start time in ms per output frame, "# total: <seconds>".  fps_num/fps_den = the SOURCE frame rate (30000/1001). */
int  amtgpu_kfm_write_timecode(const uint8_t* cadence, const uint8_t* phase, int nframes, int fps_num, int fps_den, const char* path,
                               int* nout);
/* chapter_exe output contract (CMAnalyze::readSceneChanges, CMAnalyze.hpp:411-439): header, a "----" line, "SCPos: <frame>"
 * lines; no "mute" lines (audio is out of scope) */
int  amtgpu_cm_write_chapter_exe(const int* scene_changes, int nsc, int nframes, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* AMT_GPU_H */
