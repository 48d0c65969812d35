// engine.hpp -- host runtime around the kernels: context, device buffers, evaluation engine.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

#include "eval_plan.h"
#include "eval_tiles.hpp"
#include "logo_model.hpp"

#define AMT_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// Instrumented builds only (-DAMT_TRACE_CALLS, build.py build_variant): host-side begin / duration of every C ABI call and of the
// waits inside them, written to stderr at exit as "amt_trace <name> <begin_us> <dur_us>" (tools/boundary_calls.py).
#ifdef AMT_TRACE_CALLS
#include <chrono>
#include <cstdio>
#include <vector>
struct AmtTrace {
    struct Rec { const char* name; double t0, dur; };
    std::vector<Rec> recs;
    std::mutex m;
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    ~AmtTrace() { for (const Rec& r : recs) std::fprintf(stderr, "amt_trace %s %.1f %.1f\n", r.name, r.t0 - (recs.empty() ? 0 : recs[0].t0), r.dur); }
    static AmtTrace& get() { static AmtTrace t; return t; }
};
struct AmtTraceScope {
    const char* name; double t0;
    explicit AmtTraceScope(const char* n) : name(n), t0(AmtTrace::now()) {}
    ~AmtTraceScope() { AmtTrace& t = AmtTrace::get(); std::lock_guard<std::mutex> lk(t.m); t.recs.push_back({name, t0, AmtTrace::now() - t0}); }
};
#define AMT_TRACE_CAT2(a, b) a##b
#define AMT_TRACE_CAT(a, b) AMT_TRACE_CAT2(a, b)
#define AMT_TRACE_SCOPE(n) AmtTraceScope AMT_TRACE_CAT(amt_trace_scope_, __LINE__)(n)
#else
#define AMT_TRACE_SCOPE(n) do { } while (0)
#endif

struct AmtGpuContext {
    int device = 0;
    hipStream_t stream = nullptr;       // compute stream (own or borrowed)
    hipStream_t own_stream = nullptr;
    hipStream_t copy_stream = nullptr;  // side stream for ingest
    hipEvent_t copy_done = nullptr;
    bool copies_pending = false;
    static constexpr int kRingSlots = 4;
    void* pinned = nullptr;             // pinned staging ring (kRingSlots slots): the CPU fills slots while the DMA engine drains earlier ones
    size_t pinned_bytes = 0;
    hipEvent_t slot_free[kRingSlots] = {};
    int next_slot = 0;                  // the slot being filled
    size_t slot_fill = 0;               // bytes of it handed out (small uploads share a slot: no event wait per call)
    // staging copies (pageable host memory -> pinned slot) are shared out over a few worker threads: one core's memcpy is below
    // what PCIe Gen5 x16 carries (amt_gpu_upload.hip)
    struct UploadPool;
    UploadPool* pool = nullptr;
    int upload_threads = 1;
    // host ranges the caller has page-locked through amtgpu_frames_register: uploads from inside them skip the staging ring
    std::vector<std::pair<uintptr_t, size_t>> registered;
#ifdef AMT_TRACE_CALLS
    // instrumented builds: timing events that bracket a block's GPU-side stages (first copy issued -> copies done -> the compute
    // stream's wait released -> kernels done -> results landed), read out in amtgpu_analyze_batch_host
    hipEvent_t tr_first_copy = nullptr, tr_copies_done = nullptr, tr_released = nullptr, tr_kernels = nullptr, tr_landed = nullptr;
    bool tr_block_open = false;
    // ... and GPU wall-clock stamps (100 MHz constant counter, written to pinned memory by one-thread kernels) mapped onto the host's
    // clock by a calibration at first use: when did the device START and FINISH each stage, against when the host enqueued / noticed
    unsigned long long* tr_stamps = nullptr;      // [8] pinned
    double tr_offset_us = 0;                      // host_us = gpu_ticks / 100 + tr_offset_us
    double tr_host_block_begin = 0;
#endif
    void* pinned_down = nullptr;        // pinned landing buffer of amtgpu_download_pinned
    size_t pinned_down_bytes = 0;
    hipEvent_t markers[16] = {};        // amtgpu_marker_record / _wait
    std::string err;
    // One context may be shared by several filter instances whose GetFrame runs on different AviSynth threads
    // (MT_NICE_FILTER): the staging ring, the error string and the timing spans are guarded by this lock.
    std::recursive_mutex mu;

    // optional per-kernel timing with HIP events on the launch stream (amtgpu_profile_*)
    struct ProfSpan { int name; hipEvent_t a, b; };
    bool profiling = false;
    std::vector<std::string> prof_names;
    std::vector<ProfSpan> prof_spans;
    std::vector<hipEvent_t> prof_pool;
    std::vector<double> prof_ms;        // resolved totals per name
    std::vector<long long> prof_calls;

    void bind() const { AMT_HIP(hipSetDevice(device)); }
    int prof_id(const char* name);
    hipEvent_t prof_event();
    // RAII-less helpers: begin returns a span index or -1 when profiling is off
    int prof_begin(const char* name);
    void prof_end(int span);
    void prof_resolve();
};

namespace amt {
void upload_pool_default(AmtGpuContext* c);
void context_stop_threads(AmtGpuContext* c);
// `bytes` from the device to ANY host memory, synchronously, on the context's stream: lands in the context's pinned buffer and is
// copied out from there.  Device-to-host copies never get a pageable destination: hipMemcpyAsync into unpinned memory waits for the
// stream inside the runtime, and on some hosts that wait is served at a 10 ms tick (profiles/r04_notes.md, "Boundary").
void download_via_pinned(AmtGpuContext* c, void* hdst, const void* dsrc, size_t bytes);

template <typename T> class DevBuf {
    T* p_ = nullptr;
    size_t n_ = 0;
public:
    DevBuf() = default;
    explicit DevBuf(size_t n) { alloc(n); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
    ~DevBuf() { release(); }
    void alloc(size_t n) { release(); if (n) { AMT_HIP(hipMalloc((void**)&p_, n * sizeof(T))); n_ = n; } }
    void release() { if (p_) { (void)hipFree(p_); p_ = nullptr; n_ = 0; } }
    void upload(const T* h, size_t n, hipStream_t st)
    {
        if (n > n_) alloc(n);
        if (n) {
            { AMT_TRACE_SCOPE("devbuf.upload.memcpy_h2d_pageable"); AMT_HIP(hipMemcpyAsync(p_, h, n * sizeof(T), hipMemcpyHostToDevice, st)); }
            { AMT_TRACE_SCOPE("devbuf.upload.sync"); AMT_HIP(hipStreamSynchronize(st)); }
        }
    }
    void upload(const std::vector<T>& h, hipStream_t st) { upload(h.data(), h.size(), st); }
    T* get() const { return p_; }
    size_t size() const { return n_; }
};

// tile plan of one evaluation logo resident in HBM (eval_tiles.hpp; eval_pair_kernels.hip).  slot = (band * kTileWaves + wave) * 64 + lane
struct TileLogoDev {
    const float2* kp;            // [13][nslots]  taps of the slot's mask pixel as pairs {k[2j], k[2j+1]} (k[25] = 0), pair-major
    const float2* sc;            // [32][nslots]  bin-major {scale, scale2} of the slot's mask pixel (the exact scan kernel)
    const float2* pq;            // [nslots]      {P, Q}: the pixel's response on flat level c is |P + Q c| (the linear analysis kernel: no gathers)
    const uint32_t* sinfo;       // [nslots]      tile_slot_info
    const uint32_t* pos;         // [nslots]      (y << 16) | x of the slot's mask pixel in the evaluation logo (the linear kernel's exact bin check)
    const TileDesc* tiles;       // [nbands * 8]
    const TileBandDesc* bands;   // [nbands]
    const int* tlist;            // [ntlist]  indices of the tiles that hold pixels (kernels that need no band order walk these)
    int nbands, nslots, ntlist;
    float floorResp;             // limitCorr of the logo (LogoScan.hpp:203)
    // the linear kernel's copy of everything it loads per tile, in ONE allocation (one scalar base instead of five: its loop is short of
    // scalar registers): kp at 0, then pq, sinfo, the evaluation logo's a and b planes at these byte offsets
    const char* lin;
    unsigned lin_pq, lin_sinfo, lin_a, lin_b;
};

// one evaluation logo + where its source pixels come from
struct EvalLogoSpec {
    LogoPlanes planes;       // evaluation logo (deinterlaced, or one field)
    MaskTables tables;
    int imgx = 0, imgy = 0;  // rectangle origin in the full frame
    int row0 = 0, row_step = 1, deint = 1;
    int out_off = 0;         // float offset in a frame's output record
};

// Evaluates `fades.size()` blends of every logo on every frame of a device batch:
// out[frame*out_frame_stride + spec.out_off + f] = (|.|) CorrelationScore / blackScore.
class EvalEngine {
public:
    // prof_name: label of this engine's launches in amtgpu_profile_report (no spaces)
    EvalEngine(AmtGpuContext* ctx, std::vector<EvalLogoSpec> specs, std::vector<float> fades, bool take_abs, int out_frame_stride,
               const char* prof_name = "logo_eval_fused_kernel");
    // async on ctx->stream; dout device, nframes*out_frame_stride floats
    // dframe_map (device, optional): batch frame i reads source frame dframe_map[i] of dY
    void run(const void* dY, int64_t frame_stride_bytes, int pitch, int bits, int nframes, float* dout,
             const int* dframe_map = nullptr);
    // Linear (decision-guarded) evaluation of all fades from one window evaluation of s and one of bg
    // (eval_linear_kernels.hip).  Results are within linear_error_bound(i, bits) of run()'s for logo i; same arguments as run().
    // dforce (device, optional, nframes bytes): the kernel sets the byte of every frame it could not finish in this mode (its list of
    // means next to a bin edge overflowed): the caller re-evaluates those frames exactly
    void run_linear(const void* dY, int64_t frame_stride_bytes, int pitch, int bits, int nframes, float* dout,
                    const int* dframe_map = nullptr, uint8_t* dforce = nullptr);
    // (pixel, frame, fade) pairs a wave of the linear kernel can list for the exact bin check (default 256; amtgpu_analyze_set_fixup_queue)
    void set_linear_queue(int entries) { lin_queue_ = entries; }
    // rigorous bound on |run_linear - run| for every score of logo i (rounding analysis in eval_engine.hip)
    float linear_error_bound(int logo, int bits) const;
    // exact re-evaluation of the frames listed on the device: batch slot j < *dcount reads source frame dlist[j] and its results
    // go to record dlist[j] of dout (scatter).  max_frames bounds *dcount.  async
    void run_listed(const void* dY, int64_t frame_stride_bytes, int pitch, int bits, int max_frames, const int* dlist, const int* dcount,
                    float* dout);
    int num_fades() const { return (int)fades_.size(); }
    int num_logos() const { return (int)specs_.size(); }
    const EvalLogoSpec& spec(int i) const { return specs_[i]; }
    // flops / bytes bookkeeping for the bench (algorithmic, per frame)
    double mask_pixel_evals_per_frame() const;

private:
    AmtGpuContext* ctx_;
    std::vector<EvalLogoSpec> specs_;
    std::vector<float> fades_;
    bool take_abs_;
    int out_frame_stride_;
    std::string prof_name_;
    int plane_cap_ = 0;
    int group_frames_ = 0;   // frames per workgroup (0 = pick per batch; AMTGPU_G overrides)
    std::vector<EvalBand> bands_;
    // device state
    std::vector<DevBuf<float>> d_a_, d_b_;
    std::vector<DevBuf<float2>> d_scales_, d_kslot_;
    std::vector<DevBuf<uint2>> d_slot2_;
    DevBuf<EvalLogoDev> d_logos_;
    DevBuf<EvalBand> d_bands_;
    DevBuf<float> d_fades_;
    // fades {0, 1} (the LogoFrame scan): both evaluations as one packed instruction stream (eval_pair_kernels.hip); decided once
    bool pair_eligible();
    bool pair_addressable(int pitch_bytes) const;
    bool tiles_usable(int pitch_bytes) const;
    int pair_state_ = -1;                          // -1 undecided, 0 generic kernel, 1 pair kernel
    // tile plans (eval_tiles.hpp), built when the pair kernel is chosen
    void ensure_tiles();
    bool tiles_ready_ = false;
    std::vector<DevBuf<float2>> d_tkp_, d_tsc_, d_tpq_;
    std::vector<DevBuf<uint32_t>> d_tinfo_, d_tpos_;
    std::vector<DevBuf<char>> d_tlin_;
    int lin_queue_ = 256;
    std::vector<DevBuf<TileDesc>> d_tiles_;
    std::vector<DevBuf<TileBandDesc>> d_tbands_;
    std::vector<DevBuf<int>> d_tlist_;
    DevBuf<TileLogoDev> d_tls_;
    // linear mode (built on first use)
    void ensure_linear();
    bool linear_ready_ = false;
    float vmax_unit_ = 1.0f;                       // max over logos / pixels of max(1, |a| + |b|) (x 2 for fades outside [0, 1]): window values are <= this * maxv
    std::vector<double> lin_err_corr_, lin_err_sum_;   // per logo: the two parts of the error bound, in units of (u * vmax) and u
    std::vector<double> lin_err_formula_;              // ... and what evaluating |P + Q c| instead of looking the scale up adds (absolute)
    bool lin_formula_ok_ = true;                       // every logo's responses fit the formula's preconditions (finite, floorResp > 0)
};

// kernel launcher (eval_fused_kernels.hip).  dnframes (device, optional): the number of frames actually present (<= nframes,
// which then only sizes the grid); scatter != 0: frame i's results go to record dframe_map[i] of dout; fade_chunk > 0 (with dnframes):
// a workgroup evaluates fade_chunk of the fades, the chunks of a (logo, frame group) run side by side -- a handful of listed frames
// is a latency problem (one workgroup walking every band for all fades), not a throughput one
hipError_t launch_logo_eval_fused(hipStream_t st, int bits, const EvalLogoDev* dlogos, int nlogos, const EvalBand* dbands,
                                  const float* dfades, int nfades, int fade0, const void* dY, const int* dframe_map,
                                  long long frame_stride_elems, int pitch, int nframes, int G, float* dout, int out_frame_stride,
                                  int take_abs, int plane_cap, const int* dnframes = nullptr, int scatter = 0, int fade_chunk = 0);
// eval_linear_kernels.hip
hipError_t launch_logo_eval_linear(hipStream_t st, int bits, const EvalLogoDev* dlogos, const TileLogoDev* dtls, int nlogos,
                                   const float* dfades, int nfades, int fade0, const void* dY, const int* dframe_map,
                                   long long frame_stride_elems, int pitch, int nframes, int G, float* dout, int out_frame_stride,
                                   int take_abs, float bin_eps, int qlog2, int qcap, uint8_t* dforce);
// eval_pair_kernels.hip: fades {0, 1} of every logo, bit-exact
hipError_t launch_logo_eval_pair(hipStream_t st, int bits, const EvalLogoDev* dlogos, const TileLogoDev* dtls, int nlogos,
                                 const void* dY, const int* dframe_map, long long frame_stride_elems, int pitch,
                                 int nframes, int G, float* dout, int out_frame_stride, int take_abs);
hipError_t launch_analysis_mark(hipStream_t st, const float* drec, int stride, int nframes, int ngroups, int nfades, const float* eps3,
                                int* dlist, int* dcount, const uint8_t* dforce = nullptr);
hipError_t launch_rect_range_flag(hipStream_t st, const void* dY, long long frame_stride_elems, int pitch, int imgx, int imgy, int w, int h, int bits,
                                  int nframes, uint8_t* dflag);

} // namespace amt
