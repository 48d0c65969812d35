// eval_engine.hip -- host side of the correlation kernels: table upload, band partition, chunked launches.
#include "engine.hpp"

#include <algorithm>
#include <cstdlib>

int AmtGpuContext::prof_id(const char* name)
{
    for (size_t i = 0; i < prof_names.size(); ++i) if (prof_names[i] == name) return (int)i;
    prof_names.push_back(name);
    prof_ms.push_back(0.0);
    prof_calls.push_back(0);
    return (int)prof_names.size() - 1;
}
hipEvent_t AmtGpuContext::prof_event()
{
    if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
    hipEvent_t e;
    AMT_HIP(hipEventCreate(&e));
    return e;
}
int AmtGpuContext::prof_begin(const char* name)
{
    if (!profiling) return -1;
    if (prof_spans.size() >= 4096) prof_resolve();
    ProfSpan sp{prof_id(name), prof_event(), prof_event()};
    AMT_HIP(hipEventRecord(sp.a, stream));
    prof_spans.push_back(sp);
    return (int)prof_spans.size() - 1;
}
void AmtGpuContext::prof_end(int span)
{
    if (span < 0) return;
    AMT_HIP(hipEventRecord(prof_spans[span].b, stream));
}
void AmtGpuContext::prof_resolve()
{
    for (auto& sp : prof_spans) {
        AMT_HIP(hipEventSynchronize(sp.b));
        float ms = 0;
        AMT_HIP(hipEventElapsedTime(&ms, sp.a, sp.b));
        prof_ms[sp.name] += ms;
        prof_calls[sp.name] += 1;
        prof_pool.push_back(sp.a);
        prof_pool.push_back(sp.b);
    }
    prof_spans.clear();
}

namespace amt {

namespace {
int lds_pitch(int w)
{
    int pad = 8;
    if (const char* e = std::getenv("AMTGPU_LPPAD")) pad = std::atoi(e);      // experiments
    return ((w + 31) & ~31) + pad;
}
}

EvalEngine::EvalEngine(AmtGpuContext* ctx, std::vector<EvalLogoSpec> specs, std::vector<float> fades, bool take_abs,
                       int out_frame_stride)
    : ctx_(ctx), specs_(std::move(specs)), fades_(std::move(fades)), take_abs_(take_abs), out_frame_stride_(out_frame_stride)
{
    ctx_->bind();
    if (const char* e = std::getenv("AMTGPU_PXT")) pxt_ = std::atoi(e);
    if (const char* e = std::getenv("AMTGPU_NT")) nt_ = std::atoi(e);
    if (eval_stage_per_thread(pxt_, nt_) == 0) { pxt_ = 1; nt_ = 256; }
    const int kPlaneCapMax = nt_ * eval_stage_per_thread(pxt_, nt_);   // floats per LDS plane
    const int nl = (int)specs_.size();
    const int nf = (int)fades_.size();
    std::vector<EvalLogoDev> hl(nl);
    d_a_.resize(nl); d_b_.resize(nl); d_kern_.resize(nl); d_pos_.resize(nl); d_slots_.resize(nl); d_scales_.resize(nl);
    long long off = 0;
    plane_cap_ = 0;
    for (int i = 0; i < nl; ++i) {
        EvalLogoSpec& S = specs_[i];
        const MaskTables& T = S.tables;
        const int w = S.planes.w, h = S.planes.h;
        const int lp = lds_pitch(w);
        if (w > 0xFFFF || h > 0xFFFF) throw std::runtime_error("logo too large");
        if (5 * lp > kPlaneCapMax) throw std::runtime_error("logo too wide for the evaluation kernel");
        const int cpad = std::max(kTablePad, (T.count + kTablePad - 1) / kTablePad * kTablePad);

        // run slots: greedily group horizontally adjacent mask pixels (same row, x+1) -- they are consecutive
        // in raster order -- into runs of at most pxt_ pixels; one thread evaluates one slot
        std::vector<uint32_t> slots;
        for (int m = 0; m < T.count;) {
            int n = 1;
            while (n < pxt_ && m + n < T.count && T.pos[m + n] == T.pos[m + n - 1] + 1) ++n;
            slots.push_back(((uint32_t)n << 28) | (uint32_t)m);
            m += n;
        }
        if (T.count >= (1 << 28)) throw std::runtime_error("logo too large");
        // bands: up to nt_ consecutive slots (one per thread) whose windows fit the LDS plane
        {
            const int ns = (int)slots.size();
            int s = 0;
            while (s < ns) {
                EvalBand B;
                B.logo = i; B.s0 = s;
                const int ytop = (int)(T.pos[slots[s] & 0x0FFFFFFFu] >> 16) - 2;
                int e = s;
                while (e < ns && e - s < nt_) {
                    const int ybot = (int)(T.pos[slots[e] & 0x0FFFFFFFu] >> 16) + 2;
                    if ((ybot - ytop + 1) * lp > kPlaneCapMax) break;
                    ++e;
                }
                B.nslots = e - s;
                B.y0 = ytop;
                B.nrows = (int)(T.pos[slots[e - 1] & 0x0FFFFFFFu] >> 16) + 2 - ytop + 1;
                plane_cap_ = std::max(plane_cap_, B.nrows * lp);
                bands_.push_back(B);
                s = e;
            }
        }

        // tables, re-laid for the kernel: taps and bins major, mask pixels minor
        std::vector<uint32_t> pos(cpad, T.count ? T.pos[0] : ((2u << 16) | 2u));
        std::copy(T.pos.begin(), T.pos.end(), pos.begin());
        std::vector<float> kern((size_t)25 * cpad, 0.0f);
        std::vector<float2> scales((size_t)kNumBins * cpad, float2{0.0f, 0.0f});
        for (int m = 0; m < T.count; ++m) {
            for (int t = 0; t < 25; ++t) kern[(size_t)t * cpad + m] = T.kernels[(size_t)m * 25 + t];
            for (int c = 0; c < kNumBins; ++c)
                scales[(size_t)c * cpad + m] = float2{T.scales[((size_t)m * 32 + c) * 2], T.scales[((size_t)m * 32 + c) * 2 + 1]};
        }
        if (slots.empty()) slots.push_back(0);
        d_a_[i].upload(S.planes.A(0), (size_t)w * h, ctx_->stream);
        d_b_[i].upload(S.planes.B(0), (size_t)w * h, ctx_->stream);
        d_pos_[i].upload(pos, ctx_->stream);
        d_slots_[i].upload(slots, ctx_->stream);
        d_kern_[i].upload(kern, ctx_->stream);
        d_scales_[i].upload(scales, ctx_->stream);

        EvalLogoDev& D = hl[i];
        D.a = d_a_[i].get(); D.b = d_b_[i].get(); D.pos = d_pos_[i].get(); D.slots = d_slots_[i].get();
        D.kern = d_kern_[i].get(); D.scales = d_scales_[i].get();
        D.w = w; D.h = h; D.count = T.count; D.count_pad = cpad;
        D.imgx = S.imgx; D.imgy = S.imgy; D.row0 = S.row0; D.row_step = S.row_step; D.deint = S.deint;
        D.score_off = (int)off;
        D.blackScore = T.blackScore;
        D.out_off = S.out_off;
        D.lp = lp;
        D.lp_magic = (uint32_t)((0x100000000ull + lp - 1) / lp);
        off += (long long)nf * cpad;
    }
    scores_per_frame_ = off;
    d_logos_.upload(hl, ctx_->stream);
    d_bands_.upload(bands_, ctx_->stream);
    d_fades_.upload(fades_, ctx_->stream);

    // frames per launch: large enough that each launch fills the chip many times over and the ordered-sum pass
    // has thousands of independent rows (its per-row chain of adds is serial); the scratch round trip through
    // HBM (~2 MB per frame for the 33-evaluation analysis) is far below the 8 TB/s roofline at this kernel's
    // VALU-bound frame rate.  Override with AMTGPU_SCRATCH_MB.
    long long budget_mb = 1536;
    if (const char* e = std::getenv("AMTGPU_SCRATCH_MB")) budget_mb = std::max(1LL, std::atoll(e));
    const long long per_frame_bytes = std::max(1LL, scores_per_frame_ * 4);
    chunk_frames_ = (int)std::max(1LL, std::min(65536LL, budget_mb * (1LL << 20) / per_frame_bytes));
}

double EvalEngine::mask_pixel_evals_per_frame() const
{
    double s = 0;
    for (const auto& sp : specs_) s += (double)sp.tables.count * fades_.size();
    return s;
}

void EvalEngine::run(const void* dY, int64_t frame_stride_bytes, int pitch, int bits, int nframes, float* dout,
                     const int* dframe_map)
{
    if (nframes <= 0 || specs_.empty() || bands_.empty()) return;
    ctx_->bind();
    const int es = bits <= 8 ? 1 : 2;
    if (frame_stride_bytes % es) throw std::runtime_error("frame stride not a multiple of the sample size");
    const int chunk = std::min(nframes, chunk_frames_);
    if (d_scratch_.size() < (size_t)chunk * scores_per_frame_) d_scratch_.alloc((size_t)chunk * scores_per_frame_);
    for (int f0 = 0; f0 < nframes; f0 += chunk) {
        const int n = std::min(chunk, nframes - f0);
        const uint8_t* base = static_cast<const uint8_t*>(dY) + (dframe_map ? 0 : (int64_t)f0 * frame_stride_bytes);
        int sp = ctx_->prof_begin("logo_corr_kernel");
        AMT_HIP(launch_logo_corr(ctx_->stream, bits, pxt_, nt_, d_logos_.get(), d_bands_.get(), (int)bands_.size(), d_fades_.get(),
                                 (int)fades_.size(), base, dframe_map ? dframe_map + f0 : nullptr, frame_stride_bytes / es, pitch,
                                 n, d_scratch_.get(), scores_per_frame_, plane_cap_));
        ctx_->prof_end(sp);
        sp = ctx_->prof_begin("ordered_sum_kernel");
        AMT_HIP(launch_ordered_sum(ctx_->stream, d_logos_.get(), (int)specs_.size(), (int)fades_.size(), n, d_scratch_.get(),
                                   scores_per_frame_, dout + (size_t)f0 * out_frame_stride_, out_frame_stride_, take_abs_ ? 1 : 0));
        ctx_->prof_end(sp);
    }
}

} // namespace amt
